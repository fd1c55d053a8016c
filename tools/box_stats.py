#!/usr/bin/env python3
"""Which target boxes do the patch gather's units (4x2 pixel blocks) need at the bench's finest level?  Replays the coarse levels
of the headline problem, then evaluates the warp at 640x480 in torch and histograms (box width, box height) per unit: the share
that fits the fixed-stride patch (<= 8 x 5 texels), the packed one (w h <= 36), and what falls back to direct loads."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for mode in ("prior", "after the four coarser levels", "after the full solve"):
    prob = bench.Problem(B, 2, bench.H, bench.W, bench.K, 1234, dev)
    st = prob.ba.new_state(T=prob.T0)
    iters = {"prior": [0, 0, 0, 0, 0], "after the four coarser levels": [10, 10, 10, 10, 0], "after the full solve": [10] * 5}[mode]
    prob.ba.solve(iters, st)
    lv = prob.levels[-1]
    H, W = lv.H, lv.W
    intr = prob.intr
    fx0, fy0, ox0, oy0 = (intr[:, i].view(B, 1, 1) for i in range(4))
    yy, xx = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    p = torch.stack([(xx * lv.scale - ox0) / fx0, (yy * lv.scale - oy0) / fy0, torch.ones(B, H, W, device=dev)], dim=1)
    p = p / p.norm(dim=1, keepdim=True)
    D = lv.depth.reshape(B, H, W) + (lv.basis.reshape(B, H * W, -1) @ st.Wc).reshape(B, H, W)
    Rp = torch.einsum("bij,bjhw->bihw", st.R.reshape(B, 3, 3), p)
    X = Rp * D.unsqueeze(1) + st.T.reshape(B, 3, 1, 1)
    px = fx0 / lv.scale * X[:, 0] / X[:, 2] + ox0 / lv.scale
    py = fy0 / lv.scale * X[:, 1] / X[:, 2] + oy0 / lv.scale
    x0, y0 = torch.floor(px), torch.floor(py)
    fast = (x0 >= 1) & (x0 + 2 <= W - 1) & (y0 >= 1) & (y0 + 2 <= H - 1) & (px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)
    big = 1e9
    def unit(v, fill, red):
        v = torch.where(fast, v, torch.full_like(v, fill)).reshape(B, H // 2, 2, W // 4, 4)
        return red(red(v, 4).values, 2).values
    bx0, bx1 = unit(x0, big, torch.min), unit(x0, -big, torch.max)
    by0, by1 = unit(y0, big, torch.min), unit(y0, -big, torch.max)
    ok = bx1 >= bx0
    pw, ph = (bx1 - bx0 + 4)[ok], (by1 - by0 + 4)[ok]
    n = float(ok.sum())
    fs = float(((pw <= 8) & (ph <= 5)).sum()) / n
    pk = float((pw * ph <= 36).sum()) / n
    fs6 = float(((pw <= 8) & (ph <= 6)).sum()) / n
    hist = {}
    for a, b in zip(pw.tolist()[::97], ph.tolist()[::97]):
        hist[(int(a), int(b))] = hist.get((int(a), int(b)), 0) + 1
    top = sorted(hist.items(), key=lambda kv: -kv[1])[:8]
    tot = sum(hist.values())
    print("%-32s units with taps %.3f | fit 8x5 fixed-stride %.3f, packed (w h <= 36) %.3f, 8x6 %.3f | boxes: %s" % (
        mode, n / ok.numel(), fs, pk, fs6, ", ".join("%dx%d %.0f%%" % (k[0], k[1], 100.0 * v / tot) for k, v in top)))
