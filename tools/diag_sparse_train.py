#!/usr/bin/env python3
"""Which training graph is closer to float64?  One CameraIteration / BundleIteration with gradients on sparse points: the fused node
(dense_train._SparseIteration), the lean torch graph and a float64 pure-torch restatement of the same statements
(bundlenet.py:122-278 with the resampler as a differentiable expression), per gradient tensor: max |g - g64| / max |g64|.
    PB=2 PN=4096 PC=128 PK=128 PH=96 PW=128 python tools/diag_sparse_train.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import ops  # noqa: E402
from banet_amd.bundlenet import (AngleaAxisRotation, BundleNet, CameraJacobianMatrix, DepthJacobianMatrix, VMatrix,  # noqa: E402
                                 _resampler_autograd, he_normal_lambda_weights)

DEV = "cuda:0"
B, N, C, K = (int(os.environ.get(k, d)) for k, d in (("PB", "2"), ("PN", "4096"), ("PC", "128"), ("PK", "128")))
H, W = int(os.environ.get("PH", "96")), int(os.environ.get("PW", "128"))
g = torch.Generator().manual_seed(1000 + N)
img = torch.randn(B, H, W, C, generator=g).to(DEV)
conv2 = ops.target_map(img)
pts = torch.stack([torch.rand(B, N, generator=g) * (W - 1.5) + 0.25, torch.rand(B, N, generator=g) * (H - 1.5) + 0.25], dim=-1).to(DEV)
conv1 = ops.resample(img, pts) + 0.05 * torch.randn(B, N, C, generator=g).to(DEV)
fx = torch.full((B, N), 0.8 * W, device=DEV)
fy = fx.clone()
ox = torch.full((B, N), W / 2.0, device=DEV)
oy = torch.full((B, N), H / 2.0, device=DEV)
ray = torch.stack([(pts[..., 0] - ox) / fx, (pts[..., 1] - oy) / fy, torch.ones(B, N, device=DEV)], dim=1)
p = ray / ray.norm(dim=1, keepdim=True)
D = (2.5 + torch.rand(B, N, 1, generator=g)).to(DEV)
Bs = (torch.randn(B, N, K, generator=g) / K ** 0.5).to(DEV)
R = torch.eye(3, device=DEV).repeat(B, 1, 1)
T = (0.02 * torch.randn(B, 3, 1, generator=g)).to(DEV)
Wc = (0.01 * torch.randn(B, K, 1, generator=g)).to(DEV)
cR, cT, cW = [torch.randn(x.shape, generator=g).to(DEV) for x in (R, T, Wc)]
lw0 = he_normal_lambda_weights(C, 7)


def iteration64(conv1, conv2, D, Bs, R, T, Wc, lw, bundle, l2):
    """float64 statements (the lean graph's, with the C-wide part written out)"""
    f8 = lambda x: x.double()  # noqa: E731
    fx8, fy8, ox8, oy8, p8 = f8(fx), f8(fy), f8(ox), f8(oy), f8(p)
    Dd = D + torch.matmul(Bs, Wc) if bundle else D
    Rp = torch.matmul(R, p8)
    rx, ry, rz = Rp[:, 0], Rp[:, 1], Rp[:, 2]
    RPT = Rp * Dd.transpose(1, 2) + T
    X, Y, Z = RPT[:, 0], RPT[:, 1], RPT[:, 2]
    x, y = X / Z, Y / Z
    px, py = fx8 * x + ox8, fy8 * y + oy8
    samp = _resampler_autograd(conv2, torch.stack([px, py], dim=-1))
    Hh, Ww = conv2.shape[1], conv2.shape[2]
    m = (~((px < 0) | (px > float(Ww - 1)) | (py < 0) | (py > float(Hh - 1)))).to(torch.float64)
    d = (conv1 - samp[..., :C]) * m[..., None]
    gx, gy = samp[..., C:2 * C] * m[..., None], samp[..., 2 * C:] * m[..., None]
    M11, M12, M22, g1, g2 = (gx * gx).sum(-1), (gx * gy).sum(-1), (gy * gy).sum(-1), (gx * d).sum(-1), (gy * d).sum(-1)
    absd = d.abs().sum(dim=1)
    avg = (absd / float(N)).unsqueeze(1)
    h = avg
    for i, (w, b) in enumerate(lw):
        z = torch.matmul(h, w) + b
        h = torch.tanh(z) if i == 4 else torch.nn.functional.selu(z)
    lam = torch.linalg.vector_norm(avg, dim=-1, keepdim=True) ** (2.0 + h)
    if bundle:
        lam = l2 * lam
    Jc = CameraJacobianMatrix(x, y, Z, fx8, fy8)
    MJ0 = M11.unsqueeze(-1) * Jc[:, :, 0] + M12.unsqueeze(-1) * Jc[:, :, 1]
    MJ1 = M12.unsqueeze(-1) * Jc[:, :, 0] + M22.unsqueeze(-1) * Jc[:, :, 1]
    Hcc = torch.matmul(Jc[:, :, 0].transpose(1, 2), MJ0) + torch.matmul(Jc[:, :, 1].transpose(1, 2), MJ1)
    bc = (Jc[:, :, 0] * g1.unsqueeze(-1) + Jc[:, :, 1] * g2.unsqueeze(-1)).sum(dim=1)
    nb = conv1.shape[0]
    if bundle:
        jd = DepthJacobianMatrix(rx.unsqueeze(1), ry.unsqueeze(1), rz.unsqueeze(1), x, y, Z, fx8, fy8)
        u = MJ0 * jd[..., 0:1] + MJ1 * jd[..., 1:2]
        s = M11 * jd[..., 0] ** 2 + 2.0 * M12 * jd[..., 0] * jd[..., 1] + M22 * jd[..., 1] ** 2
        r = jd[..., 0] * g1 + jd[..., 1] * g2
        Hcd = torch.matmul(u.transpose(1, 2), Bs)
        Hdd = torch.matmul(Bs.transpose(1, 2), Bs * s.unsqueeze(-1))
        bd = torch.matmul(Bs.transpose(1, 2), r.unsqueeze(-1)).squeeze(-1)
        AtA = torch.cat([torch.cat([Hcc, Hcd], dim=2), torch.cat([Hcd.transpose(1, 2), Hdd], dim=2)], dim=1)
        Atb = torch.cat([bc, bd], dim=1).unsqueeze(-1)
        diag = torch.diagonal(AtA, dim1=1, dim2=2)
        damp = torch.cat([diag[:, :-1] + 1e-5, torch.zeros(nb, 1, device=diag.device, dtype=diag.dtype)], dim=-1)
    else:
        AtA, Atb = Hcc, bc.unsqueeze(-1)
        damp = torch.diagonal(AtA, dim1=1, dim2=2) + 1e-5
    AtA = AtA + torch.diag_embed(damp * lam.squeeze(-1))
    sol = torch.linalg.solve(AtA, Atb)
    wx, wy, wz = sol[:, 0], sol[:, 1], sol[:, 2]
    dr = AngleaAxisRotation(wx, wy, wz)
    dv = VMatrix(wx.reshape(-1), wy.reshape(-1), wz.reshape(-1))
    return torch.matmul(dr, R), torch.matmul(dv, sol[:, 3:6]) + torch.matmul(dr, T), (Wc + sol[:, 6:]) if bundle else None, lam


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


for bundle in (True, False):
    res = {}
    for graph in ("f64", "lean", "fused"):
        dt = torch.float64 if graph == "f64" else torch.float32
        lw = [(w.to(DEV).to(dt).requires_grad_(True), b.to(DEV).to(dt).requires_grad_(True)) for w, b in lw0]
        leaves = [x.clone().to(dt).requires_grad_(True) for x in (conv1, conv2, D, Bs, R, T, Wc)]
        if graph == "f64":
            R2, T2, W2, lam = iteration64(*leaves, lw, bundle, 1000.0)
        else:
            net = BundleNet(lambda_weights={"0": lw})
            net.training_graph = graph
            if bundle:
                R2, T2, W2 = net.BundleIteration(leaves[0], leaves[1], fx, fy, ox, oy, p, leaves[2], leaves[3], leaves[4], leaves[5], leaves[6], 1000.0, "0")
            else:
                R2, T2 = net.CameraIteration(leaves[0], leaves[1], fx, fy, ox, oy, p, leaves[2], leaves[4], leaves[5], 1.0, "0")
        loss = (R2 * cR.to(dt)).sum() + (T2 * cT.to(dt)).sum() + ((W2 * cW.to(dt)).sum() if bundle else 0.0)
        wrt = [leaves[i] for i in ((0, 1, 2, 3, 4, 5, 6) if bundle else (0, 1, 2, 4, 5))] + [x for wb in lw for x in wb]
        grads = torch.autograd.grad(loss, wrt, allow_unused=True)
        res[graph] = ([R2, T2] + ([W2] if bundle else []), grads)
        if graph == "f64":
            print("%s: lambda %s, |T2 - T| %.3e" % ("BundleIteration" if bundle else "CameraIteration", lam.reshape(-1).tolist(), float((T2 - T.double()).abs().max())))
    names = (["conv1", "conv2", "D", "Bs", "R", "T", "W"] if bundle else ["conv1", "conv2", "D", "R", "T"]) + ["lw%d" % i for i in range(10)]
    for graph in ("lean", "fused"):
        print("  %-5s outputs vs f64: %s" % (graph, ["%.1e" % rel(a, b) for a, b in zip(res[graph][0], res["f64"][0])]))
        print("  %-5s grads   vs f64: %s" % (graph, ", ".join("%s %.1e" % (nm, rel(a, b)) for nm, a, b in zip(names, res[graph][1], res["f64"][1]) if a is not None)))
    print("  max |grad| (f64): %s" % ", ".join("%s %.1e" % (nm, float(b.abs().max())) for nm, b in zip(names, res["f64"][1])))
