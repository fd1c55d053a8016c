"""float64 torch restatement of ONE dense assembly pass (same arithmetic as
oracle.banet_oracle.bundle_iteration / legacy_camera_iteration up to the normal equations).

TEST INFRASTRUCTURE.  The numpy oracle is the reference for parity; this twin exists so the
640x480 / K=128 sizes of BASELINE.json can be checked on the GPU box in seconds (the numpy
oracle needs minutes there).  It is itself checked against the oracle on CPU at small sizes
(tests/test_torch_ref_cpu.py).
"""
import torch


def grad_fixed(img):
    """bundlenet.py:92-100; img [B,H,W,C]"""
    H, W = img.shape[1], img.shape[2]
    p = torch.nn.functional.pad(img.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect").permute(0, 2, 3, 1)
    gx = 0.5 * (p[:, 1:H + 1, 2:W + 2, :] - p[:, 1:H + 1, 0:W, :])
    gy = 0.5 * (p[:, 2:H + 2, 1:W + 1, :] - p[:, 0:H, 1:W + 1, :])
    return gx, gy


def _gather(flat, W, yy, xx):
    C = flat.shape[-1]
    return torch.gather(flat, 1, (yy * W + xx).unsqueeze(-1).expand(-1, -1, C))


def dense_assemble(intr, scale, src, tgt, depth, basis, R, T, Wc, bundle, normalize_rays, dtype=torch.float64):
    """-> AtA [B,P,P], Atb [B,P], absres [B,C], nvalid [B]   (P = 6 + K)."""
    B, H, W, C = tgt.shape
    N = H * W
    dev = tgt.device
    f = lambda x: x.to(dtype)  # noqa: E731
    src, tgt, depth, R, T, intr = f(src), f(tgt), f(depth).reshape(B, N), f(R).reshape(B, 3, 3), f(T).reshape(B, 3, 1), f(intr)
    K = 0 if basis is None else basis.shape[-1]
    vv, uu = torch.meshgrid(torch.arange(H, dtype=dtype, device=dev), torch.arange(W, dtype=dtype, device=dev),
                            indexing="ij")
    fx0, fy0, ox0, oy0 = [intr[:, i:i + 1] for i in range(4)]
    u, v = (uu.reshape(1, N) * scale), (vv.reshape(1, N) * scale)
    p = torch.stack([(u - ox0) / fx0, (v - oy0) / fy0, torch.ones(B, N, dtype=dtype, device=dev)], dim=1)
    if normalize_rays:
        p = p / torch.sqrt(torch.clamp((p * p).sum(1, keepdim=True), min=1e-12))
    fx, fy, ox, oy = fx0 / scale, fy0 / scale, ox0 / scale, oy0 / scale
    D = depth
    if K > 0:
        Bs = f(basis).reshape(B, N, K)
        D = D + torch.matmul(Bs, f(Wc).reshape(B, K, 1))[..., 0]
    Rp = torch.matmul(R, p)
    X = Rp * D.unsqueeze(1) + T
    x, y, Z = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2], X[:, 2]
    px, py = fx * x + ox, fy * y + oy
    mask = ((px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)).to(dtype)
    pxs = torch.where(mask > 0, px, torch.zeros_like(px))
    pys = torch.where(mask > 0, py, torch.zeros_like(py))
    x0f, y0f = torch.floor(pxs), torch.floor(pys)
    dx, dy = pxs - x0f, pys - y0f
    x0, y0 = x0f.long(), y0f.long()
    x1, y1 = (x0 + 1).clamp(0, W - 1), (y0 + 1).clamp(0, H - 1)
    x0, y0 = x0.clamp(0, W - 1), y0.clamp(0, H - 1)
    gxm, gym = grad_fixed(tgt)
    w00, w01, w10, w11 = (1 - dx) * (1 - dy), dx * (1 - dy), (1 - dx) * dy, dx * dy

    def samp(m):
        fl = m.reshape(B, N, C)
        return (_gather(fl, W, y0, x0) * w00.unsqueeze(-1) + _gather(fl, W, y0, x1) * w01.unsqueeze(-1)
                + _gather(fl, W, y1, x0) * w10.unsqueeze(-1) + _gather(fl, W, y1, x1) * w11.unsqueeze(-1))

    mk = mask.unsqueeze(-1)
    F2w, gx, gy = samp(tgt), samp(gxm) * mk, samp(gym) * mk
    d = (F2w - src.reshape(B, N, C)) * mk                     # legacy sign
    zero = torch.zeros_like(x)
    iz = 1.0 / Z
    Jx = fx.unsqueeze(-1) * torch.stack([x * y, -1 - x * x, y, -iz, zero, x / Z], dim=-1)
    Jy = fy.unsqueeze(-1) * torch.stack([1 + y * y, -x * y, -x, zero, -iz, y / Z], dim=-1)
    Jx, Jy = Jx * mk, Jy * mk
    if bundle:                                               # bundlenet.py:60,234: J = [-Jc | jd b], d = F1 - F2w
        d = -d
        Jx, Jy = -Jx, -Jy
        if K > 0:
            jd0 = fx * ((Rp[:, 0] - Rp[:, 2] * x) / Z) * mask
            jd1 = fy * ((Rp[:, 1] - Rp[:, 2] * y) / Z) * mask
            Jx = torch.cat([Jx, jd0.unsqueeze(-1) * Bs], dim=-1)
            Jy = torch.cat([Jy, jd1.unsqueeze(-1) * Bs], dim=-1)
    Jx = torch.nan_to_num(Jx)
    Jy = torch.nan_to_num(Jy)
    m11, m12, m22 = (gx * gx).sum(-1), (gx * gy).sum(-1), (gy * gy).sum(-1)
    g1, g2 = (gx * d).sum(-1), (gy * d).sum(-1)
    Zx = m11.unsqueeze(-1) * Jx + m12.unsqueeze(-1) * Jy
    Zy = m12.unsqueeze(-1) * Jx + m22.unsqueeze(-1) * Jy
    AtA = torch.matmul(Zx.transpose(1, 2), Jx) + torch.matmul(Zy.transpose(1, 2), Jy)
    Atb = (Jx * g1.unsqueeze(-1) + Jy * g2.unsqueeze(-1)).sum(1)
    return AtA, Atb, d.abs().sum(1), mask.sum(1)
