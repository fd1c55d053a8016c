"""Synthetic dense 2-frame scenes on the GPU (torch), same construction as oracle/synth.py:
target features = analytic smooth field, source features = the field at the GT-warped
coordinates, depth = D0 + basis.W_gt.  Used by bench.py (data: "synthetic")."""
import math

import torch

from .dense import DenseLevel


def _field_params(C, seed, device, wl_min=48.0, wl_max=320.0, nwaves=8):
    g = torch.Generator().manual_seed(seed)
    amp = (torch.rand(C, nwaves, generator=g) + 0.5) / math.sqrt(nwaves)
    wl = torch.exp(torch.rand(C, nwaves, generator=g) * (math.log(wl_max) - math.log(wl_min)) + math.log(wl_min))
    ang = torch.rand(C, nwaves, generator=g) * 2 * math.pi
    phi = torch.rand(C, nwaves, generator=g) * 2 * math.pi
    kx, ky = 2 * math.pi / wl * torch.cos(ang), 2 * math.pi / wl * torch.sin(ang)
    return [t.to(device) for t in (amp, kx, ky, phi)]


def _eval_field(params, u, v, chunk=16):
    amp, kx, ky, phi = params
    C = amp.shape[0]
    out = torch.empty(u.shape + (C,), dtype=torch.float32, device=u.device)
    for c0 in range(0, C, chunk):
        sl = slice(c0, min(C, c0 + chunk))
        ph = u[..., None, None] * kx[sl] + v[..., None, None] * ky[sl] + phi[sl]
        out[..., sl] = (amp[sl] * torch.sin(ph)).sum(-1)
    return out


def _depth0(u, v, W, H):
    return 2.8 + 0.5 * torch.sin(2 * math.pi * u / W * 1.3 + 0.4) * torch.cos(2 * math.pi * v / H * 0.9 + 0.2) \
        + 0.3 * torch.cos(2 * math.pi * (u / W + v / H) * 0.8)


def _dct_basis(u, v, W, H, K):
    order = sorted(((i + j, i, j) for i in range(32) for j in range(32) if i + j > 0))[:K]
    out = torch.empty(u.shape + (K,), dtype=torch.float32, device=u.device)
    for k, (_, i, j) in enumerate(order):
        ci = torch.cos(math.pi * (u + 0.5) * i / W) * (math.sqrt(2.0) if i else 1.0)
        cj = torch.cos(math.pi * (v + 0.5) * j / H) * (math.sqrt(2.0) if j else 1.0)
        out[..., k] = ci * cj
    return out


def _rodrigues(w):
    th = float(torch.linalg.norm(w))
    if th < 1e-12:
        return torch.eye(3)
    k = w / th
    Kx = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=torch.float32)
    return torch.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * Kx @ Kx


def make_dense_windows(B, H, W, C, K, scales, seed, device, normalize_rays=True, rot_mag=0.012, trans_mag=0.03,
                       noise=0.01, pairs=1):
    """B independent windows.  pairs = 1: 2-frame windows, returns (intr [B,4], levels [DenseLevel
    coarse->fine], gt dict).  pairs > 1: multi-frame windows (one key frame + `pairs` target frames, see
    make_multiframe_windows)."""
    if pairs > 1:
        return make_multiframe_windows(B, H, W, C, K, scales, seed, device, pairs, normalize_rays, rot_mag, trans_mag,
                                       noise)
    g = torch.Generator().manual_seed(seed)
    fx = fy = 0.8 * W
    ox, oy = W / 2.0, H / 2.0
    intr = torch.tensor([fx, fy, ox, oy], dtype=torch.float32).repeat(B, 1).to(device)
    w_gt = (torch.rand(B, 3, generator=g) * 2 - 1) * rot_mag
    t_gt = (torch.rand(B, 3, generator=g) * 2 - 1) * trans_mag
    Wc_gt = torch.randn(B, max(K, 1), generator=g)[:, :K] * 0.08 / math.sqrt(max(K, 1))
    R_gt = torch.stack([_rodrigues(w_gt[b]) for b in range(B)])
    levels = []
    for s in scales:
        Hl, Wl = H // s, W // s
        vv, uu = torch.meshgrid(torch.arange(Hl, dtype=torch.float32, device=device) * s,
                                torch.arange(Wl, dtype=torch.float32, device=device) * s, indexing="ij")
        basis1 = _dct_basis(uu, vv, W, H, K) if K > 0 else None
        D0 = _depth0(uu, vv, W, H)
        ray = torch.stack([(uu - ox) / fx, (vv - oy) / fy, torch.ones_like(uu)], dim=-1)
        if normalize_rays:
            ray = ray / torch.linalg.norm(ray, dim=-1, keepdim=True)
        src = torch.empty(B, Hl, Wl, C, dtype=torch.float32, device=device)
        tgt = torch.empty(B, Hl, Wl, C, dtype=torch.float32, device=device)
        for b in range(B):
            fld = _field_params(C, seed * 1000 + 17 + b, device)
            Dgt = D0 + (basis1 @ Wc_gt[b].to(device) if K > 0 else 0.0)
            X = (ray * Dgt[..., None]) @ R_gt[b].to(device).T + t_gt[b].to(device)
            pu = fx * X[..., 0] / X[..., 2] + ox
            pv = fy * X[..., 1] / X[..., 2] + oy
            src[b] = _eval_field(fld, pu, pv)
            tgt[b] = _eval_field(fld, uu, vv)
            if noise > 0:
                tgt[b] += torch.randn(Hl, Wl, C, generator=g).to(device) * noise if Hl * Wl * C < (1 << 22) else \
                    torch.randn(Hl, Wl, C, device=device) * noise
        depth = D0[None].repeat(B, 1, 1).contiguous()
        basis = basis1[None].repeat(B, 1, 1, 1).contiguous() if K > 0 else None
        levels.append(DenseLevel(s, src, tgt, depth, basis))
    return intr, levels, dict(R=R_gt, T=t_gt, W=Wc_gt)


def make_multiframe_windows(B, H, W, C, K, scales, seed, device, pairs, normalize_rays=True, rot_mag=0.012,
                            trans_mag=0.03, noise=0.01):
    """Same construction as make_window_scene in oracle/synth.py: key-frame features = the analytic field on
    the grid, target frame i = the field at the inverse GT warp (fixed-point iteration), tgt [B,pairs,H,W,C].
    gt: R [B,pairs,3,3], T [B,pairs,3], W [B,K]."""
    g = torch.Generator().manual_seed(seed)
    fx = fy = 0.8 * W
    ox, oy = W / 2.0, H / 2.0
    intr = torch.tensor([fx, fy, ox, oy], dtype=torch.float32).repeat(B, 1).to(device)
    w_gt = (torch.rand(B, pairs, 3, generator=g) * 2 - 1) * rot_mag
    t_gt = (torch.rand(B, pairs, 3, generator=g) * 2 - 1) * trans_mag
    Wc_gt = torch.randn(B, max(K, 1), generator=g)[:, :K] * 0.08 / math.sqrt(max(K, 1))
    R_gt = torch.stack([torch.stack([_rodrigues(w_gt[b, i]) for i in range(pairs)]) for b in range(B)])
    levels = []
    for s in scales:
        Hl, Wl = H // s, W // s
        vv, uu = torch.meshgrid(torch.arange(Hl, dtype=torch.float32, device=device) * s,
                                torch.arange(Wl, dtype=torch.float32, device=device) * s, indexing="ij")
        basis1 = _dct_basis(uu, vv, W, H, K) if K > 0 else None
        D0 = _depth0(uu, vv, W, H)
        src = torch.empty(B, Hl, Wl, C, dtype=torch.float32, device=device)
        tgt = torch.empty(B, pairs, Hl, Wl, C, dtype=torch.float32, device=device)
        for b in range(B):
            fld = _field_params(C, seed * 1000 + 17 + b, device)
            src[b] = _eval_field(fld, uu, vv)
            wb = Wc_gt[b].to(device)
            for i in range(pairs):
                Rm, tv = R_gt[b, i].to(device), t_gt[b, i].to(device)
                u, v = uu.clone(), vv.clone()
                for _ in range(8):
                    Dg = _depth0(u, v, W, H) + (_dct_basis(u, v, W, H, K) @ wb if K > 0 else 0.0)
                    ray = torch.stack([(u - ox) / fx, (v - oy) / fy, torch.ones_like(u)], dim=-1)
                    if normalize_rays:
                        ray = ray / torch.linalg.norm(ray, dim=-1, keepdim=True)
                    X = (ray * Dg[..., None]) @ Rm.T + tv
                    u = u + (uu - (fx * X[..., 0] / X[..., 2] + ox))
                    v = v + (vv - (fy * X[..., 1] / X[..., 2] + oy))
                tgt[b, i] = _eval_field(fld, u, v)
                if noise > 0:
                    tgt[b, i] += torch.randn(Hl, Wl, C, device=device) * noise
        depth = D0[None].repeat(B, 1, 1).contiguous()
        basis = basis1[None].repeat(B, 1, 1, 1).contiguous() if K > 0 else None
        levels.append(DenseLevel(s, src, tgt, depth, basis))
    return intr, levels, dict(R=R_gt, T=t_gt, W=Wc_gt)
