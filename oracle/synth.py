"""Seeded analytic synthetic scenes for the BA hot path  --  TEST INFRASTRUCTURE (numpy).

A scene is a pair (source/key frame, target frame) of C-channel feature pyramids with a
known relative pose and a known depth = D0 + basis.W_gt, built so that feature-metric BA
has a well-defined minimum at the ground truth:

  * target features  F2_l(u,v)  = field(s*u, s*v)                 (analytic smooth field)
  * source features  F1_l(u,v)  = field(pi(R_gt * ray(s*u,s*v) * D_gt(s*u,s*v) + T_gt))
  * pyramid level l has scale s = 2^l; coordinates and intrinsics are divided by s with
    no half-pixel offset, exactly as the reference does (legacy/ba.py:108-113,
    bundlenet.py:378-383).
"""
import numpy as np


def make_field(C, seed, wl_min=48.0, wl_max=320.0, nwaves=8):
    rng = np.random.RandomState(seed)
    amp = rng.uniform(0.5, 1.5, (C, nwaves)) / np.sqrt(nwaves)
    wl = np.exp(rng.uniform(np.log(wl_min), np.log(wl_max), (C, nwaves)))
    ang = rng.uniform(0, 2 * np.pi, (C, nwaves))
    kx = 2 * np.pi / wl * np.cos(ang)
    ky = 2 * np.pi / wl * np.sin(ang)
    phi = rng.uniform(0, 2 * np.pi, (C, nwaves))
    return dict(amp=amp, kx=kx, ky=ky, phi=phi)


def eval_field(f, u, v):
    """u,v [...]  (float64) -> [..., C] float64."""
    ph = u[..., None, None] * f["kx"] + v[..., None, None] * f["ky"] + f["phi"]
    return np.sum(f["amp"] * np.sin(ph), axis=-1)


def depth0(u, v, W, H):
    return 2.8 + 0.5 * np.sin(2 * np.pi * u / W * 1.3 + 0.4) * np.cos(2 * np.pi * v / H * 0.9 + 0.2) \
        + 0.3 * np.cos(2 * np.pi * (u / W + v / H) * 0.8)


def dct_basis(u, v, W, H, K):
    """K smooth, mutually near-orthogonal basis maps (low-frequency cosine products),
    unit RMS.  -> [..., K]."""
    out = []
    a = 0
    # enumerate (i,j) by increasing i+j, skipping the constant (0,0) term
    order = sorted(((i + j, i, j) for i in range(32) for j in range(32) if i + j > 0))
    for _, i, j in order[:K]:
        ci = np.cos(np.pi * (u + 0.5) * i / W) * (np.sqrt(2.0) if i else 1.0)
        cj = np.cos(np.pi * (v + 0.5) * j / H) * (np.sqrt(2.0) if j else 1.0)
        out.append(ci * cj)
        a += 1
    return np.stack(out, axis=-1)


def rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def make_pair_scene(H, W, C, K, levels, seed, normalize_rays, w_gt=None, t_gt=None,
                    Wc_gt=None, noise=0.0, dtype=np.float32):
    """Dense 2-frame scene.  levels: list of scales (e.g. [4,2,1], coarse -> fine).
    Returns dict(intr [4], R_gt, T_gt, W_gt [K], levels=[dict(scale,H,W,src,tgt,D0,basis)])."""
    rng = np.random.RandomState(seed)
    field = make_field(C, seed + 17)
    fx = fy = 0.8 * W
    ox, oy = W / 2.0, H / 2.0
    if w_gt is None:
        w_gt = rng.uniform(-1, 1, 3) * 0.012
    if t_gt is None:
        t_gt = rng.uniform(-1, 1, 3) * 0.03
    if Wc_gt is None:
        Wc_gt = rng.standard_normal(max(K, 1)) * 0.08 / np.sqrt(max(K, 1))
    Wc_gt = np.asarray(Wc_gt, np.float64)[:K]
    R = rodrigues(np.asarray(w_gt, np.float64))
    T = np.asarray(t_gt, np.float64)
    out_levels = []
    for s in levels:
        Hl, Wl = H // s, W // s
        vv, uu = np.meshgrid(np.arange(Hl, dtype=np.float64) * s, np.arange(Wl, dtype=np.float64) * s,
                             indexing="ij")
        basis = dct_basis(uu, vv, W, H, K) if K > 0 else np.zeros(uu.shape + (0,))
        D0 = depth0(uu, vv, W, H)
        Dgt = D0 + (basis @ Wc_gt if K > 0 else 0.0)
        ray = np.stack([(uu - ox) / fx, (vv - oy) / fy, np.ones_like(uu)], axis=-1)
        if normalize_rays:
            ray = ray / np.linalg.norm(ray, axis=-1, keepdims=True)
        X = (ray * Dgt[..., None]) @ R.T + T
        pu = fx * X[..., 0] / X[..., 2] + ox
        pv = fy * X[..., 1] / X[..., 2] + oy
        src = eval_field(field, pu, pv)
        tgt = eval_field(field, uu, vv)
        if noise > 0:
            tgt = tgt + rng.standard_normal(tgt.shape) * noise
        out_levels.append(dict(scale=s, H=Hl, W=Wl, src=src.astype(dtype), tgt=tgt.astype(dtype),
                               D0=D0.astype(dtype), basis=basis.astype(dtype)))
    return dict(intr=np.array([fx, fy, ox, oy], dtype), R_gt=R, T_gt=T, W_gt=Wc_gt, levels=out_levels,
                H=H, W=W, C=C, K=K)


def make_window_scene(H, W, C, K, levels, seed, pairs, normalize_rays=True, rot_mag=0.012, trans_mag=0.03,
                      Wc_gt=None, noise=0.0, dtype=np.float32):
    """Dense multi-frame window (not in the reference): ONE key frame with depth
    D0 + basis.W_gt and `pairs` target frames with their own GT poses.  Key-frame features are the
    analytic field on the pixel grid; target frame i shows the same field moved by its pose:
    F2_i(u') = field(warp_i^-1(u')), with the inverse warp found by fixed-point iteration (the
    warps are near-identity, so it contracts by ~0.05 per step).
    Returns dict(intr, R_gt [pairs,3,3], T_gt [pairs,3], W_gt, levels=[dict(scale,H,W,src,tgt[pairs],D0,basis)])."""
    rng = np.random.RandomState(seed)
    field = make_field(C, seed + 17)
    fx = fy = 0.8 * W
    ox, oy = W / 2.0, H / 2.0
    w_gt = rng.uniform(-1, 1, (pairs, 3)) * rot_mag
    t_gt = rng.uniform(-1, 1, (pairs, 3)) * trans_mag
    if Wc_gt is None:
        Wc_gt = rng.standard_normal(max(K, 1)) * 0.08 / np.sqrt(max(K, 1))
    Wc_gt = np.asarray(Wc_gt, np.float64)[:K]
    Rs = [rodrigues(w_gt[i]) for i in range(pairs)]

    def warp(i, u, v):  # key-frame pixel (full-res coordinates) -> target-frame pixel
        basis = dct_basis(u, v, W, H, K) if K > 0 else None
        Dgt = depth0(u, v, W, H) + (basis @ Wc_gt if K > 0 else 0.0)
        ray = np.stack([(u - ox) / fx, (v - oy) / fy, np.ones_like(u)], axis=-1)
        if normalize_rays:
            ray = ray / np.linalg.norm(ray, axis=-1, keepdims=True)
        X = (ray * Dgt[..., None]) @ Rs[i].T + t_gt[i]
        return fx * X[..., 0] / X[..., 2] + ox, fy * X[..., 1] / X[..., 2] + oy

    out_levels = []
    for s in levels:
        Hl, Wl = H // s, W // s
        vv, uu = np.meshgrid(np.arange(Hl, dtype=np.float64) * s, np.arange(Wl, dtype=np.float64) * s, indexing="ij")
        basis = dct_basis(uu, vv, W, H, K) if K > 0 else np.zeros(uu.shape + (0,))
        D0 = depth0(uu, vv, W, H)
        src = eval_field(field, uu, vv)
        tgts = []
        for i in range(pairs):
            u, v = uu.copy(), vv.copy()
            for _ in range(14):
                pu, pv = warp(i, u, v)
                u, v = u + (uu - pu), v + (vv - pv)
            t = eval_field(field, u, v)
            if noise > 0:
                t = t + rng.standard_normal(t.shape) * noise
            tgts.append(t.astype(dtype))
        out_levels.append(dict(scale=s, H=Hl, W=Wl, src=src.astype(dtype), tgt=np.stack(tgts), D0=D0.astype(dtype),
                               basis=basis.astype(dtype)))
    return dict(intr=np.array([fx, fy, ox, oy], dtype), R_gt=np.stack(Rs), T_gt=t_gt, W_gt=Wc_gt, levels=out_levels,
                H=H, W=W, C=C, K=K, pairs=pairs)


def make_plane_sequence(H, W, C, poses, seed, levels=(4, 2, 1), plane_n=(0.05, -0.03, 1.0), plane_d=3.0, dtype=np.float32):
    """A camera moving in front of a textured plane: every frame pair is exactly consistent, so any frame
    can serve as key frame (what the sequence driver needs).  poses: list of (w [3], t [3]) mapping frame-0
    coordinates to frame-i coordinates (frame 0 = identity).  Returns dict(intr [1,4,1], frames =
    [pyramid per frame: list of [1,H_l,W_l,C]], depths = [z-depth map [H,W] per frame], images = [H,W,3])."""
    field = make_field(C, seed + 17, wl_min=24.0, wl_max=160.0)
    fx = fy = 0.8 * W
    ox, oy = W / 2.0, H / 2.0
    n = np.asarray(plane_n, np.float64)
    n = n / np.linalg.norm(n)
    frames, depths, images = [], [], []
    for (w, t) in poses:
        R, t = rodrigues(np.asarray(w, np.float64)), np.asarray(t, np.float64)
        Rn = R @ n

        def sample(u, v):
            ray = np.stack([(u - ox) / fx, (v - oy) / fy, np.ones_like(u)], axis=-1)
            lam = (plane_d + Rn @ t) / (ray @ Rn)
            X0 = (ray * lam[..., None] - t) @ R          # R^T (X_i - t)
            return fx * X0[..., 0] / X0[..., 2] + ox, fy * X0[..., 1] / X0[..., 2] + oy, lam

        pyr = []
        for s in levels:
            vv, uu = np.meshgrid(np.arange(H // s, dtype=np.float64) * s, np.arange(W // s, dtype=np.float64) * s, indexing="ij")
            pu, pv, _ = sample(uu, vv)
            pyr.append(eval_field(field, pu, pv).astype(dtype)[None])
        vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
        pu, pv, lam = sample(uu, vv)
        frames.append(pyr)
        depths.append(lam.astype(dtype))
        img = eval_field(field, pu, pv)[..., :3]
        images.append((127.5 + 100.0 * img).astype(dtype))
    return dict(intr=np.array([fx, fy, ox, oy], dtype).reshape(1, 4, 1), frames=frames, depths=depths, images=images)
