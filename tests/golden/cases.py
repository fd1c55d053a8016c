"""Deterministic inputs for the golden-vector cases.

Shared by `make_golden.py` (which feeds them to the REFERENCE's Python running over
`oracle/tf1_shim`) and by the tests (which feed the same inputs to the oracle / the HIP
path and compare with the stored reference outputs in `golden_*.npz`).  Inputs are
regenerated from seeds; only the reference's outputs are committed.
"""
import os
import sys

import numpy as np

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import banet_oracle as orc  # noqa: E402
from oracle import synth  # noqa: E402

F32 = np.float32


def mlp_weights(C, levels, seed):
    """dict level-name -> 5 x (filters [Cin,Cout], bias [Cout])."""
    return {lv: orc.he_normal_mlp_weights(C, seed + 31 * i) for i, lv in enumerate(levels)}


def small_rotation(rng, mag):
    w = rng.uniform(-1, 1, 3) * mag
    return synth.rodrigues(w).astype(F32)[None]


def case_legacy_ci2(seed=101):
    """One CameraIteration2 call on random (not scene-consistent) data; B=1."""
    rng = np.random.RandomState(seed)
    H, W, C, N = 20, 28, 6, 200
    conv2_f = rng.standard_normal((1, H, W, C)).astype(F32)
    conv1 = rng.standard_normal((1, N, C)).astype(F32)
    intr = np.array([0.8 * W * 4, 0.8 * W * 4, W * 2.0, H * 2.0], F32).reshape(1, 4, 1)   # full-res
    pts = np.stack([rng.uniform(0, 4 * (W - 1), N), rng.uniform(0, 4 * (H - 1), N)], -1).astype(F32)[None]
    # a few points pushed outside / exactly onto the border
    pts[0, 0] = [0.0, 0.0]
    pts[0, 1] = [4 * (W - 1), 4 * (H - 1)]
    pts[0, 2] = [-3.0, 5.0]
    d = rng.uniform(1.0, 3.0, (1, N, 1)).astype(F32)
    R = small_rotation(rng, 0.02)
    T = (rng.uniform(-1, 1, (1, 3, 1)) * 0.05).astype(F32)
    return dict(conv1=conv1, conv2_f=conv2_f, intr=intr, points=pts, d=d, R=R, T=T, scale=4.0,
                mlp=mlp_weights(C, ["1"], seed), level="1")


def case_legacy_track(seed=202):
    """Tracker.trackTF on an analytic scene, 3 levels (scale 4,2,1), sparse points."""
    rng = np.random.RandomState(seed)
    H, W, C, N = 48, 64, 8, 300
    sc = synth.make_pair_scene(H, W, C, 0, [4, 2, 1], seed, normalize_rays=False,
                               w_gt=[0.010, -0.008, 0.006], t_gt=[0.02, -0.015, 0.01])
    layers = [np.stack([lv["src"], lv["tgt"]], 0) for lv in sc["levels"]]
    pts = np.stack([rng.uniform(2, W - 3, N), rng.uniform(2, H - 3, N)], -1)
    d = synth.depth0(pts[:, 0], pts[:, 1], W, H)
    return dict(layers=layers, intr=sc["intr"].reshape(1, 4, 1), points=pts.astype(F32)[None],
                d=d.astype(F32).reshape(1, N, 1), R=np.eye(3, dtype=F32)[None], T=np.zeros((1, 3, 1), F32),
                iters=[3, 5, 7], mlp=mlp_weights(C, ["1", "2", "3"], seed), R_gt=sc["R_gt"], T_gt=sc["T_gt"])


def case_bundle_fns(seed=303):
    rng = np.random.RandomState(seed)
    B, N = 2, 50
    x = rng.uniform(-0.5, 0.5, (B, N)).astype(F32)
    y = rng.uniform(-0.4, 0.4, (B, N)).astype(F32)
    Z = rng.uniform(1, 4, (B, N)).astype(F32)
    fx = np.full((B, N), 50.0, F32)
    fy = np.full((B, N), 48.0, F32)
    r = rng.uniform(-1, 1, (3, B, 1, N)).astype(F32)
    w1 = (rng.uniform(-1, 1, (1, 3)) * 0.2).astype(F32)          # B=1 for VMatrix
    w2 = (rng.uniform(-1, 1, (B, 3)) * 0.2).astype(F32)
    img = rng.standard_normal((B, 7, 9, 3)).astype(F32)
    pts = np.stack([rng.uniform(0, 30, (B, N)), rng.uniform(0, 20, (B, N))], -1).astype(F32)
    Rm = np.concatenate([small_rotation(rng, 0.5), small_rotation(rng, 0.3)], 0)
    return dict(x=x, y=y, Z=Z, fx=fx, fy=fy, r=r, w1=w1, w2=w2, img=img, pts=pts, Rm=Rm)


def case_bundle_iter(seed=404, B=1, K=5):
    """One BundleNet.CameraIteration / BundleIteration call; scene-consistent data so the
    system is well conditioned.  B=1 (VMatrix is only correct for B=1 in the reference)."""
    rng = np.random.RandomState(seed)
    H, W, C, N = 36, 48, 6, 400
    sc = synth.make_pair_scene(H, W, C, K, [1], seed, normalize_rays=True,
                               w_gt=[0.012, 0.009, -0.01], t_gt=[0.16, -0.12, 0.05])
    lv = sc["levels"][0]
    pts = np.stack([rng.uniform(1, W - 2, (B, N)), rng.uniform(1, H - 2, (B, N))], -1).astype(F32)
    pts[0, 0] = [0.0, 0.0]
    pts[0, 1] = [W - 1.0, H - 1.0]
    src = np.tile(lv["src"][None], (B, 1, 1, 1))
    tgt = np.tile(lv["tgt"][None], (B, 1, 1, 1))
    conv1 = orc.resampler(src, pts)
    conv2 = orc.target_map(tgt)
    D = orc.resampler(np.tile(lv["D0"][None, :, :, None], (B, 1, 1, 1)), pts)
    Bs = orc.resampler(np.tile(lv["basis"][None], (B, 1, 1, 1)), pts)
    intr = sc["intr"]
    fx = np.full((B, N), intr[0], F32)
    fy = np.full((B, N), intr[1], F32)
    ox = np.full((B, N), intr[2], F32)
    oy = np.full((B, N), intr[3], F32)
    p = orc.compute_coordinates(pts, fx, fy, ox, oy, normalize=True)
    R = np.tile(small_rotation(rng, 0.004), (B, 1, 1))
    T = (np.array([0.15, -0.11, 0.045]).reshape(1, 3, 1) + rng.uniform(-1, 1, (B, 3, 1)) * 0.004).astype(F32)
    Wc = (rng.standard_normal((B, K, 1)) * 0.01).astype(F32)
    return dict(conv1=conv1, conv2=conv2, fx=fx, fy=fy, ox=ox, oy=oy, p=p, D=D, Bs=Bs, R=R, T=T, W=Wc,
                mlp=mlp_weights(C, ["2"], seed), level="2")


def case_resize(seed=505, C=4, K=3, N=256):
    """BundleNet.CameraResize / BundleResize: B=2 (the two orderings of one pair), full-res
    256x320 (hard-coded in bundlenet.py:286-287,397), 4 pyramid levels scale 8,4,2,1."""
    rng = np.random.RandomState(seed)
    H, W = 256, 320
    sc = synth.make_pair_scene(H, W, C, K, [8, 4, 2, 1], seed, normalize_rays=True,
                               w_gt=[0.006, -0.004, 0.005], t_gt=[0.10, 0.05, -0.06])
    layers = [np.stack([lv["src"], lv["tgt"]], 0) for lv in sc["levels"]]
    half = sc["levels"][2]                                        # scale 2 == half resolution
    basis = np.tile(half["basis"][None], (2, 1, 1, 1))
    depth = np.tile(half["D0"][None, :, :, None], (2, 1, 1, 1))
    # points live in the un-cropped frame: x in [4,316], y in [4,236] maps onto the 320x256 maps
    pts = np.stack([rng.uniform(10, 310, (2, N)), rng.uniform(10, 230, (2, N))], -1).astype(F32)
    # choose the raw intrinsics so that the reference's crop adjustment (bundlenet.py:354-357)
    # lands on the scene's intrinsics
    fx, fy, ox, oy = [float(v) for v in sc["intr"]]
    raw = np.array([fx * 39.0 / 40.0, fy * 29.0 / 32.0, (ox + 160.0 / 39.0) * 39.0 / 40.0,
                    (oy + 128.0 / 29.0) * 29.0 / 32.0], F32)
    intr = np.tile(raw.reshape(1, 4, 1), (2, 1, 1))
    return dict(layers=layers, basis=basis, depth=depth, points=pts, intr=intr,
                mlp=mlp_weights(C, ["0", "1", "2", "3"], seed))


def case_losses(seed=606):
    """BundleNet.lossR / lossT / lossF (bundlenet.py:401-463): B = 2, a 12 x 16 depth map with a partial mask."""
    rng = np.random.RandomState(seed)
    B, H, W = 2, 12, 16
    q = rng.standard_normal((B, 4)).astype(F32)
    predQ = q / np.linalg.norm(q, axis=1, keepdims=True)
    q2 = q + 0.05 * rng.standard_normal((B, 4)).astype(F32)
    gtQ = (q2 / np.linalg.norm(q2, axis=1, keepdims=True)).astype(F32)
    predT = (rng.uniform(-1, 1, (B, 3)) * 0.1).astype(F32)
    gtT = (predT + rng.uniform(-1, 1, (B, 3)) * 0.01).astype(F32)
    predR = np.concatenate([small_rotation(rng, 0.03) for _ in range(B)], 0)
    gtR = np.concatenate([small_rotation(rng, 0.03) for _ in range(B)], 0)
    depth = rng.uniform(1.5, 3.0, (B, H, W, 1)).astype(F32)
    mask = (rng.uniform(0, 1, (B, H, W, 1)) > 0.3).astype(F32)
    intr = np.tile(np.array([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], F32).reshape(1, 4, 1), (B, 1, 1))
    return dict(predQ=predQ.astype(F32), gtQ=gtQ, predT=predT, gtT=gtT, predR=predR, gtR=gtR, depth=depth, mask=mask, intr=intr)
