#!/usr/bin/env python3
"""Random shapes for the fused backward's kernels (banet_dense_adjoint_f32 / banet_target_map_adjoint_f32) against the float64
adjoint statement (oracle/dense_adjoint.py): K from 1 to 256 (all three forms of the GEMM-shaped piece, one- and two-pixel-per-wave
kernels, 1-4 coefficients per lane), the pose-only variant, odd image sizes and channel counts.  `fuzz_backward.py [seed] [count]`."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dense_backward as tb
from oracle import dense as odense, dense_adjoint as oadj

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng0 = np.random.RandomState(seed0)
fails = 0
for it in range(count):
    H, W = int(rng0.randint(8, 40)), int(rng0.randint(8, 48))
    C = int(rng0.choice([3, 8, 16, 70, 128, 130, 192, 256]))
    camera = bool(rng0.rand() < 0.2)
    K = 4 if camera else int(rng0.choice([1, 5, 16, 33, 64, 100, 128, 129, 136, 144, 145, 160, 192, 193, 200, 255, 256]))
    seed = seed0 + it
    intr, levels, R, T, Wc, rng = tb._scene(H, W, C, K, seed)
    if camera:
        Wc = np.zeros((2, 0, 1))
    lv = tb._avoid_abs_kinks(intr, levels[0], R, T, Wc, camera=camera)
    B, P = 2, 6 + (0 if camera else K)
    G = rng.standard_normal((B, P, P))
    gb = rng.standard_normal((B, P, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    if camera:
        a["Bs"] = np.zeros(a["Bs"].shape[:2] + (0,))
    want = oadj.assembly_adjoint(a, lv64["tgt"], f32(R), f32(T), f32(Wc), f32(G), f32(gb), f32(gabs) * H * W)
    got = tb._run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, variant="bundle_camera" if camera else "bundle")
    errs = {}
    names = [("dsrc", want["dsrc"]), ("dmap3", want["dmap"]), ("dtgt", want["dtgt"]), ("ddepth", want["dD0"])]
    if not camera:
        names.append(("dbasis", want["dbasis"]))
    for name, w in names:
        g = tb.n(got[name]).reshape(w.shape)
        errs[name] = float(np.abs(g - w).max() / max(np.abs(w).max(), 1e-30))
    dpose = tb.n(got["dpose"])
    errs["dR"] = float(np.abs(dpose[:, :9] - want["dR"].reshape(B, 9)).max() / max(np.abs(want["dR"]).max(), 1e-30))
    errs["dT"] = float(np.abs(dpose[:, 9:12] - want["dT"].reshape(B, 3)).max() / max(np.abs(want["dT"]).max(), 1e-30))
    if not camera:
        errs["dW"] = float(np.abs(dpose[:, 12:] - want["dW"].reshape(B, K)).max() / max(np.abs(want["dW"]).max(), 1e-30))
    # round 6: the target-tile path (BANET_ADJOINT_FOLD_TARGET) with a random tile kernel / shape: its target gradient against the
    # float64 statement, every other output bit-equal to the row-gather path's
    shape = int(rng0.choice([0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13]))
    gf = tb._run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, variant="bundle_camera" if camera else "bundle", fold=True, tile=shape)
    errs["dtgt_tile%d" % shape] = float(np.abs(tb.n(gf["dtgt"]).reshape(want["dtgt"].shape) - want["dtgt"]).max() / max(np.abs(want["dtgt"]).max(), 1e-30))
    for k in ("dsrc", "ddepth", "dpose") + (() if camera else ("dbasis",)):
        if not torch.equal(gf[k], got[k]):
            errs["tile_vs_rows_" + k] = 1.0
    bad = {k: v for k, v in errs.items() if not (v < (5e-4 if k in ("dR", "dT", "dW") else 2e-4))}
    if bad:
        fails += 1
        print("FAIL", (H, W, C, K, camera, seed), {k: "%.2e" % v for k, v in bad.items()})
print("fused backward, %d random shapes from seed %d: %d failures" % (count, seed0, fails))
