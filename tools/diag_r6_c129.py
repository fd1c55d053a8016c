#!/usr/bin/env python3
"""dsrc of the dense adjoint vs the float64 statement at C = 129 / 130, several seeds: is 3e-2 a masked-pixel flip or a kernel defect?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import dense as odense, dense_adjoint as oadj
from test_gpu_dense_backward import _run_adjoint, _scene, n
for (H, W, C, K, seed) in [(37, 50, 129, 8, 12), (37, 50, 130, 8, 12), (37, 50, 128, 8, 12), (37, 50, 129, 8, 13), (37, 50, 65, 8, 12), (37, 50, 129, 16, 12)]:
    intr, levels, R, T, Wc, rng = _scene(H, W, C, K, seed)
    lv = levels[0]
    B, P = 2, 6 + K
    G = rng.standard_normal((B, P, P)); gb = rng.standard_normal((B, P, 1)); gabs = rng.standard_normal((B, 1, C)) * 0.1
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    want = oadj.assembly_adjoint(a, lv64["tgt"], f32(R), f32(T), f32(Wc), f32(G), f32(gb), f32(gabs) * H * W)
    got = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
    w = want["dsrc"]; g = n(got["dsrc"]).reshape(w.shape)
    e = np.abs(g - w)
    bad = np.argwhere(e.max(axis=-1) > 1e-3 * np.abs(w).max())
    print((H, W, C, K, seed), "dsrc err %.2e" % (e.max() / np.abs(w).max()), "bad pixels", len(bad), bad[:4].tolist(),
          "bad channels of the first", np.argwhere(e[tuple(bad[0])] > 1e-3 * np.abs(w).max()).ravel()[:8].tolist() if len(bad) else [])
