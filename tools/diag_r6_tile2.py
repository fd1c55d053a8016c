#!/usr/bin/env python3
"""where does adj_tile2_kernel differ from the row-gather path?  per-texel error map"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_gpu_dense_backward import _run_adjoint, _scene, n
H, W, C, K, seed = [int(x) for x in (sys.argv[2:7] if len(sys.argv) > 6 else (16, 16, 64, 16, 2))]
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
intr, levels, R, T, Wc, rng = _scene(H, W, C, K, seed)
lv = levels[0]
B, P = 2, 6 + K
G = rng.standard_normal((B, P, P)); gb = rng.standard_normal((B, P, 1)); gabs = rng.standard_normal((B, 1, C)) * 0.1
got = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, fold=True, tile=tile)
old = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
e = (got["dtgt"] - old["dtgt"]).abs().amax(dim=-1).cpu().numpy() / float(old["dtgt"].abs().max())
np.set_printoptions(linewidth=250, precision=1, suppress=False)
for b in range(B):
    print("window", b)
    for y in range(H):
        print(" ".join("%5.0e" % v if v > 1e-4 else "  .  " for v in e[b, y]))
