#!/usr/bin/env python3
"""Cycle breakdown of ba_solve_update_kernel (build with EXTRA_HIPCC_FLAGS=-DBANET_TIMING)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import dense as bdense, ops, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
B, H, W, C, K = 8, 120, 160, 128, 128
dev = torch.device("cuda:0")
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 5, dev, trans_mag=0.06)
ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(C, 1)], "bundle", 1000.0)
st = ba.new_state(T=(gt["T"] * 0.7).reshape(B, 3, 1).to(dev))
p = ba.problems[0]
out = ops.ba_assemble(p, st.R, st.T, st.Wc)
for _ in range(3):
    ops.ba_solve_update(p, ba.mlps[0], 1000.0, *out, st)
torch.cuda.synchronize()
d = st.delta[:, :8].cpu()
for i, n in enumerate(["avg+MLP", "accept+damp/load", "LU+backsub", "update", "  A diag blocks", "  B panel rows", "  C trailing", "  back-subst"]):
    print("%-18s %10.0f cycles  (%.1f us @2.3GHz)" % (n, d[:, i].mean(), d[:, i].mean() / 2300))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.ba_solve_update(p, ba.mlps[0], 1000.0, *out, st)
e1.record(); torch.cuda.synchronize()
print("solve launch: %.1f us" % (e0.elapsed_time(e1) * 50))
