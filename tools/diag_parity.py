#!/usr/bin/env python3
"""Diagnostic: bench.py's window 0, one iteration from the oracle's level-start state at the two finest levels --
GPU vs float32 oracle vs float64 oracle, per coefficient group (pose / damped depth / the undamped last coefficient)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
import bench
from banet_amd import dense as bdense
from oracle import banet_oracle as orc, dense as odense

dev = torch.device("cuda", 0)
prob = bench.Problem(32, 2, bench.H, bench.W, bench.K, 1234, dev)
lv1 = [bdense.DenseLevel(l.scale, l.src[0:1].contiguous(), l.tgt[0:1].contiguous(), l.depth[0:1].contiguous(),
                         l.basis[0:1].contiguous()) for l in prob.levels]
ba1 = bdense.DenseBA(prob.intr[0:1].contiguous(), lv1, prob.mlps, "bundle", 1000.0)
intr = prob.intr[0:1].cpu().numpy()
nlv = [dict(scale=l.scale, H=l.H, W=l.W, src=l.src.cpu().numpy(), tgt=l.tgt.cpu().numpy(), D0=l.depth.cpu().numpy(),
            basis=l.basis.cpu().numpy()) for l in lv1]
mlps = [[(np.asarray(w.cpu()), np.asarray(b.cpu())) for w, b in lw] for lw in prob.mlps]
R0 = np.eye(3, dtype=np.float32)[None]
T0 = prob.T0[0:1].cpu().numpy().reshape(1, 3, 1)
W0 = np.zeros((1, bench.K, 1), np.float32)
ref, _ = odense.bundle_chain(intr, nlv, mlps, bench.CHAIN_ITERS, R0, T0, W0)
for li in (2, 3, 4):
    r = ref[li]
    s1 = ba1.step_from(li, torch.from_numpy(r["R_start"]).to(dev), torch.from_numpy(r["T_start"]).to(dev), torch.from_numpy(r["W_start"]).to(dev))
    g = s1.delta.cpu().numpy()[0].astype(np.float64)
    o32 = r["first_delta"][0].astype(np.float64)
    a = odense.level_inputs(intr, nlv[li], True, np.float64)
    _, _, _, dbg = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"],
                                        r["R_start"].astype(np.float64), r["T_start"].astype(np.float64), r["W_start"].astype(np.float64),
                                        mlps[li], 1000.0, eq=orc.equation_construction_gemm)
    o64 = dbg["solution"][0, :, 0]
    print("level", li, "lam gpu/o32/o64", float(s1.lambda_out[0]), float(r["first_lam"][0]), float(np.asarray(dbg["lam"]).reshape(-1)[0]))
    for name, sl in (("pose", slice(0, 6)), ("depth damped", slice(6, -1)), ("last", slice(-1, None))):
        sc = np.abs(o64[sl]).max()
        print("  %-13s scale %.3e | gpu-o64 %.3e  o32-o64 %.3e  gpu-o32 %.3e (relative to the group's max)" % (
            name, sc, np.abs(g[sl] - o64[sl]).max() / sc, np.abs(o32[sl] - o64[sl]).max() / sc, np.abs(g[sl] - o32[sl]).max() / sc))
    print("  |W_start| max %.3e  last W %.3e" % (np.abs(r["W_start"]).max(), r["W_start"][0, -1, 0]))
