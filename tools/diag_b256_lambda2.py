#!/usr/bin/env python3
"""bench.py's B256 sweep problem itself: at the 160x120 level compare, for a few windows, absres / lambda of the GPU with
the float64 twin, from the GPU's own [1,1] chain state."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from banet_amd import ops
from oracle import torch_port
dev = torch.device("cuda:0")
B = int(os.environ.get("PB", "256"))
prob = bench.Problem(B, 2, 480, 640, 128, 4321, dev, 0, scales=[16, 8, 4])
ba = prob.ba
st = ba.new_state(T=prob.T0)
R, T, Wc = st.R.clone(), st.T.clone(), st.Wc.clone()
for li, lv in enumerate(prob.levels):
    AtA, Atb, absres, nvalid = ops.ba_assemble(ba.problems[li], R, T, Wc)
    s1 = ba.step_from(li, R.clone(), T.clone(), Wc.clone())
    mlp = [(w_.cpu().numpy(), b_.cpu().numpy()) for w_, b_ in prob.mlps[li]]
    print("level", lv.W, lv.H, "gather", ops.GATHER_KERNELS[ops.gather_selection(ba.problems[li])])
    for wdw in (0, 1, 2, B // 2, B - 1):
        w = slice(wdw, wdw + 1)
        A64, b64, ab64, nv64 = torch_port.window_assemble(prob.intr[w], lv.scale, lv.src[w], lv.tgt[w].unsqueeze(1), lv.depth[w], lv.basis[w], R[w].unsqueeze(1), T[w].unsqueeze(1), Wc[w])
        *_, d = torch_port.window_iteration(prob.intr[w], lv.scale, lv.src[w], lv.tgt[w].unsqueeze(1), lv.depth[w], lv.basis[w], R[w].unsqueeze(1), T[w].unsqueeze(1), Wc[w], mlp, 1000.0)
        rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
        avg_gpu = (absres[w].double() / (lv.H * lv.W)).unsqueeze(1).cpu()
        y = torch_port.lambda_mlp(avg_gpu, mlp)
        lam_gpu_abs = float((1000.0 * torch.sqrt((avg_gpu ** 2).sum(-1, keepdim=True)) ** (2.0 + y)).reshape(-1)[0])
        lg, l64 = float(s1.lambda_out[wdw]), float(d["lam"][0])
        print("  window %3d: absres %.2e AtA %.2e Atb %.2e nvalid %d/%d | lam gpu %.8g f64 %.8g rel %.2e | f64 MLP on GPU absres %.8g | y %.6f" % (
            wdw, rel(absres[wdw], ab64[0]), rel(AtA[wdw], A64[0]), rel(Atb[wdw], b64[0]), int(nvalid[wdw]), int(nv64.sum()), lg, l64, abs(lg - l64) / l64, lam_gpu_abs, float(y.reshape(-1)[0])))
    R, T, Wc = s1.R.clone(), s1.T.clone(), s1.Wc.clone()
