#!/usr/bin/env python3
"""Cycle counters of ba_syrk_direct_kernel (build with EXTRA_HIPCC_FLAGS=-DBANET_TIMING): main-loop cycles
per quad (s_memtime), wall time (s_memrealtime, 100 MHz) -> effective clock and MFMA pipe utilisation."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import _capi as capi, dense as bdense, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
B, H, W, C, K = int(os.environ.get("PB", "8")), 480, 640, 128, 128
dev = torch.device("cuda:0")
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 5, dev, trans_mag=0.06)
ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(C, 1)], "bundle", 1000.0)
p = ba.problems[0]
BF = os.environ.get("PBF", "1") == "1"   # 1: the default bf16x6 kernel, 0: the fp32-MFMA ba_syrk_direct_kernel
p.c.flags = 0 if BF else 256
L = capi.lib()
R = torch.eye(3, device=dev).repeat(B, 1, 1)
T = (gt["T"] * 0.7).reshape(B, 3, 1).to(dev)
Wc = torch.zeros(B, K, 1, device=dev)
P = 6 + K
AtA = torch.empty(B, P, P, device=dev); Atb = torch.empty(B, P, device=dev)
absres = torch.empty(B, C, device=dev); nvalid = torch.empty(B, device=dev)
nb = L.banet_ba_assemble_workspace_bytes(ctypes.byref(p.c))
ws = capi.workspace(nb, dev)
for _ in range(3):
    capi.check(L.banet_ba_assemble_f32(ctypes.byref(p.c), capi.ptr(R), capi.ptr(T), capi.ptr(Wc), capi.ptr(AtA), capi.ptr(Atb),
                                       capi.ptr(absres), capi.ptr(nvalid), ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
torch.cuda.synchronize()
pstride = 7 * K + K * K
Gs = max(1, min((H * W + 1023) // 1024, (256 + B - 1) // B))
sp = ws.view(torch.float32)[-(B * Gs * pstride + 64):]   # syrk partials sit at the end of the workspace
# locate: search the first float equal to a plausible quad count
flat = ws.view(torch.float32)
off = int(os.environ.get("POFF", "-1"))
import numpy as np
arr = flat.cpu().numpy()
nq = (H * W // (32 if BF else 4)) // (Gs * 4)
cand = np.where((arr[2:] >= nq - 1) & (arr[2:] <= nq + 1))[0]
cand = [c for c in cand if arr[c] > 1e4 and arr[c + 1] > 1e2 and (c + pstride + 2 >= len(arr) or abs(arr[c + pstride + 2] - nq) <= 1)]
base = cand[0]
rows = np.stack([arr[base + i * pstride: base + i * pstride + 3] for i in range(B * Gs)])
cyc, rt, q = rows[:, 0], rows[:, 1], rows[:, 2]
mf = 216 * 16 + 64 * 32 if BF else 1408   # MFMA pipe cycles per unit (bf16x6: a 32-pixel step; fp32: a 4-pixel quad)
print("workgroups %d  units/wave %.0f  loop cycles %.0f (%.0f per unit; MFMA pipe %d cycles -> %.0f%% busy)" % (
    len(rows), q.mean(), cyc.mean(), (cyc / q).mean(), mf, 100 * mf / (cyc / q).mean()))
print("loop wall time %.1f us  -> shader clock %.2f GHz" % (rt.mean() / 100.0, cyc.mean() / (rt.mean() * 10.0) ))
