#!/usr/bin/env python3
"""Read the BANET_TIMING cycle counters (build with EXTRA_HIPCC_FLAGS=-DBANET_TIMING) of wave 0 of
each gather workgroup: where does a 64-pixel batch spend its cycles?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import _capi as capi, dense as bdense, synth as bsynth  # noqa: E402
from banet_amd.bundlenet import he_normal_lambda_weights  # noqa: E402

B, H, W, C, K = int(os.environ.get("PB", "4")), 480, 640, 128, int(os.environ.get("PK", "128"))
dev = torch.device("cuda:0")
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 5, dev, trans_mag=0.06)
ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(C, 1)], "bundle" if K else "bundle_camera", 1000.0)
p = ba.problems[0]
PATCH = os.environ.get("PPATCH", "1") == "1"   # 1: the patch kernel (ba_gather128p_kernel), 0: the direct ba_gather128_kernel
p.c.flags = 0 if PATCH else 64
L = capi.lib()
R = torch.eye(3, device=dev).repeat(B, 1, 1)
T = (gt["T"] * 0.7).reshape(B, 3, 1).to(dev)
Wc = torch.zeros(B, max(K, 1), 1, device=dev)[:, :K].contiguous()
P = 6 + K
AtA = torch.empty(B, P, P, device=dev); Atb = torch.empty(B, P, device=dev)
absres = torch.empty(B, C, device=dev); nvalid = torch.empty(B, device=dev)
nb = L.banet_ba_assemble_workspace_bytes(ctypes.byref(p.c))
ws = capi.workspace(nb, dev)
for _ in range(3):
    capi.check(L.banet_ba_assemble_f32(ctypes.byref(p.c), capi.ptr(R), capi.ptr(T), capi.ptr(Wc) if K else None,
                                       capi.ptr(AtA), capi.ptr(Atb), capi.ptr(absres), capi.ptr(nvalid),
                                       ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
torch.cuda.synchronize()
tiles = 80 * 60
part = ws[:B * tiles * (32 + C) * 4].view(torch.float32).reshape(B, tiles, 32 + C)
t = part[:, :, 28:32].reshape(-1, 4).cpu()
MODE2 = os.environ.get("PMODE", "1") == "2"   # the library was built with -DBANET_TIMING=2
names = (["unit loop", "rim + algebra + pose sums + records", "sum|d| fold + partial", "whole tile"] if PATCH and MODE2 else
         ["unit halves: wait + box -> LDS + next issue", "unit halves: taps", "depth dot + geometry", "whole tile"] if PATCH else
         ["16 gather steps", "rim+algebra+partials", "depth dot", "whole tile"])
for i, nme in enumerate(names):
    v = t[:, i]
    print("%-22s mean %9.0f cycles/tile  (p10 %9.0f  p90 %9.0f)  %7.1f cycles/step" % (
        nme, v.mean(), v.quantile(0.1), v.quantile(0.9), v.mean() / 16))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    capi.check(L.banet_ba_assemble_f32(ctypes.byref(p.c), capi.ptr(R), capi.ptr(T), capi.ptr(Wc) if K else None,
                                       capi.ptr(AtA), capi.ptr(Atb), capi.ptr(absres), capi.ptr(nvalid),
                                       ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
e1.record(); torch.cuda.synchronize()
print("assemble (gather+syrk+reduce) %.1f us per call" % (e0.elapsed_time(e1) * 100))
