#!/usr/bin/env python3
"""Read the BANET_TIMING cycle counters of the strip gather (ba_gather128s_kernel; build with
EXTRA_HIPCC_FLAGS=-DBANET_TIMING=1 or =2 into banet_amd/lib_timing{1,2}): where does a 16x32 segment spend its cycles?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import _capi as capi, dense as bdense, ops, synth as bsynth  # noqa: E402
from banet_amd.bundlenet import he_normal_lambda_weights  # noqa: E402

B, H, W, C, K = int(os.environ.get("PB", "32")), 480, 640, 128, int(os.environ.get("PK", "128"))
MODE = int(os.environ.get("PMODE", "1"))
PP = int(os.environ.get("PP", "1"))     # target frames per window (frame-parallel workgroups when > 1)
dev = torch.device("cuda:0")
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 5, dev, trans_mag=0.06, pairs=PP)
ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(C, 1)], "bundle" if K else "bundle_camera", 1000.0)
p = ba.problems[0]
p.c.flags = ops.FORCE_STRIP_GATHER
assert ops.gather_selection(p) == 3
L = capi.lib()
R = torch.eye(3, device=dev).repeat(B, 1, 1) if PP == 1 else torch.eye(3, device=dev).repeat(B, PP, 1, 1)
T = (gt["T"] * 0.7).reshape(B, 3, 1).to(dev) if PP == 1 else (gt["T"] * 0.7).reshape(B, PP, 3, 1).to(dev)
Wc = torch.zeros(B, max(K, 1), 1, device=dev)[:, :K].contiguous()
P = 6 * PP + K
AtA = torch.empty(B, P, P, device=dev); Atb = torch.empty(B, P, device=dev)
absres = torch.empty(B, C, device=dev); nvalid = torch.empty(B, device=dev)
nb = L.banet_ba_assemble_workspace_bytes(ctypes.byref(p.c))
ws = capi.workspace(nb, dev)


def run():
    capi.check(L.banet_ba_assemble_f32(ctypes.byref(p.c), capi.ptr(R), capi.ptr(T), capi.ptr(Wc) if K else None,
                                       capi.ptr(AtA), capi.ptr(Atb), capi.ptr(absres), capi.ptr(nvalid),
                                       ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))


for _ in range(3):
    run()
torch.cuda.synchronize()
SEGH = int(os.environ.get("PSEGH", "16"))
items = ((W + 15) // 16) * ((H + SEGH - 1) // SEGH)
part = ws[:B * PP * items * (32 + C) * 4].view(torch.float32).reshape(B * PP, items, 32 + C)
t = part[:, :, 28:32].reshape(-1, 4).cpu()
names = {1: ["parked in counted waits", "four slice passes", "rim + algebra + records + partial", "whole segment"],
         2: ["depth dot", "geometry + plan", "rim + algebra + records + partial", "whole segment"],
         3: ["workgroup barriers (FP) + queue pop", "depth dot + geometry + plan (no barrier)", "rim + algebra + records + partial", "whole segment incl. item barrier"],
         4: ["steps: start -> counted wait done", "steps: window reads (issue -> returned)", "four slice passes", "whole segment"],
         5: ["steps: channel maths (both pieces)", "steps: rest (reductions, loop)", "four slice passes", "whole segment"]}[MODE]
for i, nme in enumerate(names):
    v = t[:, i]
    print("%-36s mean %9.0f cycles/segment  (p10 %9.0f  p90 %9.0f)  %5.1f %%" % (
        nme, v.mean(), v.quantile(0.1), v.quantile(0.9), 100.0 * v.mean() / t[:, 3].mean()))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
print("assemble (gather+syrk+reduce) %.1f us per call, %d windows" % (e0.elapsed_time(e1) * 100, B))
