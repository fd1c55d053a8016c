#!/usr/bin/env python3
"""Cycle breakdown of ba_syrk_kernel per tile (build with EXTRA_HIPCC_FLAGS=-DBANET_TIMING)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import _capi as capi, dense as bdense, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
B, H, W, C, K = 4, 480, 640, 128, 128
dev = torch.device("cuda:0")
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 5, dev, trans_mag=0.06)
ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(C, 1)], "bundle", 1000.0)
p = ba.problems[0]
L = capi.lib()
R = torch.eye(3, device=dev).repeat(B, 1, 1); T = (gt["T"] * 0.7).reshape(B, 3, 1).to(dev); Wc = torch.zeros(B, K, 1, device=dev)
P = 6 + K
AtA = torch.empty(B, P, P, device=dev); Atb = torch.empty(B, P, device=dev); absres = torch.empty(B, C, device=dev); nvalid = torch.empty(B, device=dev)
nb = L.banet_ba_assemble_workspace_bytes(ctypes.byref(p.c)); ws = capi.workspace(nb, dev)
for _ in range(2):
    capi.check(L.banet_ba_assemble_f32(ctypes.byref(p.c), capi.ptr(R), capi.ptr(T), capi.ptr(Wc), capi.ptr(AtA), capi.ptr(Atb),
                                       capi.ptr(absres), capi.ptr(nvalid), ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
torch.cuda.synchronize()
groups = (80 * 60 + 3) // 4
G = min(groups // 2, (1024 + B - 1) // B) & ~7
gbytes = ((B * G * (32 + C) * 4 + 255) // 256) * 256
N = H * W
rec = ws[gbytes:gbytes + B * N * 8 * 4].view(torch.float32).reshape(B, N, 8)
tiles = (N + 63) // 64
Gs = min(tiles // 4, (512 + B - 1) // B)
d = rec[:, :Gs, :4].reshape(-1, 4).cpu()
per = tiles / Gs
for i, nme in enumerate(["issue next-tile loads", "H_cd (VALU)", "H_dd (MFMA)", "wait loads+LDS store+barrier"]):
    print("%-30s %9.0f cycles/WG  %8.0f cycles/tile" % (nme, d[:, i].mean(), d[:, i].mean() / per))
print("tiles per WG %.1f, Gs=%d" % (per, Gs))
