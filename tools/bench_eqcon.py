#!/usr/bin/env python3
"""The literal op boundary: banet_equation_construction_f32 / _grad_f32 (the reference's EquationConstruction and
EquationConstructionGrad, utils.cu:150-171,420-428) -- time and HBM bandwidth on the bytes the op must move."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import ops
dev = torch.device("cuda:0")
C = 128
SHAPES = ((8, 4096, 134), (8, 76800, 134), (2, 307200, 134), (8, 76800, 6), (2, 76800, 262), (8, 76800, 262), (8, 76800, 200))
if os.environ.get("EQ_SHAPES"):      # e.g. EQ_SHAPES=8x76800x262,8x76800x134 (under rocprofv3: one shape per run keeps the stats readable)
    SHAPES = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["EQ_SHAPES"].split(","))
for B, N, P in SHAPES:
    g = torch.Generator().manual_seed(1)
    J = torch.randn(B, N, 2, P, generator=g).to(dev)
    G = torch.randn(B, N, C, 2, generator=g).to(dev)
    d = torch.randn(B, N, C, 1, generator=g).to(dev)
    g0 = torch.randn(B, P, P, generator=g).to(dev)
    g0 = g0 + g0.transpose(1, 2)
    g1 = torch.randn(B, P, 1, generator=g).to(dev)
    res = {}
    for name, fn in (("forward", lambda: ops.equation_construction_forward(J, G, d)),
                     ("grad", lambda: ops.equation_construction_grad(J, G, d, g0, g1))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    fbytes = 4.0 * B * N * (2 * P + 3 * C)             # read J, G, d once
    gbytes = 2 * fbytes                                 # read them, write their gradients
    print("B=%d N=%d C=%d P=%d: forward %.3f ms = %.0f GB/s of J+G+d (%.2f of 8 TB/s); grad %.3f ms = %.0f GB/s (%.2f)" % (
        B, N, C, P, res["forward"], fbytes / res["forward"] / 1e6, fbytes / res["forward"] / 8e9,
        res["grad"], gbytes / res["grad"] / 1e6, gbytes / res["grad"] / 8e9))
