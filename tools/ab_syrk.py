#!/usr/bin/env python3
"""A/B of ba_syrk_bf16x6_kernel (default) against ba_syrk_direct_kernel (fp32 MFMA, reserved_ bit 8)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import dense as bdense, ops, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
dev = torch.device("cuda:0")
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for (B, H, W, K, pairs) in [(2, 48, 64, 128, 1), (2, 37, 53, 64, 1), (2, 40, 56, 128, 3), (1, 120, 160, 128, 4), (2, 480, 640, 128, 1)]:
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, 128, K, [1], 5, dev, trans_mag=0.06, pairs=pairs)
    ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(128, 1)], "bundle", 1000.0)
    p = ba.problems[0]
    R = torch.eye(3, device=dev).repeat(B * pairs, 1, 1)
    T = (gt["T"] * 0.7).reshape(B * pairs, 3, 1).to(dev)
    Wc = torch.zeros(B, K, 1, device=dev)
    outs = {}
    for bits in (0, 256):
        p.c.flags = bits
        outs[bits] = [x.clone() for x in ops.ba_assemble(p, R, T, Wc)]
        again = ops.ba_assemble(p, R, T, Wc)
        assert all(torch.equal(x, y) for x, y in zip(outs[bits], again)), "not deterministic"
    P6 = 6 * pairs
    e = [rel(x, y) for x, y in zip(outs[256], outs[0])]
    edd = rel(outs[256][0][:, P6:, P6:], outs[0][0][:, P6:, P6:])
    print("%dx%d B=%d K=%d pairs=%d: bf16x6 vs fp32 MFMA: AtA %.1e (H_dd %.1e) Atb %.1e" % (W, H, B, K, pairs, e[0], edd, e[1]))
    assert e[0] < 2e-6 and e[1] < 2e-6, e
print("bf16x6 == fp32 (to fp32 rounding), deterministic")
