#!/usr/bin/env python3
"""A/B of ba_gather128p_kernel (default on large levels; reserved_ bit 7 forces its direct-load fallback)
against ba_gather128_kernel (reserved_ bit 6)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import dense as bdense, ops, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
dev = torch.device("cuda:0")
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for (B, H, W, K, rot, tr, pairs) in [(2, 48, 64, 128, 0.012, 0.06, 1), (2, 37, 53, 16, 0.012, 0.06, 1), (1, 120, 160, 0, 0.012, 0.06, 1),
                                     (2, 64, 96, 64, 0.08, 0.4, 1), (2, 40, 56, 128, 0.012, 0.06, 3), (8, 30, 40, 128, 0.012, 0.06, 1), (8, 240, 320, 128, 0.05, 0.3, 1),
                                     (4, 480, 640, 128, 0.012, 0.06, 1)]:
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, 128, K, [1], 5, dev, rot_mag=rot, trans_mag=tr, pairs=pairs)
    ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(128, 1)], "bundle" if K else "bundle_camera", 1000.0)
    p = ba.problems[0]
    R = torch.eye(3, device=dev).repeat(B * pairs, 1, 1)
    T = (gt["T"] * 0.7).reshape(B * pairs, 3, 1).to(dev)
    Wc = torch.zeros(B, max(K, 1), 1, device=dev)[:, :K]
    outs = {}
    for bits in (64, 0, 128):
        p.c.flags = bits
        outs[bits] = [x.clone() for x in ops.ba_assemble(p, R, T, Wc if K else None)]
        again = ops.ba_assemble(p, R, T, Wc if K else None)
        assert all(torch.equal(x, y) for x, y in zip(outs[bits], again)), "not deterministic"
    for bits in (0, 128):
        e = [rel(x, y) for x, y in zip(outs[bits], outs[64])]
        print("%dx%d B=%d K=%d pairs=%d rot=%.3f bits=%3d  rel.diff AtA %.1e Atb %.1e absres %.1e nvalid %.1e" % (W, H, B, K, pairs, rot, bits, *e))
        assert max(e) < 2e-5, e
print("patch kernel == direct kernel (to rounding), deterministic")
