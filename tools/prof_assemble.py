#!/usr/bin/env python3
"""Ablation timing of the fused assembly kernel at L0 (640x480, C=K=128): which phase costs what.
Uses the reserved_ field of banet_level_t as debug bits (see assemble.hip `dbg`)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import dense as bdense, ops, synth as bsynth  # noqa: E402
from banet_amd.bundlenet import he_normal_lambda_weights  # noqa: E402

B = int(os.environ.get("PB", "4"))
H, W = int(os.environ.get("PH", "480")), int(os.environ.get("PW", "640"))
C, K = 128, int(os.environ.get("PK", "128"))
only = os.environ.get("PONLY")
PP = int(os.environ.get("PP", "1"))   # target frames per window (pairs)
dev = torch.device("cuda:0")
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 5, dev, trans_mag=0.06, pairs=PP)
variant = "bundle" if K > 0 else "bundle_camera"
ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(C, 1)], variant, 1000.0)
p = ba.problems[0]
R = torch.eye(3, device=dev).repeat(B, 1, 1) if PP == 1 else torch.eye(3, device=dev).repeat(B, PP, 1, 1)
T = (gt["T"] * 0.7).reshape(B, 3, 1).to(dev) if PP == 1 else (gt["T"] * 0.7).reshape(B, PP, 3, 1).to(dev)
Wc = torch.zeros(B, max(K, 1), 1, device=dev)[:, :K]
byts = ba.algorithmic_bytes_per_iteration(0) * B
ALL = ((0, "full"), (1 << 25, "quad gather forced"), (524288, "no strip gather"), (262144 | 1024, "strip gather, 8-row segments forced"), (262144, "strip gather, 16-row segments forced"), (1 << 30, "no quad gather"), (4194304, "strip: frames looped in a wave"), (1 << 24, "fp16 two-piece SYRK forced"), (1024, "quarter tiles forced"), (512, "patch kernel forced"), (64, "direct gather kernel"), (128, "patch kernel, direct loads only"), (256, "fp32-MFMA syrk"), (16, "no quarter tiles"), (32, "generic kernel"), (1, "all taps -> texel(1,1)"), (4, "no depth dot"), (8, "no gather loop"), (12, "geometry only"), (8192, "patch gather at ONE workgroup per CU"), (16384, "patch gather, 4-step units"), (65536, "patch gather, packed patch + flat loads"), (131072, "patch gather, column-major unit order"), (2, "P: src rows -> pixels 0..3"), (3, "P: taps+src fixed"), (7, "P: taps+src fixed, no depth dot"), (15, "P: no loads but records, no tap math"), (9, "P: fixed taps, no tap math"), (5, "P: fixed taps, no depth dot"))
sel = [int(x) for x in os.environ.get("PBITS", "0").split(",")]
cfgs = [a for a in ALL if a[0] in sel]
res = {bits: [] for bits, _ in cfgs}
ROUNDS = int(os.environ.get("PROUNDS", "3"))
for rnd in range(ROUNDS):                      # round-robin over the configurations: warm-up / clock drift hits all alike
    for bits, name in cfgs:
        p.c.flags = bits
        for _ in range(2):
            ops.ba_assemble(p, R, T, Wc if K else None)
        torch.cuda.synchronize()
        ops.profile_begin(4 * int(os.environ.get("PN", "5")) + 64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = int(os.environ.get("PN", "5"))   # launches per timed burst (PN=50: sustained load, power-limited clocks)
        e0.record()
        for _ in range(n):
            ops.ba_assemble(p, R, T, Wc if K else None)
        e1.record()
        torch.cuda.synchronize()
        prof = ops.profile_end()
        res[bits].append((e0.elapsed_time(e1) / n, {("gather" if k > 0 else "syrk"): 1e3 * v[1] / v[0] / B for k, v in prof.items()}))
for bits, name in cfgs:
    ms = min(r[0] for r in res[bits])
    ker = {k: round(min(r[1][k] for r in res[bits]), 1) for k in res[bits][0][1]}
    print("%dx%d K=%d B=%d pairs=%d %-6s %8.1f us/launch  %7.1f us/window  %7.1f GB/s   kernels (best of %d): %s" % (
        W, H, K, B, PP, name, ms * 1e3, ms * 1e3 / B, byts / ms / 1e6, ROUNDS, ker))
ref = None
for bits, name in cfgs:      # the variants must agree to rounding
    p.c.flags = bits
    out = [t.clone() for t in ops.ba_assemble(p, R, T, Wc if K else None)]
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    else:
        print("  %-28s max rel diff vs first: %s" % (name, ["%.1e" % float((a - b).abs().max() / b.abs().max()) for a, b in zip(out, ref)]))
