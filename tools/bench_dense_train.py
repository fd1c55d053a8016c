#!/usr/bin/env python3
"""Training step of the dense BA layer: DenseBA.solve_differentiable (fused forward kernels, fused backward kernels of
csrc/adjoint.hip) -- forward-only solve vs forward + backward, per-kernel share of the backward, peak memory; and, at a size
the reference-layout graph can hold, the same step through BundleNet's lean training graph (ops.sample_stats + torch) on all
pixels as points.   python tools/bench_dense_train.py [B] [H] [W] [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import dense as bdense, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
IT = int(sys.argv[4]) if len(sys.argv) > 4 else 2
C = K = 128
FRAMES = int(os.environ.get("PFRAMES", "2"))
SCALES = [16, 8, 4, 2, 1]
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, SCALES, 7, dev, trans_mag=0.06, pairs=FRAMES - 1)
mlps = [[(w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)) for w, b in he_normal_lambda_weights(C, 100 + i)]
        for i in range(len(SCALES))]
for lv in levels:
    for name in ("src", "tgt", "depth", "basis"):
        setattr(lv, name, getattr(lv, name).requires_grad_(True))
ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1000.0)
for prob in ba.problems:                       # PBITS: banet_level_t.reserved_ bits for every level (A/B switches, e.g. 67108864 =
    prob.c.flags = int(os.environ.get("PBITS", "0"))   # bit 26 = the fp32-MFMA form of the backward's GEMM-shaped piece)
T0 = (gt["T"] * 0.7).reshape(B * (FRAMES - 1), 3, 1).to(dev)
iters = [IT] * len(SCALES)
leaves = [getattr(lv, n) for lv in levels for n in ("src", "tgt", "depth", "basis")] + [x for lw in mlps for wb in lw for x in wb]


def fwd_only():
    st = ba.new_state(T=T0)
    ba.solve(iters, st)


def fwd_bwd():
    R, T, Wc = ba.solve_differentiable(iters, T=T0)
    loss = R.sum() + T.sum() + Wc.sum()
    return torch.autograd.grad(loss, leaves)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, (torch.cuda.max_memory_allocated() - base) / 2 ** 30


f_ms, _ = timed(fwd_only, 3)
t_ms, mem = timed(fwd_bwd, 3)
nit = sum(iters) * B
print("dense BA training step, %d windows of %d frames %dx%d, 5 levels x %d iterations, C = K = 128:" % (B, FRAMES, W, H, IT))
print("  forward only (lm_level)                 %8.2f ms   (%.1f LM it/s)" % (f_ms, nit / f_ms * 1e3))
print("  forward + backward (solve_differentiable) %6.2f ms   (%.1f LM it/s, %.2fx the forward), peak extra memory %.2f GB"
      % (t_ms, nit / t_ms * 1e3, t_ms / f_ms, mem))
from banet_amd import dense_train
modes = dense_train.small_step_modes()       # empty when the HIP small step ran (smallstep.hip: four launches, no torch graph)
print("  small backward step (lambda MLP / damping / solve / update adjoint on [B,P,P]):",
      modes if modes else "HIP (banet_small_step_adjoint_f32)")
