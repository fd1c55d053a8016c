#!/usr/bin/env python3
"""cfg-2 solve of 8 windows: one batch on one stream vs S sub-batches on S concurrent HIP streams (does the solve kernel's
idle time -- 8 CUs busy out of 256 -- get filled by another sub-batch's gather?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import dense as bdense, ops, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
dev = torch.device("cuda:0")
B, H, W, C, K = int(os.environ.get("PB", "8")), 480, 640, 128, 128
SC, IT = [16, 8, 4, 2, 1], [10] * 5
mlps = [he_normal_lambda_weights(C, 100 + i) for i in range(5)]


def make(b, seed):
    intr, levels, gt = bsynth.make_dense_windows(b, H, W, C, K, SC, seed, dev, trans_mag=0.06)
    ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1000.0)
    st = ba.new_state(T=(gt["T"] * 0.7).reshape(b, 3, 1).to(dev))
    return ba, st, (st.R.clone(), st.T.clone(), st.Wc.clone())


def run(ba, st):
    for prob, mlp, its in zip(ba.problems, ba.mlps, IT):
        ops.lm_level(prob, mlp, ba.l2_base, its, False, st, ws=ba.ws)


def reset(st, init):
    st.R.copy_(init[0]); st.T.copy_(init[1]); st.Wc.copy_(init[2])


def timed(parts, streams, n=5):
    for it in range(2 + n):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        for (ba, st, init), s in zip(parts, streams):
            with torch.cuda.stream(s):
                reset(st, init)
                run(ba, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


one = [make(B, 1236)]
t1 = timed(one, [torch.cuda.current_stream()])
print("B=%d  one stream           %.2f ms/step  %.0f LM it/s" % (B, t1 * 1e3, B * 50 / t1))
for S in (2, 4):
    parts = [make(B // S, 1236 + i) for i in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    tS = timed(parts, streams)
    print("B=%d  %d streams x %d windows %.2f ms/step  %.0f LM it/s  (%+.1f%%)" % (B, S, B // S, tS * 1e3, B * 50 / tS, 100 * (t1 / tS - 1)))
