#!/usr/bin/env python3
"""cfg-2 solve: eager enqueue vs one captured HIP graph replayed (how much of a step is launch gaps)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import dense as bdense, ops, synth as bsynth
from banet_amd.bundlenet import he_normal_lambda_weights
dev = torch.device("cuda:0")
B, H, W, C, K = int(os.environ.get("PB", "8")), 480, 640, 128, 128
SC, IT = [16, 8, 4, 2, 1], [10] * 5
intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, SC, 1236, dev, trans_mag=0.06)
ba = bdense.DenseBA(intr, levels, [he_normal_lambda_weights(C, 100 + i) for i in range(5)], "bundle", 1000.0)
T0 = (gt["T"] * 0.7).reshape(B, 3, 1).to(dev)
st = ba.new_state(T=T0)
R0, Tc0, W0 = st.R.clone(), st.T.clone(), st.Wc.clone()
def run():
    for prob, mlp, its in zip(ba.problems, ba.mlps, IT):
        ops.lm_level(prob, mlp, ba.l2_base, its, False, st, ws=ba.ws)
def reset():
    st.R.copy_(R0); st.T.copy_(Tc0); st.Wc.copy_(W0)
for _ in range(2):
    reset(); run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    reset(); run()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 5
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    reset(); run()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
reset(); g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    reset(); g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 5
print("B=%d  eager %.2f ms/step   graph replay %.2f ms/step   (%.1f%%)" % (B, eager * 1e3, graph * 1e3, 100 * (eager - graph) / eager))
