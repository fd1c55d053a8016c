import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["x", "8", "480", "640", "2"]
exec(open(os.path.join(ROOT, "tools", "bench_dense_train.py")).read().split("f_ms, _ = timed(fwd_only, 3)")[0])
fwd_bwd(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    fwd_bwd(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
# coarse wall-clock split
import banet_amd.dense_train as dt
t0 = time.perf_counter(); R, T, Wc = ba.solve_differentiable(iters, T=T0); torch.cuda.synchronize(); t1 = time.perf_counter()
loss = R.sum() + T.sum() + Wc.sum(); g = torch.autograd.grad(loss, leaves); torch.cuda.synchronize(); t2 = time.perf_counter()
print("forward (training mode) %.1f ms, backward %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
