import os, sys, time
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from banet_amd import dense_train
B = 32
for spd in (True, False):
    dense_train.USE_SPD_SOLVE = spd
    dense_train._small_cache.clear()
    dev = torch.device("cuda:0")
    P, C, N = 134, 128, 307200
    g = torch.Generator().manual_seed(1)
    M = torch.randn(B, P, P + 20, generator=g)
    AtA = (M @ M.transpose(1, 2)).to(dev); Atb = torch.randn(B, P, generator=g).to(dev); absres = (torch.rand(B, C, generator=g) * N).to(dev)
    R = torch.eye(3).repeat(B, 1, 1).to(dev); T = torch.zeros(B, 3, 1).to(dev); W = torch.zeros(B, 128, 1).to(dev)
    gR = torch.randn(B, 3, 3, generator=g).to(dev); gT = torch.randn(B, 3, 1, generator=g).to(dev); gW = torch.randn(B, 128, 1, generator=g).to(dev)
    from banet_amd.bundlenet import he_normal_lambda_weights
    flat = [t.to(dev) for wb in he_normal_lambda_weights(C, 1) for t in wb]
    tens = [AtA, Atb, absres, R, T, W, gR, gT, gW] + flat
    for mode in ("graph", "eager"):
        os.environ["BANET_TRAIN_GRAPH"] = "1" if mode == "graph" else "0"
        dense_train._small_cache.clear()
        dense_train._small_step(tens, N, 1000.0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = dense_train._small_step(tens, N, 1000.0)
        torch.cuda.synchronize()
        print("spd", spd, mode, "%.2f ms per small step" % ((time.perf_counter() - t0) * 100), dense_train.small_step_modes())
