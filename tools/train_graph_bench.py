#!/usr/bin/env python3
"""One differentiable BundleIteration (forward + backward w.r.t. features, basis, depth, pose, lambda-MLP weights):
the fused node (round 5: the inference kernels forward, implicit differentiation + banet_dense_adjoint_f32 on the sparse layout
backward) vs the lean training graph (ops.sample_stats + block-wise normal equations) vs the reference-style graph (the reference's
statements with the EquationConstruction op): time per step and peak device memory.  PN=4096 PH=384 PW=512: the reference's
training shape (bundlenet.py:332-399)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banet_amd import ops
from banet_amd.bundlenet import BundleNet, he_normal_lambda_weights
dev = torch.device("cuda:0")
B, H, W, C, K = int(os.environ.get("PB", "4")), int(os.environ.get("PH", "120")), int(os.environ.get("PW", "160")), 128, 128
N = int(os.environ.get("PN", str(H * W)))
g = torch.Generator().manual_seed(3)
img = torch.randn(B, H, W, C, generator=g).to(dev)
conv2 = ops.target_map(img)
pts = torch.stack([torch.rand(B, N, generator=g) * (W - 3) + 1, torch.rand(B, N, generator=g) * (H - 3) + 1], dim=-1).to(dev)
conv1 = ops.resample(img, pts) + 0.05 * torch.randn(B, N, C, device=dev)
fx = torch.full((B, N), 0.8 * W, device=dev); fy = fx.clone()
ox = torch.full((B, N), W / 2.0, device=dev); oy = torch.full((B, N), H / 2.0, device=dev)
ray = torch.stack([(pts[..., 0] - ox) / fx, (pts[..., 1] - oy) / fy, torch.ones(B, N, device=dev)], dim=1)
p = ray / ray.norm(dim=1, keepdim=True)
D = 2.5 + torch.rand(B, N, 1, device=dev)
Bs = torch.randn(B, N, K, device=dev) / K ** 0.5
R = torch.eye(3, device=dev).repeat(B, 1, 1)
T = 0.01 * torch.randn(B, 3, 1, device=dev)
Wc = torch.zeros(B, K, 1, device=dev)
for graph in tuple(os.environ.get("PGRAPHS", "fused,lean,reference").split(",")):
    lw = [(w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)) for w, b in he_normal_lambda_weights(C, 7)]
    net = BundleNet(lambda_weights={"0": lw})
    net.training_graph = graph
    leaves = [x.clone().requires_grad_(True) for x in (conv1, conv2, D, Bs, T, Wc)]

    def step():
        R2, T2, W2 = net.BundleIteration(leaves[0], leaves[1], fx, fy, ox, oy, p, leaves[2], leaves[3], R, leaves[4], leaves[5], 1000.0, "0")
        loss = R2.sum() + T2.sum() + W2.sum()
        torch.autograd.grad(loss, leaves + [x for wb in lw for x in wb])
    try:
        step(); torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        print("%-9s B=%d N=%d (%dx%d map) C=%d K=%d: %.1f ms per forward+backward, peak extra memory %.0f MB" % (
            graph, B, N, W, H, C, K, (time.perf_counter() - t0) / 3 * 1e3, (torch.cuda.max_memory_allocated() - base) / 2 ** 20))
    except RuntimeError as e:
        print(graph, "failed:", str(e)[:120])
