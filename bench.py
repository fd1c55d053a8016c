#!/usr/bin/env python3
"""bench.py -- LM iterations/sec of the MI355X BA hot path on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (configs[1], "cfg-2"): 2-frame windows, 640x480, 5-level pyramid (scales 16,8,4,2,1),
C = 128 feature channels, K = 128 depth-basis coefficients, 10 LM iterations per level,
8 windows per GPU (weak scaling: every rank solves its own 8 windows; the only collective is
the all-gather of the per-window result records).  Iteration body = bundlenet.py:193-278
(BundleIteration), dense points, synthetic features / random-init lambda MLP.

One "step" = one full coarse->fine solve (50 LM iterations) of the rank's 8 windows with the
inputs already resident in HBM.  value = windows * 50 * steps / time over all ranks.
The JSON line also carries
  roofline     : the fused assembly kernel (dominant), algorithmic bytes 4*N_l*(2C+K+1) per
                 window-iteration vs the kernel time measured with HIP events inside the
                 timed region (banet_profile_begin/_end), peak 8 TB/s;
  cpu_baseline : the numpy oracle (a port of the reference's arithmetic) timed on the host
                 cores for ONE window x ONE iteration at each of the 5 levels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, C, K = 480, 640, 128, 128
SCALES = [16, 8, 4, 2, 1]
ITERS = [10, 10, 10, 10, 10]
WINDOWS_PER_GPU = 8
HBM_PEAK_GBS = 8000.0


def cpu_baseline(intr, levels, gt, mlps):
    """Oracle timing on the host: 1 window, CPU_ITERS chained BundleIterations at each of the 5 levels (10-15 s)."""
    import numpy as np
    from oracle import banet_oracle as orc, dense as odense
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    mlps = [[(np.asarray(w.cpu()), np.asarray(b.cpu())) for w, b in lw] for lw in mlps]
    R = np.eye(3, dtype=np.float32)[None]
    T = (gt["T"][0:1].numpy() * 0.7).reshape(1, 3, 1).astype(np.float32)
    Wc = np.zeros((1, K, 1), np.float32)
    total = 0.0
    CPU_ITERS = 4
    for li, lv in enumerate(levels):
        d = dict(scale=lv.scale, H=lv.H, W=lv.W, src=lv.src[0:1].cpu().numpy(), tgt=lv.tgt[0:1].cpu().numpy(),
                 D0=lv.depth[0:1].cpu().numpy(), basis=lv.basis[0:1].cpu().numpy())
        a = odense.level_inputs(intr[0:1].cpu().numpy(), d, True)          # per-level prep, not timed
        t0 = time.perf_counter()
        for _ in range(CPU_ITERS):                                         # a real chain: the state moves
            R, T, Wc, _ = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"],
                                               R, T, Wc, mlps[li], 1000.0, eq=orc.equation_construction_gemm)
        total += time.perf_counter() - t0
    return {"value": round(CPU_ITERS * len(levels) / total, 4), "unit": "LM iterations/s", "cores": int(threads),
            "kind": "port", "host_cpus": os.cpu_count(),
            "sample": "numpy oracle (oracle/banet_oracle.bundle_iteration, GEMM-arranged normal equations, "
                      "BLAS-threaded matmuls, single-threaded elementwise): 1 window x %d chained LM iterations at each of the "
                      "5 levels of the same synthetic 640x480 C=128 K=128 workload (%.1f s)" % (CPU_ITERS, total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=None, help="windows per GPU (default: 8 for cfg-2, 32 for cfg-3)")
    ap.add_argument("--frames", type=int, default=2, help="frames per window: 2 = cfg-2 (the default, BASELINE's metric "
                    "workload); 5 = cfg-3/4, the 5-frame sliding window (key frame + 4 target frames, SURVEY 8(d))")
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--basis", type=int, default=K, help="depth-basis coefficients K")
    ap.add_argument("--iters", type=int, default=ITERS[0], help="LM iterations per level")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reserved", type=int, default=0, help="development: banet_level_t.reserved_ bits for every level (A/B switches)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from banet_amd import dense as bdense, ops, parallel, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # development overrides (to exercise the multi-rank path on a one-GPU box): BANET_BENCH_DEVICE pins every rank to one
    # device, BANET_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    dev_index = int(os.environ.get("BANET_BENCH_DEVICE", local_rank))
    backend = os.environ.get("BANET_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pairs = args.frames - 1
    assert 1 <= pairs <= 7, "--frames 2..8"
    Hh, Ww, Kk = args.height, args.width, args.basis
    iters = [args.iters] * len(SCALES)
    default_shape = (Hh, Ww, Kk, args.iters) == (H, W, K, ITERS[0])
    B = args.windows if args.windows is not None else (WINDOWS_PER_GPU if pairs == 1 else 32)
    total_windows = B * world
    torch.manual_seed(1234 + rank)
    intr, levels, gt = bsynth.make_dense_windows(B, Hh, Ww, C, Kk, SCALES, 1234 + 2 + rank, dev, trans_mag=0.06, pairs=pairs)
    mlps = [he_normal_lambda_weights(C, 100 + i) for i in range(len(SCALES))]
    ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1000.0)
    for prob in ba.problems:
        prob.c.reserved_ = args.reserved
    T0 = (gt["T"] * 0.7).reshape(B * pairs, 3, 1).to(dev)   # translation prior: depth is unobservable from T = 0

    def step():
        st = ba.new_state(T=T0)
        st, counts = ba.solve(iters, st)
        rec = parallel.pack_results(st.R, st.T, st.Wc, counts)
        return parallel.gather_results(rec, total_windows), st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    launches = 2 * args.steps * sum(iters) + 8
    ops.profile_begin(launches)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full, st = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ops.profile_end()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(full).all(), "solve produced non-finite results"
    # the timed work is a real solve: it ends closer to the synthetic ground truth than the prior it started from
    Tgt = gt["T"].reshape(B * pairs, 3).to(dev)
    err0 = float((T0.reshape(B * pairs, 3) - Tgt).norm(dim=1).mean())
    err1 = float((st.T.reshape(B * pairs, 3) - Tgt).norm(dim=1).mean())
    Rgt = gt["R"].reshape(B * pairs, 3, 3).to(dev)
    rot0 = float((torch.eye(3, device=dev)[None] - Rgt).flatten(1).norm(dim=1).mean())
    rot1 = float((st.R.reshape(B * pairs, 3, 3) - Rgt).flatten(1).norm(dim=1).mean())
    assert err1 < err0 and rot1 < rot0, "the solve did not converge towards the ground truth (%g -> %g, %g -> %g)" % (
        err0, err1, rot0, rot1)

    if rank == 0:
        iters_per_step = sum(iters)
        value = total_windows * iters_per_step * args.steps / elapsed
        # roofline of the dominant kernel (ba_gather128_kernel): algorithmic bytes of the pass it streams
        # / its measured time, summed over every launch of the timed region (all levels)
        alg_bytes, kern_ms, nlaunch, per_level, syrk_ms, syrk_n = 0.0, 0.0, 0, {}, 0.0, 0
        for li, p in enumerate(ba.problems):
            cnt, ms = prof.get(p.N, (0, 0.0))
            scnt, sms = prof.get(-p.N, (0, 0.0))
            by = ba.algorithmic_bytes_per_iteration(li) * B * cnt
            alg_bytes += by
            kern_ms += ms
            nlaunch += cnt
            syrk_ms += sms
            syrk_n += scnt
            per_level["%dx%d" % (p.c.W, p.c.H)] = {"launches": cnt, "gather_avg_us": round(1e3 * ms / max(cnt, 1), 2),
                                                    "syrk_avg_us": round(1e3 * sms / max(scnt, 1), 2),
                                                    "gather_GBps": round(by / max(ms, 1e-9) / 1e6, 1)}
        achieved = alg_bytes / max(kern_ms, 1e-9) / 1e6            # GB/s
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and pairs == 1 and default_shape:   # the PMC pass was taken on cfg-2
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "LM iterations/sec (%d-frame %dx%d 5-level dense BA, %d-coeff depth basis)" % (args.frames, Ww, Hh, Kk),
            "value": round(value, 2), "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "ms_per_solve": round(1e3 * elapsed / args.steps / B, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("cfg-2: 2-frame 640x480 5-level pyramid, C=128, K=128 basis, 10 LM iters/level, "
                                    "batch %d windows per GPU, BundleIteration (bundlenet.py:193-278), dense points" % B)
                       if pairs == 1 and default_shape else
                       ("%s: %d-frame window (key frame + %d target frames sharing depth/basis, P = %d), "
                        "%dx%d 5-level pyramid, C=128, K=%d, %d LM iters/level, batch %d windows per GPU, dense points"
                        % ("cfg-3" if default_shape else "custom", args.frames, pairs, 6 * pairs + Kk, Ww, Hh, Kk, args.iters, B)),
                       "windows_total": total_windows, "iters_per_level": iters, "scales": SCALES,
                       "shape": {"H": Hh, "W": Ww, "C": C, "K": Kk, "frames": args.frames},
                       "parallelism": "windows sharded, dp%d" % world},
            "check": {"translation_error_prior": round(err0, 6), "translation_error_final": round(err1, 6),
                      "rotation_error_prior": round(rot0, 6), "rotation_error_final": round(rot1, 6),
                      "lambda_last_mean": round(float(st.lambda_out.mean()), 3),
                      "note": "random-init lambda MLP and l2_regularizer_base = 1000 (bundlenet.py:393) damp every step "
                              "heavily, so 50 iterations move the estimate only slightly; the cost per iteration does not "
                              "depend on it"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "ba_gather128p_kernel<1> (640x480 and 320x240 levels) + ba_gather128_kernel<1> (coarser levels)",
                         "launches": nlaunch, "avg_launch_us": round(1e3 * kern_ms / max(nlaunch, 1), 2),
                         "algorithmic_bytes_per_launch": round(alg_bytes / max(nlaunch, 1)),
                         "kernel_time_share": round(kern_ms / (1e3 * elapsed), 4),
                         "syrk_kernel": {"launches": syrk_n, "avg_launch_us": round(1e3 * syrk_ms / max(syrk_n, 1), 2),
                                         "time_share": round(syrk_ms / (1e3 * elapsed), 4)},
                         "pipeline_GBps": round(alg_bytes / max(kern_ms + syrk_ms, 1e-9) / 1e6, 1),
                         "per_level": per_level},
        }
        if world == 1 and not args.no_cpu_baseline and pairs == 1 and default_shape:
            out["cpu_baseline"] = cpu_baseline(intr, levels, gt, mlps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
