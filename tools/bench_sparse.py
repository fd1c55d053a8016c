#!/usr/bin/env python3
"""The reference's OWN workload (legacy/example.py:11-13,83): 512x384 frames, 3 pyramid levels (scale 4, 2, 1), C = 128
feature channels (legacy/feat.py:240-271), N = 4096 sampled points, pose only (P = 6), fixed iteration counts [5, 8, 12]
(`ba.early_termination = False`, legacy/example.py:8) -- through banet_amd.legacy.Tracker.trackTF, i.e. the generic
gather kernel on the reference's [f|gx|gy] target layout.  B = 1 (what the reference runs) and B = 64 windows.

    python tools/bench_sparse.py [--windows 1 64] [--reps 20] [--lm]      (GPU box; prints one JSON line per batch size)

Algorithmic bytes per LM iteration and window: every point reads its source row (C floats) and 4 bilinear taps of the 3C
target map: 4 * N * (C + 12 C) = 27.3 MB -- 3.4 us at 8 TB/s, so B = 1 is a launch / latency measurement, not a bandwidth one.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

H, W, C, N = 384, 512, 128, 4096
SCALES = [4, 2, 1]
ITERS = [5, 8, 12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, nargs="+", default=[1, 64])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lm", action="store_true", help="early_termination = True: the LM variant with the lambda MLP (legacy/eval.py:9)")
    args = ap.parse_args()
    import torch
    from banet_amd import legacy, ops, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    dev = torch.device("cuda", 0)
    out = []
    for B in args.windows:
        torch.manual_seed(5)
        intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, 0, SCALES, 77, dev, normalize_rays=False, trans_mag=0.04)
        layers = [torch.cat([lv.src, lv.tgt], dim=0).contiguous() for lv in levels]          # [2B,H_l,W_l,C], sources first
        g = torch.Generator().manual_seed(3)
        pts = torch.stack([torch.rand(B, N, generator=g) * (W - 9) + 4, torch.rand(B, N, generator=g) * (H - 9) + 4], dim=-1).to(dev)
        # depth at the sampled points from the finest depth map (nearest pixel: a synthetic stand-in for the sensor depth)
        d = levels[-1].depth.reshape(B, H * W).gather(1, (pts[..., 1].round().long() * W + pts[..., 0].round().long())).reshape(B, N, 1)
        intrisic = intr.reshape(B, 4, 1)
        R0 = torch.eye(3, device=dev).repeat(B, 1, 1)
        T0 = (gt["T"] * 0.5).reshape(B, 3, 1).to(dev)
        trk = legacy.Tracker(lambda_weights={str(l): he_normal_lambda_weights(C, 40 + l) for l in (1, 2, 3)}, iters=ITERS)
        legacy.early_termination = bool(args.lm)

        def solve():
            return trk.trackTF(intrisic, layers, pts, d, R0, T0, ITERS)

        for _ in range(3):
            R, T, ratio = solve()
        torch.cuda.synchronize()
        ops.profile_begin(2 * args.reps * (sum(ITERS) + 3) + 8)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            R, T, ratio = solve()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        prof = ops.profile_end()
        cnt, ms = prof.get(N, (0, 0.0))
        iters_run = sum(int(x.sum()) for x in trk.level_iters_run)      # over the B windows, last solve
        e0 = float((T0[:, :, 0] - gt["T"].to(dev)).norm(dim=1).mean())
        e1 = float((T[:, :, 0] - gt["T"].to(dev)).norm(dim=1).mean())
        alg = 4.0 * N * 13 * C * B                                       # per launch (one LM iteration of B windows)
        rec = {"workload": "reference's own: sparse N=%d points, %dx%d, 3 levels, iters %s, P=6, %s, batch %d" % (
                   N, W, H, ITERS, "early-terminated LM (CameraIteration2)" if args.lm else "fixed count (CameraIteration)", B),
               "windows": B, "ms_per_solve_batch": round(1e3 * el / args.reps, 3), "ms_per_solve": round(1e3 * el / args.reps / B, 4),
               "value": round(iters_run * args.reps / el, 1), "unit": "LM iterations/s", "iterations_per_solve": iters_run // B,
               "includes": "per-level preparation (interpolate2d2 + [f|gx|gy] target map) + LM loop, as legacy/ba.py:106-145",
               "roofline": {"bound": "hbm", "kernel": "ba_gather_kernel (generic, 3C target layout)", "launches": cnt,
                            "avg_launch_us": round(1e3 * ms / max(cnt, 1), 2), "algorithmic_bytes_per_launch": int(alg),
                            "achieved": round(alg * cnt / max(ms, 1e-9) / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(alg * cnt / max(ms, 1e-9) / 1e6 / 8000.0, 4),
                            "kernel_time_share": round(ms / (1e3 * el), 4)},
               "check": {"translation_error_prior": round(e0, 6), "translation_error_final": round(e1, 6)}}
        assert torch.isfinite(R).all() and torch.isfinite(T).all()
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del layers, levels
        torch.cuda.empty_cache()
    legacy.early_termination = True


if __name__ == "__main__":
    main()
