"""Dense-mode drivers on top of banet_oracle  --  TEST INFRASTRUCTURE ONLY.

"Dense mode"  feeds the reference's iteration bodies with every pixel of a
pyramid level as a BA point: points = the level's own grid, conv1 = the source map itself
(bilinear sampling at integer coordinates is the identity), conv2 = [f|gx|gy] of the target
map.  These helpers build exactly those reference-layout inputs and call the restated
reference iterations, so the HIP dense path is checked against the same arithmetic.
"""
import numpy as np

from . import banet_oracle as orc


def level_inputs(intr, lv, normalize, dtype=np.float32):
    """intr [B,4] full-res; lv: dict(scale,H,W,src[B,H,W,C],tgt[B,H,W,C],D0[B,H,W],basis[B,H,W,K])."""
    B = lv["src"].shape[0]
    H, W, s = lv["H"], lv["W"], dtype(lv["scale"])
    N = H * W
    vv, uu = np.meshgrid(np.arange(H, dtype=dtype), np.arange(W, dtype=dtype), indexing="ij")
    pts_full = np.stack([uu.reshape(-1) * s, vv.reshape(-1) * s], -1)[None].repeat(B, 0).astype(dtype)
    intr = intr.astype(dtype)
    fx0 = np.repeat(intr[:, 0:1], N, 1)
    fy0 = np.repeat(intr[:, 1:2], N, 1)
    ox0 = np.repeat(intr[:, 2:3], N, 1)
    oy0 = np.repeat(intr[:, 3:4], N, 1)
    p = orc.compute_coordinates(pts_full, fx0, fy0, ox0, oy0, normalize)
    out = dict(conv1=lv["src"].reshape(B, N, -1).astype(dtype), conv2=orc.target_map(lv["tgt"].astype(dtype)),
               fx=fx0 / s, fy=fy0 / s, ox=ox0 / s, oy=oy0 / s, p=p, D=lv["D0"].reshape(B, N, 1).astype(dtype))
    if lv.get("basis") is not None and lv["basis"].shape[-1] > 0:
        out["Bs"] = lv["basis"].reshape(B, N, -1).astype(dtype)
    return out


def batch_scene(scenes):
    """stack per-window scenes (oracle/synth.make_pair_scene dicts) into batched levels"""
    nl = len(scenes[0]["levels"])
    levels = []
    for i in range(nl):
        l0 = scenes[0]["levels"][i]
        levels.append(dict(scale=l0["scale"], H=l0["H"], W=l0["W"],
                           src=np.stack([s["levels"][i]["src"] for s in scenes]),
                           tgt=np.stack([s["levels"][i]["tgt"] for s in scenes]),
                           D0=np.stack([s["levels"][i]["D0"] for s in scenes]),
                           basis=np.stack([s["levels"][i]["basis"] for s in scenes])))
    intr = np.stack([s["intr"] for s in scenes])
    return intr, levels


def solve_bundle(intr, levels, mlps, iters, l2_base=1000.0, dtype=np.float32, pose_only=False):
    """Fixed-count dense BA with the bundlenet iteration bodies.  Returns final (R,T,W) and
    the list of per-iteration records (level, delta, lam, AtA, Atb)."""
    B = levels[0]["src"].shape[0]
    K = 0 if pose_only else levels[0]["basis"].shape[-1]
    R = np.tile(np.eye(3, dtype=dtype)[None], (B, 1, 1))
    T = np.zeros((B, 3, 1), dtype)
    W = np.zeros((B, K, 1), dtype)
    hist = []
    for li, (lv, n_it) in enumerate(zip(levels, iters)):
        a = level_inputs(intr, lv, True, dtype)
        for _ in range(n_it):
            if pose_only:
                R, T, dbg = orc.bundle_camera_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"],
                                                        a["p"], a["D"], R, T, mlps[li], 1.0)
                hist.append(dict(level=li, delta=dbg["motion"][:, :, 0], lam=dbg["lam"], AtA=dbg["AtA"], Atb=dbg["Atb"],
                                 avg=dbg["avg"]))
            else:
                R, T, W, dbg = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"],
                                                    a["D"], a["Bs"], R, T, W, mlps[li], l2_base)
                hist.append(dict(level=li, delta=dbg["solution"][:, :, 0], lam=dbg["lam"], AtA=dbg["AtA"],
                                 Atb=dbg["Atb"], avg=dbg["avg"], mask=dbg["mask"]))
    return R, T, W, hist


def solve_legacy(intr, levels, mlps, iters, early_termination=True, dtype=np.float32, use_qr=True, R0=None, T0=None):
    """Dense legacy tracker (legacy/ba.py CameraIteration2 / CameraIteration), one window at a
    time (the reference's accept/reject is scalar).  Returns R [B,3,3], T [B,3,1], ratio [B],
    counts [levels][B].  The loop thresholds / residual ratio are the module constants of banet_oracle
    (legacy/ba.py:6-8), read at call time; use_qr = legacy/ba.py:9."""
    B = levels[0]["src"].shape[0]
    Rs, Ts, ratios, counts = [], [], [], [[0] * B for _ in levels]
    for b in range(B):
        R = np.eye(3, dtype=dtype)[None] if R0 is None else R0[b:b + 1].astype(dtype)
        T = np.zeros((1, 3, 1), dtype) if T0 is None else T0[b:b + 1].astype(dtype)
        ratio = dtype(1.0)
        for li, (lv, n_it) in enumerate(zip(levels, iters)):
            one = {k: (v[b:b + 1] if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
            a = level_inputs(intr[b:b + 1], one, False, dtype)
            it = 0
            if early_termination:
                uw = ut = dtype(1.0)
                while it < n_it and orc.ANGLE_CHANGE < uw and orc.TRANSLATION_CHANGE < ut:
                    R, T, uw, ut, ratio, _ = orc.legacy_camera_iteration2(a["conv1"], a["conv2"], a["fx"], a["fy"],
                                                                          a["ox"], a["oy"], a["p"], a["D"], R, T, mlps[li], use_qr)
                    it += 1
            else:
                for _ in range(n_it):
                    R, T, ratio = orc.legacy_camera_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"],
                                                              a["oy"], a["p"], a["D"], R, T, use_qr)
                    it += 1
            counts[li][b] = it
        Rs.append(R[0])
        Ts.append(T[0])
        ratios.append(float(np.squeeze(ratio)))
    return np.stack(Rs), np.stack(Ts), np.array(ratios), counts


def batch_window_scene(scenes):
    """stack oracle/synth.make_window_scene dicts: tgt becomes [B,pairs,H,W,C]"""
    return batch_scene(scenes)


def solve_bundle_window(intr, levels, mlps, iters, l2_base=1000.0, dtype=np.float32, R0=None, T0=None, eq=None):
    """Fixed-count dense multi-frame window BA (banet_oracle.bundle_window_iteration).  levels[i]["tgt"] is
    [B,pairs,H,W,C].  Returns (Rs [pairs][B,3,3], Ts [pairs][B,3,1], W, hist)."""
    B, pairs = levels[0]["tgt"].shape[:2]
    K = levels[0]["basis"].shape[-1]
    Rs = [np.tile(np.eye(3, dtype=dtype)[None], (B, 1, 1)) if R0 is None else R0[:, i].astype(dtype) for i in range(pairs)]
    Ts = [np.zeros((B, 3, 1), dtype) if T0 is None else T0[:, i].astype(dtype) for i in range(pairs)]
    W = np.zeros((B, K, 1), dtype)
    hist = []
    for li, (lv, n_it) in enumerate(zip(levels, iters)):
        one = dict(lv)
        one["tgt"] = lv["tgt"][:, 0]
        a = level_inputs(intr, one, True, dtype)
        conv2s = [orc.target_map(lv["tgt"][:, i].astype(dtype)) for i in range(pairs)]
        for _ in range(n_it):
            Rs, Ts, W, dbg = orc.bundle_window_iteration(a["conv1"], conv2s, a["fx"], a["fy"], a["ox"], a["oy"], a["p"],
                                                         a["D"], a["Bs"], Rs, Ts, W, mlps[li], l2_base, eq=eq)
            hist.append(dict(level=li, delta=dbg["solution"][:, :, 0], lam=dbg["lam"], AtA=dbg["AtA"], Atb=dbg["Atb"],
                             avg=dbg["avg"]))
    return Rs, Ts, W, hist


def bundle_chain(intr, levels, mlps, iters, R0, T0, W0, l2_base=1000.0, dtype=np.float32, engine="numpy", truth=False):
    """A chained coarse->fine dense bundle solve that records the state after every level -- the sequence bench.py
    times as the CPU baseline AND compares the GPU solve with (parity at BASELINE's full size).
    engine "numpy": oracle.banet_oracle.bundle_iteration (GEMM-arranged normal equations);
    engine "torch": oracle.torch_port.bundle_iteration (float32, all host threads).
    Returns (snaps, seconds): snaps[l] = dict(R, T, W: the state after level l's last iteration; delta, lam: of that
    last iteration; R_start, T_start, W_start: the state the level started from; first_delta, first_lam: the update /
    lambda of the level's FIRST iteration, i.e. one step from the start state; with truth=True also truth_delta /
    truth_lam: that same first step recomputed by the oracle in float64 from the same start state, untimed); seconds =
    time spent inside the iterations (per-level preparation excluded)."""
    import time
    R, T, W = R0.astype(dtype), T0.astype(dtype), W0.astype(dtype)
    snaps, total = [], 0.0
    for li, (lv, n_it) in enumerate(zip(levels, iters)):
        start = dict(R_start=R.copy(), T_start=T.copy(), W_start=W.copy())
        first = None
        if engine == "numpy":
            a = level_inputs(intr, lv, True, dtype)
            t0 = time.perf_counter()
            for it in range(n_it):
                R, T, W, dbg = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                                    a["Bs"], R, T, W, mlps[li], l2_base, eq=orc.equation_construction_gemm)
                if it == 0:
                    first = (dbg["solution"][:, :, 0].copy(), np.asarray(dbg["lam"]).reshape(-1).copy())
            total += time.perf_counter() - t0
            last = (dbg["solution"][:, :, 0].copy(), np.asarray(dbg["lam"]).reshape(-1).copy())
        else:
            import torch
            from . import torch_port
            f = lambda x: torch.from_numpy(np.ascontiguousarray(x))  # noqa: E731
            ti, src, tgt, dep, bas = f(intr), f(lv["src"]), f(lv["tgt"]), f(lv["D0"]), f(lv["basis"])
            Rt, Tt, Wt = f(R), f(T), f(W)
            L = torch_port.prepare_level(ti, float(lv["scale"]), src, tgt, dep, bas)        # per-level preparation, not timed
            t0 = time.perf_counter()
            for it in range(n_it):
                Rt, Tt, Wt, dbg = torch_port.bundle_iteration(None, None, None, None, None, None, Rt, Tt, Wt, mlps[li], l2_base,
                                                              level=L)
                if it == 0:
                    first = (dbg["solution"][:, :, 0].numpy().copy(), dbg["lam"].numpy().copy())
            total += time.perf_counter() - t0
            R, T, W = Rt.numpy(), Tt.numpy(), Wt.numpy()
            last = (dbg["solution"][:, :, 0].numpy().copy(), dbg["lam"].numpy().copy())
        snap = dict(start, R=R.copy(), T=T.copy(), W=W.copy(), delta=last[0], lam=last[1], first_delta=first[0],
                    first_lam=first[1])
        if truth:
            a64 = level_inputs(intr, lv, True, np.float64)
            f8 = lambda x: np.asarray(x, np.float64)  # noqa: E731
            _, _, _, d64 = orc.bundle_iteration(a64["conv1"], a64["conv2"], a64["fx"], a64["fy"], a64["ox"], a64["oy"], a64["p"],
                                                a64["D"], a64["Bs"], f8(start["R_start"]), f8(start["T_start"]),
                                                f8(start["W_start"]), mlps[li], l2_base, eq=orc.equation_construction_gemm)
            snap.update(truth_delta=d64["solution"][:, :, 0].copy(), truth_lam=np.asarray(d64["lam"]).reshape(-1).copy())
            del a64
        snaps.append(snap)
    return snaps, total


STEP_GROUPS = (("pose", slice(0, 6)), ("depth", slice(6, -1)), ("last", slice(-1, None)))


def chain_parity(gpu_snaps, ref_snaps, gpu_steps=None):
    """Per-level parity record of a GPU chain against bundle_chain's snapshots.  Relative error = max-abs difference over
    the max-abs of the reference quantity.
      R / T / W : the carried state after the level's chained iterations (GPU vs the float32 oracle chain);
      step_<group> : ONE iteration started from the oracle's own level-start state (gpu_steps[l] = dict(delta, lam)), so both
          sides solve the same system -- the well-posed form of "updates within 1e-4 relative".  Three coefficient groups,
          each on its own scale: pose (6), the damped depth coefficients, and the UNDAMPED last coefficient
          (bundlenet.py:264-266), whose update is ~lambda times larger than the others' until it has converged and a
          difference of cancelling terms afterwards (ill-conditioned: float32 implementations legitimately differ there).
          With truth steps available (bundle_chain(truth=True)) every group carries three numbers:
             step_<g>          GPU vs the float64 oracle (the error that matters),
             step_<g>_ref32    float32 oracle vs the float64 oracle (the reference arithmetic's own rounding error),
             step_<g>_vs32     GPU vs the float32 oracle;
      update_T / update_W : the level's accumulated update (end - start); informative only in a chain."""
    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    out = []
    for li, (g, r) in enumerate(zip(gpu_snaps, ref_snaps)):
        rec = dict(R=rel(g["R"], r["R"]), T=rel(g["T"], r["T"]), W=rel(g["W"], r["W"]),
                   update_T=rel(g["T"] - r["T_start"], r["T"] - r["T_start"]),
                   update_W=rel(g["W"] - r["W_start"], r["W"] - r["W_start"]))
        if gpu_steps is not None:
            s = gpu_steps[li]
            tru = r.get("truth_delta")
            for name, sl in STEP_GROUPS:
                if tru is not None:
                    rec["step_" + name] = rel(s["delta"][:, sl], tru[:, sl])
                    rec["step_" + name + "_ref32"] = rel(r["first_delta"][:, sl], tru[:, sl])
                rec["step_" + name + "_vs32"] = rel(s["delta"][:, sl], r["first_delta"][:, sl])
            rec["step_lam"] = rel(s["lam"], r["truth_lam"] if tru is not None else r["first_lam"])
        out.append(rec)
    return out


def parity_failures(per_level, tol=1e-4):
    """The gate bench.py and the tests apply: carried state within tol of the float32 oracle chain; every single-step group
    within tol of the float64 oracle (or, without truth steps, of the float32 oracle), and never further from the float32
    oracle than max(tol, 2 x that oracle's own error)."""
    bad = []
    for li, r in enumerate(per_level):
        for k in ("R", "T", "W"):
            if not r[k] <= tol:
                bad.append((li, k, r[k]))
        for name, _ in STEP_GROUPS:
            if "step_" + name in r:
                if not r["step_" + name] <= tol:
                    bad.append((li, "step_" + name, r["step_" + name]))
                lim = max(tol, 2.0 * r["step_" + name + "_ref32"])
                if not r["step_" + name + "_vs32"] <= lim:
                    bad.append((li, "step_" + name + "_vs32", r["step_" + name + "_vs32"]))
            elif "step_" + name + "_vs32" in r and not r["step_" + name + "_vs32"] <= tol:
                bad.append((li, "step_" + name + "_vs32", r["step_" + name + "_vs32"]))
        if "step_lam" in r and not r["step_lam"] <= tol:
            bad.append((li, "step_lam", r["step_lam"]))
    return bad


def chain_repeat_timer(intr, levels, mlps, R0, T0, W0, engine, warmups, repeats, iters_per_level=1, l2_base=1000.0):
    """BASELINE.md section 2's CPU protocol: the coarse->fine chain with `iters_per_level` BundleIterations at every level,
    run `warmups` untimed + `repeats` timed times from the SAME start state (identical work every repeat); per-level
    preparation (rays, the [f|gx|gy] target map -- once per level in the reference too, bundlenet.py:376-385) is done once
    and excluded.  engine "numpy" = banet_oracle.bundle_iteration (GEMM-arranged normal equations), "torch" =
    torch_port.bundle_iteration (float32, all intra-op threads).  Returns (seconds_per_repeat list, LM iterations per repeat)."""
    import time
    dtype = np.float32
    if engine == "numpy":
        preps = [level_inputs(intr, lv, True, dtype) for lv in levels]
    else:
        import torch
        from . import torch_port
        f = lambda x: torch.from_numpy(np.ascontiguousarray(x))  # noqa: E731
        preps = [torch_port.prepare_level(f(intr), float(lv["scale"]), f(lv["src"]), f(lv["tgt"]), f(lv["D0"]), f(lv["basis"]))
                 for lv in levels]
    secs = []
    for rep in range(warmups + repeats):
        t0 = time.perf_counter()
        if engine == "numpy":
            R, T, W = R0.astype(dtype), T0.astype(dtype), W0.astype(dtype)
            for a, mlp in zip(preps, mlps):
                for _ in range(iters_per_level):
                    R, T, W, _dbg = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                                         a["Bs"], R, T, W, mlp, l2_base, eq=orc.equation_construction_gemm)
        else:
            R, T, W = f(R0.astype(dtype)), f(T0.astype(dtype)), f(W0.astype(dtype))
            for L, mlp in zip(preps, mlps):
                for _ in range(iters_per_level):
                    R, T, W, _dbg = torch_port.bundle_iteration(None, None, None, None, None, None, R, T, W, mlp, l2_base, level=L)
        dt = time.perf_counter() - t0
        if rep >= warmups:
            secs.append(dt)
    return secs, iters_per_level * len(levels)
