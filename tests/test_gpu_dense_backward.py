"""Fused backward of the dense BA path (`-m gpu`, through the C ABI): banet_dense_adjoint_f32 / banet_target_map_adjoint_f32
against the float64 statement of the same adjoint (oracle/dense_adjoint.py, itself validated against finite differences
on the CPU), bit-reproducibility, and DenseBA.solve_differentiable end to end against finite differences of the float64
ORACLE chain (oracle.dense.solve_bundle)."""
import numpy as np
import pytest
import torch

from oracle import banet_oracle as orc, dense as odense, dense_adjoint as oadj, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from banet_amd import _capi
    _capi.lib()


def t(x):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float32))).to(DEV)


def n(x):
    return x.detach().cpu().numpy().astype(np.float64)


def _scene(H, W, C, K, seed, B=2, scales=(1,)):
    scenes = [synth.make_pair_scene(H, W, C, K, list(scales), seed + b, normalize_rays=True, w_gt=[0.01, -0.008, 0.006],
                                    t_gt=[0.06, -0.04, 0.03]) for b in range(B)]
    intr, levels = odense.batch_scene(scenes)
    rng = np.random.RandomState(seed)
    R = np.stack([synth.rodrigues(0.004 * rng.standard_normal(3)) for _ in range(B)])
    T = np.stack([np.asarray(s["T_gt"]) * 0.7 for s in scenes]).reshape(B, 3, 1)
    Wc = 0.02 * rng.standard_normal((B, K, 1))
    return intr, levels, R, T, Wc, rng


def _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, variant="bundle", overwrite=False, fold=False, tile=0):
    """fold: BANET_ADJOINT_FOLD_TARGET (round 6) -- the target gradient straight from the tile kernel, no dmap3 / fold pass"""
    from banet_amd import dense as bdense, dense_train
    B, H, W, C = lv["src"].shape
    K = lv["basis"].shape[-1] if variant == "bundle" else 0
    level = bdense.DenseLevel(lv["scale"], t(lv["src"]), t(lv["tgt"]), t(lv["D0"]), t(lv["basis"]) if K else None)
    ba = bdense.DenseBA(t(intr), [level], [orc.he_normal_mlp_weights(C, 5)], variant, 1000.0)
    prob = ba.problems[0]
    fill = float("nan") if overwrite else 0.0         # BANET_ADJOINT_OVERWRITE: every entry is written, nothing is read
    out = dict(dsrc=torch.full((B, H * W, C), fill, device=DEV), dmap3=torch.full((B, H, W, 3 * C), fill, device=DEV),
               ddepth=torch.full((B, H * W), fill, device=DEV), dbasis=torch.full((B, H * W, K), fill, device=DEV))
    dtgt = torch.full((B, H, W, C), fill, device=DEV)
    if fold:
        dpose, _ = dense_train.dense_adjoint(prob, t(R), t(T), t(Wc), t(G), t(gb).reshape(B, -1), t(gabs).reshape(B, -1),
                                            out["dsrc"], dtgt, out["ddepth"], out["dbasis"], overwrite=overwrite, fold=True,
                                            extra_flags=dense_train.ADJOINT_TILE_SHAPE(tile))
        torch.cuda.synchronize()
        return dict(dsrc=out["dsrc"], ddepth=out["ddepth"], dbasis=out["dbasis"], dpose=dpose, dtgt=dtgt)
    dpose, _ = dense_train.dense_adjoint(prob, t(R), t(T), t(Wc), t(G), t(gb).reshape(B, -1), t(gabs).reshape(B, -1),
                                        out["dsrc"], out["dmap3"], out["ddepth"], out["dbasis"], overwrite=overwrite)
    dense_train.target_map_adjoint(out["dmap3"], dtgt, overwrite=overwrite)
    torch.cuda.synchronize()
    return dict(dsrc=out["dsrc"], dmap3=out["dmap3"], ddepth=out["ddepth"], dbasis=out["dbasis"], dpose=dpose, dtgt=dtgt)


@pytest.mark.parametrize("H,W,C,K,seed", [(24, 32, 6, 5, 3), (48, 64, 128, 128, 7), (30, 41, 70, 33, 11), (9, 11, 3, 1, 5),
                                          (16, 16, 64, 16, 2), (17, 33, 130, 40, 9), (20, 24, 256, 64, 4), (15, 21, 128, 128, 6),
                                          (9, 7, 16, 8, 8)])
def test_dense_adjoint_kernels_match_the_float64_statement(H, W, C, K, seed):
    intr, levels, R, T, Wc, rng = _scene(H, W, C, K, seed)
    lv = levels[0]
    B, P = 2, 6 + K
    G = rng.standard_normal((B, P, P))             # not symmetric on purpose
    gb = rng.standard_normal((B, P, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    # float32-rounded inputs on both sides
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    N = H * W
    want = oadj.assembly_adjoint(a, lv64["tgt"], f32(R), f32(T), f32(Wc), f32(G), f32(gb), f32(gabs) * N)   # gavg = N gabs
    got = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
    assert want["fwd"]["mask"].mean() > 0.5
    pairs = [("dsrc", want["dsrc"]), ("dmap3", want["dmap"]), ("dtgt", want["dtgt"]), ("ddepth", want["dD0"]),
             ("dbasis", want["dbasis"])]
    for name, w in pairs:
        g = n(got[name]).reshape(w.shape)
        err = np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
        assert err < 2e-4, (name, err)
    dpose = n(got["dpose"])
    for name, sl, w in (("dR", slice(0, 9), want["dR"].reshape(B, 9)), ("dT", slice(9, 12), want["dT"].reshape(B, 3)),
                        ("dW", slice(12, None), want["dW"].reshape(B, K))):
        err = np.abs(dpose[:, sl] - w).max() / max(np.abs(w).max(), 1e-30)
        assert err < 5e-4, (name, err)


def test_dense_adjoint_is_bit_reproducible():
    intr, levels, R, T, Wc, rng = _scene(48, 64, 128, 32, 5)
    B, K, C = 2, 32, 128
    G = rng.standard_normal((B, 6 + K, 6 + K))
    gb = rng.standard_normal((B, 6 + K, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    a = _run_adjoint(intr, levels[0], R, T, Wc, G, gb, gabs)
    for _ in range(2):
        b = _run_adjoint(intr, levels[0], R, T, Wc, G, gb, gabs)
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_solve_differentiable_matches_the_fused_forward_and_oracle_finite_differences():
    from banet_amd import dense as bdense
    H, W, C, K, B = 48, 64, 16, 8, 2
    iters = [2, 2]
    intr, levels, _, T0, _, rng = _scene(H, W, C, K, 21, B=B, scales=(2, 1))
    mlps = [orc.he_normal_mlp_weights(C, 40 + i, np.float64) for i in range(2)]
    cR, cT, cW = rng.standard_normal((B, 3, 3)), rng.standard_normal((B, 3, 1)), rng.standard_normal((B, K, 1))

    def oracle_loss(levels_, mlps_):
        R, T, Wn, _ = _oracle_chain(levels_, mlps_)
        return float((cR * R).sum() + (cT * T).sum() + (cW * Wn).sum())

    def _oracle_chain(levels_, mlps_):
        R = np.tile(np.eye(3)[None], (B, 1, 1))
        T = T0.astype(np.float64).copy()
        Wn = np.zeros((B, K, 1))
        for li, lv in enumerate(levels_):
            a = odense.level_inputs(np.asarray(intr, np.float64), lv, True, np.float64)
            for _ in range(iters[li]):
                R, T, Wn, _ = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                                   a["Bs"], R, T, Wn, mlps_[li], 1000.0)
        return R, T, Wn, None

    lv64 = [{k: (np.asarray(v, np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
            for lv in levels]
    # ---- GPU: forward values equal the fused solve, gradients by the fused backward
    tl = [bdense.DenseLevel(lv["scale"], t(lv["src"]).requires_grad_(True), t(lv["tgt"]).requires_grad_(True),
                            t(lv["D0"]).requires_grad_(True), t(lv["basis"]).requires_grad_(True)) for lv in levels]
    tm = [[(t(w).requires_grad_(True), t(b).requires_grad_(True)) for w, b in lw] for lw in mlps]
    ba = bdense.DenseBA(t(intr), tl, tm, "bundle", 1000.0)
    R, T, Wn = ba.solve_differentiable(iters, T=t(T0))
    st = ba.new_state(T=t(T0))
    ba.solve(iters, st)
    assert torch.allclose(R, st.R, atol=1e-6) and torch.allclose(T, st.T, atol=1e-6) and torch.allclose(Wn, st.Wc, atol=1e-6)
    Ro, To, Wo, _ = _oracle_chain(lv64, mlps)
    for got, want in ((R, Ro), (T, To), (Wn, Wo)):
        assert np.abs(n(got) - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-6)
    loss = (t(cR) * R).sum() + (t(cT) * T).sum() + (t(cW) * Wn).sum()
    loss.backward()
    torch.cuda.synchronize()

    # ---- float64 oracle: central differences along random directions
    def fd(apply, shape, eps):
        d = rng.standard_normal(shape)
        d /= np.linalg.norm(d)
        vals = sorted((apply(e * d) - apply(-e * d)) / (2 * e) for e in (eps, 3 * eps, 10 * eps))
        return d, vals[1]

    checks = []
    for li in range(2):
        for key, tens in (("src", tl[li].src), ("tgt", tl[li].tgt), ("D0", tl[li].depth), ("basis", tl[li].basis)):
            def apply(dl, li=li, key=key):
                l2 = [dict(x) for x in lv64]
                l2[li][key] = lv64[li][key] + dl
                return oracle_loss(l2, mlps)
            d, num = fd(apply, lv64[li][key].shape, 1e-5)
            checks.append(("%s[%d]" % (key, li), num, float((n(tens.grad) * d).sum())))
        for wi in (0, 4):
            def apply(dl, li=li, wi=wi):
                m2 = [[(w.copy(), b.copy()) for w, b in lw] for lw in mlps]
                m2[li][wi] = (m2[li][wi][0] + dl, m2[li][wi][1])
                return oracle_loss(lv64, m2)
            d, num = fd(apply, mlps[li][wi][0].shape, 1e-4)
            checks.append(("mlp[%d][%d]" % (li, wi), num, float((n(tm[li][wi][0].grad) * d).sum())))
    scale = max(abs(c[1]) for c in checks)
    for name, num, ana in checks:
        assert abs(num - ana) <= 2e-2 * max(abs(num), abs(ana)) + 1e-4 * scale, (name, num, ana)


def test_solve_differentiable_multi_frame_windows_match_oracle_finite_differences():
    """Round 3: the fused backward on multi-frame windows (cfg-3's shape class: one key frame + 3 target frames sharing depth /
    basis / W, P = 18 + K): forward values equal the fused window solve and the float64 oracle chain
    (banet_oracle.bundle_window_iteration), gradients w.r.t. the key-frame features, EVERY target frame, the depth, the basis
    and the lambda weights against central differences of that oracle chain."""
    from banet_amd import dense as bdense
    H, W, C, K, B, pairs = 40, 48, 16, 26, 2, 3            # P = 44: the backward's solves run on banet_spd_solve_f32
    iters = [2, 1]
    scenes = [synth.make_window_scene(H, W, C, K, [2, 1], 61 + b, pairs, rot_mag=0.012, trans_mag=0.04) for b in range(B)]
    intr, levels = odense.batch_window_scene(scenes)
    rng = np.random.RandomState(5)
    T0 = (np.stack([s["T_gt"] for s in scenes]) * 0.7).reshape(B, pairs, 3, 1)
    mlps = [orc.he_normal_mlp_weights(C, 40 + i, np.float64) for i in range(2)]
    cR, cT, cW = rng.standard_normal((B, pairs, 3, 3)), rng.standard_normal((B, pairs, 3, 1)), rng.standard_normal((B, K, 1))

    def _oracle_chain(levels_, mlps_):
        Rs = [np.tile(np.eye(3)[None], (B, 1, 1)) for _ in range(pairs)]
        Ts = [T0[:, i].astype(np.float64).copy() for i in range(pairs)]
        Wn = np.zeros((B, K, 1))
        for li, lv in enumerate(levels_):
            one = dict(lv)
            one["tgt"] = lv["tgt"][:, 0]
            a = odense.level_inputs(np.asarray(intr, np.float64), one, True, np.float64)
            conv2s = [orc.target_map(lv["tgt"][:, i].astype(np.float64)) for i in range(pairs)]
            for _ in range(iters[li]):
                Rs, Ts, Wn, _ = orc.bundle_window_iteration(a["conv1"], conv2s, a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                                            a["Bs"], Rs, Ts, Wn, mlps_[li], 1000.0)
        return np.stack(Rs, 1), np.stack(Ts, 1), Wn

    def oracle_loss(levels_, mlps_):
        R, T, Wn = _oracle_chain(levels_, mlps_)
        return float((cR * R).sum() + (cT * T).sum() + (cW * Wn).sum())

    lv64 = [{k: (np.asarray(v, np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
            for lv in levels]
    tl = [bdense.DenseLevel(lv["scale"], t(lv["src"]).requires_grad_(True), t(lv["tgt"]).requires_grad_(True),
                            t(lv["D0"]).requires_grad_(True), t(lv["basis"]).requires_grad_(True)) for lv in levels]
    tm = [[(t(w).requires_grad_(True), t(b).requires_grad_(True)) for w, b in lw] for lw in mlps]
    ba = bdense.DenseBA(t(intr), tl, tm, "bundle", 1000.0)
    assert ba.pairs == pairs
    R, T, Wn = ba.solve_differentiable(iters, T=t(T0))
    st = ba.new_state(T=t(T0.reshape(B * pairs, 3, 1)))
    ba.solve(iters, st)
    assert torch.allclose(R, st.R, atol=1e-6) and torch.allclose(T, st.T, atol=1e-6) and torch.allclose(Wn, st.Wc, atol=1e-6)
    Ro, To, Wo = _oracle_chain(lv64, mlps)
    for got, want in ((R, Ro), (T, To), (Wn, Wo)):
        assert np.abs(n(got) - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-6)
    loss = (t(cR) * R).sum() + (t(cT) * T).sum() + (t(cW) * Wn).sum()
    loss.backward()
    torch.cuda.synchronize()

    def fd(apply, shape, eps):
        d = rng.standard_normal(shape)
        d /= np.linalg.norm(d)
        vals = sorted((apply(e * d) - apply(-e * d)) / (2 * e) for e in (eps, 3 * eps, 10 * eps))
        return d, vals[1]

    checks = []
    for li in range(2):
        for key, tens in (("src", tl[li].src), ("D0", tl[li].depth), ("basis", tl[li].basis)):
            def apply(dl, li=li, key=key):
                l2 = [dict(x) for x in lv64]
                l2[li][key] = lv64[li][key] + dl
                return oracle_loss(l2, mlps)
            d, num = fd(apply, lv64[li][key].shape, 1e-5)
            checks.append(("%s[%d]" % (key, li), num, float((n(tens.grad) * d).sum())))
        for i in range(pairs):                                   # every target frame on its own
            def apply(dl, li=li, i=i):
                l2 = [dict(x) for x in lv64]
                tg = lv64[li]["tgt"].copy()
                tg[:, i] += dl
                l2[li]["tgt"] = tg
                return oracle_loss(l2, mlps)
            d, num = fd(apply, lv64[li]["tgt"][:, i].shape, 1e-5)
            checks.append(("tgt[%d][frame %d]" % (li, i), num, float((n(tl[li].tgt.grad)[:, i] * d).sum())))
        for wi in (0, 4):
            def apply(dl, li=li, wi=wi):
                m2 = [[(w.copy(), b.copy()) for w, b in lw] for lw in mlps]
                m2[li][wi] = (m2[li][wi][0] + dl, m2[li][wi][1])
                return oracle_loss(lv64, m2)
            d, num = fd(apply, mlps[li][wi][0].shape, 1e-4)
            checks.append(("mlp[%d][%d]" % (li, wi), num, float((n(tm[li][wi][0].grad) * d).sum())))
    scale = max(abs(c[1]) for c in checks)
    for name, num, ana in checks:
        assert abs(num - ana) <= 2e-2 * max(abs(num), abs(ana)) + 1e-4 * scale, (name, num, ana)
    # bit-reproducible
    for x in [tl[0].src, tl[1].tgt, tl[1].basis]:
        x.grad_first = x.grad.clone()
        x.grad = None
    R2, T2, W2 = ba.solve_differentiable(iters, T=t(T0))
    ((t(cR) * R2).sum() + (t(cT) * T2).sum() + (t(cW) * W2).sum()).backward()
    for x in [tl[0].src, tl[1].tgt, tl[1].basis]:
        assert torch.equal(x.grad, x.grad_first)


def _avoid_abs_kinks(intr, lv, R, T, Wc, camera=False):
    """|d| is not differentiable at d = 0 and float32 / float64 may land on different sides of it (one such entry moves dsrc by
    2 |gabs|): nudge the source features of the few entries whose float64 residual is below 1e-5 away from the kink."""
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    if camera:
        a["Bs"] = np.zeros(a["Bs"].shape[:2] + (0,))
    F = oadj.forward_lean(a, lv64["tgt"], f32(R), f32(T), f32(Wc))
    near = (np.abs(F["diff"]) < 1e-5) & F["mask"][..., None]
    lv = dict(lv)
    src = np.array(lv["src"], np.float32)
    src.reshape(near.shape)[near] += 1e-3
    lv["src"] = src
    return lv


@pytest.mark.parametrize("H,W,C,K,seed", [(20, 24, 8, 136, 3), (20, 24, 128, 256, 5), (18, 22, 70, 200, 7), (16, 20, 256, 192, 9),
                                          (17, 19, 12, 255, 4), (16, 20, 200, 192, 9), (16, 20, 130, 192, 9), (16, 20, 66, 250, 9)])
def test_dense_adjoint_kernels_with_more_than_128_depth_coefficients(H, W, C, K, seed):
    """Round 3: 128 < K <= 256 (cfg-5's K = 256) -- the seed block of the GEMM-shaped piece no longer fits the LDS and is taken
    in column chunks (adj_basis_wide_kernel), the per-pixel kernel runs with 3-4 coefficients per lane."""
    intr, levels, R, T, Wc, rng = _scene(H, W, C, K, seed)
    lv = _avoid_abs_kinks(intr, levels[0], R, T, Wc)
    B, P = 2, 6 + K
    G = rng.standard_normal((B, P, P))
    gb = rng.standard_normal((B, P, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    N = H * W
    want = oadj.assembly_adjoint(a, lv64["tgt"], f32(R), f32(T), f32(Wc), f32(G), f32(gb), f32(gabs) * N)
    got = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
    again = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
    for k in got:
        assert torch.equal(got[k], again[k]), k                  # bit-reproducible
    assert want["fwd"]["mask"].mean() > 0.5
    for name, w in (("dsrc", want["dsrc"]), ("dmap3", want["dmap"]), ("dtgt", want["dtgt"]), ("ddepth", want["dD0"]),
                    ("dbasis", want["dbasis"])):
        g = n(got[name]).reshape(w.shape)
        err = np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
        assert err < 2e-4, (name, err)
    dpose = n(got["dpose"])
    for name, sl, w in (("dR", slice(0, 9), want["dR"].reshape(B, 9)), ("dT", slice(9, 12), want["dT"].reshape(B, 3)),
                        ("dW", slice(12, None), want["dW"].reshape(B, K))):
        err = np.abs(dpose[:, sl] - w).max() / max(np.abs(w).max(), 1e-30)
        assert err < 5e-4, (name, err)


@pytest.mark.parametrize("H,W,C,seed", [(24, 32, 128, 3), (21, 27, 6, 5), (18, 22, 130, 7), (16, 20, 256, 9)])
def test_pose_only_adjoint_kernels_match_the_float64_statement(H, W, C, seed):
    """Round 3: the pose-only variant (BANET_BUNDLE_CAMERA, bundlenet.py:122-191) through banet_dense_adjoint_f32: K = 0, no
    basis / coefficient pointers, dpose [B,12]."""
    intr, levels, R, T, _, rng = _scene(H, W, C, 4, seed)
    lv = _avoid_abs_kinks(intr, levels[0], R, T, np.zeros((2, 0, 1)), camera=True)
    B = 2
    G = rng.standard_normal((B, 6, 6))
    gb = rng.standard_normal((B, 6, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    a["Bs"] = np.zeros(a["Bs"].shape[:2] + (0,))
    N = H * W
    want = oadj.assembly_adjoint(a, lv64["tgt"], f32(R), f32(T), np.zeros((B, 0, 1)), f32(G), f32(gb), f32(gabs) * N)
    got = _run_adjoint(intr, lv, R, T, np.zeros((B, 0, 1)), G, gb, gabs, variant="bundle_camera")
    assert got["dpose"].shape == (B, 12)
    for name, w in (("dsrc", want["dsrc"]), ("dmap3", want["dmap"]), ("dtgt", want["dtgt"]), ("ddepth", want["dD0"])):
        g = n(got[name]).reshape(w.shape)
        err = np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
        assert err < 2e-4, (name, err)
    dpose = n(got["dpose"])
    for name, sl, w in (("dR", slice(0, 9), want["dR"].reshape(B, 9)), ("dT", slice(9, 12), want["dT"].reshape(B, 3))):
        err = np.abs(dpose[:, sl] - w).max() / max(np.abs(w).max(), 1e-30)
        assert err < 5e-4, (name, err)


def test_solve_differentiable_pose_only_variant_matches_oracle_finite_differences():
    """Round 3: DenseBA("bundle_camera").solve_differentiable -- the reference's CameraIteration levels (bundlenet.py:122-191,
    :376-385) trained through the fused path: forward equals the fused solve and the float64 oracle chain, gradients w.r.t. the
    feature maps, the depth and the lambda weights equal central differences of that chain."""
    from banet_amd import dense as bdense
    H, W, C, B = 48, 64, 16, 2
    iters = [2, 2]
    intr, levels, _, T0, _, rng = _scene(H, W, C, 4, 31, B=B, scales=(2, 1))
    mlps = [orc.he_normal_mlp_weights(C, 50 + i, np.float64) for i in range(2)]
    cR, cT = rng.standard_normal((B, 3, 3)), rng.standard_normal((B, 3, 1))

    def _oracle_chain(levels_, mlps_):
        R = np.tile(np.eye(3)[None], (B, 1, 1))
        T = T0.astype(np.float64).copy()
        for li, lv in enumerate(levels_):
            a = odense.level_inputs(np.asarray(intr, np.float64), lv, True, np.float64)
            for _ in range(iters[li]):
                R, T, _ = orc.bundle_camera_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                                      R, T, mlps_[li])
        return R, T

    def oracle_loss(levels_, mlps_):
        R, T = _oracle_chain(levels_, mlps_)
        return float((cR * R).sum() + (cT * T).sum())

    lv64 = [{k: (np.asarray(v, np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
            for lv in levels]
    tl = [bdense.DenseLevel(lv["scale"], t(lv["src"]).requires_grad_(True), t(lv["tgt"]).requires_grad_(True),
                            t(lv["D0"]).requires_grad_(True)) for lv in levels]
    tm = [[(t(w).requires_grad_(True), t(b).requires_grad_(True)) for w, b in lw] for lw in mlps]
    ba = bdense.DenseBA(t(intr), tl, tm, "bundle_camera")
    R, T, Wn = ba.solve_differentiable(iters, T=t(T0))
    assert Wn.numel() == 0
    st = ba.new_state(T=t(T0))
    ba.solve(iters, st)
    assert torch.allclose(R, st.R, atol=1e-6) and torch.allclose(T, st.T, atol=1e-6)
    Ro, To = _oracle_chain(lv64, mlps)
    for got, want in ((R, Ro), (T, To)):
        assert np.abs(n(got) - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-6)
    ((t(cR) * R).sum() + (t(cT) * T).sum()).backward()
    torch.cuda.synchronize()

    def fd(apply, shape, eps):
        d = rng.standard_normal(shape)
        d /= np.linalg.norm(d)
        vals = sorted((apply(e * d) - apply(-e * d)) / (2 * e) for e in (eps, 3 * eps, 10 * eps))
        return d, vals[1]

    checks = []
    for li in range(2):
        for key, tens in (("src", tl[li].src), ("tgt", tl[li].tgt), ("D0", tl[li].depth)):
            def apply(dl, li=li, key=key):
                l2 = [dict(x) for x in lv64]
                l2[li][key] = lv64[li][key] + dl
                return oracle_loss(l2, mlps)
            d, num = fd(apply, lv64[li][key].shape, 1e-5)
            checks.append(("%s[%d]" % (key, li), num, float((n(tens.grad) * d).sum())))
        for wi in (0, 4):
            def apply(dl, li=li, wi=wi):
                m2 = [[(w.copy(), b.copy()) for w, b in lw] for lw in mlps]
                m2[li][wi] = (m2[li][wi][0] + dl, m2[li][wi][1])
                return oracle_loss(lv64, m2)
            d, num = fd(apply, mlps[li][wi][0].shape, 1e-4)
            checks.append(("mlp[%d][%d]" % (li, wi), num, float((n(tm[li][wi][0].grad) * d).sum())))
    scale = max(abs(c[1]) for c in checks)
    for name, num, ana in checks:
        assert abs(num - ana) <= 2e-2 * max(abs(num), abs(ana)) + 1e-4 * scale, (name, num, ana)


def test_solve_differentiable_with_200_depth_coefficients_matches_oracle_finite_differences():
    """Round 3: the whole differentiable level at K = 200 (P = 206: the backward's small solves leave the LDS-resident
    banet_spd_solve_f32 for torch.linalg.solve_ex, the adjoint kernels take their K > 128 forms)."""
    from banet_amd import dense as bdense
    H, W, C, K, B = 32, 40, 8, 200, 2
    iters = [2]
    intr, levels, _, T0, _, rng = _scene(H, W, C, K, 77, B=B, scales=(1,))
    mlps = [orc.he_normal_mlp_weights(C, 60, np.float64)]
    cR, cT, cW = rng.standard_normal((B, 3, 3)), rng.standard_normal((B, 3, 1)), rng.standard_normal((B, K, 1))

    def oracle_loss(levels_):
        R = np.tile(np.eye(3)[None], (B, 1, 1))
        T = T0.astype(np.float64).copy()
        Wn = np.zeros((B, K, 1))
        a = odense.level_inputs(np.asarray(intr, np.float64), levels_[0], True, np.float64)
        for _ in range(iters[0]):
            R, T, Wn, _ = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"],
                                               R, T, Wn, mlps[0], 1000.0)
        return float((cR * R).sum() + (cT * T).sum() + (cW * Wn).sum()), (R, T, Wn)

    lv64 = [{k: (np.asarray(v, np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
            for lv in levels]
    tl = [bdense.DenseLevel(lv["scale"], t(lv["src"]).requires_grad_(True), t(lv["tgt"]).requires_grad_(True),
                            t(lv["D0"]).requires_grad_(True), t(lv["basis"]).requires_grad_(True)) for lv in levels]
    ba = bdense.DenseBA(t(intr), tl, [[(t(w), t(b)) for w, b in mlps[0]]], "bundle", 1000.0)
    R, T, Wn = ba.solve_differentiable(iters, T=t(T0))
    _, (Ro, To, Wo) = oracle_loss(lv64)
    for got, want in ((R, Ro), (T, To), (Wn, Wo)):
        assert np.abs(n(got) - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-6)
    ((t(cR) * R).sum() + (t(cT) * T).sum() + (t(cW) * Wn).sum()).backward()
    torch.cuda.synchronize()
    checks = []
    for key, tens in (("src", tl[0].src), ("tgt", tl[0].tgt), ("D0", tl[0].depth), ("basis", tl[0].basis)):
        d = rng.standard_normal(lv64[0][key].shape)
        d /= np.linalg.norm(d)
        vals = []
        for e in (1e-5, 3e-5, 1e-4):
            lp, lm = [dict(x) for x in lv64], [dict(x) for x in lv64]
            lp[0][key] = lv64[0][key] + e * d
            lm[0][key] = lv64[0][key] - e * d
            vals.append((oracle_loss(lp)[0] - oracle_loss(lm)[0]) / (2 * e))
        checks.append((key, sorted(vals)[1], float((n(tens.grad) * d).sum())))
    scale = max(abs(c[1]) for c in checks)
    for name, num, ana in checks:
        assert abs(num - ana) <= 2e-2 * max(abs(num), abs(ana)) + 1e-4 * scale, (name, num, ana)


@pytest.mark.parametrize("H,W,C,K,variant", [(24, 32, 128, 128, "bundle"), (17, 33, 130, 40, "bundle"), (20, 24, 256, 64, "bundle"),
                                             (16, 20, 128, 256, "bundle"), (9, 11, 3, 1, "bundle"), (24, 32, 128, 0, "bundle_camera"),
                                             (21, 27, 6, 0, "bundle_camera")])
def test_dense_adjoint_overwrite_mode_equals_accumulation_into_zeros(H, W, C, K, variant):
    """BANET_ADJOINT_OVERWRITE (banet_dense_adjoint_ex_f32): on NaN-filled buffers the call writes every entry of dsrc / dmap3 /
    ddepth / dbasis -- zeros where nothing contributes -- with the bits the accumulating call leaves in zero-filled buffers.  One
    window is rotated so that most of its pixels are masked, one is translated out of view (every pixel masked, no texel hit)."""
    intr, levels, R, T, Wc, rng = _scene(H, W, C, max(K, 1), 17, B=3)
    lv = levels[0]
    R, T = R.copy(), T.copy()
    R[1] = synth.rodrigues(np.array([0.0, 0.35, 0.05]))
    T[2] = np.array([[60.0], [0.0], [0.0]])
    B = 3
    if variant == "bundle_camera":
        Wc = np.zeros((B, 0, 1))
    P = 6 + K
    G = rng.standard_normal((B, P, P))
    gb = rng.standard_normal((B, P, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    acc = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, variant=variant)
    ow = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, variant=variant, overwrite=True)
    for k in acc:
        assert torch.isfinite(ow[k]).all(), k
        assert torch.equal(acc[k], ow[k]), k
    assert float(acc["dsrc"][2].abs().max()) == 0.0 and float(acc["dmap3"][2].abs().max()) == 0.0   # the empty window


@pytest.mark.parametrize("B,N,C,H,W", [(2, 700, 128, 24, 32), (1, 4096, 70, 48, 64)])
def test_sample_stats_grad_deterministic_variant_matches_the_atomic_one_and_is_bit_reproducible(B, N, C, H, W):
    """banet_sample_stats_grad_det_f32 (the default of the reference-layout training graph): same gradients as the
    float-atomic scatter to rounding, identical bits run to run; points outside the image, on the last row / column and
    several points per target cell included."""
    from banet_amd import ops
    g = torch.Generator().manual_seed(5)
    conv1 = torch.randn(B, N, C, generator=g).to(DEV)
    conv2 = torch.randn(B, H, W, 3 * C, generator=g).to(DEV)
    px = (torch.rand(B, N, generator=g) * (W + 3) - 2).to(DEV)
    py = (torch.rand(B, N, generator=g) * (H + 3) - 2).to(DEV)
    px[:, :8] = float(W - 1)                      # x1 = W is outside the image (weight 0)
    py[:, 8:16] = float(H - 1)
    px[:, 16:48] = 5.25                           # 32 points in one cell
    py[:, 16:48] = 7.5
    dstats = torch.randn(B, N, 8, generator=g).to(DEV)
    dabs = torch.randn(B, C, generator=g).to(DEV)
    a = ops.sample_stats_grad(conv1, conv2, px, py, dstats, dabs, deterministic=True)
    b = ops.sample_stats_grad(conv1, conv2, px, py, dstats, dabs, deterministic=True)
    c = ops.sample_stats_grad(conv1, conv2, px, py, dstats, dabs, deterministic=False)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    for x, y in zip(a, c):
        assert float((x - y).abs().max()) <= 1e-5 * max(float(y.abs().max()), 1e-30)


def test_dense_adjoint_with_most_pixels_outside_the_image_and_an_empty_window():
    """A large rotation pushes most of window 0's pixels out of the target image (masked: no contribution anywhere), window 1
    is translated out of view (every pixel masked): finite gradients, equal to the float64 statement; the empty window's are 0."""
    H, W, C, K = 24, 32, 16, 8
    intr, levels, R, T, Wc, rng = _scene(H, W, C, K, 13)
    lv = levels[0]
    R = R.copy()
    R[0] = synth.rodrigues(np.array([0.0, 0.35, 0.05]))
    T = T.copy()
    T[1] = np.array([[60.0], [0.0], [0.0]])                      # every pixel projects far outside the image
    B, P = 2, 6 + K
    G = rng.standard_normal((B, P, P))
    gb = rng.standard_normal((B, P, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    want = oadj.assembly_adjoint(a, lv64["tgt"], f32(R), f32(T), f32(Wc), f32(G), f32(gb), f32(gabs) * H * W)
    frac = want["fwd"]["mask"].mean(axis=1)
    assert 0.02 < frac[0] < 0.7 and frac[1] == 0.0, frac
    got = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
    for name, w in (("dsrc", want["dsrc"]), ("dtgt", want["dtgt"]), ("ddepth", want["dD0"]), ("dbasis", want["dbasis"])):
        g = n(got[name]).reshape(w.shape)
        assert np.isfinite(g).all()
        assert np.abs(g - w).max() <= 2e-4 * max(np.abs(w).max(), 1e-30), name
        assert np.abs(g[1]).max() == 0.0, name
    assert np.isfinite(n(got["dpose"])).all() and np.abs(n(got["dpose"])[1]).max() == 0.0


@pytest.mark.parametrize("B,P", [(3, 38), (8, 134), (2, 150), (1, 32)])
def test_spd_solve_matches_float64(B, P):
    """banet_spd_solve_f32 (the blocked LDL^T of the update kernel as an op): the backward's two solves per iteration."""
    from banet_amd import dense_train
    g = torch.Generator().manual_seed(P)
    M = torch.randn(B, P, P + 20, generator=g, dtype=torch.float64)
    A = M @ M.transpose(1, 2) + 0.5 * torch.eye(P, dtype=torch.float64)
    b = torch.randn(B, P, 1, generator=g, dtype=torch.float64)
    want = torch.linalg.solve(A, b)
    got = dense_train.spd_solve(A.float().to(DEV), b.float().to(DEV)).double().cpu()
    assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max())
    with pytest.raises(Exception):
        dense_train.spd_solve(torch.eye(8, device=DEV)[None], torch.ones(1, 8, 1, device=DEV))       # P < 32: unsupported, loudly
