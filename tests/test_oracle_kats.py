"""Known-answer tests of the oracle itself (CPU): the internal consistency checks the
reference affords."""
import numpy as np

from oracle import banet_oracle as orc, dense as odense, synth


def test_two_reference_formulations_agree():
    """utils.cu GEMM chain == legacy/ba.py:282-283 pure-TF twin == the GEMM arrangement."""
    rng = np.random.RandomState(0)
    J = rng.standard_normal((2, 50, 2, 11))
    G = rng.standard_normal((2, 50, 9, 2))
    d = rng.standard_normal((2, 50, 9, 1))
    a1, b1 = orc.equation_construction(J, G, d)
    a2, b2 = orc.equation_construction_tf_twin(J, G, d)
    a3, b3 = orc.equation_construction_gemm(J, G, d)
    np.testing.assert_allclose(a1, a2, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b1, b2, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a1, a3, rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(b1, b3, rtol=1e-12, atol=1e-11)


def test_grad_op_matches_finite_differences_for_symmetric_g0():
    rng = np.random.RandomState(1)
    B, N, C, P = 1, 4, 3, 5
    J = rng.standard_normal((B, N, 2, P))
    G = rng.standard_normal((B, N, C, 2))
    d = rng.standard_normal((B, N, C, 1))
    g0 = rng.standard_normal((B, P, P))
    g0 = g0 + np.swapaxes(g0, 1, 2)
    g1 = rng.standard_normal((B, P, 1))

    def loss(J, G, d):
        a, b = orc.equation_construction(J, G, d)
        return (a * g0).sum() + (b * g1).sum()

    dJ, dG, dd = orc.equation_construction_grad(J, G, d, g0, g1)
    eps = 1e-6
    for arr, grad in ((J, dJ), (G, dG), (d, dd)):
        it = np.nditer(arr, flags=["multi_index"])
        for _ in range(12):
            idx = tuple(rng.randint(0, s) for s in arr.shape)
            a = arr.copy()
            a[idx] += eps
            b = arr.copy()
            b[idx] -= eps
            args = [a if x is arr else x for x in (J, G, d)], [b if x is arr else x for x in (J, G, d)]
            fd = (loss(*args[0]) - loss(*args[1])) / (2 * eps)
            assert abs(fd - grad[idx]) < 1e-5 * max(1.0, abs(fd))


def test_jacobians_are_derivatives_of_the_warp():
    """legacy Jc = -d(px,py)/d(xi) under the left-multiplicative update; jd = +d(px,py)/dD."""
    rng = np.random.RandomState(2)
    N = 6
    p = np.stack([rng.uniform(-.4, .4, N), rng.uniform(-.3, .3, N), np.ones(N)])[None]
    p = p / np.linalg.norm(p, axis=1, keepdims=True)
    D = rng.uniform(2, 4, (1, N, 1))
    R = synth.rodrigues(np.array([0.02, -0.01, 0.03]))[None]
    T = np.array([0.05, -0.02, 0.04]).reshape(1, 3, 1)
    fx = np.full((1, N), 50.0)
    fy = np.full((1, N), 48.0)
    ox = np.full((1, N), 16.0)
    oy = np.full((1, N), 12.0)
    w = orc.warp(R, T, p, D, fx, fy, ox, oy)
    Jc = orc.camera_jacobian(w["x"], w["y"], w["Z"], fx, fy, +1)
    jd = orc.depth_jacobian(w["rx"], w["ry"], w["rz"], w["x"], w["y"], w["Z"], fx, fy)
    eps = 1e-6
    for k in range(6):
        xi = np.zeros(6)
        xi[k] = eps
        dr = orc.angle_axis_rotation(xi[None, :3], False) if k < 3 else np.eye(3)[None]
        Rn = np.matmul(dr, R)
        Tn = np.matmul(dr, T) + xi[3:].reshape(1, 3, 1)
        wn = orc.warp(Rn, Tn, p, D, fx, fy, ox, oy)
        # the solve yields delta = (J^T J)^-1 J^T d with d = F2w - F1, i.e. a DEcrease along +J
        np.testing.assert_allclose((wn["px"] - w["px"]) / eps, -Jc[:, :, 0, k], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose((wn["py"] - w["py"]) / eps, -Jc[:, :, 1, k], rtol=2e-4, atol=2e-4)
    wn = orc.warp(R, T, p, D + eps, fx, fy, ox, oy)
    np.testing.assert_allclose((wn["px"] - w["px"]) / eps, jd[:, :, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose((wn["py"] - w["py"]) / eps, jd[:, :, 1], rtol=1e-4, atol=1e-5)


def test_resampler_equals_interpolate2d_inside_the_mask():
    rng = np.random.RandomState(3)
    img = rng.standard_normal((2, 9, 11, 4)).astype(np.float32)
    x = rng.uniform(-2, 12, (2, 60)).astype(np.float32)
    y = rng.uniform(-2, 10, (2, 60)).astype(np.float32)
    x[0, :3] = [0.0, 10.0, 10.0]
    y[0, :3] = [0.0, 8.0, 3.5]
    s1, m = orc.interpolate2d(img, x, y)
    s2 = orc.resampler(img, np.stack([x, y], -1))
    assert 0 < m.sum() < m.size
    np.testing.assert_allclose(s1 * m, s2 * m, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(m[..., 0], orc.bundlenet_mask(x, y, 9, 11))


def test_float32_and_float64_oracles_agree_and_converge():
    sc = synth.make_pair_scene(48, 64, 8, 4, [2, 1], 7, normalize_rays=True, w_gt=[0.01, -0.008, 0.006],
                               t_gt=[0.06, -0.04, 0.03])
    intr, levels = odense.batch_scene([sc])
    mlps = [orc.he_normal_mlp_weights(8, 5 + i) for i in range(2)]
    out = {}
    for dt in (np.float32, np.float64):
        R = np.eye(3, dtype=dt)[None]
        T = (np.asarray(sc["T_gt"]) * 0.7).reshape(1, 3, 1).astype(dt)
        W = np.zeros((1, 4, 1), dt)
        for li, lv in enumerate(levels):
            a = odense.level_inputs(intr, lv, True, dt)
            for _ in range(4):
                R, T, W, _ = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"],
                                                  a["D"], a["Bs"], R, T, W, mlps[li], 1000.0)
        out[dt] = (R, T, W)
    np.testing.assert_allclose(out[np.float32][1], out[np.float64][1], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(out[np.float32][2], out[np.float64][2], rtol=2e-3, atol=2e-5)
    assert np.abs(out[np.float64][1][0, :, 0] - sc["T_gt"]).max() < 0.3 * np.abs(np.asarray(sc["T_gt"])).max() * 0.3 + 5e-3


def test_window_iteration_extends_bundle_iteration():
    """The multi-frame window (not in the reference) reuses the reference's per-pair functions:
    with one pair it IS bundle_iteration (bit for bit); with several the normal matrix is block-arrowhead
    and equals the sum of the per-pair normal equations embedded at their pose slots."""
    from oracle import dense as od, synth
    sc = [synth.make_window_scene(24, 32, 8, 4, [1], 9 + i, 3) for i in range(2)]
    intr, levels = od.batch_window_scene(sc)
    lv = levels[0]
    one = dict(lv)
    one["tgt"] = lv["tgt"][:, 0]
    a = od.level_inputs(intr, one, True)
    mlp = orc.he_normal_mlp_weights(8, 3)
    B, pairs, K = 2, 3, 4
    R = [np.tile(np.eye(3, dtype=np.float32)[None], (B, 1, 1)) for _ in range(pairs)]
    T = [(np.stack([s["T_gt"][i] for s in sc]) * 0.7).astype(np.float32).reshape(B, 3, 1) for i in range(pairs)]
    W = np.zeros((B, K, 1), np.float32)
    conv2s = [orc.target_map(lv["tgt"][:, i]) for i in range(pairs)]
    r1 = orc.bundle_iteration(a["conv1"], conv2s[0], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"], R[0], T[0], W, mlp, 1000.0)
    rw = orc.bundle_window_iteration(a["conv1"], conv2s[:1], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"], R[:1], T[:1], W, mlp, 1000.0)
    np.testing.assert_array_equal(r1[0], rw[0][0])
    np.testing.assert_array_equal(r1[1], rw[1][0])
    np.testing.assert_array_equal(r1[2], rw[2])
    full = orc.bundle_window_iteration(a["conv1"], conv2s, a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"], R, T, W, mlp, 1000.0)[3]
    A = full["AtA"].astype(np.float64)
    want = np.zeros_like(A)
    rhs = np.zeros(full["Atb"].shape, np.float64)
    P6 = 6 * pairs
    for i in range(pairs):
        d = orc.bundle_iteration(a["conv1"], conv2s[i], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"], R[i], T[i], W, mlp, 1000.0)[3]
        Ai, bi = d["AtA"].astype(np.float64), d["Atb"].astype(np.float64)
        sl = slice(6 * i, 6 * i + 6)
        want[:, sl, sl] += Ai[:, :6, :6]
        want[:, sl, P6:] += Ai[:, :6, 6:]
        want[:, P6:, sl] += Ai[:, 6:, :6]
        want[:, P6:, P6:] += Ai[:, 6:, 6:]
        rhs[:, sl] += bi[:, :6]
        rhs[:, P6:] += bi[:, 6:]
    assert np.abs(A - want).max() / np.abs(want).max() < 1e-5
    assert np.abs(full["Atb"] - rhs).max() / np.abs(rhs).max() < 1e-5
    for i in range(pairs):
        for j in range(pairs):
            if i != j:
                assert not A[:, 6 * i:6 * i + 6, 6 * j:6 * j + 6].any()
