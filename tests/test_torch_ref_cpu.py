"""The float64 torch twin used for the full-size GPU checks agrees with the numpy oracle."""
import numpy as np
import torch

import torch_ref
from oracle import banet_oracle as orc, dense as odense, synth


def _scene(K, C=6, H=20, W=28, big_motion=False):
    t = [0.5, -0.3, 0.2] if big_motion else [0.06, -0.04, 0.03]
    w = [0.08, -0.06, 0.05] if big_motion else [0.01, -0.008, 0.006]
    sc = synth.make_pair_scene(H, W, C, K, [1], 3, normalize_rays=True, w_gt=w, t_gt=t)
    return sc, odense.batch_scene([sc])


def test_bundle_normal_equations_match_oracle():
    for big in (False, True):
        sc, (intr, levels) = _scene(5, big_motion=big)
        lv = levels[0]
        a = odense.level_inputs(intr, lv, True, np.float64)
        R = synth.rodrigues(np.array([0.004, 0.002, -0.003]))[None]
        T = (np.asarray(sc["T_gt"]) * 0.8).reshape(1, 3, 1)
        Wc = np.full((1, 5, 1), 0.01)
        mlp = orc.he_normal_mlp_weights(6, 1)
        _, _, _, dbg = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                            a["Bs"], R, T, Wc, mlp, 1000.0)
        tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))  # noqa: E731
        AtA, Atb, absres, nvalid = torch_ref.dense_assemble(tt(intr), 1.0, tt(lv["src"]), tt(lv["tgt"]), tt(lv["D0"]),
                                                            tt(lv["basis"]), tt(R), tt(T), tt(Wc), True, True)
        assert 0 < nvalid.item() <= 20 * 28 and nvalid.item() == dbg["mask"].sum()
        if big:
            assert nvalid.item() < 20 * 28                    # some pixels really leave the image
        np.testing.assert_allclose(AtA.numpy(), dbg["AtA"], rtol=1e-9, atol=1e-9 * np.abs(dbg["AtA"]).max())
        np.testing.assert_allclose(Atb.numpy()[..., None], dbg["Atb"], rtol=1e-9, atol=1e-9 * np.abs(dbg["Atb"]).max())
        np.testing.assert_allclose(absres.numpy() / (20 * 28), dbg["avg"][:, 0], rtol=1e-10)


def test_legacy_normal_equations_match_oracle():
    sc = synth.make_pair_scene(20, 28, 6, 0, [1], 3, normalize_rays=False, w_gt=[0.01, -0.008, 0.006],
                               t_gt=[0.06, -0.04, 0.03])
    intr, levels = odense.batch_scene([sc])
    lv = levels[0]
    a = odense.level_inputs(intr, lv, False, np.float64)
    R, T = np.eye(3)[None], np.zeros((1, 3, 1))
    mlp = orc.he_normal_mlp_weights(6, 1)
    *_, dbg = orc.legacy_camera_iteration2(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                           R, T, mlp)
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))  # noqa: E731
    AtA, Atb, absres, nvalid = torch_ref.dense_assemble(tt(intr), 1.0, tt(lv["src"]), tt(lv["tgt"]), tt(lv["D0"]), None,
                                                        tt(R), tt(T), None, False, False)
    np.testing.assert_allclose(AtA.numpy(), dbg["AtA"], rtol=1e-9, atol=1e-9 * np.abs(dbg["AtA"]).max())
    np.testing.assert_allclose(Atb.numpy()[..., None], dbg["Atb"], rtol=1e-9, atol=1e-9 * np.abs(dbg["Atb"]).max())


def test_legacy_tracker_helper_methods_match_oracle():
    """the small methods of legacy/ba.py's Tracker (Jacobian without the bundlenet minus sign, V matrix, unguarded
    Rodrigues, k = 1 conv) as pure torch functions, against the numpy oracle"""
    import numpy as np
    import torch
    from banet_amd import legacy
    from oracle import banet_oracle as orc
    rng = np.random.RandomState(2)
    B, N = 2, 7
    x, y = rng.uniform(-0.5, 0.5, (B, N)).astype(np.float32), rng.uniform(-0.5, 0.5, (B, N)).astype(np.float32)
    Z = rng.uniform(1, 3, (B, N)).astype(np.float32)
    fx, fy = np.full((B, N), 300.0, np.float32), np.full((B, N), 280.0, np.float32)
    w1, b1 = rng.standard_normal((4, 8)).astype(np.float32), rng.standard_normal(8).astype(np.float32)
    tr = legacy.Tracker(lambda_weights={"2": [(torch.from_numpy(w1), torch.from_numpy(b1))]})
    T = torch.from_numpy
    J = tr.CameraJacobianMatrix(T(x), T(y), T(Z), T(fx), T(fy)).numpy()
    np.testing.assert_allclose(J, orc.camera_jacobian(x, y, Z, fx, fy, +1), rtol=1e-6, atol=1e-6)
    w = rng.uniform(-0.3, 0.3, (B, 3)).astype(np.float32)
    Rm = tr.AngleaAxisRotation(T(w[:, 0:1]), T(w[:, 1:2]), T(w[:, 2:3])).numpy()
    np.testing.assert_allclose(Rm, orc.angle_axis_rotation(w, clamp_theta=False), rtol=1e-5, atol=1e-6)
    V = tr.VMatrix(T(w[:, 0:1]), T(w[:, 1:2]), T(w[:, 2:3])).numpy()
    np.testing.assert_allclose(V, orc.vmatrix(w), rtol=1e-5, atol=1e-6)
    assert np.isnan(tr.AngleaAxisRotation(torch.zeros(1, 1), torch.zeros(1, 1), torch.zeros(1, 1)).numpy()).any()   # as the reference
    xin = rng.standard_normal((B, 1, 4)).astype(np.float32)
    out = tr.conv1d(T(xin), 8, "lambda_2_1", activation=torch.nn.functional.selu).numpy()
    np.testing.assert_allclose(out, orc.selu(xin @ w1 + b1), rtol=1e-5, atol=1e-6)


def test_bundle_chain_engines_agree_and_parity_record_shape():
    """oracle.dense.bundle_chain: the numpy oracle and the float32 torch port (bench.py's two CPU baselines) walk the same
    chain; chain_parity of one against the other is the record bench.py emits for the GPU."""
    import numpy as np
    from oracle import banet_oracle as orc, dense as odense, synth
    C, K = 16, 8
    sc = synth.make_pair_scene(24, 32, C, K, [2, 1], 11, normalize_rays=True, w_gt=[0.01, -0.008, 0.006], t_gt=[0.06, -0.04, 0.03])
    intr, levels = odense.batch_scene([sc])
    mlps = [orc.he_normal_mlp_weights(C, 5 + i) for i in range(2)]
    R0 = np.eye(3, dtype=np.float32)[None]
    T0 = (np.asarray(sc["T_gt"]) * 0.7).reshape(1, 3, 1).astype(np.float32)
    W0 = np.zeros((1, K, 1), np.float32)
    a, sec_a = odense.bundle_chain(intr, levels, mlps, [3, 2], R0, T0, W0, engine="numpy", truth=True)
    b, sec_b = odense.bundle_chain(intr, levels, mlps, [3, 2], R0, T0, W0, engine="torch")
    assert sec_a > 0 and sec_b > 0 and len(a) == len(b) == 2
    steps = [dict(delta=s["first_delta"], lam=s["first_lam"]) for s in b]
    par = odense.chain_parity(b, a, steps)
    for r in par:
        assert {"R", "T", "W", "step_pose", "step_depth", "step_last", "step_last_ref32", "step_last_vs32", "step_lam"} <= set(r)
    assert odense.parity_failures(par, 1e-4) == [], odense.parity_failures(par, 1e-4)
    np.testing.assert_array_equal(a[1]["R_start"], a[0]["R"])          # levels chain: a level starts where the previous ended
    # the gate trips on a wrong update
    steps[1]["delta"] = steps[1]["delta"] * 1.01
    assert odense.parity_failures(odense.chain_parity(b, a, steps), 1e-4) != []


def test_window_twin_matches_oracle_window_iteration():
    """oracle.torch_port.window_assemble / window_iteration (the float64 twin the full-size multi-frame GPU checks use)
    against banet_oracle.bundle_window_iteration: same normal equations, lambda, solution and updated state."""
    from oracle import torch_port
    C, K, pairs, H, W = 6, 5, 3, 20, 28
    sc = synth.make_window_scene(H, W, C, K, [1], 9, pairs, rot_mag=0.012, trans_mag=0.04)
    intr, levels = odense.batch_window_scene([sc])
    lv = levels[0]
    one = dict(lv)
    one["tgt"] = lv["tgt"][:, 0]
    a = odense.level_inputs(intr, one, True, np.float64)
    conv2s = [orc.target_map(lv["tgt"][:, i].astype(np.float64)) for i in range(pairs)]
    rng = np.random.RandomState(4)
    Rs = [synth.rodrigues(rng.uniform(-0.004, 0.004, 3))[None] for _ in range(pairs)]
    Ts = [(np.asarray(sc["T_gt"])[i] * 0.8).reshape(1, 3, 1) for i in range(pairs)]
    Wc = rng.uniform(-0.01, 0.01, (1, K, 1))
    mlp = orc.he_normal_mlp_weights(C, 1)
    Rn, Tn, Wn, dbg = orc.bundle_window_iteration(a["conv1"], conv2s, a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                                  a["Bs"], Rs, Ts, Wc, mlp, 1000.0)
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float64)))  # noqa: E731
    R2, T2, W2, d2 = torch_port.window_iteration(tt(intr), 1.0, tt(lv["src"]), tt(lv["tgt"]), tt(lv["D0"]), tt(lv["basis"]),
                                                 tt(np.stack(Rs, 1)), tt(np.stack(Ts, 1)), tt(Wc), mlp, 1000.0)
    np.testing.assert_allclose(d2["AtA"].numpy(), dbg["AtA"], rtol=1e-9, atol=1e-9 * np.abs(dbg["AtA"]).max())
    np.testing.assert_allclose(d2["Atb"].numpy()[..., None], dbg["Atb"], rtol=1e-9, atol=1e-9 * np.abs(dbg["Atb"]).max())
    np.testing.assert_allclose(d2["lam"].numpy(), np.asarray(dbg["lam"]).reshape(-1), rtol=1e-9)
    np.testing.assert_allclose(d2["solution"].numpy(), dbg["solution"][:, :, 0], rtol=1e-7, atol=1e-9 * np.abs(dbg["solution"]).max())
    np.testing.assert_allclose(R2.numpy(), np.stack(Rn, 1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(T2.numpy(), np.stack(Tn, 1), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(W2.numpy(), Wn, rtol=1e-8, atol=1e-12)


def test_sparse_iteration_twin_matches_the_numpy_oracle():
    """oracle/torch_port.sparse_iteration (the differentiable float64 statement the GPU tests measure training gradients against)
    == banet_oracle.bundle_iteration / bundle_camera_iteration on sampled points, incl. points that leave the image."""
    from oracle import torch_port
    rng = np.random.RandomState(3)
    sc = synth.make_pair_scene(24, 32, 6, 5, [1], 9, normalize_rays=True, w_gt=[0.02, -0.015, 0.01], t_gt=[0.12, -0.08, 0.05])
    intr, levels = odense.batch_scene([sc])
    lv = levels[0]
    N = 300
    pts = np.stack([rng.uniform(0.2, 30.8, (1, N)), rng.uniform(0.2, 22.8, (1, N))], axis=-1)
    fx = np.repeat(intr[:, 0:1], N, 1); fy = np.repeat(intr[:, 1:2], N, 1)
    ox = np.repeat(intr[:, 2:3], N, 1); oy = np.repeat(intr[:, 3:4], N, 1)
    p = orc.compute_coordinates(pts, fx, fy, ox, oy, True)
    conv1 = orc.resampler(lv["src"].astype(np.float64), pts)
    conv2 = orc.target_map(lv["tgt"].astype(np.float64))
    D = orc.resampler(lv["D0"][..., None].astype(np.float64), pts)
    Bs = orc.resampler(lv["basis"].astype(np.float64), pts)
    R = synth.rodrigues(np.array([0.01, 0.004, -0.006]))[None]
    T = (np.asarray(sc["T_gt"]) * 0.6).reshape(1, 3, 1)
    Wc = rng.standard_normal((1, 5, 1)) * 0.02
    mlp = orc.he_normal_mlp_weights(6, 4, np.float64)
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float64)))  # noqa: E731
    lw = [(tt(w).reshape(w.shape[-2], w.shape[-1]), tt(b).reshape(-1)) for w, b in mlp]
    R2, T2, W2, dbg = orc.bundle_iteration(conv1, conv2, fx, fy, ox, oy, p, D, Bs, R, T, Wc, mlp, 1000.0)
    assert 0 < dbg["mask"].sum() < N                          # some points really leave the image
    r2, t2, w2 = torch_port.sparse_iteration(tt(conv1), tt(conv2), tt(D), tt(Bs), tt(R), tt(T), tt(Wc), lw, True, 1000.0,
                                              tt(fx), tt(fy), tt(ox), tt(oy), tt(p))
    for a, b in ((r2, R2), (t2, T2), (w2, W2)):
        np.testing.assert_allclose(a.numpy(), b, rtol=1e-8, atol=1e-10 * max(np.abs(b).max(), 1.0))
    Rc, Tc, _ = orc.bundle_camera_iteration(conv1, conv2, fx, fy, ox, oy, p, D, R, T, mlp)
    rc, tc, _ = torch_port.sparse_iteration(tt(conv1), tt(conv2), tt(D), None, tt(R), tt(T), None, lw, False, 1.0,
                                            tt(fx), tt(fy), tt(ox), tt(oy), tt(p))
    np.testing.assert_allclose(rc.numpy(), Rc, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(tc.numpy(), Tc, rtol=1e-8, atol=1e-10)
