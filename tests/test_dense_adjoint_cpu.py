"""The hand-derived adjoint of the dense bundle assembly (oracle/dense_adjoint.py) -- the statement the fused HIP backward
is checked against -- validated on the CPU: its forward equals the oracle's BundleIteration assembly, and its gradients
equal central finite differences of that forward in float64."""
import numpy as np
import pytest

from oracle import banet_oracle as orc, dense as odense, dense_adjoint as adj, synth


def _scene(H=24, W=32, C=6, K=5, seed=3, B=2):
    scenes = [synth.make_pair_scene(H, W, C, K, [1], seed + b, normalize_rays=True, w_gt=[0.01, -0.008, 0.006],
                                    t_gt=[0.06, -0.04, 0.03]) for b in range(B)]
    intr, levels = odense.batch_scene(scenes)
    lv = levels[0]
    rng = np.random.RandomState(seed)
    R = np.stack([synth.rodrigues(0.004 * rng.standard_normal(3)) for _ in range(B)]).astype(np.float64)
    T = np.stack([np.asarray(s["T_gt"]) * 0.7 for s in scenes]).reshape(B, 3, 1).astype(np.float64)
    Wc = 0.02 * rng.standard_normal((B, K, 1))
    return intr, lv, R, T, Wc, rng


def _phi(intr, lv, R, T, Wc, G, gb, gavg):
    a = odense.level_inputs(intr, lv, True, np.float64)
    F = adj.forward_lean(a, lv["tgt"], R, T, Wc)
    return float((G * F["AtA"]).sum() + (gb * F["Atb"]).sum() + (gavg * F["avg"]).sum())


def test_forward_lean_equals_the_oracle_iteration_assembly():
    intr, lv, R, T, Wc, rng = _scene()
    a = odense.level_inputs(intr, lv, True, np.float64)
    mlp = orc.he_normal_mlp_weights(lv["src"].shape[-1], 5, np.float64)
    _, _, _, dbg = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"],
                                        R, T, Wc, mlp, 1000.0)
    F = adj.forward_lean(a, lv["tgt"], R, T, Wc)
    for k in ("AtA", "Atb", "avg"):
        np.testing.assert_allclose(F[k], dbg[k], rtol=1e-10, atol=1e-10 * np.abs(dbg[k]).max())


@pytest.mark.parametrize("seed", [3, 11])
def test_assembly_adjoint_matches_finite_differences(seed):
    intr, lv, R, T, Wc, rng = _scene(seed=seed)
    B, H, W, C = lv["src"].shape
    K = lv["basis"].shape[-1]
    P = 6 + K
    G = rng.standard_normal((B, P, P))            # deliberately NOT symmetric
    gb = rng.standard_normal((B, P, 1))
    gavg = rng.standard_normal((B, 1, C))
    lv = {k: (np.asarray(v, np.float64) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv, True, np.float64)
    out = adj.assembly_adjoint(a, lv["tgt"], R, T, Wc, G, gb, gavg)
    assert out["fwd"]["mask"].mean() > 0.5

    def fd(apply, shape, eps):
        """central differences at three steps; the one closest to the median is returned (phi is a sum of ~1e3 terms of
        mixed sign: the smallest steps are round-off limited, the largest see the bilinear kinks)"""
        d = rng.standard_normal(shape)
        d /= np.linalg.norm(d)
        vals = sorted((apply(e * d) - apply(-e * d)) / (2 * e) for e in (eps, 3 * eps, 10 * eps))
        return d, vals[1]

    def with_level(key, delta):
        l2 = dict(lv)
        l2[key] = lv[key] + delta
        return _phi(intr, l2, R, T, Wc, G, gb, gavg)

    checks = []
    for key, gname, shape in (("src", "dsrc", (B, H, W, C)), ("tgt", "dtgt", (B, H, W, C)), ("D0", "dD0", (B, H, W)),
                              ("basis", "dbasis", (B, H, W, K))):
        for _ in range(2):
            d, num = fd(lambda dl: with_level(key, dl), shape, 3e-6)
            checks.append((key, num, float((out[gname].reshape(shape) * d).sum())))
    d, num = fd(lambda dl: _phi(intr, lv, R + dl, T, Wc, G, gb, gavg), (B, 3, 3), 1e-7)
    checks.append(("R", num, float((out["dR"] * d).sum())))
    d, num = fd(lambda dl: _phi(intr, lv, R, T + dl, Wc, G, gb, gavg), (B, 3, 1), 1e-7)
    checks.append(("T", num, float((out["dT"] * d).sum())))
    d, num = fd(lambda dl: _phi(intr, lv, R, T, Wc + dl, G, gb, gavg), (B, K, 1), 1e-7)
    checks.append(("W", num, float((out["dW"] * d).sum())))
    for name, num, ana in checks:
        assert abs(num - ana) <= 1e-4 * max(abs(num), abs(ana)) + 1e-7, (name, num, ana)


def test_solve_update_graph_equals_the_oracle_iteration_tail():
    """banet_amd.dense_train.solve_update_graph (the small differentiable part of the fused backward: lambda MLP, damping,
    solve, SE(3) / W update on the saved normal equations) reproduces oracle.bundle_iteration from its own AtA / Atb / avg."""
    import torch
    from banet_amd import dense_train
    intr, lv, R, T, Wc, rng = _scene()
    C = lv["src"].shape[-1]
    N = lv["src"].shape[1] * lv["src"].shape[2]
    a = odense.level_inputs(intr, lv, True, np.float64)
    mlp = orc.he_normal_mlp_weights(C, 5, np.float64)
    Rn, Tn, Wn, dbg = orc.bundle_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"],
                                           R, T, Wc, mlp, 1000.0)
    tt = lambda v: torch.from_numpy(np.asarray(v, np.float64))
    layers = [(tt(w), tt(b)) for w, b in mlp]
    R2, T2, W2 = dense_train.solve_update_graph(tt(dbg["AtA"]), tt(dbg["Atb"][..., 0]), tt(dbg["avg"][:, 0] * N), N, tt(R), tt(T),
                                                tt(Wc), layers, 1000.0)
    for got, want in ((R2, Rn), (T2, Tn), (W2, Wn)):
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-9, atol=1e-12)


def test_small_step_graph_of_a_multi_frame_window_matches_the_oracle_update():
    """dense_train.solve_update_graph(pairs = 3) -- the differentiable small part of the window iteration the fused backward
    differentiates -- reproduces banet_oracle.bundle_window_iteration's update from the oracle's own AtA / Atb / sum|d|
    (float64, CPU), including the per-frame SE(3) updates and their order in the solution vector."""
    import numpy as np
    import torch
    from banet_amd import dense_train
    from oracle import banet_oracle as orc, dense as odense, synth
    C, K, pairs, H, W = 6, 5, 3, 16, 20
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
    mlp = orc.he_normal_mlp_weights(C, 1, np.float64)
    Rn, Tn, Wn, dbg = orc.bundle_window_iteration(a["conv1"], conv2s, a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"],
                                                  a["Bs"], Rs, Ts, Wc, mlp, 1000.0)
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float64)))  # noqa: E731
    N = H * W
    absres = tt(dbg["avg"][:, 0]) * (N * pairs)
    layers = [(tt(w), tt(b)) for w, b in mlp]
    R2, T2, W2 = dense_train.solve_update_graph(tt(dbg["AtA"]), tt(dbg["Atb"][..., 0]), absres, N, tt(np.stack(Rs, 1)),
                                                tt(np.stack(Ts, 1)), tt(Wc), layers, 1000.0, pairs=pairs)
    np.testing.assert_allclose(R2.numpy(), np.stack(Rn, 1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(T2.numpy(), np.stack(Tn, 1), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(W2.numpy(), Wn, rtol=1e-8, atol=1e-12)


def _camera_inputs(intr, lv):
    """level inputs of the pose-only CameraIteration: the bundle level with an EMPTY depth basis (K = 0)"""
    a = odense.level_inputs(intr, lv, True, np.float64)
    a["Bs"] = np.zeros(a["Bs"].shape[:2] + (0,))
    return a


def test_pose_only_small_step_and_assembly_adjoint():
    """The pose-only variant (bundlenet.py:122-191) through the same statements with K = 0: forward_lean reproduces
    oracle.bundle_camera_iteration's normal equations, solve_update_graph(camera=True) its update (all six coefficients damped,
    no l2 base), and the adjoint matches finite differences for the pose, the depth and the feature maps."""
    import torch
    from banet_amd import dense_train
    intr, lv, R, T, _, rng = _scene(seed=7)
    lv = {k: (np.asarray(v, np.float64) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    B, H, W, C = lv["src"].shape
    N = H * W
    a = _camera_inputs(intr, lv)
    W0 = np.zeros((B, 0, 1))
    mlp = orc.he_normal_mlp_weights(C, 5, np.float64)
    Rn, Tn, dbg = orc.bundle_camera_iteration(a["conv1"], a["conv2"], a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], R, T, mlp)
    F = adj.forward_lean(a, lv["tgt"], R, T, W0)
    for k in ("AtA", "Atb", "avg"):
        np.testing.assert_allclose(F[k], dbg[k], rtol=1e-10, atol=1e-10 * np.abs(dbg[k]).max())
    tt = lambda v: torch.from_numpy(np.asarray(v, np.float64))
    layers = [(tt(w), tt(b)) for w, b in mlp]
    R2, T2, W2 = dense_train.solve_update_graph(tt(dbg["AtA"]), tt(dbg["Atb"][..., 0]), tt(dbg["avg"][:, 0] * N), N, tt(R), tt(T),
                                                tt(W0), layers, 1.0, camera=True)
    np.testing.assert_allclose(R2.numpy(), Rn, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(T2.numpy(), Tn, rtol=1e-9, atol=1e-12)
    assert W2.shape == (B, 0, 1)
    # the small step's gradients exist for an empty coefficient vector
    g = dense_train._small_grads(tt(dbg["AtA"]), tt(dbg["Atb"][..., 0]), tt(dbg["avg"][:, 0] * N), tt(R).reshape(B, 1, 3, 3),
                                 tt(T).reshape(B, 1, 3, 1), tt(W0), torch.ones(B, 1, 3, 3, dtype=torch.float64),
                                 torch.ones(B, 1, 3, 1, dtype=torch.float64), torch.zeros(B, 0, 1, dtype=torch.float64),
                                 [x for wb in layers for x in wb], N, 1.0, camera=True)
    assert g[0].shape == (B, 6, 6) and g[5].shape == (B, 0, 1) and all(torch.isfinite(x).all() for x in g)

    G = rng.standard_normal((B, 6, 6))
    gb = rng.standard_normal((B, 6, 1))
    gavg = rng.standard_normal((B, 1, C))
    out = adj.assembly_adjoint(a, lv["tgt"], R, T, W0, G, gb, gavg)
    assert out["dbasis"].shape == (B, N, 0) and out["dW"].shape == (B, 0, 1)

    def phi(lv_, R_, T_):
        F_ = adj.forward_lean(_camera_inputs(intr, lv_), lv_["tgt"], R_, T_, W0)
        return float((G * F_["AtA"]).sum() + (gb * F_["Atb"]).sum() + (gavg * F_["avg"]).sum())

    def fd(apply, shape, eps):
        d = rng.standard_normal(shape)
        d /= np.linalg.norm(d)
        vals = sorted((apply(e * d) - apply(-e * d)) / (2 * e) for e in (eps, 3 * eps, 10 * eps))
        return d, vals[1]

    checks = []
    for key, gname, shape in (("src", "dsrc", (B, H, W, C)), ("tgt", "dtgt", (B, H, W, C)), ("D0", "dD0", (B, H, W))):
        def with_level(dl, key=key):
            l2 = dict(lv)
            l2[key] = lv[key] + dl
            return phi(l2, R, T)
        d, num = fd(with_level, shape, 3e-6)
        checks.append((key, num, float((out[gname].reshape(shape) * d).sum())))
    d, num = fd(lambda dl: phi(lv, R + dl, T), (B, 3, 3), 1e-7)
    checks.append(("R", num, float((out["dR"] * d).sum())))
    d, num = fd(lambda dl: phi(lv, R, T + dl), (B, 3, 1), 1e-7)
    checks.append(("T", num, float((out["dT"] * d).sum())))
    for name, num, ana in checks:
        assert abs(num - ana) <= 1e-4 * max(abs(num), abs(ana)) + 1e-7, (name, num, ana)
