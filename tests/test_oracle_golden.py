"""The oracle is pinned against outputs of the REFERENCE's own Python (executed over
oracle/tf1_shim by tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import cases
from oracle import banet_oracle as orc

TOL = dict(rtol=2e-5, atol=2e-6)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _level_inputs(c):
    N = c["points"].shape[1]
    intr = c["intr"]
    fx0, fy0 = np.tile(intr[:, 0], (1, N)), np.tile(intr[:, 1], (1, N))
    ox0, oy0 = np.tile(intr[:, 2], (1, N)), np.tile(intr[:, 3], (1, N))
    p = orc.compute_coordinates(c["points"], fx0, fy0, ox0, oy0, normalize=False)
    s = np.float32(c["scale"])
    return p, fx0 / s, fy0 / s, ox0 / s, oy0 / s


def test_legacy_subfunctions(golden_dir):
    g = _g(golden_dir, "golden_legacy_ci2.npz")
    c = cases.case_legacy_ci2()
    p, fx, fy, ox, oy = _level_inputs(c)
    np.testing.assert_allclose(p, g["p"], **TOL)
    conv2 = orc.target_map(c["conv2_f"])
    np.testing.assert_array_equal(conv2, g["conv2"])
    samp, mask = orc.interpolate2d(conv2, g["px"], g["py"])
    np.testing.assert_array_equal(mask, g["mask"])
    assert 0 < mask.sum() < mask.size          # the case exercises both inside and outside
    np.testing.assert_allclose(samp, g["samp"], **TOL)
    np.testing.assert_allclose(orc.interpolate2d2(c["conv2_f"], c["points"] / np.float32(c["scale"])),
                               g["samp2"], **TOL)
    w = orc.warp(c["R"], c["T"], p, c["d"], fx, fy, ox, oy)
    np.testing.assert_allclose(w["px"], g["px"], rtol=1e-5, atol=1e-4)
    J = orc.camera_jacobian(w["x"], w["y"], w["Z"], fx, fy, +1)
    np.testing.assert_allclose(J, g["J"], rtol=1e-4, atol=1e-4)


def test_legacy_camera_iteration2(golden_dir):
    g = _g(golden_dir, "golden_legacy_ci2.npz")
    c = cases.case_legacy_ci2()
    p, fx, fy, ox, oy = _level_inputs(c)
    conv2 = orc.target_map(c["conv2_f"])
    R, T, uw, ut, ratio, _ = orc.legacy_camera_iteration2(c["conv1"], conv2, fx, fy, ox, oy, p, c["d"],
                                                          c["R"], c["T"], c["mlp"][c["level"]])
    np.testing.assert_allclose(R, g["R"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(T, g["T"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose([uw, ut, ratio], [g["uw"], g["ut"], g["ratio"]], rtol=1e-4)
    R1, T1, ratio1 = orc.legacy_camera_iteration(c["conv1"], conv2, fx, fy, ox, oy, p, c["d"], c["R"], c["T"])
    np.testing.assert_allclose(R1, g["R1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(T1, g["T1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ratio1, g["ratio1"], rtol=1e-6)


def test_legacy_track(golden_dir):
    g = _g(golden_dir, "golden_legacy_track.npz")
    c = cases.case_legacy_track()
    R, T, ratio, its = orc.legacy_track(c["intr"], c["layers"], c["points"], c["d"], c["R"], c["T"],
                                        c["iters"], c["mlp"], early_termination=True)
    assert list(its) == list(g["iters"])                        # iteration count identical
    np.testing.assert_allclose(R, g["R"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(T, g["T"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ratio, g["ratio"], rtol=1e-5)
    # and the BA actually recovers the scene's pose
    assert np.abs(R[0] - c["R_gt"]).max() < 2e-4 and np.abs(T[0, :, 0] - c["T_gt"]).max() < 5e-4
    R, T, ratio, its = orc.legacy_track(c["intr"], c["layers"], c["points"], c["d"], c["R"], c["T"],
                                        c["iters"], c["mlp"], early_termination=False)
    assert list(its) == c["iters"]
    np.testing.assert_allclose(R, g["Rs"][-1], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(T, g["Ts"][-1], rtol=1e-4, atol=1e-6)


def test_bundlenet_functions(golden_dir):
    g = _g(golden_dir, "golden_bundle_fns.npz")
    c = cases.case_bundle_fns()
    np.testing.assert_allclose(orc.camera_jacobian(c["x"], c["y"], c["Z"], c["fx"], c["fy"], -1), g["Jc"], **TOL)
    np.testing.assert_allclose(orc.depth_jacobian(c["r"][0][:, 0], c["r"][1][:, 0], c["r"][2][:, 0], c["x"],
                                                  c["y"], c["Z"], c["fx"], c["fy"]), g["jd"], **TOL)
    np.testing.assert_allclose(orc.angle_axis_rotation(c["w1"], True), g["rot1"], **TOL)
    np.testing.assert_allclose(orc.angle_axis_rotation(c["w2"], True), g["rot2"], **TOL)
    np.testing.assert_allclose(orc.angle_axis_rotation(c["w1"], False), g["rotL"], **TOL)
    np.testing.assert_allclose(orc.vmatrix(c["w1"]), g["V1"], **TOL)
    np.testing.assert_allclose(orc.vmatrix(c["w1"]), g["VL"], **TOL)
    orc.VMATRIX_REFERENCE_BATCH_LAYOUT = True
    try:
        np.testing.assert_allclose(orc.vmatrix(c["w2"]), g["V2bug"], **TOL)
    finally:
        orc.VMATRIX_REFERENCE_BATCH_LAYOUT = False
    assert np.abs(orc.vmatrix(c["w2"]) - g["V2bug"]).max() > 1e-3      # the B>1 layout really differs
    np.testing.assert_allclose(orc.rotation2quaternion(c["Rm"]), g["q"], **TOL)
    np.testing.assert_array_equal(orc.grad_fixed(c["img"]), g["g"])
    B, N = c["x"].shape
    p = orc.compute_coordinates(c["pts"], c["fx"], c["fy"], np.full((B, N), 16.0, np.float32),
                                np.full((B, N), 10.0, np.float32), normalize=True)
    np.testing.assert_allclose(p, g["p"], **TOL)


def test_bundle_iterations(golden_dir):
    g = _g(golden_dir, "golden_bundle_iter.npz")
    c = cases.case_bundle_iter()
    a = [c[k] for k in ("conv1", "conv2", "fx", "fy", "ox", "oy", "p", "D")]
    R, T, _ = orc.bundle_camera_iteration(*a, c["R"], c["T"], c["mlp"][c["level"]], 1.0)
    np.testing.assert_allclose(R, g["Rc"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(T, g["Tc"], rtol=1e-4, atol=1e-6)
    R, T, W, _ = orc.bundle_iteration(*a, c["Bs"], c["R"], c["T"], c["W"], c["mlp"][c["level"]], 1000.0)
    np.testing.assert_allclose(R, g["R"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(T, g["T"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(W, g["W"], rtol=1e-4, atol=1e-6)


def test_resize_drivers(golden_dir):
    g = _g(golden_dir, "golden_resize.npz")
    c = cases.case_resize()
    orc.VMATRIX_REFERENCE_BATCH_LAYOUT = True       # goldens are B=2: replay the reference's layout
    try:
        Rs, Ts = orc.camera_resize(c["intr"], c["layers"], c["points"], c["depth"], c["mlp"])
        Rb, Tb, Db = orc.bundle_resize(c["intr"], c["layers"], c["points"], c["basis"], c["depth"], c["mlp"],
                                       init_rotation=Rs[-1], init_translation=Ts[-1])
    finally:
        orc.VMATRIX_REFERENCE_BATCH_LAYOUT = False
    np.testing.assert_allclose(np.stack(Rs), g["Rs"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(np.stack(Ts), g["Ts"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(np.stack(Rb), g["Rb"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(np.stack(Tb), g["Tb"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(np.stack([d[:, ::8, ::8, 0] for d in Db]), g["Db"], rtol=1e-4, atol=1e-5)
