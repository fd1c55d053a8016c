"""Host logic of the keyframe sequence driver and its oracle twin (no GPU needed)."""
import numpy as np

from oracle import banet_oracle as orc, sequence as oseq, synth


def test_sobel_matches_scipy_and_point_selection_follows_the_reference_rule():
    from scipy import ndimage
    from banet_amd import sequence
    rng = np.random.RandomState(0)
    img = rng.uniform(0, 255, (20, 30, 3)).astype(np.float32)
    want = np.stack([ndimage.sobel(img[..., c], axis=1, mode="mirror") for c in range(3)], -1)   # cv2 BORDER_REFLECT_101
    np.testing.assert_allclose(sequence.sobel_x(img), want, rtol=0, atol=1e-3)
    depth = rng.uniform(0.5, 3, (20, 30)).astype(np.float32)
    depth[:, :5] = 0.0                                            # invalid depth: never selected
    pts, d = sequence.valid_point_and_depth(img, depth, 200, 50.0, np.random.RandomState(1))
    assert pts.shape == (1, 200, 2) and d.shape == (1, 200, 1) and (d > 1e-5).all() and (pts[..., 0] >= 5).all()
    gx = sequence.sobel_x(img)
    mag = np.sqrt(2.0 * np.sum(gx * gx, -1))                      # the reference takes the x-derivative twice
    xi, yi = pts[0, :, 0].astype(int), pts[0, :, 1].astype(int)
    assert (mag[yi, xi] > 50.0).all()
    np.testing.assert_array_equal(d[0, :, 0], depth[yi, xi])


def test_oracle_sequence_recovers_motion_and_switches_keyframes():
    from banet_amd import sequence
    H, W, C, N = 96, 128, 8, 512
    poses = [((0, 0, 0), (0, 0, 0))] + [((0.004 * i, -0.003 * i, 0.002 * i), (0.02 * i, -0.012 * i, 0.008 * i)) for i in range(1, 5)]
    seq = synth.make_plane_sequence(H, W, C, poses, 3)
    mlps = {str(l): orc.he_normal_mlp_weights(C, 40 + l) for l in (1, 2, 3)}

    def select(i):
        return sequence.valid_point_and_depth(seq["images"][i], seq["depths"][i], N, 5.0, np.random.RandomState(100 + i))

    out = oseq.run_sequence(seq["intr"], seq["frames"], [0.0, 0.04, 0.08, 0.12, 0.16], select, mlps, [5, 8, 8])
    assert [o["new_keyframe"] for o in out] == [False, False, True, False]
    for i in (0, 1, 2):                                           # tracked against key frame 0
        assert np.abs(out[i]["translation"].ravel() - np.asarray(poses[i + 1][1])).max() < 2e-3
        assert np.abs(out[i]["rotation"][0] - synth.rodrigues(np.asarray(poses[i + 1][0], float))).max() < 1e-3
    # frame 4 is tracked against key frame 3: relative motion, chained into the global pose (seq_example.py:168-169)
    np.testing.assert_allclose(out[3]["globalRotation"], np.matmul(out[3]["rotation"], out[2]["globalRotation"]), atol=1e-6)
    assert all(1 <= c <= m for o in out for c, m in zip(o["iters"], [5, 8, 8]))


def test_lambda_weight_import_round_trip(tmp_path):
    """the reference's lambda-MLP variables (bundlenet.py:102-110: lambda_<level>_<i>_filters [1,Cin,Cout], _biases [Cout])
    import from / export to an .npz unchanged, scope prefixes and ':0' suffixes ignored; malformed sets are rejected"""
    import numpy as np
    import pytest
    import torch
    from banet_amd import bundlenet as bn
    C = 8
    lw = {"3": bn.he_normal_lambda_weights(C, 5), "0": bn.he_normal_lambda_weights(C, 6)}
    var = bn.lambda_weights_to_variables(lw)
    assert var["lambda_3_2_filters"].shape == (1, 2 * C, 4 * C) and var["lambda_0_5_biases"].shape == (1,)
    path = tmp_path / "lambda.npz"
    np.savez(path, **{"lambda_%s/%s:0" % (k.split("_")[1] + "_" + k.split("_")[2], k): v for k, v in var.items()})
    back = bn.lambda_weights_from_variables(dict(np.load(path)))
    assert sorted(back) == ["0", "3"]
    for lev in lw:
        for (w0, b0), (w1, b1) in zip(lw[lev], back[lev]):
            assert torch.equal(w0, w1) and torch.equal(b0, b1)
    del var["lambda_3_4_biases"]
    with pytest.raises(KeyError):
        bn.lambda_weights_from_variables(var)


def test_rotation_to_quaternion_and_tum_line():
    """the TUM trajectory line of seq_example.py:174-177: camera centre + quaternion (x, y, z, w) of the transposed global rotation"""
    import numpy as np
    from banet_amd import sequence as seq
    from oracle import synth
    rng = np.random.RandomState(3)
    for _ in range(50):
        w = rng.uniform(-1, 1, 3) * rng.choice([1e-4, 0.3, 3.0])
        R = synth.rodrigues(w).astype(np.float64)
        x, y, z, ww = seq.rotation_to_quaternion_xyzw(R)
        assert abs(x * x + y * y + z * z + ww * ww - 1) < 1e-12 and ww >= 0
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                       [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                       [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])
        assert np.abs(Rq - R).max() < 1e-6
    R180 = np.diag([1.0, -1.0, -1.0])                                   # trace = -1: the diagonal branch
    q = seq.rotation_to_quaternion_xyzw(R180)
    assert np.allclose(np.abs(q), [1, 0, 0, 0], atol=1e-12)
    line = seq.tum_line(1305031102.175304, [0.1, -0.2, 0.3], [0.0, 0.0, 0.0, 1.0])
    assert line.split() == ["1305031102.175304", "0.1", "-0.2", "0.3", "0.0", "0.0", "0.0", "1.0"]


def test_gt_consistent_point_selection_and_pose_errors():
    """legacy/eval.py:102-147 (vectorised here) against a direct restatement of the reference's double loop, and the
    error metrics of eval.py:225-234"""
    import numpy as np
    from banet_amd import sequence as seq
    from oracle import synth
    rng = np.random.RandomState(4)
    H, W = 24, 32
    K = np.array([[30.0, 30.0, W / 2.0, H / 2.0]])
    jj = np.arange(W, dtype=np.float32)[None, :, None] + np.zeros((H, 1, 3), np.float32)
    img1 = (128 + 100 * np.sin(0.25 * jj + np.array([0.0, 0.7, 1.9], np.float32))).astype(np.float32)   # smooth, strong x-gradients
    img1 += rng.rand(H, W, 3).astype(np.float32) * 4
    img2 = img1 + rng.randn(H, W, 3).astype(np.float32) * 30 * (rng.rand(H, W, 1) < 0.3)   # some pixels fail the colour test
    d1 = (1.5 + rng.rand(H, W)).astype(np.float32)
    d1[rng.rand(H, W) < 0.1] = 0.0
    d2 = (2.0 + 0.0 * d1) * (1 + 0.12 * rng.randn(H, W)).astype(np.float32)            # some fail the depth test
    R = synth.rodrigues(np.array([0.01, -0.02, 0.015])).astype(np.float64)
    t = np.array([0.02, -0.01, 0.03])

    def reference_loop():                                                   # eval.py:112-141 statement by statement
        dx = seq.sobel_x(img1)
        dxy = np.sqrt(np.sum(np.square(dx), axis=-1) + np.sum(np.square(dx), axis=-1))
        pts, deps = [], []
        k = K.flatten()
        for i in range(H):
            for j in range(W):
                if d1[i, j] < 1e-5 or dxy[i, j] < 80:
                    continue
                px, py = (j - k[2]) / k[0], (i - k[3]) / k[1]
                rot = np.matmul(R, d1[i, j] * np.reshape(np.asarray([px, py, 1.0]), [3, 1])).flatten() + t
                px, py = rot[0] / rot[2] * k[0] + k[2], rot[1] / rot[2] * k[1] + k[3]
                if int(py) < 0 or int(py) >= H or int(px) < 0 or int(px) >= W:
                    continue
                if np.linalg.norm(img1[i, j, :] - img2[int(py), int(px), :]) > 64:
                    continue
                if abs(rot[2] - d2[int(py), int(px)]) / rot[2] > 0.2:
                    continue
                pts.append([j, i])
                deps.append(d1[i, j])
        return np.asarray(pts, np.float32), np.asarray(deps, np.float32)

    want_p, want_d = reference_loop()
    assert 20 < len(want_p) < H * W
    num = 64
    got_p, got_d = seq.valid_point_and_depth2(img1, img2, d1, d2, R, t, K, num, np.random.RandomState(9))
    pick = np.random.RandomState(9).randint(0, len(want_p), num)
    np.testing.assert_array_equal(got_p[0], want_p[pick])
    np.testing.assert_array_equal(got_d[0, :, 0], want_d[pick])
    e = seq.pose_errors(R, t, R, t)
    assert e["rotation_error_deg"] < 1e-3 and e["translation_error"] == 0.0
    R2 = synth.rodrigues(np.array([0.0, 0.0, 0.1])).astype(np.float64)
    e = seq.pose_errors(R2, t, np.eye(3), t * 0)
    assert abs(e["rotation_error_deg"] - 0.1 * 180 / 3.14) < 1e-6 and abs(e["rotation_deg"] - 0.1 * 180 / 3.14) < 1e-6
    assert abs(e["translation_error"] - np.linalg.norm(t)) < 1e-12
