"""Round-4 GPU parity tests (`-m gpu`, all through the C ABI of libbanet_hip.so).

* banet_ba_assemble_mask_f32: the per-pixel in-image mask bits of every gather kernel (generic / direct / patch / strip, two-frame
  and multi-frame windows) -- same sums as banet_ba_assemble_f32 bit for bit, the bits add up to nvalid, agree between the kernels
  and with the float64 oracle's mask (bundlenet.py:155,231);
* the float64 twin evaluated with a forced mask (oracle/torch_port.py, mask_override) reproduces its own result when handed its
  own mask, and moves by one pixel's worth when one bit is flipped -- the mechanism bench.py's sweep gate relies on.
"""
import numpy as np
import pytest
import torch

from oracle import banet_oracle as orc, torch_port

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
GENERIC, DIRECT, PATCH, STRIP, STRIP_PAIR_LOOP = 32, 64 | 524288, 512, 262144, 262144 | (1 << 22)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from banet_amd import _capi
    _capi.lib()


def n(x):
    return x.detach().cpu().numpy()


def t(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).to(DEV)


@pytest.mark.parametrize("H,W,K,pairs,big", [(48, 64, 32, 1, True), (40, 56, 128, 3, True), (37, 53, 0, 2, True), (64, 96, 64, 4, False)])
def test_mask_output_of_every_gather_kernel(H, W, K, pairs, big):
    from banet_amd import dense as bdense, ops
    from oracle import dense as odense, synth
    B, C = 2, 128
    scenes = [synth.make_window_scene(H, W, C, K, [1], 500 + b, pairs, rot_mag=0.012 * (6 if big else 1),
                                      trans_mag=0.05 * (6 if big else 1)) for b in range(B)]
    intr, levels = odense.batch_window_scene(scenes)
    lv = levels[0]
    rng = np.random.RandomState(3)
    R = np.stack([[synth.rodrigues(rng.uniform(-1, 1, 3) * 0.004) for _ in range(pairs)] for _ in range(B)]).astype(np.float32)
    T = (np.stack([s["T_gt"] for s in scenes]) * 0.8).reshape(B, pairs, 3, 1).astype(np.float32)
    Wc = (rng.standard_normal((B, K, 1)) * 0.01).astype(np.float32)
    mlps = [orc.he_normal_mlp_weights(C, 9)]
    tl = [bdense.DenseLevel(l["scale"], t(l["src"]), t(l["tgt"]), t(l["D0"]), t(l["basis"]) if l["basis"].shape[-1] else None)
          for l in levels]
    ba = bdense.DenseBA(t(intr), tl, mlps, "bundle" if K else "bundle_camera", 1000.0)
    masks = {}
    for bits in (GENERIC, DIRECT, PATCH, STRIP) + ((STRIP_PAIR_LOOP,) if pairs > 1 else ()):
        ba.problems[0].c.reserved_ = bits
        plain = ops.ba_assemble(ba.problems[0], t(R), t(T), t(Wc) if K else None)
        withm = ops.ba_assemble(ba.problems[0], t(R), t(T), t(Wc) if K else None, return_mask=True)
        for x, y in zip(plain, withm[:4]):                         # writing the mask changes no sum
            assert torch.equal(x, y)
        m = n(withm[4])
        assert m.shape == (B, pairs, H * W) and m.max() <= 1       # every pixel written (the buffer starts at 255)
        np.testing.assert_array_equal(m.reshape(B, -1).sum(1).astype(np.float32), n(withm[3]))
        masks[bits] = m
    ba.problems[0].c.reserved_ = 0
    for bits, m in masks.items():
        np.testing.assert_array_equal(m, masks[DIRECT])             # the same float32 geometry in every kernel
    # against the float64 oracle's mask: identical except for pixels on the border (none expected at these seeds: <= 1 tolerated)
    one = dict(lv)
    one["tgt"] = lv["tgt"][:, 0]
    a = odense.level_inputs(intr, one, True, np.float64)
    for i in range(pairs):
        if K:
            conv2s = [orc.target_map(lv["tgt"][:, j].astype(np.float64)) for j in range(pairs)]
            dbg = orc.bundle_window_iteration(a["conv1"], conv2s, a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"],
                                              [R[:, j].astype(np.float64) for j in range(pairs)],
                                              [T[:, j].astype(np.float64) for j in range(pairs)], Wc.astype(np.float64), mlps[0], 1000.0)[3]
            m64 = np.stack([np.asarray(mm).reshape(B, -1) for mm in dbg["mask"]], 1)
            assert (m64 > 0).sum() < B * pairs * H * W or not big
            assert ((m64 > 0) != (masks[STRIP] > 0)).sum() <= 1
            break


def test_twin_with_forced_mask():
    """torch_port.window_iteration(mask_override=...): its own mask reproduces its own result; one flipped bit moves lambda by
    about one pixel's worth (what a float32 / float64 disagreement on a border pixel does)."""
    from banet_amd import synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    B, H, W, C, K, pairs = 1, 60, 80, 128, 32, 2
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 77, DEV, trans_mag=0.3, pairs=pairs)
    lv = levels[0]
    mlp = [(n(w_), n(b_)) for w_, b_ in he_normal_lambda_weights(C, 100)]
    R = torch.eye(3, device=DEV).repeat(B, pairs, 1, 1)
    T = (gt["T"] * 0.7).reshape(B, pairs, 3, 1).to(DEV)
    Wc = torch.zeros(B, K, 1, device=DEV)
    args = (intr, lv.scale, lv.src, lv.tgt, lv.depth, lv.basis, R, T, Wc, mlp, 1000.0)
    *_, d = torch_port.window_iteration(*args)
    m = d["mask"]
    assert m.shape == (B, pairs, H * W) and 0 < int(m.sum()) < m.numel()
    *_, d2 = torch_port.window_iteration(*args, mask_override=m.to(torch.uint8))
    assert torch.equal(d["solution"], d2["solution"]) and torch.equal(d["lam"], d2["lam"])
    assert torch.equal(d2["mask"], m)                               # "mask" stays the evaluation's OWN mask
    m2 = m.clone()
    idx = int(m2[0, 0].nonzero()[0])
    m2[0, 0, idx] = False
    *_, d3 = torch_port.window_iteration(*args, mask_override=m2.to(torch.uint8))
    dl = float((d3["lam"] - d["lam"]).abs() / d["lam"].abs())
    assert 0 < dl < 50.0 / (H * W * pairs)
