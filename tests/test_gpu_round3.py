"""Round-3 GPU parity tests (`-m gpu`, all through the C ABI of libbanet_hip.so).

* BASELINE.json's multi-frame configurations at THEIR full size, with the production kernel selection (no forced
  switches): one 640x480 level of cfg-3's 5-frame windows (C = K = 128, P = 152) and one 1280x960 level of cfg-5's
  8-frame windows (K = 256, P = 298) against the float64 twin of the oracle's window iteration
  (oracle/torch_port.window_iteration, pinned to banet_oracle.bundle_window_iteration on the CPU in
  tests/test_torch_ref_cpu.py): normal equations, lambda, and ONE update from an identical state per coefficient group;
* the SYRK's bf16x6 emulation with basis columns spanning 2^12 in magnitude (a dropped cross term of the split shows up
  as an error of 2^-16 .. 2^-8 of an entry's own scale, far above the gate);
* the losses and rotation2quaternion of bundlenet.py:6-15,401-463 ON THE GPU against the reference's own golden output;
* the second scene of bench.py's in-line parity record (another seed, stronger motion).
"""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import banet_oracle as orc, torch_port

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from banet_amd import _capi
    _capi.lib()                                   # fail loudly if the HIP library is missing


def n(x):
    return x.detach().cpu().numpy()


def relerr(got, want):
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def _window_level_check(B, H, W, C, K, pairs, seed, tol_step=1e-4, flags=0, expect_sel=None, last_yardstick=False):
    """one full-resolution level of B multi-frame windows: assembly and one LM update vs the float64 twin.
    flags: banet_level_t.flags for the level (forces a kernel selection); expect_sel: the (gather, SYRK) selection asserted;
    last_yardstick: the undamped last depth coefficient (bundlenet.py:266: `[diag[:-1] + 1e-5, 0] * lambda`) is the quotient of two
    cancelling sums once the other coefficients have converged -- gate it like bench.py does, at max(tol_step, 2 x what the SAME
    statements lose in float32 at this state) instead of tol_step alone (the yardstick is printed)"""
    from banet_amd import dense as bdense, ops, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    torch.manual_seed(seed)
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], seed, DEV, trans_mag=0.06, pairs=pairs)
    lv = levels[0]
    mlps = [he_normal_lambda_weights(C, 100)]
    ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1000.0)
    assert ba.pairs == pairs and ba.problems[0].P == 6 * pairs + K
    ba.problems[0].c.flags = flags
    if expect_sel is not None:
        assert (ops.gather_selection(ba.problems[0]), ops.syrk_selection(ba.problems[0])) == tuple(expect_sel)
    g = torch.Generator().manual_seed(seed + 1)
    R = torch.stack([torch.stack([bsynth._rodrigues((torch.rand(3, generator=g) * 2 - 1) * 0.003) for _ in range(pairs)])
                     for _ in range(B)]).to(DEV)
    T = (gt["T"] * 0.7).reshape(B, pairs, 3, 1).to(DEV)
    Wc = (torch.randn(B, K, 1, generator=g) * 0.004).to(DEV)
    AtA, Atb, absres, nvalid = ops.ba_assemble(ba.problems[0], R, T, Wc)
    st = ba.step_from(0, R.reshape(B * pairs, 3, 3).clone(), T.reshape(B * pairs, 3, 1).clone(), Wc.clone())
    torch.cuda.synchronize()
    out = []
    for b in range(B):                              # one window at a time (float64 memory)
        sl = slice(b, b + 1)
        R2, T2, W2, d = torch_port.window_iteration(intr[sl], lv.scale, lv.src[sl], lv.tgt[sl], lv.depth[sl], lv.basis[sl],
                                                    R[sl], T[sl], Wc[sl], [(n(w), n(bb)) for w, bb in mlps[0]], 1000.0)
        A64, b64 = n(d["AtA"][0]), n(d["Atb"][0])
        eA, eb = relerr(n(AtA[b]), A64), relerr(n(Atb[b]), b64)
        assert eA < 3e-5 and eb < 3e-5, (b, eA, eb)
        got = n(AtA[b])
        np.testing.assert_array_equal(got, got.T)
        sol = n(d["solution"][0])
        dl = n(st.delta[b])
        o = 6 * pairs
        e_lam = relerr(n(st.lambda_out[b:b + 1]), n(d["lam"]))
        e_pose, e_depth, e_last = relerr(dl[:o], sol[:o]), relerr(dl[o:-1], sol[o:-1]), relerr(dl[-1:], sol[-1:])
        tol_last = tol_step
        if last_yardstick:
            *_, d32 = torch_port.window_iteration(intr[sl], lv.scale, lv.src[sl], lv.tgt[sl], lv.depth[sl], lv.basis[sl], R[sl], T[sl],
                                                  Wc[sl], [(n(w), n(bb)) for w, bb in mlps[0]], 1000.0, dtype=torch.float32)
            ref32_last = relerr(n(d32["solution"][0])[-1:], sol[-1:])
            tol_last = max(tol_step, 2.0 * ref32_last)
            del d32
            out.append(("float32 twin's own error on the last coefficient", ref32_last))
        out.append((eA, eb, e_lam, e_pose, e_depth, e_last))
        assert e_lam < 1e-4 and e_pose < tol_step and e_depth < tol_step and e_last < tol_last, (b, out[-2:])
        assert relerr(n(st.R[b]), n(R2[0])) < 1e-5 and relerr(n(st.T[b]), n(T2[0])) < 1e-4 and relerr(n(st.Wc[b]), n(W2[0])) < 1e-4
        del d, R2, T2, W2
        torch.cuda.empty_cache()
    print("window level %dx%d K=%d pairs=%d: (AtA, Atb, lam, pose, depth, last) =" % (W, H, K, pairs), out)
    return ba


def test_cfg3_full_size_level_matches_float64_twin():
    """configs[2]: 5-frame windows, 640x480, C = K = 128 -- the finest level at full size on the kernels a TWO-window launch
    selects by itself (patch gather with the pair loop inside a tile, exact bf16 SYRK).  The batch-32 production selection
    (frame-parallel strip gather + fp16 two-piece SYRK) at full size: tests/test_gpu_round5.py."""
    ba = _window_level_check(2, 480, 640, 128, 128, 4, 4711)
    assert ba.problems[0].N == 640 * 480


def test_cfg5_full_size_level_matches_float64_twin():
    """configs[4]: 8-frame windows, 1280x960, K = 256 (P = 298) -- the finest level at full size (K = 256 patch gather,
    syrk_wide.hip jobs, the solve with its matrix in the caller's workspace)."""
    ba = _window_level_check(1, 960, 1280, 128, 256, 7, 4712)
    assert ba.problems[0].N == 1280 * 960 and ba.problems[0].P == 298


@pytest.mark.parametrize("K,H,W", [(128, 96, 128), (64, 48, 64)])
def test_syrk_bf16x6_with_basis_columns_spanning_4096x(K, H, W):
    """The depth block runs on the bf16 pipe as six products of an exact three-way split (syrk.hip).  A dropped or
    mis-scaled cross term (hi x mid, hi x lo, mid x mid) is relative to the PRODUCT of two entries' own scales, so basis
    columns of very different magnitude expose it: column k is scaled by 2^(12 k / (K-1)).  Every entry of AtA / Atb must
    agree with the float64 twin to 3e-5 of sqrt(A_ii A_jj) -- its own scale, not the matrix maximum."""
    from banet_amd import dense as bdense, ops, synth as bsynth
    B, C = 2, 128
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 91, DEV, trans_mag=0.06)
    lv = levels[0]
    scale = torch.pow(2.0, 12.0 * torch.arange(K, dtype=torch.float64) / (K - 1)).to(torch.float32).to(DEV)
    perm = torch.randperm(K, generator=torch.Generator().manual_seed(5)).to(DEV)
    lv.basis = (lv.basis * scale[perm]).contiguous()          # large and small columns interleaved across the 16-wide blocks
    ba = bdense.DenseBA(intr, levels, [orc.he_normal_mlp_weights(C, 5)], "bundle", 1000.0)
    R = torch.eye(3, device=DEV).repeat(B, 1, 1)
    T = (gt["T"] * 0.7).reshape(B, 3, 1).to(DEV)
    Wc = torch.zeros(B, K, 1, device=DEV)
    AtA, Atb, absres, nvalid = ops.ba_assemble(ba.problems[0], R, T, Wc)
    for b in range(B):
        r = torch_port.dense_assemble(intr[b:b + 1], lv.scale, lv.src[b:b + 1], lv.tgt[b:b + 1], lv.depth[b:b + 1],
                                      lv.basis[b:b + 1], R[b:b + 1], T[b:b + 1], Wc[b:b + 1], True, True)
        A64, b64 = n(r[0][0]), n(r[1][0])
        dg = np.sqrt(np.maximum(np.diag(A64), 1e-300))
        eA = np.abs(n(AtA[b]).astype(np.float64) - A64) / np.outer(dg, dg)
        assert eA.max() < 3e-5, (b, eA.max(), np.unravel_index(eA.argmax(), eA.shape))
        # Atb_i against its Cauchy-Schwarz scale sqrt(A_ii) * |residual|
        rs = float(np.sqrt((n(absres[b]).astype(np.float64) ** 2).sum()))
        eb = np.abs(n(Atb[b]).astype(np.float64) - b64) / (dg * max(rs, 1e-30))
        assert eb.max() < 3e-5, (b, eb.max())
        assert dg[6:].max() / dg[6:].min() > 1000.0           # the dynamic range really is in the matrix


def test_opt_in_three_product_syrk_is_a_separate_less_exact_path():
    """reserved_ bit 29 (ops.SYRK_THREE_PRODUCTS): the K = 128 SYRK with the three largest of the six bf16 products.  It is NOT the
    product path: the default leaves the bits of AtA untouched, the opt-in differs from it -- by no more than the two-piece split
    allows (each entry within 1e-4 of sqrt(A_ii A_jj), against 3e-5 vs float64 for the exact form) -- and one LM step from it stays
    within the headline tolerance of the float64 twin."""
    from banet_amd import dense as bdense, ops, synth as bsynth
    B, C, K, H, W = 2, 128, 128, 120, 160
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [1], 23, DEV, trans_mag=0.06)
    ba = bdense.DenseBA(intr, levels, [orc.he_normal_mlp_weights(C, 5)], "bundle", 1000.0)
    p = ba.problems[0]
    R = torch.eye(3, device=DEV).repeat(B, 1, 1)
    T = (gt["T"] * 0.7).reshape(B, 3, 1).to(DEV)
    Wc = torch.zeros(B, K, 1, device=DEV)
    base = [x.clone() for x in ops.ba_assemble(p, R, T, Wc)]
    again = [x.clone() for x in ops.ba_assemble(p, R, T, Wc)]
    p.c.flags = ops.SYRK_THREE_PRODUCTS
    fast = [x.clone() for x in ops.ba_assemble(p, R, T, Wc)]
    st_fast = ba.step_from(0, R, T, Wc)
    p.c.flags = 0
    st_base = ba.step_from(0, R, T, Wc)
    assert all(torch.equal(a, b) for a, b in zip(base, again))
    assert not torch.equal(base[0], fast[0])                         # a different kernel ran
    assert torch.equal(base[0][:, :6, :6], fast[0][:, :6, :6])       # the pose block comes from the gather: untouched
    for b in range(B):
        A, F = n(base[0][b]).astype(np.float64), n(fast[0][b]).astype(np.float64)
        dg = np.sqrt(np.maximum(np.diag(A), 1e-300))
        assert (np.abs(F - A) / np.outer(dg, dg)).max() < 1e-4
    d = (st_fast.delta - st_base.delta).abs().max() / st_base.delta.abs().max()
    assert float(d) < 1e-4, float(d)


def test_losses_and_quaternion_on_the_gpu_match_the_reference(golden_dir):
    """SURVEY 8(f) rank 4 (bundlenet.py:6-15,401-463) on cuda:0 against the reference's own output
    (tests/golden/golden_losses.npz, golden_bundle_fns.npz), and their gradients against the CPU result."""
    from banet_amd.bundlenet import BundleNet, rotation2quaternion
    c = cases.case_losses()
    g = np.load(os.path.join(golden_dir, "golden_losses.npz"))
    tt = {k: torch.from_numpy(v).to(DEV) for k, v in c.items()}
    net = BundleNet()
    np.testing.assert_allclose(float(net.lossR(tt["predQ"], tt["gtQ"])), g["lossR"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(float(net.lossT(tt["predT"], tt["gtT"])), g["lossT"], rtol=1e-6)
    lf = net.lossF(tt["intr"], tt["depth"], tt["mask"], tt["predR"], tt["predT"], tt["gtR"], tt["gtT"])
    assert lf.device.type == "cuda"
    np.testing.assert_allclose(float(lf), g["lossF"], rtol=2e-5)
    q = rotation2quaternion(tt["predR"])
    assert q.device.type == "cuda"
    np.testing.assert_allclose(n(q), orc.rotation2quaternion(c["predR"]), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(n((q * q).sum(1)), 1.0, rtol=1e-5)
    # gradients on the GPU == gradients on the CPU
    grads = {}
    for dev in ("cpu", DEV):
        t2 = {k: torch.from_numpy(v).to(dev) for k, v in c.items()}
        pT, pR = t2["predT"].clone().requires_grad_(True), t2["predR"].clone().requires_grad_(True)
        loss = net.lossF(t2["intr"], t2["depth"], t2["mask"], pR, pT, t2["gtR"], t2["gtT"]) + net.lossT(pT, t2["gtT"]) \
            + net.lossR(rotation2quaternion(pR), t2["gtQ"])
        loss.backward()
        grads[dev] = (n(pT.grad), n(pR.grad))
    np.testing.assert_allclose(grads[DEV][0], grads["cpu"][0], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(grads[DEV][1], grads["cpu"][1], rtol=2e-4, atol=1e-7)


# ======================================================================================
# the strip gather (gather128s.hip)
# ======================================================================================
def t(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).to(DEV)


def _torch_levels(levels):
    from banet_amd import dense as bdense
    return [bdense.DenseLevel(lv["scale"], t(lv["src"]), t(lv["tgt"]), t(lv["D0"]),
                              t(lv["basis"]) if lv["basis"].shape[-1] else None) for lv in levels]


STRIP, DIRECT, STRIP_ALL_DIRECT, SEG32 = 262144, 64 | 524288, 262144 | (1 << 20), 1 << 21
PAIR_LOOP = 1 << 22      # multi-frame strip gather: frames looped over inside one wave (A/B)
SEG8 = 1024              # with STRIP: 8-row segments (mid-size two-frame launches)


@pytest.mark.parametrize("H,W,K,big,pairs", [(48, 64, 128, False, 1),      # whole segments, unit-scale footprints: all from the window
                                             (40, 56, 16, True, 1),        # large motion: fallback rows, rim, masked pixels
                                             (37, 53, 0, True, 2),         # ragged strips and segments, pose only, 2 target frames
                                             (64, 96, 64, False, 3),       # 3 target frames share the depth dot
                                             (50, 70, 128, True, 4),       # cfg-3's window: 4 waves per segment, large motion
                                             (36, 48, 256, False, 7),      # cfg-5's window: 7 waves per segment, K = 256
                                             (70, 45, 256, False, 1),      # K = 256 (two basis chunks per row)
                                             (33, 21, 32, False, 1)])      # the narrowest map the window fits (21 texels)
def test_strip_gather_kernel_matches_oracle(H, W, K, big, pairs):
    """ba_gather128s_kernel forced at oracle-sized inputs, against the direct C = 128 kernel (same arithmetic per pixel, the
    channel sums added slice by slice), against itself with every pixel row on the window-less path, and against the
    float64 oracle."""
    from banet_amd import dense as bdense, ops
    from oracle import dense as odense, synth
    B, C = 2, 128
    scenes = [synth.make_window_scene(H, W, C, K, [1], 300 + b, pairs, rot_mag=0.012 * (6 if big else 1),
                                      trans_mag=0.05 * (6 if big else 1)) for b in range(B)]
    intr, levels = odense.batch_window_scene(scenes)
    lv = levels[0]
    rng = np.random.RandomState(8)
    R = np.stack([[synth.rodrigues(rng.uniform(-1, 1, 3) * 0.004) for _ in range(pairs)] for _ in range(B)]).astype(np.float32)
    T = (np.stack([s["T_gt"] for s in scenes]) * 0.8).reshape(B, pairs, 3, 1).astype(np.float32)
    Wc = (rng.standard_normal((B, K, 1)) * 0.01).astype(np.float32)
    mlps = [orc.he_normal_mlp_weights(C, 9)]
    ba = bdense.DenseBA(t(intr), _torch_levels(levels), mlps, "bundle" if K else "bundle_camera", 1000.0)
    outs = {}
    variants = (STRIP, DIRECT, STRIP_ALL_DIRECT, STRIP | SEG32, STRIP | SEG8) + ((STRIP | PAIR_LOOP,) if pairs > 1 else ())
    for bits in variants:
        ba.problems[0].c.flags = bits
        assert ops.gather_selection(ba.problems[0]) == (1 if bits == DIRECT else 3)
        outs[bits] = [n(x) for x in ops.ba_assemble(ba.problems[0], t(R), t(T), t(Wc) if K else None)]
        again = [n(x) for x in ops.ba_assemble(ba.problems[0], t(R), t(T), t(Wc) if K else None)]
        for x, y in zip(outs[bits], again):                       # bit-reproducible run to run
            np.testing.assert_array_equal(x, y)
    ba.problems[0].c.flags = 0
    for name, x, y in zip(("AtA", "Atb", "absres", "nvalid"), outs[STRIP], outs[DIRECT]):
        assert relerr(x, y) < 3e-6, (name, relerr(x, y))
    for name, x, y in zip(("AtA", "Atb", "absres", "nvalid"), outs[STRIP_ALL_DIRECT], outs[STRIP]):
        assert relerr(x, y) < 1e-6, (name, relerr(x, y))          # window and direct taps read the same texels
    for name, x, y in zip(("AtA", "Atb", "absres", "nvalid"), outs[STRIP | SEG32], outs[STRIP]):
        assert relerr(x, y) < 3e-6, (name, relerr(x, y))          # 32-row segments: the same sums, other partial rows
    for name, x, y in zip(("AtA", "Atb", "absres", "nvalid"), outs[STRIP | SEG8], outs[STRIP]):
        assert relerr(x, y) < 3e-6, (name, relerr(x, y))          # 8-row segments likewise
    np.testing.assert_array_equal(outs[STRIP][3], outs[DIRECT][3])
    if pairs > 1:     # frame-parallel workgroups (the default) == the frames looped over inside one wave, bit for bit
        for x, y in zip(outs[STRIP], outs[STRIP | PAIR_LOOP]):
            np.testing.assert_array_equal(x, y)
    one = dict(lv)
    one["tgt"] = lv["tgt"][:, 0]
    a = odense.level_inputs(intr, one, True, np.float64)
    R64 = [R[:, i].astype(np.float64) for i in range(pairs)]
    T64 = [T[:, i].astype(np.float64) for i in range(pairs)]
    if K:
        conv2s = [orc.target_map(lv["tgt"][:, i].astype(np.float64)) for i in range(pairs)]
        dbg = orc.bundle_window_iteration(a["conv1"], conv2s, a["fx"], a["fy"], a["ox"], a["oy"], a["p"], a["D"], a["Bs"],
                                          R64, T64, Wc.astype(np.float64), mlps[0], 1000.0)[3]
        assert relerr(outs[STRIP][0], dbg["AtA"]) < 3e-5 and relerr(outs[STRIP][1][..., None], dbg["Atb"]) < 3e-5
        nv = sum(m.sum(axis=(1, 2)) for m in dbg["mask"])
        assert np.abs(outs[STRIP][3] - nv).max() <= 1
        assert relerr(outs[STRIP][2] / (H * W * pairs), dbg["avg"][:, 0]) < 1e-5
        if big:
            assert (nv < H * W * pairs).all()                     # some pixels really leave the image
    else:
        for i in range(pairs):
            d = orc.bundle_camera_iteration(a["conv1"], orc.target_map(lv["tgt"][:, i].astype(np.float64)), a["fx"], a["fy"],
                                            a["ox"], a["oy"], a["p"], a["D"], R64[i], T64[i], mlps[0], 1.0)[2]
            assert relerr(outs[STRIP][0][:, 6 * i:6 * i + 6, 6 * i:6 * i + 6], d["AtA"]) < 3e-5


def test_strip_gather_full_size_batch_selection_and_twin():
    """The production selection: at the metric's batch (32 windows) the 640x480 level runs the strip gather; window 0 and
    window 31 of the batch against the float64 twin (normal equations and one LM update), and a whole 5-level solve of the
    batch is bit-identical run to run."""
    from banet_amd import dense as bdense, ops, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    B, H, W, C, K = 32, 480, 640, 128, 128
    torch.manual_seed(5)
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [4, 1], 990, DEV, trans_mag=0.06)
    mlps = [he_normal_lambda_weights(C, 100 + i) for i in range(2)]
    ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1000.0)
    assert ops.gather_selection(ba.problems[1]) == 3
    R = torch.eye(3, device=DEV).repeat(B, 1, 1)
    T = (gt["T"] * 0.7).reshape(B, 3, 1).to(DEV)
    Wc = torch.zeros(B, K, 1, device=DEV)
    lv = levels[1]
    AtA, Atb, absres, nvalid = ops.ba_assemble(ba.problems[1], R, T, Wc)
    st = ba.step_from(1, R.clone(), T.clone(), Wc.clone())
    for b in (0, 31):
        sl = slice(b, b + 1)
        R2, T2, W2, d = torch_port.window_iteration(intr[sl], lv.scale, lv.src[sl], lv.tgt[sl].unsqueeze(1), lv.depth[sl], lv.basis[sl],
                                                    R[sl].unsqueeze(1), T[sl].unsqueeze(1), Wc[sl], [(n(w_), n(b_)) for w_, b_ in mlps[1]], 1000.0)
        assert relerr(n(AtA[b]), n(d["AtA"][0])) < 3e-5 and relerr(n(Atb[b]), n(d["Atb"][0])) < 3e-5
        assert float(nvalid[b]) == float(d["nvalid"][0])
        sol, dl = n(d["solution"][0]), n(st.delta[b])
        assert relerr(dl[:6], sol[:6]) < 1e-4 and relerr(dl[6:-1], sol[6:-1]) < 1e-4 and relerr(dl[-1:], sol[-1:]) < 1e-4
        assert relerr(n(st.lambda_out[b:b + 1]), n(d["lam"])) < 1e-4
    T0 = (gt["T"] * 0.7).reshape(B, 3, 1).to(DEV)
    a = ba.solve([3, 3], ba.new_state(T=T0.clone()))[0]
    ra, ta, wa = a.R.clone(), a.T.clone(), a.Wc.clone()
    b2 = ba.solve([3, 3], ba.new_state(T=T0.clone()))[0]
    assert torch.equal(ra, b2.R) and torch.equal(ta, b2.T) and torch.equal(wa, b2.Wc)


# ======================================================================================
# the conjugate-gradient solve of the damped systems (solve.hip::pcg_schur_solve) and its LDL^T fallback
# ======================================================================================
@pytest.mark.parametrize("l2_base,pairs", [(1000.0, 1), (30.0, 1), (1.0, 1), (1e-3, 1), (1000.0, 3), (0.05, 3)])
def test_cg_solve_and_ldlt_fallback_match_the_oracle_lu(l2_base, pairs):
    """bundlenet.py:264-267: tf.matrix_solve on the damped normal matrix.  The update kernel runs Jacobi-preconditioned
    conjugate gradients on the damped block + an exact Schur step for the undamped last coefficient, and falls back to the
    blocked LDL^T when CG does not converge (weak damping: small l2_regularizer_base).  From strong damping (the layer's
    own 1000: CG in a handful of products) to practically none (fallback): the update equals the float64 LU solution of the
    same damped system to 1e-4 per coefficient group, and equals what the LDL^T alone gives (reserved_ bit 23)."""
    from banet_amd import dense as bdense, ops
    from oracle import dense as odense, synth
    B, H, W, C, K = 2, 40, 56, 128, 64
    scenes = [synth.make_window_scene(H, W, C, K, [1], 410 + b, pairs, rot_mag=0.012, trans_mag=0.05) for b in range(B)]
    intr, levels = odense.batch_window_scene(scenes)
    rng = np.random.RandomState(3)
    R = np.stack([[synth.rodrigues(rng.uniform(-1, 1, 3) * 0.004) for _ in range(pairs)] for _ in range(B)]).astype(np.float32)
    T = (np.stack([s["T_gt"] for s in scenes]) * 0.8).reshape(B, pairs, 3, 1).astype(np.float32)
    Wc = (rng.standard_normal((B, K, 1)) * 0.01).astype(np.float32)
    mlps = [orc.he_normal_mlp_weights(C, 9)]
    ba = bdense.DenseBA(t(intr), _torch_levels(levels), mlps, "bundle", l2_base)
    AtA, Atb, absres, nvalid = ops.ba_assemble(ba.problems[0], t(R), t(T), t(Wc))
    sols = {}
    for bits in (0, 1 << 23):
        ba.problems[0].c.flags = bits
        st = ba.new_state(t(R.reshape(B * pairs, 3, 3)), t(T.reshape(B * pairs, 3, 1)), t(Wc))
        ops.ba_solve_update(ba.problems[0], ba.mlps[0], l2_base, AtA, Atb, absres, nvalid, st)
        sols[bits] = (n(st.delta).astype(np.float64), float(n(st.lambda_out)[0]))
    ba.problems[0].c.flags = 0
    o = 6 * pairs
    for b in range(B):
        lam = float(n(st.lambda_out)[b])
        A64 = n(AtA)[b].astype(np.float64)
        damp = (np.diag(A64) + 1e-5) * lam
        damp[-1] = 0.0
        truth = np.linalg.solve(A64 + np.diag(damp), n(Atb)[b].astype(np.float64))
        for name, sl in (("pose", slice(0, o)), ("depth", slice(o, -1)), ("last", slice(-1, None))):
            scale = np.abs(truth[sl]).max()
            for bits in sols:
                err = np.abs(sols[bits][0][b][sl] - truth[sl]).max() / scale
                assert err < 1e-4, (l2_base, pairs, b, name, bits, err, lam)
    # (both are within 1e-4 of the float64 solution per group; against each other: 5e-5 of the largest entry -- the weakly damped
    # cases, l2_base <= 1, are conditioned ~1e3 and sit at 2-3e-5 depending on the rounding of the assembled matrix)
    assert np.abs(sols[0][0] - sols[1 << 23][0]).max() <= 5e-5 * np.abs(sols[1 << 23][0]).max()


def test_strip_gather_under_the_legacy_early_terminated_lm():
    """legacy/ba.py's early-terminated LM (device-side loop control, per-window `active` flags, un-normalised rays, pose only)
    with the strip gather forced on every level (C = 128): iteration counts identical to the oracle's, pose within 1e-4 --
    and identical counts to the tile kernels on the same problem."""
    from banet_amd import dense as bdense
    from oracle import dense as odense, synth
    B, H, W, C = 3, 48, 64, 128
    scenes = [synth.make_pair_scene(H, W, C, 0, [2, 1], 51 + b, normalize_rays=False, w_gt=[0.01 * (1 + b), -0.008, 0.006],
                                    t_gt=[0.06, -0.04 * (1 + 0.5 * b), 0.03]) for b in range(B)]
    intr, levels = odense.batch_scene(scenes)
    mlps = [orc.he_normal_mlp_weights(C, 5 + i) for i in range(2)]
    iters = [5, 7]
    R, T, ratio, counts = odense.solve_legacy(intr, levels, mlps, iters, early_termination=True)
    got = {}
    for bits in (STRIP, DIRECT):
        ba = bdense.DenseBA(t(intr), _torch_levels(levels), mlps, "legacy_lm")
        for p in ba.problems:
            p.c.flags = bits
        st, cnt = ba.solve(iters, early_termination=True)
        got[bits] = ([[int(v) for v in c] for c in cnt], n(st.R), n(st.T))
    assert got[STRIP][0] == counts, (got[STRIP][0], counts)
    assert got[DIRECT][0] == counts
    assert relerr(got[STRIP][1], R) < 1e-4 and relerr(got[STRIP][2], T) < 1e-4
    assert relerr(got[STRIP][1], got[DIRECT][1]) < 1e-5 and relerr(got[STRIP][2], got[DIRECT][2]) < 1e-5
