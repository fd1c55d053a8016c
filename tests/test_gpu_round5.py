"""Round-5 GPU parity tests (`-m gpu`, all through the C ABI of libbanet_hip.so).

* the batch-32 PRODUCTION selection of the multi-frame configurations (frame-parallel strip gather + the fp16 two-piece SYRK) at
  BASELINE's full sizes against the float64 twin: 2 windows of 640x480 x 4 target frames (configs[2]) and 1 window of
  1280x960 x 7 target frames, K = 256 (configs[4]) -- until round 4 these ran at full size only inside bench.py's sweep record;
* banet_level_t.policy = BANET_POLICY_BATCH_INVARIANT: the same window solved in batches of 1 / 8 / 32 is bit-identical (kernels,
  SYRK form and summation split decided from the level alone); under the default policy the batches may select different kernels
  and agree to 1e-5.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from banet_amd import _capi
    _capi.lib()                                   # fail loudly if the HIP library is missing


def n(x):
    return x.detach().cpu().numpy()


def test_cfg3_production_selection_full_size_matches_float64_twin():
    """configs[2] at batch 32 runs ba_gather128s_kernel<1, 4, true> (frame-parallel workgroups: wave p of a workgroup takes target
    frame p of the same 16x16 segment) and the fp16 two-piece SYRK on its 640x480 level.  Two windows at full size with exactly
    that selection forced (BANET_FLAG_FORCE_STRIP_GATHER | BANET_FLAG_SYRK_F16): assembly (every entry of AtA / Atb, 3e-5 of
    scale), lambda, and one LM update per coefficient group (1e-4, north_star) against the float64 twin of the oracle's window
    iteration (bundlenet.py:206-278 per pair over one conv1)."""
    from banet_amd import ops
    from test_gpu_round3 import _window_level_check
    ba = _window_level_check(2, 480, 640, 128, 128, 4, 4713, flags=ops.FORCE_STRIP_GATHER | ops.SYRK_F16, expect_sel=(3, 4))
    assert ba.problems[0].N == 640 * 480
    # ... and that forced selection IS what a batch of 32 selects by itself (host-side plan, no launch)
    c = ba.problems[0].c
    c.flags, keep = 0, c.B
    c.B = 32
    try:
        assert (ops.gather_selection(ba.problems[0]), ops.syrk_selection(ba.problems[0])) == (3, 4)
    finally:
        c.B = keep


def test_cfg5_production_selection_full_size_matches_float64_twin():
    """configs[4]'s per-GPU share (8 windows x 8 frames, 1280x960, K = 256, P = 298) runs the frame-parallel strip gather with two
    128-coefficient chunks per basis row (ba_gather128s_kernel<2, 4, true>, 7 waves per workgroup) and the wide-basis SYRK jobs in
    their fp16 form.  One window at full size with that selection forced, against the float64 twin.  (Pose / damped depth
    coefficients / lambda at 1e-4 like everywhere; the undamped last coefficient at max(1e-4, 2 x the float32 twin's own error):
    at this state the fp16 form measured 1.9e-4 on it, pose 6e-7, depth 1.3e-6 -- round 5, first GPU run.)"""
    from banet_amd import ops
    from test_gpu_round3 import _window_level_check
    ba = _window_level_check(1, 960, 1280, 128, 256, 7, 4714, flags=ops.FORCE_STRIP_GATHER | ops.SYRK_F16, expect_sel=(3, 4),
                             last_yardstick=True)
    assert ba.problems[0].N == 1280 * 960 and ba.problems[0].P == 298
    c = ba.problems[0].c
    c.flags, keep = 0, c.B
    c.B = 8
    try:
        assert (ops.gather_selection(ba.problems[0]), ops.syrk_selection(ba.problems[0])) == (3, 4)
    finally:
        c.B = keep


def _solve_prefix(intr, levels, mlps, nb, policy, iters, T0):
    """the first nb windows of the batch as a launch of their own (from the translation prior T0: from T = 0 the depth Jacobian
    is identically zero and the undamped last coefficient of bundlenet.py:266 diverges)"""
    from banet_amd import dense as bdense, ops
    sub = [bdense.DenseLevel(l.scale, l.src[:nb].contiguous(), l.tgt[:nb].contiguous(), l.depth[:nb].contiguous(),
                             l.basis[:nb].contiguous()) for l in levels]
    ba = bdense.DenseBA(intr[:nb].contiguous(), sub, mlps, "bundle", 1000.0)
    for p in ba.problems:
        p.c.policy = policy
    sel = [(ops.gather_selection(p), ops.syrk_selection(p)) for p in ba.problems]
    pairs = ba.pairs
    st = ba.new_state(T=T0[:nb].reshape(nb * pairs, 3, 1).clone())
    st, counts = ba.solve(iters, state=st)
    torch.cuda.synchronize()
    return sel, n(st.R), n(st.T), n(st.Wc), [n(c) for c in counts]


@pytest.mark.parametrize("pairs", [1, 4])
def test_batch_invariant_policy_is_bit_identical_across_batch_sizes(pairs):
    """A window's result must not depend on who shares its launch when the caller asks for that (multi-GPU callers: a shard of 8
    and a shard of 32 windows of the same batch).  320x240 + 160x120 levels, C = K = 128: under BANET_POLICY_BATCH_INVARIANT
    batches of 1 / 8 / 32 run the kernels a batch of 32 runs (strip gather and, on the 320x240 level, the fp16 two-piece SYRK with
    8 partial rows per window) and window 0's pose and depth coefficients are bit-identical; under the default policy the
    selections differ with the batch (4x4-item gather / exact SYRK at one window) and the results agree to rounding: pose 1e-5,
    depth coefficients 1e-4 (measured 2.4e-5)."""
    from banet_amd import _capi, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    B, H, W, C, K = 32, 240, 320, 128, 128
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [2, 1], 8800 + pairs, DEV, trans_mag=0.05, pairs=pairs)
    mlps = [he_normal_lambda_weights(C, 7 + i) for i in range(2)]
    iters = [3, 3]
    T0 = (gt["T"] * 0.7).reshape(B, pairs, 3, 1).to(DEV)
    runs = {nb: _solve_prefix(intr, levels, mlps, nb, _capi.POLICY_BATCH_INVARIANT, iters, T0) for nb in (32, 8, 1)}
    sel32 = runs[32][0]
    assert sel32[1] == (3, 4), sel32                          # the 320x240 level: strip gather + fp16 SYRK, as at batch 32
    assert all(np.isfinite(x).all() for x in runs[32][1:4])
    for nb in (8, 1):
        assert runs[nb][0] == sel32, (nb, runs[nb][0], sel32)
        for a, b in zip(runs[nb][1:4], runs[32][1:4]):
            np.testing.assert_array_equal(a, b[:a.shape[0]])  # every window of the small batch: the bits of the large one
    # default policy: selections follow the launch, results agree to rounding
    d32 = _solve_prefix(intr, levels, mlps, 32, _capi.POLICY_THROUGHPUT, iters, T0)
    d1 = _solve_prefix(intr, levels, mlps, 1, _capi.POLICY_THROUGHPUT, iters, T0)
    assert d32[0] == sel32 and d1[0] != sel32, (d32[0], d1[0])
    for a, b in zip(d1[1:4], d32[1:4]):       # R, T: 1e-5; the depth coefficients (|Wc| ~ 1e-3 after six iterations): north_star's 1e-4
        ref = b[:a.shape[0]].astype(np.float64)           # (measured, round 5: R / T <= 1e-5, Wc 2.4e-5 of its largest entry at 4 target frames)
        tol = 1e-5 if a.shape[-2:] in ((3, 3), (3, 1)) else 1e-4
        assert np.abs(a.astype(np.float64) - ref).max() <= tol * max(np.abs(ref).max(), 1e-30)
    for a, b in zip(d32[1:4], runs[32][1:4]):                 # at the canonical batch both policies are the same launch
        np.testing.assert_array_equal(a, b)
