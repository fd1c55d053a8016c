"""Round-5 GPU parity tests (`-m gpu`, all through the C ABI of libbanet_hip.so).

* the batch-32 PRODUCTION selection of the multi-frame configurations (frame-parallel strip gather + the fp16 two-piece SYRK) at
  BASELINE's full sizes against the float64 twin: 2 windows of 640x480 x 4 target frames (configs[2]) and 1 window of
  1280x960 x 7 target frames, K = 256 (configs[4]) -- until round 4 these ran at full size only inside bench.py's sweep record;
* banet_level_t.policy = BANET_POLICY_BATCH_INVARIANT: the same window solved in batches of 1 / 8 / 32 is bit-identical (kernels,
  SYRK form and summation split decided from the level alone); under the default policy the batches may select different kernels
  and agree to rounding;
* the sparse-point training iteration as ONE autograd node on the fused kernels (BundleNet.training_graph = "fused": forward = the
  inference path, backward = implicit differentiation of the small step + banet_dense_adjoint_f32 on the reference's sparse layout)
  against the lean torch graph at the reference's training shape (N = 4096 points, C = K = 128); the finite-difference check
  against the float64 oracle is tests/test_gpu_parity.py::test_training_graph_matches_fused_forward_and_finite_difference_gradients.
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


def _sparse_iteration64(*args):
    """the float64 yardstick of both float32 training graphs: oracle/torch_port.sparse_iteration -- bundlenet.py:122-278 on sparse
    points as differentiable torch statements, written with the ORACLE's own helpers (round 6: it used to be built from the product's
    torch helpers) and pinned to the numpy oracle on the CPU (tests/test_torch_ref_cpu.py)"""
    from oracle import torch_port
    return torch_port.sparse_iteration(*args)


@pytest.mark.parametrize("B,N,C,K,H,W", [(2, 4096, 128, 128, 96, 128),      # the reference's training shape class (bundlenet.py:332-399)
                                         (1, 777, 70, 33, 40, 56),          # ragged C / K (masked lanes)
                                         (2, 1500, 32, 200, 48, 64),        # K > 128: the wide seed blocks
                                         (3, 513, 16, 8, 20, 24),           # small everything, three windows
                                         (1, 2048, 128, 64, 60, 80),        # K = 64
                                         (2, 300, 256, 32, 24, 32)])        # C = 256 (four channel chunks per lane)
def test_fused_sparse_training_iteration_equals_the_lean_graph(B, N, C, K, H, W):
    """BundleNet.BundleIteration / CameraIteration with gradients: training_graph "fused" (one autograd node on the fused kernels,
    dense_train._SparseIteration) vs "lean" (ops.sample_stats + the normal equations by block in torch, FD-checked against the
    float64 oracle in test_gpu_parity.py), both against the same statements in float64 (pure torch): same updates (1e-4), and the
    gradients w.r.t. conv1, the [f|gx|gy] map, D, the basis, R, T, W and the ten lambda-weight tensors within 2e-3 of each
    gradient's largest entry (or twice the lean graph's own float32 error, whichever is larger); the fused gradients are
    bit-reproducible run to run (no float atomics)."""
    from banet_amd import ops
    from banet_amd.bundlenet import BundleNet, he_normal_lambda_weights
    g = torch.Generator().manual_seed(1000 + N)
    img = torch.randn(B, H, W, C, generator=g).to(DEV)
    conv2 = ops.target_map(img)
    pts = torch.stack([torch.rand(B, N, generator=g) * (W - 1.5) + 0.25, torch.rand(B, N, generator=g) * (H - 1.5) + 0.25], dim=-1).to(DEV)
    conv1 = ops.resample(img, pts) + 0.05 * torch.randn(B, N, C, generator=g).to(DEV)
    fx = torch.full((B, N), 0.8 * W, device=DEV)
    fy = fx.clone()
    ox = torch.full((B, N), W / 2.0, device=DEV)
    oy = torch.full((B, N), H / 2.0, device=DEV)
    ray = torch.stack([(pts[..., 0] - ox) / fx, (pts[..., 1] - oy) / fy, torch.ones(B, N, device=DEV)], dim=1)
    p = ray / ray.norm(dim=1, keepdim=True)
    D = (2.5 + torch.rand(B, N, 1, generator=g)).to(DEV)
    Bs = (torch.randn(B, N, K, generator=g) / K ** 0.5).to(DEV)
    R = torch.eye(3, device=DEV).repeat(B, 1, 1)
    T = (0.02 * torch.randn(B, 3, 1, generator=g)).to(DEV)          # (projections move by ~1 px: some points leave the image)
    Wc = (0.01 * torch.randn(B, K, 1, generator=g)).to(DEV)
    cR, cT, cW = [torch.randn(x.shape, generator=g).to(DEV) for x in (R, T, Wc)]
    res = {}
    for graph in ("f64", "lean", "fused", "fused"):
        dt = torch.float64 if graph == "f64" else torch.float32
        lw = [(w.to(DEV).to(dt).requires_grad_(True), b.to(DEV).to(dt).requires_grad_(True)) for w, b in he_normal_lambda_weights(C, 7)]
        leaves = [x.clone().to(dt).requires_grad_(True) for x in (conv1, conv2, D, Bs, R, T, Wc)]
        if graph == "f64":
            R2, T2, W2 = _sparse_iteration64(*leaves, lw, True, 1000.0, fx, fy, ox, oy, p)
            Rc, Tc, _ = _sparse_iteration64(*leaves, lw, False, 1.0, fx, fy, ox, oy, p)
        else:
            net = BundleNet(lambda_weights={"0": lw})
            net.training_graph = graph
            R2, T2, W2 = net.BundleIteration(leaves[0], leaves[1], fx, fy, ox, oy, p, leaves[2], leaves[3], leaves[4], leaves[5], leaves[6], 1000.0, "0")
            Rc, Tc = net.CameraIteration(leaves[0], leaves[1], fx, fy, ox, oy, p, leaves[2], leaves[4], leaves[5], 1.0, "0")
        loss = (R2 * cR.to(dt)).sum() + (T2 * cT.to(dt)).sum() + (W2 * cW.to(dt)).sum()
        grads = torch.autograd.grad(loss, leaves + [x for wb in lw for x in wb], retain_graph=graph == "f64")
        gc = torch.autograd.grad((Rc * cR.to(dt)).sum() + (Tc * cT.to(dt)).sum(),
                                 [leaves[0], leaves[1], leaves[2], leaves[4], leaves[5]] + [x for wb in lw for x in wb])
        out = ([n(x) for x in (R2, T2, W2, Rc, Tc)], [n(x) for x in grads] + [n(x) for x in gc])
        if graph == "fused" and "fused" in res:
            for a, b in zip(out[1], res["fused"][1]):
                np.testing.assert_array_equal(a, b)                  # bit-reproducible
        res[graph] = out

    def rel(a, b):
        return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-300))
    for a, b, c in zip(res["fused"][0], res["lean"][0], res["f64"][0]):
        assert rel(a, c) < 1e-4 and rel(a, b) < 1e-4, (rel(a, c), rel(a, b))
    # gradients: both float32 graphs against the float64 statements.  The fused node must be within 2e-3 of each gradient's
    # largest entry -- or, where float32 itself cannot do better (the pose-only iteration's feature / depth gradients are ~1e-6
    # next to pose gradients of ~1: differences of cancelling sums; the lean graph is 4e-1 off there too, tools/diag_sparse_train.py),
    # no worse than twice the lean graph's own error.
    for i, (a, b, c) in enumerate(zip(res["fused"][1], res["lean"][1], res["f64"][1])):
        assert a.shape == c.shape and np.isfinite(a).all()
        assert rel(a, c) < max(2e-3, 2.0 * rel(b, c)), (i, rel(a, c), rel(b, c))


def test_split_coarse_levels_equal_one_batch_under_the_invariant_policy():
    """DenseBA.split_coarse (opt-in): the coarsest level (<= 1200 pixels) of a batch of >= 16 windows runs as two half batches on two
    HIP streams that share the state tensors.  Under BANET_POLICY_BATCH_INVARIANT a window's bits do not depend on its launch, so the
    split solve must equal the one-batch solve bit for bit -- which also checks the stream fork / join and the state views."""
    from banet_amd import dense as bdense, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    B, H, W, C, K = 16, 60, 80, 128, 128
    intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, [2, 1], 9100, DEV, trans_mag=0.05)
    mlps = [he_normal_lambda_weights(C, 17 + i) for i in range(2)]
    T0 = (gt["T"] * 0.7).reshape(B, 3, 1).to(DEV)
    out = []
    for split in (False, True, True):
        ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1000.0, batch_invariant=True)
        ba.split_coarse = split
        st, counts = ba.solve([4, 3], ba.new_state(T=T0.clone()))
        torch.cuda.synchronize()
        assert (not split) or 0 in ba._parts                      # the 40x30 level really ran as two halves
        out.append([n(st.R), n(st.T), n(st.Wc), n(st.delta), n(st.lambda_out)] + [n(c) for c in counts])
    assert all(np.isfinite(x).all() for x in out[0])
    for other in out[1:]:
        for a, b in zip(out[0], other):
            np.testing.assert_array_equal(a, b)
