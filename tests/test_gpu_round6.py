"""Round 6 (`-m gpu`, through the C ABI): the backward's target-tile kernel (BANET_ADJOINT_FOLD_TARGET, adjoint.hip::adj_tile_kernel:
the target map's gradient accumulated per 8x8 texel tile in LDS -- no 3C adjoint rows, no [f|gx|gy] map adjoint, no fold pass)
against the float64 statement of the adjoint (oracle/dense_adjoint.py) and against the round-5 path (rows + per-texel gather +
banet_target_map_adjoint_f32), bit-reproducibility, overwrite == accumulate-into-zeros, windows with most / all pixels masked,
and a collapsed warp (hundreds of pixels in one texel cell: the big-cell sort)."""
import numpy as np
import pytest
import torch

from oracle import dense as odense, dense_adjoint as oadj, synth
from test_gpu_dense_backward import _run_adjoint, _scene, n, t, DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from banet_amd import _capi
    _capi.lib()


def _want(intr, lv, R, T, Wc, G, gb, gabs, H, W):
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    lv64 = {k: (f32(v) if isinstance(v, np.ndarray) else v) for k, v in lv.items()}
    a = odense.level_inputs(intr, lv64, True, np.float64)
    return oadj.assembly_adjoint(a, lv64["tgt"], f32(R), f32(T), f32(Wc), f32(G), f32(gb), f32(gabs) * H * W)


@pytest.mark.parametrize("tile", [0, 1, 2, 4, 5, 9, 11, 12])
@pytest.mark.parametrize("H,W,C,K,seed", [(24, 32, 6, 5, 3), (48, 64, 128, 128, 7), (30, 41, 70, 33, 11), (9, 11, 3, 1, 5),
                                          (16, 16, 64, 16, 2), (17, 33, 130, 40, 9), (20, 24, 256, 64, 4), (15, 21, 128, 128, 6),
                                          (9, 7, 16, 8, 8), (37, 50, 131, 8, 12), (26, 19, 255, 4, 13)])
def test_target_tile_adjoint_matches_the_float64_statement_and_the_row_gather_path(H, W, C, K, seed, tile):
    """every (channel chunks, vector width) instantiation: even C <= 128 / <= 256 (two channels per lane), odd C <= 64 / <= 128 / <= 256"""
    intr, levels, R, T, Wc, rng = _scene(H, W, C, K, seed)
    lv = levels[0]
    B, P = 2, 6 + K
    G = rng.standard_normal((B, P, P))
    gb = rng.standard_normal((B, P, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    want = _want(intr, lv, R, T, Wc, G, gb, gabs, H, W)
    # sign(d) (the adjoint of sum |d|) is undecidable in float32 where the float64 residual is below rounding: no upstream
    # gradient on the channels that hold such an element (a handful at most; the smooth synthetic features cross zero somewhere)
    tiny = (np.abs(want["fwd"]["diff"]) < 1e-6) & want["fwd"]["mask"][..., None]
    if tiny.any():
        gabs = gabs * (~tiny.any(axis=1))[:, None, :]
        assert (gabs != 0).mean() > 0.5
        want = _want(intr, lv, R, T, Wc, G, gb, gabs, H, W)
    got = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, fold=True, tile=tile)
    old = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
    assert want["fwd"]["mask"].mean() > 0.5
    # the other outputs do not depend on the mode at all; the target gradient only through the summation order
    for k in ("dsrc", "ddepth", "dbasis", "dpose"):
        assert torch.equal(got[k], old[k]), k
    scale = float(old["dtgt"].abs().max())
    assert float((got["dtgt"] - old["dtgt"]).abs().max()) <= 2e-5 * scale
    for name, w in (("dsrc", want["dsrc"]), ("dtgt", want["dtgt"]), ("ddepth", want["dD0"]), ("dbasis", want["dbasis"])):
        g = n(got[name]).reshape(w.shape)
        err = np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
        assert err < 2e-4, (name, err)


def test_target_tile_adjoint_is_bit_reproducible_and_overwrite_equals_accumulation():
    H, W, C, K = 48, 64, 128, 32
    intr, levels, R, T, Wc, rng = _scene(H, W, C, K, 5, B=3)
    R, T = R.copy(), T.copy()
    R[1] = synth.rodrigues(np.array([0.0, 0.35, 0.05]))       # most pixels masked
    T[2] = np.array([[60.0], [0.0], [0.0]])                   # every pixel masked: no texel hit
    B = 3
    G = rng.standard_normal((B, 6 + K, 6 + K))
    gb = rng.standard_normal((B, 6 + K, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    a = _run_adjoint(intr, levels[0], R, T, Wc, G, gb, gabs, fold=True)
    for _ in range(2):
        b = _run_adjoint(intr, levels[0], R, T, Wc, G, gb, gabs, fold=True)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    ow = _run_adjoint(intr, levels[0], R, T, Wc, G, gb, gabs, fold=True, overwrite=True)      # NaN-filled buffers: every entry written
    for k in a:
        assert torch.isfinite(ow[k]).all(), k
        assert torch.equal(a[k], ow[k]), k
    assert float(a["dtgt"][2].abs().max()) == 0.0 and float(a["dsrc"][2].abs().max()) == 0.0
    want = _want(intr, levels[0], R, T, Wc, G, gb, gabs, H, W)
    frac = want["fwd"]["mask"].mean(axis=1)
    assert 0.02 < frac[1] < 0.7 and frac[2] == 0.0, frac
    w = want["dtgt"]
    assert np.abs(n(a["dtgt"]).reshape(w.shape) - w).max() <= 2e-4 * np.abs(w).max()


def test_target_tile_adjoint_with_a_collapsed_warp():
    """A translation of 60 scene units along the optical axis shrinks a whole window into a few target cells around the principal
    point: hundreds of pixels per cell -- the big-cell queue / rank sort and long per-cell loops -- and still the float64 statement,
    bit-reproducibly."""
    H, W, C, K = 24, 32, 16, 8
    intr, levels, R, T, Wc, rng = _scene(H, W, C, K, 21)
    lv = levels[0]
    T = T.copy()
    T[:, 2, 0] += 60.0
    B, P = 2, 6 + K
    G = rng.standard_normal((B, P, P))
    gb = rng.standard_normal((B, P, 1))
    gabs = rng.standard_normal((B, 1, C)) * 0.1
    want = _want(intr, lv, R, T, Wc, G, gb, gabs, H, W)
    assert want["fwd"]["mask"].mean() > 0.5
    got = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, fold=True)
    again = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs, fold=True)
    old = _run_adjoint(intr, lv, R, T, Wc, G, gb, gabs)
    w = want["dtgt"]
    hit = (np.abs(w).max(axis=-1) > 0).reshape(B, -1).sum(axis=1)
    assert hit.max() <= 64, hit                               # the window's footprint really is a handful of texels
    assert np.abs(n(got["dtgt"]).reshape(w.shape) - w).max() <= 5e-4 * np.abs(w).max()
    assert torch.equal(got["dtgt"], again["dtgt"])
    assert float((got["dtgt"] - old["dtgt"]).abs().max()) <= 1e-4 * float(old["dtgt"].abs().max())


def test_solve_differentiable_fold_and_row_gather_backward_agree(monkeypatch):
    """DenseBA.solve_differentiable end to end (2 levels x 2 iterations, two-frame and 3-frame windows): the default backward
    (target-tile kernel) against the round-5 one (BANET_ADJOINT_FOLD=0) -- equal to rounding in every gradient."""
    from banet_amd import dense as bdense, dense_train, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    for frames in (2, 3):
        B, H, W, C, K = 2, 48, 64, 32, 16
        scales = [2, 1]
        intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, scales, 7, torch.device(DEV), trans_mag=0.06, pairs=frames - 1)
        mlps = [[(w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for w, b in he_normal_lambda_weights(C, 100 + i)]
                for i in range(len(scales))]
        for lv in levels:
            for name in ("src", "tgt", "depth", "basis"):
                setattr(lv, name, getattr(lv, name).requires_grad_(True))
        ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1.0)
        T0 = (gt["T"] * 0.7).reshape(B * (frames - 1), 3, 1).to(DEV)
        leaves = [getattr(lv, nm) for lv in levels for nm in ("src", "tgt", "depth", "basis")] + [x for lw in mlps for wb in lw for x in wb]
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setattr(dense_train, "FOLD_MODE", mode)
            Rr, Tt, Ww = ba.solve_differentiable([2, 2], T=T0)
            loss = (Rr * torch.arange(Rr.numel(), device=DEV).reshape(Rr.shape).float().cos()).sum() + Tt.sum() + (Ww * 0.5).sum()
            res[mode] = torch.autograd.grad(loss, leaves)
        for x, y in zip(res["1"], res["0"]):
            assert torch.isfinite(x).all()
            assert float((x - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1e-30)


def test_multi_frame_backward_reusing_the_depth_seed_products_is_bit_equal(monkeypatch):
    """Multi-frame windows: the adjoint calls for target frames 2.. of an iteration carry BANET_ADJOINT_REUSE_DEPTH_SEED (z2 = 2 S_dd b,
    zeta, e of the first frame's call are still in the workspace; only q = S_cd b is computed) -- every gradient bit-equal to the
    calls that recompute everything (BANET_ADJOINT_REUSE=0), at K = 32 / 128 (the bf16 form) and K = 16 (flag ignored)."""
    from banet_amd import dense as bdense, dense_train, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    for frames, K in ((3, 32), (5, 128), (3, 16)):
        B, H, W, C = 2, 48, 64, 32
        scales = [2, 1]
        intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, scales, 11, torch.device(DEV), trans_mag=0.06, pairs=frames - 1)
        mlps = [[(w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for w, b in he_normal_lambda_weights(C, 300 + i)]
                for i in range(len(scales))]
        for lv in levels:
            for name in ("src", "tgt", "depth", "basis"):
                setattr(lv, name, getattr(lv, name).requires_grad_(True))
        ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1.0)
        T0 = (gt["T"] * 0.7).reshape(B * (frames - 1), 3, 1).to(DEV)
        leaves = [getattr(lv, nm) for lv in levels for nm in ("src", "tgt", "depth", "basis")] + [x for lw in mlps for wb in lw for x in wb]
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setattr(dense_train, "REUSE_MODE", mode)
            Rr, Tt, Ww = ba.solve_differentiable([2, 2], T=T0)
            loss = (Rr * torch.arange(Rr.numel(), device=DEV).reshape(Rr.shape).float().cos()).sum() + Tt.sum() + (Ww * 0.5).sum()
            res[mode] = torch.autograd.grad(loss, leaves)
        for x, y in zip(res["1"], res["0"]):
            assert torch.isfinite(x).all()
            assert torch.equal(x, y)


@pytest.mark.parametrize("variant,B,C,K,pairs,l2", [("bundle", 3, 128, 128, 1, 1000.0), ("bundle", 2, 32, 40, 3, 1.0),
                                                    ("bundle_camera", 4, 16, 0, 1, 1.0), ("bundle", 2, 70, 33, 1, 10.0),
                                                    ("bundle", 1, 256, 64, 2, 1000.0)])
def test_small_step_adjoint_kernels_match_the_float64_torch_graph(variant, B, C, K, pairs, l2):
    """banet_small_step_adjoint_f32 (csrc/smallstep.hip: lambda MLP forward + backward, damping, the solve by implicit
    differentiation, the SE(3) / W update adjoint) against autograd through dense_train.solve_update_graph -- the statements
    bundlenet.py:165-190 / 241-276 -- evaluated in FLOAT64 on the same inputs: every output, incl. the ten lambda-weight
    gradients (accumulated over two calls) and the direct dL/d(R, T)."""
    from banet_amd import dense_train, ops
    from banet_amd.bundlenet import he_normal_lambda_weights
    g = torch.Generator().manual_seed(1000 + C + K)
    P, N = 6 * pairs + K, 4000
    camera = variant == "bundle_camera"
    M = torch.randn(B, P, P + 30, generator=g, dtype=torch.float64)
    AtA = (M @ M.transpose(1, 2)) * 3.0
    Atb = torch.randn(B, P, generator=g, dtype=torch.float64) * 2.0
    absres = (torch.rand(B, C, generator=g, dtype=torch.float64) * 0.2 + 0.01) * N * pairs
    R = torch.stack([torch.from_numpy(synth.rodrigues(0.05 * np.random.RandomState(i).standard_normal(3))) for i in range(B * pairs)]).reshape(B, pairs, 3, 3)
    T = torch.randn(B, pairs, 3, 1, generator=g, dtype=torch.float64) * 0.1
    Wc = torch.randn(B, K, 1, generator=g, dtype=torch.float64) * 0.05
    gR, gT = torch.randn(B, pairs, 3, 3, generator=g, dtype=torch.float64), torch.randn(B, pairs, 3, 1, generator=g, dtype=torch.float64)
    gW = torch.randn(B, K, 1, generator=g, dtype=torch.float64)
    layers = [(w.double(), b.double()) for w, b in he_normal_lambda_weights(C, 7)]
    # float64 statement: autograd through the graph
    leaves = [t.clone().requires_grad_(True) for t in (AtA, Atb, absres, R, T, Wc)]
    lw = [t.clone().requires_grad_(True) for wb in layers for t in (wb[0].reshape(wb[0].shape[-2], wb[0].shape[-1]), wb[1].reshape(-1))]
    R2, T2, W2 = dense_train.solve_update_graph(leaves[0], leaves[1], leaves[2], N, leaves[3], leaves[4], leaves[5],
                                                [(lw[2 * i], lw[2 * i + 1]) for i in range(5)], l2, pairs=pairs, camera=camera)
    with torch.no_grad():       # the forward's solution, as the update kernel would leave it in banet_state_t.delta
        lam_in = torch.linalg.vector_norm(absres / (N * pairs), dim=-1)
    outs, seeds = ([R2, T2], [gR, gT]) if camera else ([R2, T2, W2], [gR, gT, gW])
    want = torch.autograd.grad(outs, leaves + lw, seeds, allow_unused=True)
    sol = (W2 - leaves[5]).detach().reshape(B, K) if K else torch.zeros(B, 0, dtype=torch.float64)
    # pose part of sol: re-solve (the graph does not expose it)
    with torch.no_grad():
        h = (absres / (N * pairs)).unsqueeze(1)
        a0 = h
        for i, (w, b) in enumerate(layers):
            z = torch.matmul(h, w.reshape(w.shape[-2], w.shape[-1])) + b.reshape(-1)
            h = torch.tanh(z) if i == 4 else torch.nn.functional.selu(z)
        lam = torch.linalg.vector_norm(a0, dim=-1, keepdim=True) ** (2.0 + h)
        if not camera:
            lam = l2 * lam
        diag = torch.diagonal(AtA, dim1=1, dim2=2)
        damp = diag + 1e-5 if camera else torch.cat([diag[:, :-1] + 1e-5, torch.zeros(B, 1, dtype=torch.float64)], -1)
        delta = torch.linalg.solve(AtA + torch.diag_embed(damp * lam.squeeze(-1)), Atb.unsqueeze(-1)).reshape(B, P)
    dev = torch.device(DEV)
    mlp = ops.MlpWeights([(w.float(), b.float()) for w, b in layers], dev)
    assert dense_train.SmallStepHip.supported(variant, B, N, C, K, pairs, dev)
    hs = dense_train.SmallStepHip(variant, B, N, C, K, pairs, mlp, l2, dev)
    c = lambda x: x.float().to(dev)
    for _ in range(2):          # twice: the weight gradients accumulate
        got = hs(c(AtA), c(Atb), c(absres), c(delta), c(R), c(T), c(gR), c(gT), c(gW))
    torch.cuda.synchronize()
    names = ["gAtA", "gAtb", "gabs", "dR", "dT"]
    for name, gv, wv in zip(names, got, want[:5]):
        wv = wv.reshape(gv.shape)
        if name == "gAtA":      # the adjoint kernels symmetrise it; compare the symmetric parts
            gv = 0.5 * (gv + gv.transpose(1, 2))
            wv = 0.5 * (wv + wv.transpose(1, 2))
        err = float((gv.double().cpu() - wv).abs().max()) / max(float(wv.abs().max()), 1e-30)
        assert err < 2e-4, (name, err)
    for i, (gv, wv) in enumerate(zip(hs.glayers, want[6:])):
        err = float((gv.double().cpu() - 2.0 * wv.reshape(gv.shape)).abs().max()) / max(float(wv.abs().max()) * 2.0, 1e-30)
        assert err < 5e-4, ("lambda weight %d" % i, err)


def test_solve_differentiable_with_the_hip_small_step_equals_the_torch_small_step(monkeypatch):
    """DenseBA.solve_differentiable: the backward's small step on csrc/smallstep.hip against the torch graph it replaces
    (BANET_SMALL_STEP_HIP=0), two-frame and 3-frame windows -- every gradient to rounding.  (The two differ in how the forward's solution enters:
    the kernels take the update kernel's own `delta`, the torch graph solves the damped system again.)"""
    from banet_amd import dense as bdense, dense_train, synth as bsynth
    from banet_amd.bundlenet import he_normal_lambda_weights
    for frames in (2, 3):
        B, H, W, C, K = 2, 48, 64, 32, 32
        scales = [2, 1]
        intr, levels, gt = bsynth.make_dense_windows(B, H, W, C, K, scales, 9, torch.device(DEV), trans_mag=0.06, pairs=frames - 1)
        mlps = [[(w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for w, b in he_normal_lambda_weights(C, 100 + i)]
                for i in range(len(scales))]
        for lv in levels:
            for name in ("src", "tgt", "depth", "basis"):
                setattr(lv, name, getattr(lv, name).requires_grad_(True))
        ba = bdense.DenseBA(intr, levels, mlps, "bundle", 1000.0)
        T0 = (gt["T"] * 0.7).reshape(B * (frames - 1), 3, 1).to(DEV)
        leaves = [getattr(lv, nm) for lv in levels for nm in ("src", "tgt", "depth", "basis")] + [x for lw in mlps for wb in lw for x in wb]
        res = {}
        for mode in (True, False):
            monkeypatch.setattr(dense_train, "SMALL_STEP_HIP", mode)
            Rr, Tt, Ww = ba.solve_differentiable([2, 2], T=T0)
            loss = (Rr * torch.arange(Rr.numel(), device=DEV).reshape(Rr.shape).float().cos()).sum() + Tt.sum() + (Ww * 0.5).sum()
            res[mode] = torch.autograd.grad(loss, leaves)
        for x, y in zip(res[True], res[False]):
            assert torch.isfinite(x).all()
            assert float((x - y).abs().max()) <= 2e-4 * max(float(y.abs().max()), 1e-30)


@pytest.mark.parametrize("B,N,C,K", [(1, 4096, 128, 0), (2, 1000, 70, 33), (3, 777, 128, 128)])
def test_sparse_gather_with_16_point_items_equals_the_64_point_items(B, N, C, K):
    """ba_gather_kernel on sparse points (the reference's tracker / training layout): latency-bound launches cut the wave items to 16
    points (round 6: 4x the waves, the reference's own N = 4096 batch-1 tracker 2.06 -> 1.35 ms per solve); flags bit 1 keeps 64.
    Same sums to rounding (another grouping of the partial rows), the in-image counts exactly; ragged N (a last item with empty slots)."""
    from banet_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + N)
    H, W = 48, 64
    img = torch.randn(B, H, W, C, generator=g).to(DEV)
    conv2 = ops.target_map(img)
    pts = torch.stack([torch.rand(B, N, generator=g) * (W + 2) - 1.5, torch.rand(B, N, generator=g) * (H + 2) - 1.5], dim=-1).to(DEV)
    conv1 = ops.resample(img, pts.clamp(min=0.0)) + 0.05 * torch.randn(B, N, C, generator=g).to(DEV)
    fx = torch.full((B, N), 0.8 * W, device=DEV)
    ox, oy = torch.full((B, N), W / 2.0, device=DEV), torch.full((B, N), H / 2.0, device=DEV)
    ray = torch.stack([(pts[..., 0] - ox) / fx, (pts[..., 1] - oy) / fx, torch.ones(B, N, device=DEV)], dim=1)
    p = (ray / ray.norm(dim=1, keepdim=True)).contiguous()
    D = (2.5 + torch.rand(B, N, generator=g)).to(DEV)
    Bs = (torch.randn(B, N, K, generator=g) / max(K, 1) ** 0.5).to(DEV) if K else None
    R = torch.eye(3, device=DEV).repeat(B, 1, 1)
    T = (0.02 * torch.randn(B, 3, 1, generator=g)).to(DEV)
    Wc = (0.01 * torch.randn(B, K, 1, generator=g)).to(DEV) if K else None
    outs = {}
    for bits in (0, 2):
        prob = ops.LevelProblem("bundle" if K else "bundle_camera", conv1, conv2, D, H, W, C, basis=Bs, rays=p, fx=fx, fy=fx.clone(), ox=ox, oy=oy,
                                dense=False, tgt_has_grad=True)
        prob.c.flags = bits
        outs[bits] = [x.double().cpu() for x in ops.ba_assemble(prob, R, T, Wc)]
    torch.cuda.synchronize()
    assert 0 < float(outs[0][3].min()) and float(outs[0][3].max()) < N                 # some points are outside the image
    assert torch.equal(outs[0][3], outs[2][3])                                          # in-image counts: exact
    for a, b_ in zip(outs[0][:3], outs[2][:3]):
        assert float((a - b_).abs().max()) <= 2e-5 * float(b_.abs().max())
