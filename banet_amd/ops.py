"""Operator layer: the reference's custom ops (`util.equation_construction`,
`util.equation_construction_grad`, bundlenet.py:76-82) and the fused entry points, each a
thin call into libbanet_hip.so through its C ABI.  GPU only -- no CPU / eager fallback.
"""
import ctypes
from typing import Tuple

import torch

from . import _capi as capi

_VARIANT_OF = {"legacy_lm": capi.LEGACY_LM, "legacy_fixed": capi.LEGACY_FIXED,
               "bundle_camera": capi.BUNDLE_CAMERA, "bundle": capi.BUNDLE}


# --------------------------------------------------------------------------------------
# EquationConstruction (+Grad)                      utils.cu:150-171 / :420-428
# --------------------------------------------------------------------------------------
def _eq_shapes(jacobian, gradient, difference):
    B, N, two, P = jacobian.shape
    Bg, Ng, C, two2 = gradient.shape
    if two != 2 or two2 != 2 or (Bg, Ng) != (B, N) or tuple(difference.shape) != (B, N, C, 1):
        raise capi.BanetError("equation_construction: expected J[B,N,2,P], G[B,N,C,2], d[B,N,C,1]; got %s %s %s"
                              % (tuple(jacobian.shape), tuple(gradient.shape), tuple(difference.shape)))
    return B, N, C, P


def equation_construction_forward(jacobian, gradient, difference):
    J, G, d = capi.f32c(jacobian), capi.f32c(gradient), capi.f32c(difference)
    B, N, C, P = _eq_shapes(J, G, d)
    L = capi.lib()
    left = torch.empty((B, P, P), dtype=torch.float32, device=J.device)
    right = torch.empty((B, P, 1), dtype=torch.float32, device=J.device)
    nb = L.banet_equation_construction_workspace_bytes(B, N, C, P)
    if nb == 0:
        raise capi.BanetError("equation_construction: unsupported shape B=%d N=%d C=%d P=%d" % (B, N, C, P))
    ws = capi.workspace(nb, J.device)
    capi.check(L.banet_equation_construction_f32(capi.ptr(J), capi.ptr(G), capi.ptr(d), capi.ptr(left), capi.ptr(right),
                                                 B, N, C, P, ctypes.c_void_p(ws.data_ptr()),
                                                 ws.numel(), capi.stream()))
    return left, right


def equation_construction_grad(jacobian, gradient, difference, left_grad, right_grad):
    """`util.equation_construction_grad` (bundlenet.py:78,81)."""
    J, G, d = capi.f32c(jacobian), capi.f32c(gradient), capi.f32c(difference)
    g0, g1 = capi.f32c(left_grad), capi.f32c(right_grad)
    B, N, C, P = _eq_shapes(J, G, d)
    L = capi.lib()
    gJ, gG, gd = torch.empty_like(J), torch.empty_like(G), torch.empty_like(d)
    nb = L.banet_equation_construction_grad_workspace_bytes(B, N, C, P)      # 0: no workspace needed for this shape
    ws = capi.workspace(nb, J.device) if nb else None
    capi.check(L.banet_equation_construction_grad_f32(capi.ptr(J), capi.ptr(G), capi.ptr(d), capi.ptr(g0), capi.ptr(g1),
                                                      capi.ptr(gJ), capi.ptr(gG), capi.ptr(gd), B, N, C, P,
                                                      ctypes.c_void_p(ws.data_ptr()) if nb else None, ws.numel() if nb else 0,
                                                      capi.stream()))
    return gJ, gG, gd


class _EquationConstruction(torch.autograd.Function):
    """forward = EquationConstruction, backward = EquationConstructionGrad: the pairing the
    reference registers with @ops.RegisterGradient (bundlenet.py:79-82)."""

    @staticmethod
    def forward(ctx, jacobian, gradient, difference, symmetric_grad):
        ctx.save_for_backward(jacobian, gradient, difference)
        ctx.symmetric_grad = bool(symmetric_grad)
        return equation_construction_forward(jacobian, gradient, difference)

    @staticmethod
    def backward(ctx, left_grad, right_grad):
        J, G, d = ctx.saved_tensors
        if ctx.symmetric_grad:
            left_grad = 0.5 * (left_grad + left_grad.transpose(1, 2))
        return equation_construction_grad(J, G, d, left_grad, right_grad) + (None,)


def equation_construction(jacobian, gradient, difference, symmetric_grad=False):
    """`util.equation_construction(jacobian=, gradient=, difference=)` -> (AtA [B,P,P], Atb [B,P,1]).
    The registered gradient is the reference's (utils.cu:648-657: dA = 2 A g0 + d g1^T), which is the
    true gradient only for a symmetric upstream g0 = dL/dAtA.  symmetric_grad=True feeds it
    (g0 + g0^T)/2 instead -- the exact gradient for any g0, identical to the reference's whenever the
    reference's is exact."""
    return _EquationConstruction.apply(jacobian, gradient, difference, symmetric_grad)


# torch.ops.banet.equation_construction / equation_construction_grad: the same two kernels registered with the
# dispatcher (the counterpart of REGISTER_OP + @ops.RegisterGradient, utils.cu:150-171,420-428 and
# bundlenet.py:79-82).  Device type "cuda" only (= HIP here): a CPU tensor raises NotImplementedError.
@torch.library.custom_op("banet::equation_construction", mutates_args=(), device_types="cuda")
def _eq_op(jacobian: torch.Tensor, gradient: torch.Tensor, difference: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return equation_construction_forward(jacobian, gradient, difference)


@_eq_op.register_fake
def _eq_op_fake(jacobian, gradient, difference):
    B, _, _, P = jacobian.shape
    return jacobian.new_empty((B, P, P)), jacobian.new_empty((B, P, 1))


@torch.library.custom_op("banet::equation_construction_grad", mutates_args=(), device_types="cuda")
def _eq_grad_op(jacobian: torch.Tensor, gradient: torch.Tensor, difference: torch.Tensor, left_grad: torch.Tensor,
                right_grad: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return equation_construction_grad(jacobian, gradient, difference, left_grad, right_grad)


@_eq_grad_op.register_fake
def _eq_grad_op_fake(jacobian, gradient, difference, left_grad, right_grad):
    return torch.empty_like(jacobian), torch.empty_like(gradient), torch.empty_like(difference)


def _eq_op_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _eq_op_backward(ctx, left_grad, right_grad):
    J, G, d = ctx.saved_tensors
    return torch.ops.banet.equation_construction_grad(J, G, d, left_grad.contiguous(), right_grad.contiguous())


_eq_op.register_autograd(_eq_op_backward, setup_context=_eq_op_setup)


# --------------------------------------------------------------------------------------
# fused path
# --------------------------------------------------------------------------------------
class MlpWeights:
    """Device copies of the five k=1 conv layers lambda_<level>_<i>_{filters,biases}
    (bundlenet.py:102-110,168-172): filters [Cin,Cout] row-major, biases [Cout]."""

    def __init__(self, layers, device, C=None):
        if len(layers) != 5:
            raise capi.BanetError("the lambda predictor has 5 layers, got %d" % len(layers))
        self.w = [torch.as_tensor(w, dtype=torch.float32).reshape(w.shape[-2], w.shape[-1]).contiguous().to(device)
                  for w, _ in layers]
        self.b = [torch.as_tensor(b, dtype=torch.float32).reshape(-1).contiguous().to(device) for _, b in layers]
        self.C = int(self.w[0].shape[0])
        self.check(self.C if C is None else C)
        self.c = capi.Mlp()
        for i in range(5):
            self.c.w[i] = self.w[i].data_ptr()
            self.c.b[i] = self.b[i].data_ptr()


    def check(self, C):
        """The solve kernel reads the layers with the fixed extents C -> 2C -> 4C -> 2C -> C -> 1
        (bundlenet.py:168-172): anything else would be an out-of-bounds device read, so refuse it here."""
        dims = [C, 2 * C, 4 * C, 2 * C, C, 1]
        for i in range(5):
            if tuple(self.w[i].shape) != (dims[i], dims[i + 1]) or tuple(self.b[i].shape) != (dims[i + 1],):
                raise capi.BanetError("lambda MLP layer %d: expected filters [%d,%d] and biases [%d] for C = %d feature "
                                      "channels, got %s and %s" % (i + 1, dims[i], dims[i + 1], dims[i + 1], C,
                                                                   tuple(self.w[i].shape), tuple(self.b[i].shape)))


class MlpCache:
    """Device copies of per-level lambda weights, rebuilt whenever a level's weights are replaced or modified in place
    (checkpoint reload, optimizer step): keyed on the identity and autograd version counter of every weight tensor;
    weights held as numpy arrays (no version counter) are re-uploaded on every call."""

    def __init__(self):
        self._d = {}

    def get(self, lambda_weights, level, device):
        level = str(level)
        if level not in lambda_weights:
            raise KeyError("no lambda weights for level %r (set lambda_weights[level])" % (level,))
        layers = lambda_weights[level]
        flat = [t for pair in layers for t in pair]
        versioned = all(torch.is_tensor(t) for t in flat)
        sig = tuple((id(t), t._version) for t in flat) if versioned else None
        key = (level, str(device))
        hit = self._d.get(key)
        if hit is None or sig is None or hit[0] != sig:
            # the entry keeps the source tensors alive: their id()s cannot be recycled by new tensors (del + reload from a
            # checkpoint at the same addresses with _version 0) while the signature still names them
            hit = (sig, MlpWeights(layers, device), flat)
            self._d[key] = hit
        return hit[1]

    def clear(self):
        self._d.clear()


def lm_params(angle_change=None, translation_change=None, residual_ratio=None, qr=None):
    """banet_lm_params_t: the reference's module globals legacy/ba.py:5-9 as a value (None = the reference's default)."""
    p = capi.LmParams()
    capi.lib().banet_lm_params_default(ctypes.byref(p))
    if angle_change is not None:
        p.angle_change = float(angle_change)
    if translation_change is not None:
        p.translation_change = float(translation_change)
    if residual_ratio is not None:
        p.residual_ratio = float(residual_ratio)
    if qr is not None:
        p.solver = capi.SOLVER_QR if qr else capi.SOLVER_INVERSE
    return p


class LmState:
    """Per-window LM state (banet_state_t): R [B,3,3], T [B,3,1], Wc [B,K,1] + diagnostics.
    Multi-frame windows (pairs > 1): R [B,pairs,3,3], T [B,pairs,3,1]."""

    def __init__(self, R, T, Wc=None, P=6, pairs=1):
        dev = R.device
        B = R.shape[0]
        shape = (B,) if pairs == 1 else (B, pairs)
        self.R = capi.f32c(R).clone().reshape(*shape, 3, 3)
        self.T = capi.f32c(T).clone().reshape(*shape, 3, 1)
        self.Wc = capi.f32c(Wc).clone() if Wc is not None else None
        self.iters = torch.zeros(B, dtype=torch.int32, device=dev)
        self.ratio = torch.zeros(B, dtype=torch.float32, device=dev)
        self.lambda_out = torch.zeros(B, dtype=torch.float32, device=dev)
        self.delta = torch.zeros(B, P, dtype=torch.float32, device=dev)
        self.c = capi.State()
        self.c.R = self.R.data_ptr()
        self.c.T = self.T.data_ptr()
        self.c.Wc = self.Wc.data_ptr() if self.Wc is not None else None
        self.c.iters = self.iters.data_ptr()
        self.c.ratio = self.ratio.data_ptr()
        self.c.lambda_out = self.lambda_out.data_ptr()
        self.c.delta = self.delta.data_ptr()


class LevelProblem:
    """One banet_level_t plus the tensors it points at (kept alive here)."""

    def __init__(self, variant, src, tgt, depth, H, W, C, basis=None, rays=None, fx=None, fy=None, ox=None, oy=None,
                 intr=None, scale=1.0, dense=False, tgt_has_grad=True, normalize_rays=False, pairs=1):
        """pairs > 1: multi-frame window, tgt [B,pairs,H,W,C(3C)] (banet_level_t.pairs)"""
        v = _VARIANT_OF[variant] if isinstance(variant, str) else int(variant)
        keep = [capi.f32c(x) if x is not None else None for x in (src, tgt, depth, basis, rays, fx, fy, ox, oy, intr)]
        src, tgt, depth, basis, rays, fx, fy, ox, oy, intr = keep
        self.keep = keep
        B = tgt.shape[0]
        N = depth.numel() // B
        K = 0 if basis is None else basis.shape[-1]
        lv = capi.Level()
        lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = B, N, C, K, H, W
        lv.variant, lv.dense, lv.tgt_has_grad = v, int(dense), int(tgt_has_grad)
        lv.normalize_rays, lv.scale = int(normalize_rays), float(scale)
        lv.pairs = int(pairs)
        for name, t in (("src", src), ("tgt", tgt), ("depth", depth), ("basis", basis), ("rays", rays), ("fx", fx),
                        ("fy", fy), ("ox", ox), ("oy", oy), ("intr", intr)):
            setattr(lv, name, None if t is None else t.data_ptr())
            if t is not None and not t.is_cuda:
                raise capi.BanetError("banet_amd runs on the GPU only (%s is on %s)" % (name, t.device))
        expect_tgt = B * int(pairs) * H * W * C * (3 if tgt_has_grad else 1)
        if tgt.numel() != expect_tgt or src.numel() != B * N * C:
            raise capi.BanetError("level tensors have inconsistent sizes")
        self.c = lv
        self.B, self.N, self.C, self.K, self.P, self.pairs = B, N, C, K, 6 * int(pairs) + K, int(pairs)
        self.device = tgt.device


GATHER_KERNELS = {0: "ba_gather_kernel", 1: "ba_gather128_kernel", 2: "ba_gather128p_kernel", 3: "ba_gather128s_kernel",
                  4: "ba_gather128q_kernel"}
FORCE_QUAD_GATHER, NO_QUAD_GATHER = 1 << 25, 1 << 30   # banet_hip.h: BANET_DEV_FORCE_QUAD_GATHER / _NO_QUAD_GATHER
FORCE_PATCH_GATHER, FORCE_STRIP_GATHER = 512, 262144     # banet_hip.h: BANET_DEV_FORCE_PATCH_GATHER / _STRIP_GATHER (parity checks at small batch sizes)
SYRK_THREE_PRODUCTS = 1 << 29   # banet_hip.h: BANET_DEV_SYRK_THREE_PRODUCTS -- opt-in: the K = 128 SYRK with the three largest bf16 products only (~2^-16 per product instead of fp32-exact)


SYRK_F16, NO_SYRK_F16 = 1 << 24, -2147483648          # banet_hip.h: BANET_DEV_SYRK_F16 / BANET_DEV_NO_SYRK_F16
SYRK_KERNELS = {0: "ba_syrk_kernel", 1: "ba_syrk_direct_kernel", 2: "ba_syrk_bf16x6_kernel (bf16 x 3 pieces, 6 products)",
                3: "syrk_wide.hip jobs", 4: "ba_syrk_bf16x6_kernel (fp16 x 2 pieces, 3 products, scaled)", -1000: "none (K = 0)"}


def syrk_selection(level):
    """banet_syrk_selection: the depth-block contraction kernel banet_lm_level_f32 runs for this level AND batch size"""
    rc = capi.lib().banet_syrk_selection(ctypes.byref(level.c))
    if rc < 0 and rc != -1000:
        capi.check(rc)
    return rc


def gather_selection(level):
    """banet_gather_selection: 0 generic / 1 C=128 direct / 2 LDS patches / 3 strip segments / 4 4x4-pixel items, for this level AND batch size"""
    rc = capi.lib().banet_gather_selection(ctypes.byref(level.c))
    if rc < 0:
        capi.check(rc)
    return rc


def ba_assemble(level, R, T, Wc=None, return_mask=False):
    """banet_ba_assemble_f32 -> (AtA [B,P,P], Atb [B,P], absres [B,C], nvalid [B]); return_mask: banet_ba_assemble_mask_f32,
    additionally the in-image mask bit the kernel decided for every pixel, uint8 [B, pairs, N] (parity diagnostics)."""
    L = capi.lib()
    dev = level.device
    B, P, C = level.B, level.P, level.C
    AtA = torch.empty((B, P, P), dtype=torch.float32, device=dev)
    Atb = torch.empty((B, P), dtype=torch.float32, device=dev)
    absres = torch.empty((B, C), dtype=torch.float32, device=dev)
    nvalid = torch.empty((B,), dtype=torch.float32, device=dev)
    nb = L.banet_ba_assemble_workspace_bytes(ctypes.byref(level.c))
    if nb == 0:
        raise capi.BanetError("ba_assemble: unsupported level shape")
    ws = capi.workspace(nb, dev)
    R, T = capi.f32c(R), capi.f32c(T)
    Wc = capi.f32c(Wc) if Wc is not None else None
    if return_mask:
        pairs = max(int(level.c.pairs), 1)
        mask = torch.full((B, pairs, int(level.c.N)), 255, dtype=torch.uint8, device=dev)
        capi.check(L.banet_ba_assemble_mask_f32(ctypes.byref(level.c), capi.ptr(R), capi.ptr(T), capi.ptr(Wc), capi.ptr(AtA),
                                                capi.ptr(Atb), capi.ptr(absres), capi.ptr(nvalid), ctypes.c_void_p(mask.data_ptr()),
                                                ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
        return AtA, Atb, absres, nvalid, mask
    capi.check(L.banet_ba_assemble_f32(ctypes.byref(level.c), capi.ptr(R), capi.ptr(T), capi.ptr(Wc), capi.ptr(AtA),
                                       capi.ptr(Atb), capi.ptr(absres), capi.ptr(nvalid),
                                       ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
    return AtA, Atb, absres, nvalid


def ba_solve_update(level, mlp, l2_base, AtA, Atb, absres, nvalid, state, ws=None):
    """banet_ba_solve_update_f32: lambda, damping, solve, SE(3)/W update (in place on `state`).  Systems whose matrix does
    not fit the LDS (P > ~190) go through banet_ba_solve_update_ws_f32 with a workspace (`ws`, or one allocated here)."""
    L = capi.lib()
    if mlp is not None:
        mlp.check(level.C)
    nb = L.banet_ba_solve_update_workspace_bytes(ctypes.byref(level.c))
    if nb == 0:
        capi.check(L.banet_ba_solve_update_f32(ctypes.byref(level.c), ctypes.byref(mlp.c) if mlp is not None else None,
                                               float(l2_base), capi.ptr(AtA), capi.ptr(Atb), capi.ptr(absres),
                                               capi.ptr(nvalid), ctypes.byref(state.c), capi.stream()))
        return None
    if ws is None or ws.numel() < nb:
        ws = capi.workspace(nb, level.device)
    capi.check(L.banet_ba_solve_update_ws_f32(ctypes.byref(level.c), ctypes.byref(mlp.c) if mlp is not None else None,
                                              float(l2_base), capi.ptr(AtA), capi.ptr(Atb), capi.ptr(absres),
                                              capi.ptr(nvalid), ctypes.byref(state.c), ctypes.c_void_p(ws.data_ptr()),
                                              ws.numel(), capi.stream()))
    return ws


def lm_level(level, mlp, l2_base, max_iters, early_termination, state, ws=None, params=None):
    """banet_lm_level_ex_f32: the whole LM loop of one pyramid level, enqueued without host sync.
    params: an `lm_params(...)` value (None = the reference's defaults, legacy/ba.py:5-9)."""
    L = capi.lib()
    if mlp is not None:
        mlp.check(level.C)
    nb = L.banet_lm_level_workspace_bytes(ctypes.byref(level.c))
    if nb == 0:
        raise capi.BanetError("lm_level: unsupported level shape")
    if ws is None or ws.numel() < nb:
        ws = capi.workspace(nb, level.device)
    capi.check(L.banet_lm_level_ex_f32(ctypes.byref(level.c), ctypes.byref(mlp.c) if mlp is not None else None,
                                       float(l2_base), int(max_iters), int(bool(early_termination)),
                                       ctypes.byref(params) if params is not None else None,
                                       ctypes.byref(state.c), ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
    return ws


def lm_level_workspace_bytes(level):
    return capi.lib().banet_lm_level_workspace_bytes(ctypes.byref(level.c))


def profile_begin(max_launches):
    capi.check(capi.lib().banet_profile_begin(int(max_launches)))


def profile_end(max_tags=16):
    """-> {points_per_window: (launches, total_ms)} for the assembly kernel since profile_begin"""
    pts = (ctypes.c_int32 * max_tags)()
    cnt = (ctypes.c_int32 * max_tags)()
    ms = (ctypes.c_double * max_tags)()
    nt = ctypes.c_int32(0)
    capi.check(capi.lib().banet_profile_end(max_tags, pts, cnt, ms, ctypes.byref(nt)))
    return {int(pts[i]): (int(cnt[i]), float(ms[i])) for i in range(nt.value)}


# --------------------------------------------------------------------------------------
# per-level preparation (banet_resample_f32 / banet_target_map_f32 / banet_depth_output_f32)
# --------------------------------------------------------------------------------------
def resample(data, warp, clamp=False):
    """data [B,H,W,C], warp [B,N,2] -> [B,N,C].  clamp=False: tf.contrib.resampler.resampler
    (bundlenet.py:290,320,343-344,385); clamp=True: interpolate2d2 (legacy/utils_python.py:177-232)."""
    data, warp = capi.f32c(data), capi.f32c(warp)
    B, H, W, C = data.shape
    N = warp.shape[1]
    if tuple(warp.shape) != (B, N, 2):
        raise capi.BanetError("resample: expected data [B,H,W,C] and warp [B,N,2]; got %s %s" % (tuple(data.shape), tuple(warp.shape)))
    out = torch.empty((B, N, C), dtype=torch.float32, device=data.device)
    capi.check(capi.lib().banet_resample_f32(capi.ptr(data), capi.ptr(warp), capi.ptr(out), B, N, C, H, W,
                                             1 if clamp else 0, capi.stream()))
    return out


def target_map(img):
    """[B,H,W,C] -> [B,H,W,3C] = [f | gx | gy] (grad_fixed + concat: bundlenet.py:92-100,323-324; legacy/ba.py:116-118)."""
    img = capi.f32c(img)
    B, H, W, C = img.shape
    out = torch.empty((B, H, W, 3 * C), dtype=torch.float32, device=img.device)
    capi.check(capi.lib().banet_target_map_f32(capi.ptr(img), capi.ptr(out), B, H, W, C, capi.stream()))
    return out


def depth_output(init_depth, basis, Wc):
    """init_depth [B,...] + basis [B,N,K] . Wc [B,K,1] -> same shape as init_depth (bundlenet.py:397)."""
    init, basis, Wc = capi.f32c(init_depth), capi.f32c(basis), capi.f32c(Wc)
    B, K = basis.shape[0], basis.shape[-1]
    N = basis.numel() // (B * K)
    if init.numel() != B * N or Wc.numel() != B * K:
        raise capi.BanetError("depth_output: inconsistent shapes")
    out = torch.empty_like(init)
    capi.check(capi.lib().banet_depth_output_f32(capi.ptr(init), capi.ptr(basis), capi.ptr(Wc), capi.ptr(out), B, N, K,
                                                 capi.stream()))
    return out


# --------------------------------------------------------------------------------------
# differentiable layer support: per-pixel sampling statistics (sstats.hip)
# --------------------------------------------------------------------------------------
def sample_stats_forward(conv1, conv2, px, py):
    """banet_sample_stats_f32 -> (stats [B,N,8] = (M11, M12, M22, g1, g2, mask, 0, 0), absd [B,C] = sum_n |d|)."""
    conv1, conv2, px, py = capi.f32c(conv1), capi.f32c(conv2), capi.f32c(px), capi.f32c(py)
    B, N, C = conv1.shape
    H, W = conv2.shape[1], conv2.shape[2]
    if conv2.shape[0] != B or conv2.shape[3] != 3 * C or tuple(px.shape) != (B, N) or tuple(py.shape) != (B, N):
        raise capi.BanetError("sample_stats: expected conv1 [B,N,C], conv2 [B,H,W,3C], px / py [B,N]")
    L = capi.lib()
    G = L.banet_sample_stats_blocks(N)
    stats = torch.empty((B, N, 8), dtype=torch.float32, device=conv1.device)
    part = torch.empty((B, G, C), dtype=torch.float32, device=conv1.device)
    capi.check(L.banet_sample_stats_f32(capi.ptr(conv1), capi.ptr(conv2), capi.ptr(px), capi.ptr(py), B, N, C, H, W,
                                        capi.ptr(stats), capi.ptr(part), capi.stream()))
    return stats, part.sum(dim=1)


DETERMINISTIC_GRADIENTS = True   # sample_stats backward: fixed-order per-texel gather (True) or float-atomic scatter (False)


def sample_stats_grad(conv1, conv2, px, py, dstats, dabs, deterministic=None):
    """banet_sample_stats_grad_det_f32 (default: bit-reproducible) / banet_sample_stats_grad_f32 (float-atomic scatter)
    -> (dconv1 [B,N,C], dconv2 [B,H,W,3C], dpos [B,N,2])."""
    conv1, conv2, px, py = capi.f32c(conv1), capi.f32c(conv2), capi.f32c(px), capi.f32c(py)
    dstats, dabs = capi.f32c(dstats), capi.f32c(dabs)
    B, N, C = conv1.shape
    H, W = conv2.shape[1], conv2.shape[2]
    dconv1 = torch.empty_like(conv1)
    dconv2 = torch.zeros_like(conv2)
    dpos = torch.empty((B, N, 2), dtype=torch.float32, device=conv1.device)
    if DETERMINISTIC_GRADIENTS if deterministic is None else deterministic:
        L = capi.lib()
        nb = L.banet_sample_stats_grad_workspace_bytes(B, N, C, H, W)
        if nb == 0:
            raise capi.BanetError("sample_stats_grad: unsupported shape")
        ws = capi.workspace(nb, conv1.device)
        capi.check(L.banet_sample_stats_grad_det_f32(capi.ptr(conv1), capi.ptr(conv2), capi.ptr(px), capi.ptr(py), B, N, C, H, W,
                                                     capi.ptr(dstats), capi.ptr(dabs), capi.ptr(dconv1), capi.ptr(dconv2),
                                                     capi.ptr(dpos), ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
        return dconv1, dconv2, dpos
    capi.check(capi.lib().banet_sample_stats_grad_f32(capi.ptr(conv1), capi.ptr(conv2), capi.ptr(px), capi.ptr(py), B, N, C, H, W,
                                                      capi.ptr(dstats), capi.ptr(dabs), capi.ptr(dconv1), capi.ptr(dconv2),
                                                      capi.ptr(dpos), capi.stream()))
    return dconv1, dconv2, dpos


class _SampleStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, conv1, conv2, px, py):
        ctx.save_for_backward(conv1, conv2, px, py)
        stats, absd = sample_stats_forward(conv1, conv2, px, py)
        ctx.mark_non_differentiable(stats[..., 5:])
        return stats[..., :5], stats[..., 5], absd

    @staticmethod
    def backward(ctx, dstats5, _dmask, dabs):
        conv1, conv2, px, py = ctx.saved_tensors
        B, N, _ = conv1.shape
        dstats = torch.zeros((B, N, 8), dtype=torch.float32, device=conv1.device)
        dstats[..., :5] = dstats5
        dconv1, dconv2, dpos = sample_stats_grad(conv1, conv2, px, py, dstats, dabs.contiguous())
        return dconv1, dconv2, dpos[..., 0], dpos[..., 1]


def sample_stats(conv1, conv2, px, py):
    """Differentiable (M, g, mask, sum |d|) of bundlenet.py:230-243 without materialising samp / diff / grad:
    returns (stats [B,N,5] = (M11, M12, M22, g1, g2), mask [B,N], absd [B,C])."""
    return _SampleStats.apply(conv1, conv2, px, py)
