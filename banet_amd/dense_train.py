"""Differentiable dense BA: the fused forward kernels plus the fused backward of banet_amd/csrc/adjoint.hip
(reference: the TF graph of bundlenet.py:193-278 differentiated by tf.gradients + the registered
EquationConstructionGrad, bundlenet.py:79-82, utils.cu:465-694).

One autograd node per pyramid LEVEL (all of its fixed-count iterations, as bundlenet.py:376-397 unrolls them):
  forward   iteration by iteration with the product kernels (banet_ba_assemble_f32 + banet_ba_solve_update_f32), keeping the
            per-iteration state (R, T, Wc) and the small outputs of the assembly (AtA, Atb, sum |d|);
  backward  iterations in reverse: (a) the small part -- lambda MLP, damping, solve, SE(3)/W update -- is re-evaluated as
            a torch graph on the saved [B,P,P] / [B,P] / [B,C] tensors (the damped solve and its implicit-function gradient
            on banet_spd_solve_f32: lam = A^-1 g, dA = -lam x^T), giving dL/d(AtA, Atb, sum|d|), the direct dL/d(R, T, Wc) and the lambda-weight
            gradients; (b) banet_dense_adjoint_f32 turns the former into gradients of the feature maps, depth, basis and
            pose, accumulated over the iterations in place; (c) once per level banet_target_map_adjoint_f32 folds the
            [f|gx|gy] map adjoint into the target map's gradient.
No J / G / d / samp tensors exist in either direction and nothing is scattered with float atomics: gradients are
bit-reproducible.  Supported: the `bundle` variant with K <= 256 and the pose-only `bundle_camera` variant (bundlenet.py:122-191:
no depth basis, P = 6, every coefficient damped), C <= 256, two-frame AND multi-frame windows (round 3): a window's
normal equations are the sum of its target frames' two-frame terms embedded in the block-arrowhead matrix (AtA = sum_i E_i AtA_i
E_i^T), so its backward is the two-frame adjoint once per target frame on the sub-blocks
dL/dAtA_i = E_i^T (dL/dAtA) E_i, dL/dAtb_i = E_i^T dL/dAtb -- the source / depth / basis gradients accumulate over the frames, every
target frame gets its own map adjoint and pose gradient.
"""
import ctypes
import os
import warnings

import torch

from . import _capi as capi
from . import ops
from .bundlenet import AngleaAxisRotation, VMatrix


import os as _os

USE_SPD_SOLVE = _os.environ.get("BANET_SPD_SOLVE", "1") != "0"   # backward: the damped SPD system on banet_spd_solve_f32 (0: torch.linalg.solve_ex, A/B)


def _to_param(t, dev):
    return t.to(dev) if torch.is_tensor(t) else torch.as_tensor(t, dtype=torch.float32, device=dev)


def solve_update_graph(AtA, Atb, absres, N, R, T, Wc, layers, l2_base, solve=torch.linalg.solve, pairs=1, camera=False):
    """bundlenet.py:241-276 after the EquationConstruction op, as differentiable torch statements on the small tensors:
    avg -> lambda MLP -> damping (last coefficient undamped) -> matrix_solve -> SE(3) / W update.  pairs > 1: the multi-frame
    window of SURVEY.md 8(d) (banet_oracle.bundle_window_iteration): residual averaged over all frames, parameter order
    [pose_1 .. pose_pairs, depth], R [B,pairs,3,3], T [B,pairs,3,1].  camera: the pose-only CameraIteration,
    bundlenet.py:165-190 -- no l2_regularizer_base, all six coefficients damped (:181-182), Wc is [B,0,1]."""
    nb = AtA.shape[0]
    avg = (absres / float(N * pairs)).unsqueeze(1)                                   # :243
    h = avg
    for i, (w, b) in enumerate(layers):
        z = torch.matmul(h, w) + b
        h = torch.tanh(z) if i == 4 else torch.nn.functional.selu(z)
    lam = torch.linalg.vector_norm(avg, dim=-1, keepdim=True) ** (2.0 + h)           # :249
    diag = torch.diagonal(AtA, dim1=1, dim2=2)
    if camera:
        damp = diag + 1e-5                                                           # :181-182
    else:
        lam = l2_base * lam                                                          # :252-253
        damp = torch.cat([diag[:, :-1] + 1e-5, torch.zeros(nb, 1, device=diag.device, dtype=diag.dtype)], dim=-1)   # :266
    A = AtA + torch.diag_embed(damp * lam.squeeze(-1))
    sol = solve(A, Atb.unsqueeze(-1))                                                # :267
    Rn, Tn = [], []
    Rv, Tv = R.reshape(nb, pairs, 3, 3), T.reshape(nb, pairs, 3, 1)
    for i in range(pairs):
        o = 6 * i
        wx, wy, wz = sol[:, o + 0], sol[:, o + 1], sol[:, o + 2]
        dr = AngleaAxisRotation(wx, wy, wz)
        dv = VMatrix(wx.reshape(-1), wy.reshape(-1), wz.reshape(-1))
        Rn.append(torch.matmul(dr, Rv[:, i]))
        Tn.append(torch.matmul(dv, sol[:, o + 3:o + 6]) + torch.matmul(dr, Tv[:, i]))
    Rn, Tn = torch.stack(Rn, 1).reshape(R.shape), torch.stack(Tn, 1).reshape(T.shape)
    return Rn, Tn, Wc + sol[:, 6 * pairs:]


def spd_solve(A, b):
    """banet_spd_solve_f32: x [B,P,1] = A^-1 b for symmetric positive definite A [B,P,P] (blocked LDL^T in LDS, P >= 32)."""
    A, b = capi.f32c(A), capi.f32c(b)
    B, P = A.shape[0], A.shape[1]
    x = torch.empty((B, P, 1), dtype=torch.float32, device=A.device)
    capi.check(capi.lib().banet_spd_solve_f32(capi.ptr(A), capi.ptr(b), capi.ptr(x), B, P, capi.stream()))
    return x


class _SolveSPD(torch.autograd.Function):
    """x = A^-1 b for the damped normal matrix (symmetric positive definite) on the library's LDL^T kernel; backward = the
    implicit-function gradient with the same kernel: lam = A^-1 g (A = A^T), dA = -lam x^T, db = lam."""

    @staticmethod
    def forward(ctx, A, b):
        x = spd_solve(A, b)
        ctx.save_for_backward(A, x)
        return x

    @staticmethod
    def backward(ctx, g):
        A, x = ctx.saved_tensors
        lam = spd_solve(A, g)
        return -torch.matmul(lam, x.transpose(-1, -2)), lam


class _SolveNoCheck(torch.autograd.Function):
    """x = A^-1 b without the host-side `info` check of torch.linalg.solve (a device->host sync per call, and illegal inside
    a captured graph); backward = the implicit-function gradient: lam = A^-T g, dA = -lam x^T, db = lam."""

    @staticmethod
    def forward(ctx, A, b):
        x = torch.linalg.solve_ex(A, b, check_errors=False).result
        ctx.save_for_backward(A, x)
        return x

    @staticmethod
    def backward(ctx, g):
        A, x = ctx.saved_tensors
        lam = torch.linalg.solve_ex(A.transpose(-1, -2), g, check_errors=False).result
        return -torch.matmul(lam, x.transpose(-1, -2)), lam


def _small_grads(AtA, Atb, absres, R, T, Wc, gR, gT, gW, flat, N, l2_base, pairs=1, camera=False):
    """dL/d(AtA, Atb, sum|d|, R, T, Wc, lambda weights) of one iteration's small part, given dL/d(R', T', W')."""
    with torch.enable_grad():
        leaves = [t.detach().requires_grad_(True) for t in (AtA, Atb, absres, R, T, Wc)]
        lw = [t.detach().requires_grad_(True) for t in flat]
        spd_ok = AtA.is_cuda and 32 <= AtA.shape[-1] <= 180 and USE_SPD_SOLVE      # banet_spd_solve_f32 keeps the matrix in LDS
        R2, T2, W2 = solve_update_graph(leaves[0], leaves[1], leaves[2], N, leaves[3], leaves[4], leaves[5],
                                        [(lw[2 * i], lw[2 * i + 1]) for i in range(5)], l2_base,
                                        solve=_SolveSPD.apply if spd_ok else _SolveNoCheck.apply, pairs=pairs, camera=camera)
        outs, seeds = ([R2, T2], [gR, gT]) if camera else ([R2, T2, W2], [gR, gT, gW])      # camera: W2 = Wc, an empty tensor
        grads = torch.autograd.grad(outs, leaves + lw, seeds, allow_unused=True)
    return [g if g is not None else torch.zeros_like(t) for g, t in zip(grads, leaves + lw)]


class _SmallStepGraph:
    """The small part of a backward iteration as ONE captured HIP graph (torch.cuda.CUDAGraph): ~150 kernels on [B,P,P] /
    [B,C] tensors per iteration are launch-bound when issued one by one from Python.  Static input / output buffers; falls
    back to eager execution if capture is not possible (BANET_TRAIN_GRAPH=0 disables it)."""

    def __init__(self, shapes, dev, N, l2_base, pairs=1, camera=False):
        import os
        self.N, self.l2, self.pairs, self.camera = N, l2_base, pairs, camera
        self.inp = [torch.zeros(s, dtype=torch.float32, device=dev) for s in shapes]
        self.graph, self.out, self.error = None, None, None
        mode = os.environ.get("BANET_TRAIN_GRAPH", "1")
        # Captured for small batches only: there the ~150 small launches are CPU-bound (2 windows: 37.6 -> 27.3 ms per training
        # step); at 8 windows eager and replayed run the same, and at 32 windows replaying the graph between the large adjoint
        # kernels was pathologically slow on ROCm 7.2 (192 -> 370-510 ms per step, cause not found) -- BANET_TRAIN_GRAPH=2 forces it.
        if mode == "0" or (shapes[0][0] > 8 and mode != "2"):
            self.error = "disabled" if mode == "0" else "batch > 8"
            return
        for t in self.inp[3:4]:
            t.copy_(torch.eye(3, device=dev).expand_as(t))          # a valid rotation / SPD system for the warm-up (broadcasts over frames)
        self.inp[0].copy_(torch.eye(shapes[0][-1], device=dev).expand_as(self.inp[0]))
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._run()
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._run()
            self.graph, self.out = g, out
        except Exception as e:                                       # capture not possible here: eager execution
            self.graph, self.out, self.error = None, None, repr(e)
            torch.cuda.synchronize(dev)

    def _run(self):
        i = self.inp
        return _small_grads(i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9:], self.N, self.l2, self.pairs, self.camera)

    def __call__(self, tensors):
        if self.graph is None:
            return _small_grads(*tensors[:9], tensors[9:], self.N, self.l2, self.pairs, self.camera)
        for dst, src in zip(self.inp, tensors):
            dst.copy_(src.reshape(dst.shape))
        self.graph.replay()
        return [o.clone() for o in self.out]


_small_cache = {}          # shape key -> _SmallStepGraph (static buffers + a private graph memory pool each): bounded, see _small_step
def _cache_limit():
    """BANET_SMALL_STEP_CACHE: at least 1 (0 / negative / non-integer values fall back to the default instead of breaking the eviction loop)"""
    try:
        return max(1, int(os.environ.get("BANET_SMALL_STEP_CACHE", "32")))
    except ValueError:
        return 32


_SMALL_CACHE_MAX = _cache_limit()   # >= levels x configurations in flight (5 levels x 2 configs = 10)
_evict_warned = False


def clear_small_step_cache():
    """Drop the captured small-step graphs and their static buffers (multi-resolution / multi-batch training keeps at most
    _SMALL_CACHE_MAX of them alive anyway)."""
    _small_cache.clear()


def small_step_modes():
    """{shape key: "graph" | "eager (<reason>)"} of the small backward steps built so far (diagnostics / benchmarks)."""
    return {str(k[0][0]): ("graph" if v.graph is not None else "eager (%s)" % (v.error or "capture failed")) for k, v in _small_cache.items()}


def _small_step(tensors, N, l2_base, pairs=1, camera=False):
    dev = tensors[0].device
    key = (tuple(tuple(t.shape) for t in tensors), str(dev), int(N), float(l2_base), int(pairs), bool(camera))
    st = _small_cache.pop(key, None)
    if st is None:
        st = _SmallStepGraph([tuple(t.shape) for t in tensors], dev, N, l2_base, pairs, camera)
        while len(_small_cache) >= _SMALL_CACHE_MAX:          # least recently used first (dicts keep insertion order)
            old = next(iter(_small_cache))
            _small_cache.pop(old)
            global _evict_warned
            if not _evict_warned:                              # once per process: multi-resolution training would repeat it every step
                _evict_warned = True
                warnings.warn("banet_amd.dense_train: small-step graph cache full (%d entries): evicting %s; the next use of that "
                              "shape re-captures its graph (raise BANET_SMALL_STEP_CACHE; further evictions are silent)"
                              % (_SMALL_CACHE_MAX, old[0][0]), RuntimeWarning)
    _small_cache[key] = st
    return st(tensors)


# the backward's small step (lambda MLP / damping / solve / SE(3) update adjoint) as four HIP launches (banet_small_step_adjoint_f32,
# csrc/smallstep.hip, round 6) instead of the ~150-launch torch graph above; BANET_SMALL_STEP_HIP=0 keeps the torch graph (A/B)
SMALL_STEP_HIP = os.environ.get("BANET_SMALL_STEP_HIP", "1") != "0"


class SmallStepHip:
    """Per level-backward state of the HIP small step: the accumulated lambda-weight gradients (ten tensors, zero-initialised;
    every call adds to them on the device) and the workspace."""

    def __init__(self, variant, B, N, C, K, pairs, mlp, l2_base, dev):
        self.args = (ops._VARIANT_OF[variant] if isinstance(variant, str) else int(variant), int(B), int(N), int(C), int(K), int(pairs))
        self.l2, self.mlp, self.dev = float(l2_base), mlp, dev
        nb = capi.lib().banet_small_step_adjoint_workspace_bytes(*self.args)
        if nb == 0:
            raise capi.BanetError("small step: unsupported shape")
        self.ws = capi.workspace(nb, dev)
        dims = [C, 2 * C, 4 * C, 2 * C, C, 1]
        sizes = []
        for i in range(5):
            sizes += [dims[i] * dims[i + 1], dims[i + 1]]
        offs = [0]
        for n_ in sizes:
            offs.append(offs[-1] + (n_ + 3) // 4 * 4)                     # 16-byte aligned views of ONE zero-filled buffer (one fill launch)
        flatbuf = torch.zeros(offs[-1], dtype=torch.float32, device=dev)
        self.glayers = []
        for i in range(5):
            self.glayers += [flatbuf[offs[2 * i]:offs[2 * i] + sizes[2 * i]].view(dims[i], dims[i + 1]),
                             flatbuf[offs[2 * i + 1]:offs[2 * i + 1] + sizes[2 * i + 1]]]
        self.g = capi.Mlp()
        for i in range(5):
            self.g.w[i] = self.glayers[2 * i].data_ptr()
            self.g.b[i] = self.glayers[2 * i + 1].data_ptr()

    @staticmethod
    def supported(variant, B, N, C, K, pairs, dev):
        v = ops._VARIANT_OF[variant] if isinstance(variant, str) else int(variant)
        return (SMALL_STEP_HIP and torch.device(dev).type == "cuda" and
                capi.lib().banet_small_step_adjoint_workspace_bytes(v, int(B), int(N), int(C), int(K), int(pairs)) > 0)

    def __call__(self, AtA, Atb, absres, delta, R, T, gR, gT, gW):
        """-> (gAtA [B,P,P], gAtb [B,P], gabs [B,C], dR [B,pairs,3,3], dT [B,pairs,3,1]); dL/dWc = gW is the caller's."""
        v, B, N, C, K, pairs = self.args
        P = 6 * pairs + K
        dev = self.dev
        f = capi.f32c
        ins = [f(AtA), f(Atb), f(absres), f(delta), f(R), f(T), f(gR), f(gT), f(gW) if K else None]
        gAtA = torch.empty((B, P, P), dtype=torch.float32, device=dev)
        gAtb = torch.empty((B, P), dtype=torch.float32, device=dev)
        gabs = torch.empty((B, C), dtype=torch.float32, device=dev)
        dR = torch.empty((B, pairs, 3, 3), dtype=torch.float32, device=dev)
        dT = torch.empty((B, pairs, 3, 1), dtype=torch.float32, device=dev)
        capi.check(capi.lib().banet_small_step_adjoint_f32(
            v, B, N, C, K, pairs, self.l2, ctypes.byref(self.mlp.c), *[capi.ptr(x) if x is not None else None for x in ins],
            capi.ptr(gAtA), capi.ptr(gAtb), capi.ptr(gabs), capi.ptr(dR), capi.ptr(dT), ctypes.byref(self.g),
            ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel(), capi.stream()))
        return gAtA, gAtb, gabs, dR, dT


ADJOINT_OVERWRITE, ADJOINT_OVERWRITE_MAP = 1, 2      # banet_hip.h: BANET_ADJOINT_OVERWRITE, BANET_ADJOINT_OVERWRITE_MAP
ADJOINT_FOLD_TARGET = 4                               # banet_hip.h: BANET_ADJOINT_FOLD_TARGET
ADJOINT_REUSE_DEPTH_SEED = 8                          # banet_hip.h: BANET_ADJOINT_REUSE_DEPTH_SEED (target frames 2.. of a multi-frame window)
REUSE_MODE = os.environ.get("BANET_ADJOINT_REUSE", "1")   # "0": every frame's call recomputes z2 / zeta / e (A/B; same bits)


def ADJOINT_TILE_SHAPE(k):
    """banet_hip.h: BANET_ADJOINT_TILE_SHAPE(k) -- development switch (A/B): 1 .. 5 = adj_tile_kernel with 8x4 / 4x4 / 8x2 / 8x7 / 4x2 texel tiles,
    8 .. 13 = adj_tile2_kernel (two visits per wave instruction) with 8x4 / 8x3 / 8x5 / 8x7 / 16x3 / 16x2; 0 = the default"""
    return (int(k) & 15) << 4


# the backward of a dense level writes the target map's gradient per texel tile (round 6) instead of 3C adjoint rows + a per-texel
# gather + the [f|gx|gy] map adjoint + its fold: BANET_ADJOINT_FOLD=0 keeps the round-5 path (A/B); BANET_ADJOINT_TILE=k the tile shape (A/B)
FOLD_MODE = os.environ.get("BANET_ADJOINT_FOLD", "1")
TILE_SHAPE = int(os.environ.get("BANET_ADJOINT_TILE", "0"))


def dense_adjoint(problem, R, T, Wc, gAtA, gAtb, gabs, dsrc, dmap3, ddepth, dbasis, ws=None, overwrite=False, overwrite_map=None,
                  fold=False, extra_flags=0):
    """banet_dense_adjoint_ex_f32 -> dpose [B, 12 + K]; dsrc / dmap3 / ddepth / dbasis are accumulated in place, or -- overwrite:
    the first call on fresh (uninitialised) buffers -- written, every entry (overwrite_map: dmap3 on its own; default = overwrite).
    fold (BANET_ADJOINT_FOLD_TARGET, dense levels): `dmap3` is the target map's gradient [B,H,W,C] itself."""
    overwrite_map = overwrite if overwrite_map is None else overwrite_map
    L = capi.lib()
    flags = ((ADJOINT_OVERWRITE if overwrite else 0) | (ADJOINT_OVERWRITE_MAP if overwrite_map else 0) |
             (ADJOINT_FOLD_TARGET if fold else 0) | int(extra_flags))
    nb = L.banet_dense_adjoint_workspace_bytes_ex(ctypes.byref(problem.c), flags)
    if nb == 0:
        raise capi.BanetError("dense_adjoint: unsupported level (bundle with 1 <= K <= 256 or bundle_camera, dense two-frame windows, C <= 256)")
    if ws is None or ws.numel() < nb:
        ws = capi.workspace(nb, problem.device)
    dpose = torch.empty((problem.B, 12 + problem.K), dtype=torch.float32, device=problem.device)
    args = [capi.f32c(x) for x in (R, T, Wc, gAtA, gAtb, gabs)]          # (K = 0: Wc / dbasis are empty, never dereferenced)
    capi.check(L.banet_dense_adjoint_ex_f32(ctypes.byref(problem.c), *[capi.ptr(x) for x in args], capi.ptr(dsrc), capi.ptr(dmap3),
                                            capi.ptr(ddepth), capi.ptr(dbasis), capi.ptr(dpose), flags,
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(), capi.stream()))
    return dpose, ws


def target_map_adjoint(dmap3, dimg, overwrite=False):
    """banet_target_map_adjoint_ex_f32: dimg [B,H,W,C] += (overwrite: =) the adjoint of banet_target_map_f32 applied to dmap3 [B,H,W,3C]."""
    B, H, W, C = dimg.shape
    capi.check(capi.lib().banet_target_map_adjoint_ex_f32(capi.ptr(dmap3), capi.ptr(dimg), B, H, W, C, ADJOINT_OVERWRITE if overwrite else 0,
                                                          capi.stream()))
    return dimg


def _pair_problems(ba, li):
    """Two-frame views of a multi-frame level, one per target frame (the adjoint kernels take [B,H,W,C] target maps): contiguous
    copies of tgt[:, i], rebuilt for every backward pass (the level tensors may have changed since the last one)."""
    lv, prob = ba.levels[li], ba.problems[li]
    B, H, W, C = lv.B, lv.H, lv.W, lv.C
    out = []
    for i in range(prob.pairs):
        tgt_i = lv.tgt.detach()[:, i].contiguous()
        out.append(ops.LevelProblem(ba.variant, lv.src.detach(), tgt_i, lv.depth.detach().reshape(B, H * W), H, W, C,
                                    basis=None if lv.basis is None else lv.basis.detach().reshape(B, H * W, -1), intr=ba.intr,
                                    scale=lv.scale, dense=True,
                                    tgt_has_grad=False, normalize_rays=True, pairs=1))
    return out


class _LevelSolve(torch.autograd.Function):
    """All fixed-count iterations of one pyramid level."""

    @staticmethod
    def forward(ctx, ba, li, n_iter, src, tgt, depth, basis, R, T, Wc, *flat_layers):
        prob, mlp = ba.problems[li], ba.mlps[li]
        pairs = prob.pairs
        st = ops.LmState(R.detach(), T.detach(), Wc.detach(), P=6 * pairs + ba.K, pairs=pairs)
        saved = []
        sws = None                                     # P > ~190: the solve's matrix lives in a workspace, shared by the iterations
        for _ in range(n_iter):
            Ri, Ti, Wi = st.R.clone(), st.T.clone(), st.Wc.clone()
            AtA, Atb, absres, nvalid = ops.ba_assemble(prob, st.R, st.T, st.Wc)
            sws = ops.ba_solve_update(prob, mlp, ba.l2_base, AtA, Atb, absres, nvalid, st, sws)
            saved.append((Ri, Ti, Wi, AtA, Atb, absres, st.delta.clone()))
        ctx.ba, ctx.li, ctx.saved = ba, li, saved
        ctx.layers = flat_layers
        ctx.shapes = (src.shape, tgt.shape, depth.shape, None if basis is None else basis.shape, R.shape, T.shape)
        return st.R.clone().reshape(R.shape), st.T.clone().reshape(T.shape), st.Wc.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gR, gT, gW):
        ba, li = ctx.ba, ctx.li
        prob = ba.problems[li]
        pairs = prob.pairs
        dev = prob.device
        camera = ba.variant == "bundle_camera"
        B, N, C, K, H, W = prob.B, prob.N, prob.C, prob.K, prob.c.H, prob.c.W
        pprobs = [prob] if pairs == 1 else _pair_problems(ba, li)
        # no zero-fills: the first adjoint call that touches a buffer writes it (BANET_ADJOINT_OVERWRITE), the later ones accumulate --
        # 25 GB of fills and as many bytes of reads per 32-window 640x480 level
        # the target gradient per texel tile: no 3C rows, no [f|gx|gy] map adjoint, no fold pass -- where the library takes the level
        # in that mode (dense layout, a window's maps below 4 GB: the tile kernels use 32-bit byte offsets); else the round-5 path
        fold = FOLD_MODE != "0" and capi.lib().banet_dense_adjoint_workspace_bytes_ex(ctypes.byref(pprobs[0].c), ADJOINT_FOLD_TARGET) != 0
        xflags = ADJOINT_TILE_SHAPE(TILE_SHAPE)
        dsrc = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        dmap3 = [torch.empty((B, H, W, C if fold else 3 * C), dtype=torch.float32, device=dev) for _ in range(pairs)]
        ddepth = torch.empty((B, N), dtype=torch.float32, device=dev)
        dbasis = torch.empty((B, N, K), dtype=torch.float32, device=dev)
        flat = ctx.layers
        gR = torch.zeros(B, pairs, 3, 3, device=dev) if gR is None else gR.reshape(B, pairs, 3, 3)
        gT = torch.zeros(B, pairs, 3, 1, device=dev) if gT is None else gT.reshape(B, pairs, 3, 1)
        gW = torch.zeros(B, K, 1, device=dev) if gW is None else gW.reshape(B, K, 1)
        o = 6 * pairs
        ws = None
        first = True
        if not ctx.saved:                                  # (no iteration ran: nothing writes the buffers)
            for t in [dsrc, ddepth, dbasis] + dmap3:
                t.zero_()
        hip_small = SmallStepHip(ba.variant, B, N, C, K, pairs, ba.mlps[li], ba.l2_base, dev) \
            if SmallStepHip.supported(ba.variant, B, N, C, K, pairs, dev) else None
        glayers = None if hip_small is not None else [torch.zeros_like(t) for t in flat]
        for Ri, Ti, Wi, AtA, Atb, absres, delta in reversed(ctx.saved):
            Rv, Tv = Ri.reshape(B, pairs, 3, 3), Ti.reshape(B, pairs, 3, 1)
            if hip_small is not None:       # four launches; the lambda-weight gradients accumulate on the device
                gAtA, gAtb, gabs, dR, dT = hip_small(AtA, Atb, absres, delta, Rv, Tv, gR, gT, gW)
                gR, gT = dR, dT             # (gW: W' = W + sol, the upstream gradient passes through)
                gW = gW.clone()
            else:
                grads = _small_step([AtA, Atb, absres, Rv, Tv, Wi, gR, gT, gW] + [t.detach() for t in flat], N, ba.l2_base, pairs, camera)
                gAtA, gAtb, gabs, dR, dT, dW = grads[:6]
                for acc, g in zip(glayers, grads[6:]):
                    acc += g
                gR, gT, gW = dR.reshape(B, pairs, 3, 3).clone(), dT.reshape(B, pairs, 3, 1).clone(), dW.reshape(B, K, 1).clone()
            for i in range(pairs):
                if pairs == 1:
                    gA_i, gb_i = gAtA, gAtb
                else:       # E_i^T (dL/dAtA) E_i: the frame's pose block, its cross blocks with the depth block, the depth block
                    idx = torch.cat([torch.arange(6 * i, 6 * i + 6, device=dev), torch.arange(o, o + K, device=dev)])
                    gA_i = gAtA.index_select(1, idx).index_select(2, idx).contiguous()
                    gb_i = gAtb.index_select(1, idx).contiguous()
                # dmap3[i] is fresh for every frame of the first iteration; dsrc / ddepth / dbasis are shared by the frames
                dpose, ws = dense_adjoint(pprobs[i], Rv[:, i].contiguous(), Tv[:, i].contiguous(), Wi, gA_i, gb_i, gabs, dsrc,
                                          dmap3[i], ddepth, dbasis, ws, overwrite=first and i == 0, overwrite_map=first, fold=fold,
                                          extra_flags=xflags | (ADJOINT_REUSE_DEPTH_SEED if (i > 0 and REUSE_MODE != "0") else 0))
                gR[:, i] += dpose[:, 0:9].reshape(B, 3, 3)
                gT[:, i] += dpose[:, 9:12].reshape(B, 3, 1)
                gW += dpose[:, 12:].reshape(B, K, 1)
            first = False
        if fold:                                           # dmap3[i] IS the frame's target gradient
            dtgt = dmap3[0].view(B, 1, H, W, C) if pairs == 1 else torch.stack(dmap3, 1)
        elif pairs == 1:                                   # written in place: no fill, no copy
            dtgt = torch.empty((B, 1, H, W, C), dtype=torch.float32, device=dev)
            target_map_adjoint(dmap3[0], dtgt.view(B, H, W, C), overwrite=True)
        else:
            dtgt = torch.empty((B, pairs, H, W, C), dtype=torch.float32, device=dev)
            for i in range(pairs):
                di = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
                target_map_adjoint(dmap3[i], di, overwrite=True)
                dtgt[:, i] = di
        if hip_small is not None:
            glayers = [g.reshape(t.shape) for g, t in zip(hip_small.glayers, flat)]
        s_src, s_tgt, s_dep, s_bas, s_R, s_T = ctx.shapes
        return (None, None, None, dsrc.reshape(s_src), dtgt.reshape(s_tgt), ddepth.reshape(s_dep),
                None if s_bas is None else dbasis.reshape(s_bas), gR.reshape(s_R), gT.reshape(s_T), gW) + tuple(glayers)


def solve_differentiable(ba, levels, lambda_weights, iters_per_level, R=None, T=None, Wc=None):
    """Differentiable DenseBA.solve with fixed iteration counts: `levels` = the DenseLevel objects `ba` was built from (their
    src / tgt / depth / basis tensors may require grad), `lambda_weights` = per level five (filters, biases) pairs (tensors
    that may require grad, or arrays).  Returns (R [B,3,3], T [B,3,1], Wc [B,K,1]) attached to the autograd graph
    (multi-frame windows: R [B,pairs,3,3], T [B,pairs,3,1]; the pose-only `bundle_camera` variant: K = 0, Wc is empty)."""
    if ba.variant not in ("bundle", "bundle_camera"):
        raise capi.BanetError("solve_differentiable: bundle / bundle_camera variants only")
    dev = ba.intr.device
    B, K, pairs = ba.B, ba.K, ba.pairs
    if pairs == 1:
        R = torch.eye(3, device=dev).repeat(B, 1, 1) if R is None else R
        T = torch.zeros(B, 3, 1, device=dev) if T is None else T
    else:
        R = torch.eye(3, device=dev).repeat(B, pairs, 1, 1) if R is None else R.reshape(B, pairs, 3, 3)
        T = torch.zeros(B, pairs, 3, 1, device=dev) if T is None else T.reshape(B, pairs, 3, 1)
    Wc = torch.zeros(B, K, 1, device=dev) if Wc is None else Wc
    for li, (lv, lw, n_it) in enumerate(zip(levels, lambda_weights, iters_per_level)):
        if int(n_it) <= 0:
            continue
        flat = []
        for w, b in lw:
            w = _to_param(w, dev)
            flat += [w.reshape(w.shape[-2], w.shape[-1]), _to_param(b, dev).reshape(-1)]
        ba.mlps[li] = ops.MlpWeights([(flat[2 * i].detach(), flat[2 * i + 1].detach()) for i in range(5)], dev)
        R, T, Wc = _LevelSolve.apply(ba, li, int(n_it), lv.src, lv.tgt, lv.depth, lv.basis, R, T, Wc, *flat)
    return R, T, Wc


# ---- the sparse-point training iteration (round 5): what the reference actually trains on --------------------------------------
class _SparseIteration(torch.autograd.Function):
    """ONE BundleIteration / CameraIteration on N sampled points in the reference's own layout (bundlenet.py:122-278 as called from
    bundlenet.py:332-399: conv1 [B,N,C], the [f|gx|gy] target map [B,H,W,3C], rays p [B,3,N] and per-point level intrinsics) as a
    single autograd node on the fused kernels:
      forward   banet_ba_assemble_f32 (gather + SYRK + reduce) and banet_ba_solve_update_f32 -- the inference path, 4-5 launches;
      backward  the small step (lambda MLP, damping, solve by implicit differentiation, SE(3) / W update: _small_step, as the dense
                path's) and banet_dense_adjoint_f32 on the sparse level (adjoint.hip, adj_pixel_kernel<.., SP = true>): gradients
                of conv1, the [f|gx|gy] map, D, the basis, R, T, W and the lambda weights.  No samp / diff / grad / J tensors, no
                float atomics (the map adjoint is gathered per texel in a fixed order): bit-reproducible.
    fx / fy / ox / oy / p are data (bundlenet.py:112-120 computes them from the sampled points): no gradient."""

    @staticmethod
    def forward(ctx, variant, mlp, l2_base, last, conv1, conv2, D, Bs, R, T, W, fx, fy, ox, oy, p, *flat_layers):
        nb, H, Wd, C3 = conv2.shape
        C = conv1.shape[2]
        K = 0 if Bs is None else Bs.shape[-1]
        prob = ops.LevelProblem(variant, conv1.detach(), conv2.detach(), D.detach(), H, Wd, C, basis=None if Bs is None else Bs.detach(),
                                rays=p.detach(), fx=fx.detach(), fy=fy.detach(), ox=ox.detach(), oy=oy.detach(), dense=False,
                                tgt_has_grad=True)
        # the update kernel works in place on a private copy of the state; the inputs themselves are the state before the update
        R0, T0 = capi.f32c(R.detach()), capi.f32c(T.detach())
        W0 = capi.f32c(W.detach().reshape(nb, K, 1)) if K else torch.zeros(nb, 0, 1, device=conv1.device)
        st = ops.LmState(R0.clone(), T0.clone(), None if K == 0 else W0.clone(), P=6 + K)
        AtA, Atb, absres, nvalid = ops.ba_assemble(prob, st.R, st.T, st.Wc)
        ops.ba_solve_update(prob, mlp, l2_base, AtA, Atb, absres, nvalid, st)
        ctx.prob, ctx.l2, ctx.camera = prob, float(l2_base), K == 0
        ctx.variant, ctx.mlp = variant, mlp
        ctx.saved = (R0, T0, W0, AtA, Atb, absres, st.delta)          # (st is private to this node: no copies)
        ctx.flat = flat_layers
        ctx.shapes = (conv1.shape, conv2.shape, D.shape, None if Bs is None else Bs.shape, R.shape, T.shape, None if W is None else W.shape)
        last.update(AtA=AtA, Atb=Atb, lam=st.lambda_out, delta=st.delta)
        # the problem aliases the (detached) inputs: remember their versions, so that an in-place change between forward and
        # backward is an error here as it would be for tensors kept with save_for_backward
        ctx.inputs = [t for t in (conv1, conv2, D, Bs, fx, fy, ox, oy, p, R, T, W) if torch.is_tensor(t)]
        ctx.versions = [t._version for t in ctx.inputs]
        Wn = st.Wc.reshape(W.shape) if K else None
        return st.R.reshape(R.shape), st.T.reshape(T.shape), Wn

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gR, gT, gW):
        prob = ctx.prob
        if any(t._version != v for t, v in zip(ctx.inputs, ctx.versions)):
            raise RuntimeError("one of the tensors the fused BA iteration reads (features, target map, depth, basis, rays, intrinsics) "
                               "was modified in place between its forward and its backward")
        B, N, C, K = prob.B, prob.N, prob.C, prob.K
        H, Wd = prob.c.H, prob.c.W
        dev = prob.device
        R0, T0, W0, AtA, Atb, absres, delta = ctx.saved
        gR = torch.zeros(B, 1, 3, 3, device=dev) if gR is None else gR.reshape(B, 1, 3, 3)
        gT = torch.zeros(B, 1, 3, 1, device=dev) if gT is None else gT.reshape(B, 1, 3, 1)
        gW = torch.zeros(B, K, 1, device=dev) if (gW is None or K == 0) else gW.reshape(B, K, 1)
        flat = ctx.flat
        if SmallStepHip.supported(ctx.variant, B, N, C, K, 1, dev):       # four launches (csrc/smallstep.hip)
            hs = SmallStepHip(ctx.variant, B, N, C, K, 1, ctx.mlp, ctx.l2, dev)
            gAtA, gAtb, gabs, dR, dT = hs(AtA, Atb, absres, delta, R0.reshape(B, 1, 3, 3), T0.reshape(B, 1, 3, 1), gR, gT, gW)
            dW, glayers = gW, tuple(g.reshape(t.shape) for g, t in zip(hs.glayers, flat))
        else:
            grads = _small_step([AtA, Atb, absres, R0.reshape(B, 1, 3, 3), T0.reshape(B, 1, 3, 1), W0, gR.contiguous(), gT.contiguous(),
                                 gW.contiguous()] + [t.detach() for t in flat], N, ctx.l2, 1, ctx.camera)
            gAtA, gAtb, gabs, dR, dT, dW = grads[:6]
            glayers = tuple(grads[6:])
        dsrc = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        dmap3 = torch.empty((B, H, Wd, 3 * C), dtype=torch.float32, device=dev)
        ddepth = torch.empty((B, N), dtype=torch.float32, device=dev)
        dbasis = torch.empty((B, N, K), dtype=torch.float32, device=dev)
        dpose, _ = dense_adjoint(prob, R0, T0, W0, gAtA, gAtb, gabs, dsrc, dmap3, ddepth, dbasis, None, overwrite=True)
        dR = dR.reshape(B, 3, 3) + dpose[:, 0:9].reshape(B, 3, 3)
        dT = dT.reshape(B, 3, 1) + dpose[:, 9:12].reshape(B, 3, 1)
        s1, s2, sD, sB, sR, sT, sW = ctx.shapes
        dWn = None if K == 0 else (dW.reshape(B, K, 1) + dpose[:, 12:].reshape(B, K, 1)).reshape(sW)
        ctx.prob = ctx.inputs = None          # (release the aliased inputs now, not when the graph node dies)
        return (None, None, None, None, dsrc.reshape(s1), dmap3.reshape(s2), ddepth.reshape(sD), None if sB is None else dbasis.reshape(sB),
                dR.reshape(sR), dT.reshape(sT), dWn, None, None, None, None, None) + glayers


def sparse_iteration_supported(conv1, conv2, Bs, D=None, R=None, T=None):
    """what the fused sparse node accepts (adjoint.hip): float32 CUDA tensors on ONE device, C <= 256, K <= 256, not C > 128 together
    with K > 128, a [f|gx|gy] map of at least 4 x 4 texels (the gather plans refuse smaller ones: the caller falls back to the lean
    torch graph instead of raising)"""
    C = conv1.shape[-1]
    K = 0 if Bs is None else Bs.shape[-1]
    tensors = [t for t in (conv1, conv2, Bs, D, R, T) if torch.is_tensor(t)]
    same = all(t.is_cuda and t.device == conv1.device and t.dtype == torch.float32 for t in tensors)
    return (same and conv2.dim() == 4 and conv2.shape[1] >= 4 and conv2.shape[2] >= 4 and 1 <= C <= 256 and K <= 256 and
            not (C > 128 and K > 128) and conv2.shape[-1] == 3 * C)


def sparse_iteration(variant, mlp, l2_base, conv1, conv2, D, Bs, R, T, W, fx, fy, ox, oy, p, layers):
    """-> (R', T', W' or None, diagnostics): see _SparseIteration.  layers: five (filters [Cin,Cout], biases [Cout]) pairs (tensors)."""
    flat = []
    for w, b in layers:
        flat += [w.reshape(w.shape[-2], w.shape[-1]), b.reshape(-1)]
    last = {}
    R2, T2, W2 = _SparseIteration.apply(variant, mlp, l2_base, last, conv1, conv2, D, Bs, R, T, W, fx, fy, ox, oy, p, *flat)
    return R2, T2, W2, last
