"""Host-side mirror of /root/reference/bundlenet.py for PyTorch-ROCm callers.

Same module functions, same `BundleNet` method names, argument order and tensor layouts
(NHWC feature maps, `[B,N,.]` per-point tensors, `fx..` as `[B,N]`, `intrisic [B,4,1]`), so an
encoder/decoder CNN written against the reference can feed this layer unchanged.  The
iteration bodies (`CameraIteration`, `BundleIteration`) do NOT build a graph of small ops:
each is one fused assembly launch + one solve/update launch in libbanet_hip.so.

Differences from the reference, on purpose:
  * `VMatrix` applies the B=1 semantics to every item (the reference's stack/reshape is
    only correct for B=1);
  * a zero rotation update yields identity instead of 0/0 = NaN;
  * NaN projections are masked out in both variants.
"""
import math

import torch

from . import ops

__all__ = ["rotation2quaternion", "AngleaAxisRotation", "VMatrix", "CameraJacobianMatrix", "DepthJacobianMatrix",
           "equation_construction", "equation_construction_grad", "equation_construction_gradient", "resampler", "BundleNet",
           "lambda_weights_from_variables", "lambda_weights_to_variables"]

equation_construction = ops.equation_construction            # bundlenet.py:77
equation_construction_grad = ops.equation_construction_grad  # bundlenet.py:78


def equation_construction_gradient(inputs, left_grad, right_grad):
    """bundlenet.py:79-82, the function the reference registers with @ops.RegisterGradient("EquationConstruction"):
    `inputs` = the op's (jacobian, gradient, difference) -> their gradients.  (Registered here through
    torch.autograd / torch.library in banet_amd/ops.py.)"""
    return ops.equation_construction_grad(inputs[0], inputs[1], inputs[2], left_grad, right_grad)


def rotation2quaternion(R, name=None):
    """bundlenet.py:6-15"""
    diag = 1.0 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    q0 = torch.sqrt(diag) / 2.0
    q1 = (R[:, 2, 1] - R[:, 1, 2]) / (4.0 * q0)
    q2 = (R[:, 0, 2] - R[:, 2, 0]) / (4.0 * q0)
    q3 = (R[:, 1, 0] - R[:, 0, 1]) / (4.0 * q0)
    q = torch.stack([q0, q1, q2, q3], dim=1)
    return q / torch.sqrt(torch.clamp((q * q).sum(dim=1, keepdim=True), min=1e-12))


def AngleaAxisRotation(wx, wy, wz, name=None):
    """bundlenet.py:17-37; wx,wy,wz [B,1] -> [B,3,3]"""
    ones = torch.ones_like(wx)
    theta = torch.clamp(torch.sqrt(wx * wx + wy * wy + wz * wz), min=1e-6)
    wx, wy, wz = wx / theta, wy / theta, wz / theta
    c, s = torch.cos(theta), torch.sin(theta)
    m = torch.stack([c + wx * wx * (ones - c), wz * s + wx * wy * (ones - c), -wy * s + wx * wz * (ones - c),
                     wx * wy * (ones - c) - wz * s, c + wy * wy * (ones - c), wx * s + wy * wz * (ones - c),
                     wy * s + wx * wz * (ones - c), -wx * s + wy * wz * (ones - c), c + wz * wz * (ones - c)], dim=-1)
    return m.reshape(-1, 3, 3).transpose(1, 2)


def VMatrix(wx, wy, wz, name=None):
    """bundlenet.py:39-46, per item."""
    wx, wy, wz = wx.reshape(-1), wy.reshape(-1), wz.reshape(-1)
    theta = torch.sqrt(wx * wx + wy * wy + wz * wz)
    c, s = torch.cos(theta), torch.sin(theta)
    z = torch.zeros_like(wx)
    K = torch.stack([z, -wz, wy, wz, z, -wx, -wy, wx, z], dim=-1).reshape(-1, 3, 3)
    a = ((1 - c) / (theta * theta))[:, None, None]
    b = ((theta - s) / (theta * theta * theta))[:, None, None]
    eye = torch.eye(3, dtype=wx.dtype, device=wx.device)[None]
    return eye + a * K + b * torch.matmul(K, K)


def CameraJacobianMatrix(x, y, Z, fx, fy, name=None):
    """bundlenet.py:49-61 -> [B,N,2,6] (note the leading minus)."""
    xy = x * y
    xx = -1.0 - x * x
    x_z = x / Z
    yy = 1.0 + y * y
    y_z = y / Z
    iz = 1.0 / Z
    zeros = torch.zeros_like(xy)
    dx = fx.unsqueeze(-1) * torch.stack([xy, xx, y, -iz, zeros, x_z], dim=2)
    dy = fy.unsqueeze(-1) * torch.stack([yy, -xy, -x, zeros, -iz, y_z], dim=2)
    return -torch.stack([dx, dy], dim=2)


def DepthJacobianMatrix(rx, ry, rz, x, y, Z, fx, fy, name=None):
    """bundlenet.py:63-74; rx.. [B,1,N] -> [B,N,2]"""
    rx, ry, rz = rx.squeeze(1), ry.squeeze(1), rz.squeeze(1)
    dx = fx * ((rx - rz * x) / Z)
    dy = fy * ((ry - rz * y) / Z)
    return torch.stack([dx, dy], dim=2)


def resampler(data, warp):
    """tf.contrib.resampler.resampler: bilinear, zero padding.  data [B,H,W,C], warp [B,N,2] (x,y)
    -> [B,N,C] (bundlenet.py:290,320,343-344,385) -- HIP kernel ba_resample_kernel; when `data` or `warp`
    requires grad, the differentiable expression with the TF op's gradients (map and coordinates)."""
    return _resample(data, warp)


def _resampler_autograd(data, warp):
    """tf.contrib.resampler.resampler as a differentiable torch expression (gradients w.r.t. the map
    and, through the bilinear weights, w.r.t. the coordinates -- the TF op's gradient).  Only used by
    the training graph below; the forward-only paths use ops.resample (HIP)."""
    B, H, W, C = data.shape
    x, y = warp[..., 0], warp[..., 1]
    ok = (x > -1.0) & (y > -1.0) & (x < W) & (y < H)
    xs = torch.where(ok, x, torch.zeros_like(x))
    ys = torch.where(ok, y, torch.zeros_like(y))
    fx_, fy_ = torch.floor(xs), torch.floor(ys)
    cx, cy = fx_ + 1, fy_ + 1
    dx, dy = cx - xs, cy - ys
    flat = data.reshape(B, H * W, C)

    def tap(xi, yi):
        xi, yi = xi.long(), yi.long()
        inside = (xi >= 0) & (yi >= 0) & (xi <= W - 1) & (yi <= H - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).unsqueeze(-1).expand(-1, -1, C)
        v = torch.gather(flat, 1, idx)
        return torch.where(inside.unsqueeze(-1), v, torch.zeros_like(v))

    out = (dx * dy).unsqueeze(-1) * tap(fx_, fy_) + ((1 - dx) * (1 - dy)).unsqueeze(-1) * tap(cx, cy) \
        + (dx * (1 - dy)).unsqueeze(-1) * tap(fx_, cy) + ((1 - dx) * dy).unsqueeze(-1) * tap(cx, fy_)
    return torch.where(ok.unsqueeze(-1), out, torch.zeros_like(out))


def _grad_fixed_autograd(img):
    """bundlenet.py:92-100 as a differentiable torch expression: REFLECT pad by one pixel, central differences
    0.5 (f[x+1] - f[x-1]) -> [gx | gy] (exactly zero on the one-pixel rim).  img [B,H,W,C] -> [B,H,W,2C]."""
    x = torch.nn.functional.pad(img.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect")
    gx = 0.5 * (x[:, :, 1:-1, 2:] - x[:, :, 1:-1, :-2])
    gy = 0.5 * (x[:, :, 2:, 1:-1] - x[:, :, :-2, 1:-1])
    return torch.cat([gx, gy], dim=1).permute(0, 2, 3, 1)


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)


def _resample(data, warp):
    """resampler(): the HIP kernel when nothing upstream needs a gradient, the differentiable expression otherwise
    (the reference differentiates through tf.contrib.resampler w.r.t. both arguments)."""
    return _resampler_autograd(data, warp) if _wants_grad(data, warp) else ops.resample(data, warp, clamp=False)


def _target_map(img):
    """[f | gx | gy] (bundlenet.py:323-324): HIP kernel, or the differentiable expression when img needs a gradient."""
    return torch.cat([img, _grad_fixed_autograd(img)], dim=-1) if _wants_grad(img) else ops.target_map(img)


def _depth_output(init_depth, basis, W):
    """init_depth + basis . W (bundlenet.py:397): HIP kernel, or differentiable w.r.t. init_depth, basis and W."""
    if _wants_grad(init_depth, basis, W):
        nb, K = basis.shape[0], basis.shape[-1]
        return init_depth + torch.matmul(basis.reshape(nb, -1, K), W.reshape(nb, K, 1)).reshape(init_depth.shape)
    return ops.depth_output(init_depth, basis.reshape(basis.shape[0], -1, basis.shape[-1]), W)


def he_normal_lambda_weights(C, seed, device="cpu"):
    """lambda_<level>_<i> weights as bundlenet.py:105-106 creates them (he_normal, zero bias)."""
    g = torch.Generator().manual_seed(seed)
    dims = [C, 2 * C, 4 * C, 2 * C, C, 1]
    out = []
    for i in range(5):
        std = math.sqrt(2.0 / dims[i]) / 0.87962566103423978
        w = torch.clamp(torch.randn(dims[i], dims[i + 1], generator=g), -2, 2) * std
        out.append((w.to(device), torch.zeros(dims[i + 1], device=device)))
    return out


def lambda_weights_from_variables(variables):
    """Import the reference's lambda-MLP variables (bundlenet.py:102-110,168-172): a mapping name -> array holding
    `lambda_<level>_<i>_filters` [1, Cin, Cout] (or [Cin, Cout]) and `lambda_<level>_<i>_biases` [Cout] for i = 1..5, e.g.
    an .npz exported from a TF-1.x checkpoint.  Variable-scope prefixes (`lambda_<level>_<i>/...`, anything before the
    last '/') and a ':0' suffix are ignored.  Returns {level: [(filters [Cin,Cout], biases [Cout]) x 5]} -- the
    `lambda_weights` argument of BundleNet / legacy.Tracker."""
    found = {}
    for name, arr in variables.items():
        base = str(name).split("/")[-1].split(":")[0]
        parts = base.split("_")
        if len(parts) != 4 or parts[0] != "lambda" or parts[3] not in ("filters", "biases") or not parts[2].isdigit():
            continue
        t = torch.as_tensor(arr, dtype=torch.float32)
        if parts[3] == "filters":
            if t.dim() == 3 and t.shape[0] == 1:
                t = t[0]
            if t.dim() != 2:
                raise ValueError("%s: expected a [1, Cin, Cout] conv1d filter, got %s" % (name, tuple(t.shape)))
        elif t.dim() != 1:
            raise ValueError("%s: expected [Cout] biases, got %s" % (name, tuple(t.shape)))
        found.setdefault(parts[1], {}).setdefault(int(parts[2]), {})[parts[3]] = t.contiguous()
    out = {}
    for level, layers in found.items():
        seq = []
        for i in range(1, 6):
            lay = layers.get(i, {})
            if "filters" not in lay or "biases" not in lay:
                raise KeyError("lambda_%s_%d: filters / biases missing" % (level, i))
            if lay["filters"].shape[1] != lay["biases"].shape[0] or (seq and seq[-1][0].shape[1] != lay["filters"].shape[0]):
                raise ValueError("lambda_%s_%d: inconsistent layer shapes" % (level, i))
            seq.append((lay["filters"], lay["biases"]))
        out[level] = seq
    return out


def lambda_weights_to_variables(lambda_weights):
    """Inverse of lambda_weights_from_variables: {name: numpy array} with the reference's names and shapes."""
    out = {}
    for level, seq in lambda_weights.items():
        for i, (w, b) in enumerate(seq, 1):
            out["lambda_%s_%d_filters" % (level, i)] = w.detach().cpu().numpy()[None]
            out["lambda_%s_%d_biases" % (level, i)] = b.detach().cpu().numpy()
    return out


class BundleNet:
    """bundlenet.py:86-463.  `lambda_weights[level]` holds that level's five conv1d layers
    as (filters [Cin,Cout], biases [Cout]) pairs -- the reference's TF variables
    `lambda_<level>_<i>_filters/_biases`."""

    def __init__(self, is_training=True, reuse_variables=None, lambda_weights=None):
        self.is_training = is_training
        self.reuse_variables = reuse_variables
        self.lambda_weights = dict(lambda_weights or {})
        self._mlp_cache = ops.MlpCache()
        # training graph: True = exact gradients (the upstream dL/dAtA is symmetrised before the op's
        # backward); False = the reference's registered gradient verbatim (inexact: matrix_solve's
        # gradient w.r.t. AtA is not symmetric, utils.cu:648-657 assumes it is)
        self.exact_gradients = True
        # training graph: "fused" (round 5, the default) = the whole iteration as ONE autograd node on the fused kernels -- forward
        # = the inference path (gather + SYRK + solve), backward = the small step by implicit differentiation + the fused adjoint
        # of the assembly on the sparse layout (dense_train._SparseIteration); "lean" = C-wide statements as one HIP op with a HIP
        # adjoint, normal equations by block in torch (no samp / diff / grad / J tensors); both always exact gradients;
        # "reference" = the reference's statements one by one with the EquationConstruction op and its registered gradient
        self.training_graph = "fused"

    # -- small helpers kept for API parity --------------------------------------------
    def grad_fixed(self, input, name=None):
        """bundlenet.py:92-100"""
        if _wants_grad(input):
            return _grad_fixed_autograd(input)
        C = input.shape[-1]
        return ops.target_map(input)[..., C:]          # [gx | gy] of ba_target_map_kernel

    def conv1d(self, x, num_out_layers, name, activation=torch.nn.functional.elu):
        """bundlenet.py:102-110 (kernel width 1).  x [B,L,Cin]; weights looked up by name
        `lambda_<level>_<i>` in self.lambda_weights."""
        _, level, i = name.split("_")
        w, b = self.lambda_weights[level][int(i) - 1]
        assert w.shape[-1] == num_out_layers
        return activation(torch.matmul(x, w.to(x.device)) + b.to(x.device))

    def computeCoordinates(self, points2d, fx, fy, ox, oy):
        """bundlenet.py:112-120 -> p [B,3,N] (unit rays)"""
        x = ((points2d[:, :, 0] - ox) / fx).unsqueeze(1)
        y = ((points2d[:, :, 1] - oy) / fy).unsqueeze(1)
        p = torch.cat([x, y, torch.ones_like(x)], dim=1)
        return p / torch.sqrt(torch.clamp((p * p).sum(dim=1, keepdim=True), min=1e-12))

    def _mlp(self, level, device):
        return self._mlp_cache.get(self.lambda_weights, level, device)

    # -- the two iteration bodies -------------------------------------------------------
    def CameraIteration(self, conv1, conv2, fx, fy, ox, oy, p, D, R, T, l2_regularizer_base=None, level=None):
        """bundlenet.py:122-191: one pose-only GN/LM step -> (updatedR, updatedT)."""
        if torch.is_grad_enabled() and any(torch.is_tensor(x) and x.requires_grad
                                           for x in (conv1, conv2, D, R, T) + self._lambda_tensors(level)):
            R2, T2, _ = self._training_iteration(conv1, conv2, fx, fy, ox, oy, p, D, None, R, T, None, l2_regularizer_base,
                                                 level)
            return R2, T2
        B, H, W, C3 = conv2.shape
        C = conv1.shape[2]
        lv = ops.LevelProblem("bundle_camera", conv1, conv2, D, H, W, C, rays=p, fx=fx, fy=fy, ox=ox, oy=oy,
                              dense=False, tgt_has_grad=True)
        st = ops.LmState(R, T, None, P=6)
        AtA, Atb, absres, nvalid = ops.ba_assemble(lv, st.R, st.T)
        ops.ba_solve_update(lv, self._mlp(level, conv1.device), 1.0, AtA, Atb, absres, nvalid, st)
        self.last = dict(AtA=AtA, Atb=Atb, lam=st.lambda_out, delta=st.delta)
        return st.R, st.T

    def BundleIteration(self, conv1, conv2, fx, fy, ox, oy, p, D, B, R, T, W, l2_regularizer_base=None, level=None):
        """bundlenet.py:193-278: pose + depth-basis step -> (updatedR, updatedT, updatedW)."""
        if torch.is_grad_enabled() and any(torch.is_tensor(x) and x.requires_grad
                                           for x in (conv1, conv2, D, B, R, T, W) + self._lambda_tensors(level)):
            return self._training_iteration(conv1, conv2, fx, fy, ox, oy, p, D, B, R, T, W, l2_regularizer_base, level)
        nb, H, Wd, C3 = conv2.shape
        C = conv1.shape[2]
        K = B.shape[-1]
        lv = ops.LevelProblem("bundle", conv1, conv2, D, H, Wd, C, basis=B, rays=p, fx=fx, fy=fy, ox=ox, oy=oy,
                              dense=False, tgt_has_grad=True)
        st = ops.LmState(R, T, W.reshape(nb, K, 1), P=6 + K)
        AtA, Atb, absres, nvalid = ops.ba_assemble(lv, st.R, st.T, st.Wc)
        l2 = 1.0 if l2_regularizer_base is None else float(l2_regularizer_base)
        ops.ba_solve_update(lv, self._mlp(level, conv1.device), l2, AtA, Atb, absres, nvalid, st)
        self.last = dict(AtA=AtA, Atb=Atb, lam=st.lambda_out, delta=st.delta)
        return st.R, st.T, st.Wc

    # -- the training graph -------------------------------------------------------------------
    def _lambda_tensors(self, level):
        lw = self.lambda_weights.get(str(level))
        return tuple(t for pair in lw for t in pair if torch.is_tensor(t)) if lw else ()

    def _training_iteration(self, *args):
        if self.training_graph == "fused" and self.exact_gradients:
            conv1, conv2, fx, fy, ox, oy, p, D, B, R, T, W, l2, level = args
            from . import dense_train
            data_grad = any(torch.is_tensor(x) and x.requires_grad for x in (fx, fy, ox, oy, p))     # (data in the reference: no gradient path)
            # a regulariser base that is itself a differentiable tensor needs the torch graph (the fused node takes it as a number)
            l2_grad = torch.is_tensor(l2) and l2.requires_grad
            if not data_grad and not l2_grad and dense_train.sparse_iteration_supported(conv1, conv2, B, D, R, T):
                lw = self.lambda_weights[str(level)]
                layers = [(w if torch.is_tensor(w) else torch.as_tensor(w), b if torch.is_tensor(b) else torch.as_tensor(b)) for w, b in lw]
                layers = [(w.to(conv1.device), b.to(conv1.device)) for w, b in layers]
                bundle = B is not None
                l2v = (1.0 if l2 is None else float(l2)) if bundle else 1.0            # CameraIteration ignores the argument (:122-191)
                R2, T2, W2, last = dense_train.sparse_iteration("bundle" if bundle else "bundle_camera", self._mlp(level, conv1.device), l2v,
                                                                conv1, conv2, D, B, R, T, W, fx, fy, ox, oy, p, layers)
                self.last = last          # (AtA, Atb, lam, delta: as the inference path leaves them)
                return R2, T2, W2
            return self._iteration_autograd_lean(*args)
        if self.training_graph in ("lean", "fused") and self.exact_gradients:
            return self._iteration_autograd_lean(*args)
        return self._iteration_autograd(*args)

    def _iteration_autograd(self, conv1, conv2, fx, fy, ox, oy, p, D, B, R, T, W, l2_regularizer_base, level):
        """The reference's differentiable graph, statement for statement (bundlenet.py:193-278, and
        :122-191 when B is None): standard differentiable tensor ops + the custom op
        `equation_construction`, whose backward is the HIP EquationConstructionGrad kernel
        (bundlenet.py:79-82) -- exactly the reference's split between TF autodiff and utils.cu.
        Taken automatically when an input or a lambda weight requires grad; the forward-only calls
        use the fused kernels.  Same values as the fused path (tested at 1e-4)."""
        nb, npix, C = conv1.shape
        H, Wd = conv2.shape[1], conv2.shape[2]
        bundle = B is not None
        if bundle:
            D = D + torch.matmul(B, W)                                                   # :207
        Rp = torch.matmul(R, p)
        rx, ry, rz = Rp[:, 0], Rp[:, 1], Rp[:, 2]
        RPT = Rp * D.transpose(1, 2) + T                                                # :210-213
        X, Y, Z = RPT[:, 0], RPT[:, 1], RPT[:, 2]
        x, y = X / Z, Y / Z
        px, py = fx * x + ox, fy * y + oy
        samp = _resampler_autograd(conv2, torch.stack([px, py], dim=-1))                # :230
        mask = (~((px < 0) | (px > float(Wd - 1)) | (py < 0) | (py > float(H - 1)))).to(conv1.dtype)[..., None, None]
        diff = (conv1 - samp[:, :, 0:C]).unsqueeze(-1) * mask                           # :234,238
        grad = torch.stack([samp[:, :, C:2 * C], samp[:, :, 2 * C:3 * C]], dim=-1) * mask
        avg = diff.squeeze(-1).abs().mean(dim=1, keepdim=True)                          # :243
        h = avg
        lw = self.lambda_weights[str(level)]
        for i, (w, b) in enumerate(lw):
            z = torch.matmul(h, w.to(h.device)) + b.to(h.device)
            h = torch.tanh(z) if i == 4 else torch.nn.functional.selu(z)
        lam = torch.linalg.vector_norm(avg, dim=-1, keepdim=True) ** (2.0 + h)          # :249
        if bundle and l2_regularizer_base is not None:                                   # :252-253; CameraIteration (:122-191)
            lam = l2_regularizer_base * lam                                             # ignores the argument
        J = CameraJacobianMatrix(x, y, Z, fx, fy)                                       # :259
        if bundle:
            jd = DepthJacobianMatrix(rx.unsqueeze(1), ry.unsqueeze(1), rz.unsqueeze(1), x, y, Z, fx, fy)
            J = torch.cat([J, jd.unsqueeze(-1) * B.unsqueeze(-2)], dim=-1)              # :260-261
        AtA, Atb = ops.equation_construction(J, grad, diff, symmetric_grad=self.exact_gradients)   # :263 (HIP fwd + bwd)
        diag = torch.diagonal(AtA, dim1=1, dim2=2)
        if bundle:
            damp = torch.cat([diag[:, :-1] + 1e-5, torch.zeros(nb, 1, device=diag.device)], dim=-1)   # :266
        else:
            damp = diag + 1e-5                                                          # :182
        AtA = AtA + torch.diag_embed(damp * lam.squeeze(-1))
        sol = torch.linalg.solve(AtA, Atb)                                              # :267
        wx, wy, wz = sol[:, 0], sol[:, 1], sol[:, 2]
        dr = AngleaAxisRotation(wx, wy, wz)
        dv = VMatrix(wx.reshape(-1), wy.reshape(-1), wz.reshape(-1))
        updatedR = torch.matmul(dr, R)
        updatedT = torch.matmul(dv, sol[:, 3:6]) + torch.matmul(dr, T)
        updatedW = W + sol[:, 6:] if bundle else None
        self.last = dict(AtA=AtA, Atb=Atb, lam=lam.reshape(-1), delta=sol[..., 0])
        return updatedR, updatedT, updatedW

    def _iteration_autograd_lean(self, conv1, conv2, fx, fy, ox, oy, p, D, B, R, T, W, l2_regularizer_base, level):
        """The same iteration as _iteration_autograd (same values, same gradients) without its large intermediates:
        the C-wide statements (:230-243) are one HIP op with a HIP adjoint (ops.sample_stats: per-pixel M = G^T G,
        g = G^T d, mask and sum |d|), and J^T (G^T G) J / J^T G^T d (:263, utils.cu:331-414) are formed from M, g and the
        per-pixel Jacobians by block -- pose block by small per-pixel algebra, depth blocks as basis GEMMs -- instead of
        materialising samp [B,N,3C], diff, grad and J [B,N,2,P].  Everything else is the reference's statements."""
        nb, npix, C = conv1.shape
        bundle = B is not None
        if bundle:
            D = D + torch.matmul(B, W)                                                   # :207
        Rp = torch.matmul(R, p)
        rx, ry, rz = Rp[:, 0], Rp[:, 1], Rp[:, 2]
        RPT = Rp * D.transpose(1, 2) + T                                                # :210-213
        X, Y, Z = RPT[:, 0], RPT[:, 1], RPT[:, 2]
        x, y = X / Z, Y / Z
        px, py = fx * x + ox, fy * y + oy
        stats, _mask, absd = ops.sample_stats(conv1, conv2, px, py)                      # :230-239 (HIP fwd + bwd)
        avg = (absd / float(npix)).unsqueeze(1)                                         # :243
        h = avg
        lw = self.lambda_weights[str(level)]
        for i, (w, b) in enumerate(lw):
            z = torch.matmul(h, w.to(h.device)) + b.to(h.device)
            h = torch.tanh(z) if i == 4 else torch.nn.functional.selu(z)
        lam = torch.linalg.vector_norm(avg, dim=-1, keepdim=True) ** (2.0 + h)          # :249
        if bundle and l2_regularizer_base is not None:                                   # :252-253; CameraIteration (:122-191)
            lam = l2_regularizer_base * lam                                             # ignores the argument
        Jc = CameraJacobianMatrix(x, y, Z, fx, fy)                                      # [B,N,2,6]  :259
        M11, M12, M22, g1, g2 = (stats[..., i] for i in range(5))
        MJ0 = M11.unsqueeze(-1) * Jc[:, :, 0] + M12.unsqueeze(-1) * Jc[:, :, 1]         # rows of M Jc  [B,N,6]
        MJ1 = M12.unsqueeze(-1) * Jc[:, :, 0] + M22.unsqueeze(-1) * Jc[:, :, 1]
        Hcc = torch.matmul(Jc[:, :, 0].transpose(1, 2), MJ0) + torch.matmul(Jc[:, :, 1].transpose(1, 2), MJ1)   # [B,6,6]
        bc = (Jc[:, :, 0] * g1.unsqueeze(-1) + Jc[:, :, 1] * g2.unsqueeze(-1)).sum(dim=1)                         # [B,6]
        if bundle:
            jd = DepthJacobianMatrix(rx.unsqueeze(1), ry.unsqueeze(1), rz.unsqueeze(1), x, y, Z, fx, fy)   # [B,N,2]
            u = MJ0 * jd[..., 0:1] + MJ1 * jd[..., 1:2]                                 # Jc^T M jd  [B,N,6]
            s = M11 * jd[..., 0] ** 2 + 2.0 * M12 * jd[..., 0] * jd[..., 1] + M22 * jd[..., 1] ** 2      # jd^T M jd
            r = jd[..., 0] * g1 + jd[..., 1] * g2                                       # jd^T g
            Hcd = torch.matmul(u.transpose(1, 2), B)                                    # [B,6,K]
            Hdd = torch.matmul(B.transpose(1, 2), B * s.unsqueeze(-1))                  # [B,K,K]
            bd = torch.matmul(B.transpose(1, 2), r.unsqueeze(-1)).squeeze(-1)           # [B,K]
            AtA = torch.cat([torch.cat([Hcc, Hcd], dim=2), torch.cat([Hcd.transpose(1, 2), Hdd], dim=2)], dim=1)
            Atb = torch.cat([bc, bd], dim=1).unsqueeze(-1)
        else:
            AtA, Atb = Hcc, bc.unsqueeze(-1)
        diag = torch.diagonal(AtA, dim1=1, dim2=2)
        if bundle:
            damp = torch.cat([diag[:, :-1] + 1e-5, torch.zeros(nb, 1, device=diag.device)], dim=-1)   # :266
        else:
            damp = diag + 1e-5                                                          # :182
        AtA = AtA + torch.diag_embed(damp * lam.squeeze(-1))
        sol = torch.linalg.solve(AtA, Atb)                                              # :267
        wx, wy, wz = sol[:, 0], sol[:, 1], sol[:, 2]
        dr = AngleaAxisRotation(wx, wy, wz)
        dv = VMatrix(wx.reshape(-1), wy.reshape(-1), wz.reshape(-1))
        updatedR = torch.matmul(dr, R)
        updatedT = torch.matmul(dv, sol[:, 3:6]) + torch.matmul(dr, T)
        updatedW = W + sol[:, 6:] if bundle else None
        self.last = dict(AtA=AtA, Atb=Atb, lam=lam.reshape(-1), delta=sol[..., 0])
        return updatedR, updatedT, updatedW

    # -- level drivers ------------------------------------------------------------------
    @staticmethod
    def _crop(points):
        x = 320 * (points[..., 0:1] - 4) / 312
        y = 256 * (points[..., 1:2] - 4) / 232
        return torch.cat([x, y], dim=-1)

    def _crop_intrinsics(self, intrisic, npixels):
        self.fx = 40.0 * intrisic[:, 0].repeat(1, npixels) / 39.0
        self.fy = 32.0 * intrisic[:, 1].repeat(1, npixels) / 29.0
        self.ox = (40.0 * intrisic[:, 2].repeat(1, npixels) / 39.0) - (160.0 / 39.0)
        self.oy = (32.0 * intrisic[:, 3].repeat(1, npixels) / 29.0) - (128.0 / 29.0)

    @staticmethod
    def _swap_halves(x):
        n = x.shape[0]
        return torch.cat([x[n // 2:n], x[0:n // 2]], dim=0)

    def CameraResize(self, intrisic, layers, points, _depths, reuse_variables=False):
        """bundlenet.py:280-329: 4 levels (scale 8,4,2,1) x 1 CameraIteration."""
        self.reuse_variables = reuse_variables
        _points = self._crop(points)
        d = resampler(_depths.detach(), _points / 2)
        nbatch, npixels = layers[-1].shape[0], points.shape[1]
        self._crop_intrinsics(intrisic, npixels)
        p = self.computeCoordinates(_points, self.fx, self.fy, self.ox, self.oy)
        R = torch.eye(3, device=points.device).repeat(nbatch, 1, 1)
        T = torch.zeros(nbatch, 3, 1, device=points.device)
        rotations, translations = [], []
        for level in range(0, 4):
            scale = 2 ** (3 - level)
            layer1 = resampler(layers[level], _points / scale)
            layer2 = _target_map(self._swap_halves(layers[level]))         # [f | gx | gy], differentiable when needed
            R, T = self.CameraIteration(layer1, layer2, self.fx / scale, self.fy / scale, self.ox / scale,
                                        self.oy / scale, p, d, R, T, 1.0, str(level))
            rotations.append(R)
            translations.append(T)
        return rotations, translations

    def BundleResize(self, intrisic, layers, points, basis, init_depth, init_rotation=None, init_translation=None,
                     reuse_variables=False):
        """bundlenet.py:332-399: levels 2,3 (scale 2,1) x 1 BundleIteration."""
        self.reuse_variables = reuse_variables
        _points = self._crop(points)
        depths = init_depth.detach()
        d = resampler(depths, _points / 2)
        b = resampler(basis, _points / 2)
        nbatch, npixels, nbasis = layers[-1].shape[0], points.shape[1], basis.shape[-1]
        self._crop_intrinsics(intrisic, npixels)
        p = self.computeCoordinates(_points, self.fx, self.fy, self.ox, self.oy)
        dev = points.device
        R = torch.eye(3, device=dev).repeat(nbatch, 1, 1) if init_rotation is None else init_rotation
        T = torch.zeros(nbatch, 3, 1, device=dev) if init_translation is None else init_translation
        W = torch.zeros(nbatch, nbasis, 1, device=dev)
        out_R, out_T, out_D = [], [], []
        Hh, Wh = init_depth.shape[1], init_depth.shape[2]
        for level in range(2, 4):
            scale = 2 ** (3 - level)
            layer1 = resampler(layers[level], _points / scale)
            layer2 = _target_map(self._swap_halves(layers[level]))         # [f | gx | gy], differentiable when needed
            R, T, W = self.BundleIteration(layer1, layer2, self.fx / scale, self.fy / scale, self.ox / scale,
                                           self.oy / scale, p, d, b, R, T, W, 1000.0, str(level))
            out_R.append(R)
            out_T.append(T)
            out_D.append(_depth_output(init_depth, basis, W))              # :397, differentiable w.r.t. init_depth / basis / W
        return out_R, out_T, out_D

    # -- losses (bundlenet.py:401-463) ---------------------------------------------------
    def lossR(self, predQ, gtQ):
        return torch.mean(1.0 - (predQ * gtQ).sum(dim=1))            # tf.losses.cosine_distance

    def lossT(self, predT, gtT):
        return torch.mean(torch.abs(predT - gtT))                     # the second definition wins (:411)

    def lossF(self, intrisic, depth, mask, predR, predT, gtR, gtT):
        nbatch, height, width = depth.shape[0], depth.shape[1], depth.shape[2]
        npixels = height * width
        dev = depth.device
        mask = mask.reshape(nbatch, npixels)
        # locals, as the reference (:441-445): lossF must not disturb the intrinsics a Resize call left on self
        fx = 40.0 * intrisic[:, 0].repeat(1, npixels) / 39.0
        fy = 32.0 * intrisic[:, 1].repeat(1, npixels) / 29.0
        ox = (40.0 * intrisic[:, 2].repeat(1, npixels) / 39.0) - (160.0 / 39.0)
        oy = (32.0 * intrisic[:, 3].repeat(1, npixels) / 29.0) - (128.0 / 29.0)
        ys, xs = torch.meshgrid(torch.arange(height, device=dev), torch.arange(width, device=dev), indexing="ij")
        pts = torch.stack([xs.reshape(-1).float(), ys.reshape(-1).float()], dim=-1)[None].repeat(nbatch, 1, 1)
        p = self.computeCoordinates(pts, fx, fy, ox, oy)

        def flow(R, T):
            X = torch.matmul(R, p) * depth.reshape(nbatch, 1, npixels) + T.reshape(nbatch, 3, 1)
            return fx * X[:, 0] / X[:, 2] + ox, fy * X[:, 1] / X[:, 2] + oy

        fxp, fyp = flow(predR, predT)
        fxg, fyg = flow(gtR, gtT)
        valid, total = mask.sum(), float(npixels * nbatch)
        return (total / valid) * (torch.mean(torch.abs(fxp - fxg) * mask) / width +
                                  torch.mean(torch.abs(fyp - fyg) * mask) / width)
