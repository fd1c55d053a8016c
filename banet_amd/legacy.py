"""Host-side mirror of /root/reference/legacy/ba.py (`Tracker`) for PyTorch-ROCm callers.

`trackTF` keeps the reference's signature and semantics (3 levels, scale 4/2/1, pose carried
across levels, early-terminated LM with accept/reject) but the whole per-level loop is
enqueued on the device by `banet_lm_level_f32` -- no per-iteration host round trip, and the
CheckUpdate pass of iteration k doubles as the assembly pass of iteration k+1.
The feature CNN of the reference (`feat.py`) is out of scope: `layers` are given.
"""
import torch

from . import ops

# module-level switches, as legacy/ba.py:5-9.  Like the reference's drivers (legacy/example.py:8, legacy/eval.py:9) callers
# overwrite them (`legacy.early_termination = False`); every Tracker call reads them at call time and hands them to
# banet_lm_level_ex_f32 as a banet_lm_params_t.
early_termination = True
angle_change = 0.002 * (3.14 / 180.0)
translation_change = 0.0002
residual_ratio = 1.0
qr = True


def _params():
    return ops.lm_params(angle_change, translation_change, residual_ratio, qr)


def interpolate2d2(imgs, p):
    """legacy/utils_python.py:177-232: clamped bilinear, no mask.  imgs [B,H,W,C], p [B,N,2]
    -> [B,N,C] -- HIP kernel ba_resample_kernel (clamp mode)."""
    return ops.resample(imgs, p, clamp=True)


class Tracker:
    """legacy/ba.py:15-482 without the TF session / CNN: the BA part only."""

    def __init__(self, lambda_weights=None, iters=(3, 5, 7)):
        self.lambda_weights = dict(lambda_weights or {})
        self.iters = list(iters)
        self._mlp_cache = ops.MlpCache()

    def _mlp(self, level, device):
        return self._mlp_cache.get(self.lambda_weights, level, device)

    def grad_fixed(self, input, name=None):
        """legacy/ba.py:17-25"""
        C = input.shape[-1]
        return ops.target_map(input)[..., C:]          # [gx | gy] of ba_target_map_kernel

    def computeCoordinates(self, points2d, fx, fy, ox, oy):
        """legacy/ba.py:27-34 (rays NOT normalised)"""
        x = ((points2d[:, :, 0] - ox) / fx).unsqueeze(1)
        y = ((points2d[:, :, 1] - oy) / fy).unsqueeze(1)
        return torch.cat([x, y, torch.ones_like(x)], dim=1)

    def CameraJacobianMatrix(self, x, y, Z, fx, fy, name=None):
        """legacy/ba.py:36-48 -> [B,N,2,6]: the same entries as bundlenet.py:49-61 WITHOUT its leading minus."""
        from . import bundlenet
        return -bundlenet.CameraJacobianMatrix(x, y, Z, fx, fy)

    def VMatrix(self, wx, wy, wz, name=None):
        """legacy/ba.py:51-58 (applied per item; the reference's stack-on-axis-0 form is only correct for B = 1)."""
        from . import bundlenet
        return bundlenet.VMatrix(wx, wy, wz)

    def AngleaAxisRotation(self, wx, wy, wz, name=None):
        """legacy/ba.py:60-80: Rodrigues' formula, dividing by theta unguarded (NaN for a zero rotation, as in the
        reference; the fused kernels return the identity there)."""
        ones = torch.ones_like(wx)
        theta = torch.sqrt(wx * wx + wy * wy + wz * wz)
        wx, wy, wz = wx / theta, wy / theta, wz / theta
        c, s = torch.cos(theta), torch.sin(theta)
        m = torch.stack([c + wx * wx * (ones - c), wz * s + wx * wy * (ones - c), -wy * s + wx * wz * (ones - c),
                         wx * wy * (ones - c) - wz * s, c + wy * wy * (ones - c), wx * s + wy * wz * (ones - c),
                         wy * s + wx * wz * (ones - c), -wx * s + wy * wz * (ones - c), c + wz * wz * (ones - c)], dim=-1)
        return m.reshape(-1, 3, 3).transpose(1, 2)

    def conv1d(self, x, num_out_layers, name, activation=torch.nn.functional.elu):
        """legacy/ba.py conv1d (k = 1): `name` = lambda_<level>_<i>, weights from self.lambda_weights[level]."""
        _, level, i = name.split("_")
        w, b = self.lambda_weights[level][int(i) - 1]
        return activation(torch.matmul(x, w.to(x.device)) + b.to(x.device))

    def _level(self, variant, conv1, conv2, fx, fy, ox, oy, p, D):
        B, H, W, C3 = conv2.shape
        return ops.LevelProblem(variant, conv1, conv2, D, H, W, conv1.shape[2], rays=p, fx=fx, fy=fy, ox=ox, oy=oy,
                                dense=False, tgt_has_grad=True)

    def CameraIteration(self, conv1, conv2, fx, fy, ox, oy, p, D, R, T):
        """legacy/ba.py:148-214 -> (updatedR, updatedT, ratio)"""
        lv = self._level("legacy_fixed", conv1, conv2, fx, fy, ox, oy, p, D)
        st = ops.LmState(R, T, None, 6)
        ops.lm_level(lv, None, 1.0, 1, False, st, params=_params())
        return st.R, st.T, st.ratio

    def CameraIteration2(self, conv1, conv2, fx, fy, ox, oy, p, D, R, T, level):
        """legacy/ba.py:226-345 -> (R, T, update_w, update_t, ratio): one LM step including
        its accept/reject test (two evaluation rounds on the device)."""
        lv = self._level("legacy_lm", conv1, conv2, fx, fy, ox, oy, p, D)
        st = ops.LmState(R, T, None, 6)
        ops.lm_level(lv, self._mlp(level, conv1.device), 1.0, 1, True, st, params=_params())
        accepted = ((st.R - ops.capi.f32c(R).reshape(-1, 3, 3)).abs().amax(dim=(1, 2)) > 0) | \
                   ((st.T - ops.capi.f32c(T).reshape(-1, 3, 1)).abs().amax(dim=(1, 2)) > 0)
        uw = torch.where(accepted, st.delta[:, 0:3].norm(dim=1), torch.zeros_like(st.ratio))
        ut = torch.where(accepted, st.delta[:, 3:6].norm(dim=1), torch.zeros_like(st.ratio))
        return st.R, st.T, uw, ut, st.ratio

    def trackTF(self, intrisic, layers, points, d, initR, initT, level_iters):
        """legacy/ba.py:85-145.  layers: 3 maps coarse->fine, each [2B,H_l,W_l,C] with the
        source frames first and the target frames second (the reference has B=1).
        Returns (R, T, ratio) and leaves the per-level iteration counts in self.level_iters_run."""
        npixels = points.shape[1]
        nb = layers[-1].shape[0] // 2
        fx0 = intrisic[:, 0].repeat(1, npixels)
        fy0 = intrisic[:, 1].repeat(1, npixels)
        ox0 = intrisic[:, 2].repeat(1, npixels)
        oy0 = intrisic[:, 3].repeat(1, npixels)
        p = self.computeCoordinates(points, fx0, fy0, ox0, oy0)
        st = ops.LmState(initR, initT, None, 6)
        self.level_iters_run = []
        for level in range(1, 4):
            scale = 2 ** (3 - level)
            layer1 = interpolate2d2(layers[level - 1][0:nb], points / scale)
            layer2 = ops.target_map(layers[level - 1][nb:2 * nb])          # [f | gx | gy], legacy/ba.py:116-118
            variant = "legacy_lm" if early_termination else "legacy_fixed"
            lv = self._level(variant, layer1, layer2, fx0 / scale, fy0 / scale, ox0 / scale, oy0 / scale, p, d)
            mlp = self._mlp(level, points.device) if early_termination else None
            ops.lm_level(lv, mlp, 1.0, level_iters[level - 1], early_termination, st, params=_params())
            self.level_iters_run.append(st.iters.clone())
        return st.R, st.T, st.ratio
