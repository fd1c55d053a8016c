"""Keyframe sequence driver: the caller of the legacy tracker, legacy/seq_example.py:72-83 (point
selection) and :150-208 (keyframe loop), without file IO, OpenCV or the CNN -- frames arrive as
feature pyramids (what `legacy/feat.py` would produce) plus, for key frames, the RGB image and the
depth map used to pick the BA points.  Host logic in Python like the reference's; every tracking
call goes to `Tracker.trackTF`, i.e. to libbanet_hip.so.
"""
import numpy as np
import torch

from . import legacy


def sobel_x(image):
    """cv2.Sobel(image, CV_32F, 1, 0, ksize=3) with OpenCV's default BORDER_REFLECT_101.  image [H,W,3]."""
    p = np.pad(image.astype(np.float32), [(1, 1), (1, 1), (0, 0)], mode="reflect")
    d = p[:, 2:] - p[:, :-2]
    return d[:-2] + 2.0 * d[1:-1] + d[2:]


def valid_point_and_depth(image, depth, num_points, thres, rng):
    """legacy/seq_example.py:72-83: points with a strong image gradient and a valid depth, sampled with
    replacement.  The reference computes the x-derivative twice (its `dy` is `Sobel(...,1,0)` too,
    :73-74); reproduced.  -> points [1,num,2] (x,y) float32, depths [1,num,1]."""
    H, W = depth.shape
    dx = sobel_x(image)
    dy = dx
    dxy = np.sqrt(np.sum(np.square(dx), axis=-1) + np.sum(np.square(dy), axis=-1)).flatten()
    d = depth.flatten()
    x, y = np.meshgrid(np.linspace(0, W - 1, W), np.linspace(0, H - 1, H))
    xy = np.stack((x.flatten(), y.flatten()), axis=-1).astype(np.float32)
    index = np.logical_and(np.greater(dxy, thres), np.greater(d, 1e-5))
    d, p = d[index], xy[index, :]
    pick = rng.randint(0, p.shape[0], num_points)
    return np.reshape(p[pick, :], (1, num_points, 2)), np.reshape(d[pick], (1, num_points, 1)).astype(np.float32)


def valid_point_and_depth2(image1, image2, depth1, depth2, rotation, translation, intrinsics, num_points, rng,
                           grad_thres=80.0, color_thres=64.0, depth_tol=0.2):
    """legacy/eval.py:102-147: BA points for a frame pair with known relative pose -- strong gradient and valid depth in
    frame 1, and consistent with frame 2 under the ground-truth motion: the projection falls inside the image, the
    colours differ by less than `color_thres`, the depths agree within `depth_tol` (relative).  Vectorised; the reference's
    double loop visits the pixels in the same row-major order, so the candidate list is identical.  As there, the
    gradient test uses the x-derivative twice (:109-110) and positions are truncated with int().  Sampled with
    replacement -> points [1,num,2] (x,y) float32, depths [1,num,1]."""
    H, W = depth1.shape
    fx, fy, ox, oy = [float(v) for v in np.asarray(intrinsics).flatten()[:4]]
    dx = sobel_x(image1)
    dxy = np.sqrt(2.0 * np.sum(np.square(dx), axis=-1))
    jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    d1 = depth1.astype(np.float64)
    ray = np.stack([(jj - ox) / fx, (ii - oy) / fy, np.ones_like(jj)], axis=-1) * d1[..., None]
    X = ray @ np.asarray(rotation, dtype=np.float64).reshape(3, 3).T + np.asarray(translation, dtype=np.float64).reshape(3)
    with np.errstate(divide="ignore", invalid="ignore"):
        px = X[..., 0] / X[..., 2] * fx + ox
        py = X[..., 1] / X[..., 2] * fy + oy
    ok = (depth1 >= 1e-5) & (dxy >= grad_thres) & np.isfinite(px) & np.isfinite(py)
    pxi = np.where(ok, px, 0.0).astype(np.int64)                      # int(): truncation toward zero, like the reference
    pyi = np.where(ok, py, 0.0).astype(np.int64)
    ok &= (pyi >= 0) & (pyi < H) & (pxi >= 0) & (pxi < W)
    pxi, pyi = np.clip(pxi, 0, W - 1), np.clip(pyi, 0, H - 1)
    c2 = image2.astype(np.float32)[pyi, pxi]
    ok &= np.linalg.norm(image1.astype(np.float32) - c2, axis=-1) <= color_thres
    with np.errstate(divide="ignore", invalid="ignore"):
        ok &= np.abs(X[..., 2] - depth2[pyi, pxi]) / X[..., 2] <= depth_tol
    idx = np.flatnonzero(ok.flatten())                                  # row-major = the reference's loop order
    pts = np.stack([idx % W, idx // W], axis=-1).astype(np.float32)
    dep = depth1.flatten()[idx].astype(np.float32)
    pick = rng.randint(0, idx.size, num_points)
    return pts[pick].reshape(1, num_points, 2), dep[pick].reshape(1, num_points, 1)


def pose_errors(rotation, translation, rotation_gt, translation_gt):
    """legacy/eval.py:225-234 (and example.py:112-121): rotation error in degrees from the quaternion dot product (the
    reference's `2*180*arccos(.)/3.14`, with its 3.14), the predicted rotation angle, the translation error norm and the
    predicted translation norm."""
    def quat_wxyz(R):   # eval.py:25-34 rotation2quaternion3D
        R = np.asarray(R, dtype=np.float64).reshape(3, 3)
        q0 = np.sqrt(1.0 + R[0, 0] + R[1, 1] + R[2, 2]) / 2.0
        q = np.array([q0, (R[2, 1] - R[1, 2]) / (4.0 * q0), (R[0, 2] - R[2, 0]) / (4.0 * q0), (R[1, 0] - R[0, 1]) / (4.0 * q0)])
        return q / np.linalg.norm(q)
    qp = quat_wxyz(rotation)
    x, y, z, w = rotation_to_quaternion_xyzw(rotation_gt)
    qg = np.array([w, x, y, z])                                          # w >= 0 as eval.py:222-223 enforces
    tp, tg = np.asarray(translation, dtype=np.float64).flatten(), np.asarray(translation_gt, dtype=np.float64).flatten()
    return dict(rotation_error_deg=2 * 180 * np.arccos(np.clip(np.dot(qg, qp), -1.0, 1.0)) / 3.14,
                rotation_deg=2 * 180 * np.arccos(np.clip(qp[0], -1.0, 1.0)) / 3.14,
                translation_error=float(np.linalg.norm(tg - tp)), translation_norm=float(np.linalg.norm(tp)))


def rotation_to_quaternion_xyzw(R):
    """Unit quaternion (x, y, z, w), w >= 0, of a 3x3 rotation matrix (float64, Shepperd's branch on the largest of
    trace / diagonal entries) -- what seq_example.py:176 obtains from quaternion.from_rotation_matrix for its TUM line
    (q and -q denote the same rotation; the sign is fixed here by w >= 0)."""
    m = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2.0
        q = np.array([(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2.0
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (m[j, i] + m[i, j]) / s
        q[k] = (m[k, i] + m[i, k]) / s
        q[3] = (m[k, j] - m[j, k]) / s
    q /= np.linalg.norm(q)
    return q if q[3] >= 0 else -q


def tum_line(stamp, camera, quaternion_xyzw):
    """'timestamp tx ty tz qx qy qz qw' -- the trajectory line seq_example.py:177 prints (TUM RGB-D format)."""
    return " ".join(repr(float(v)) for v in (stamp,) + tuple(camera) + tuple(quaternion_xyzw))


class KeyframeTracker:
    """State machine of legacy/seq_example.py:150-208.  `track` returns the frame's global pose
    (rotation, translation as the reference chains them), camera centre and whether the frame became
    the new key frame (keep_ratio < 0.8 or more than 0.1 s since the key frame, :191)."""

    def __init__(self, tracker, intrinsics, iters=(5, 8, 8), num_points=4096, thres=120.0, min_keep_ratio=0.8,
                 max_gap=0.1, rng=None, device="cuda:0"):
        self.tracker, self.iters = tracker, list(iters)
        self.intrinsics = torch.as_tensor(intrinsics, dtype=torch.float32, device=device).reshape(1, 4, 1)
        self.num_points, self.thres = num_points, thres
        self.min_keep_ratio, self.max_gap = min_keep_ratio, max_gap
        self.rng = rng if rng is not None else np.random.RandomState(0)
        self.device = device
        self.globalRotations, self.globalTranslations = [], []
        self.keyframeIndex, self.frameIndex = 0, 0

    def _reset_init(self):
        self.initR = torch.eye(3, device=self.device).reshape(1, 3, 3)
        self.initT = torch.zeros(1, 3, 1, device=self.device)

    def _select(self, image, depth):
        pts, d = valid_point_and_depth(image, depth, self.num_points, self.thres, self.rng)
        self.points = torch.from_numpy(pts).to(self.device)
        self.depths = torch.from_numpy(d).to(self.device)

    def start(self, layers, image, depth, stamp):
        """first frame = first key frame (seq_example.py:137-148)"""
        self.key_layers, self.key_stamp = layers, float(stamp)
        self._select(image, depth)
        self._reset_init()
        self.globalRotations = [torch.eye(3, device=self.device).reshape(1, 3, 3)]
        self.globalTranslations = [torch.zeros(1, 3, 1, device=self.device)]
        self.keyframeIndex, self.frameIndex = 0, 0

    def track(self, layers, image, depth, stamp):
        self.frameIndex += 1
        both = [torch.cat([k, f], dim=0) for k, f in zip(self.key_layers, layers)]
        rotation, translation, keep_ratio = self.tracker.trackTF(self.intrinsics, both, self.points, self.depths,
                                                                 self.initR, self.initT, self.iters)
        rotation, translation = rotation.clone(), translation.clone()
        keep_ratio = float(keep_ratio.reshape(-1)[0])
        gR = torch.matmul(rotation, self.globalRotations[self.keyframeIndex])                       # :168
        gT = torch.matmul(rotation, translation) + self.globalTranslations[self.keyframeIndex]      # :169 (as written)
        self.globalRotations.append(gR)
        self.globalTranslations.append(gT)
        camera = -torch.matmul(gR.transpose(1, 2).double(), gT.double()).flatten()                   # :174-175
        quat = rotation_to_quaternion_xyzw(gR.transpose(1, 2).double()[0].cpu().numpy())             # :176
        switched = keep_ratio < self.min_keep_ratio or (float(stamp) - self.key_stamp) > self.max_gap   # :191
        if switched:
            self.keyframeIndex = self.frameIndex
            self.key_layers, self.key_stamp = layers, float(stamp)
            self._select(image, depth)
            self._reset_init()
        else:
            self.initR, self.initT = rotation, translation
        return dict(rotation=rotation, translation=translation, keep_ratio=keep_ratio, globalRotation=gR,
                    globalTranslation=gT, camera=camera, quaternion=quat,
                    tum=tum_line(stamp, camera.cpu().numpy(), quat), new_keyframe=switched, iters=[int(c[0]) for c in
                                                                                         self.tracker.level_iters_run])
