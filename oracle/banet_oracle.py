"""CPU oracle for the BANet bundle-adjustment hot path  --  TEST INFRASTRUCTURE ONLY.

This module restates, in plain numpy, the arithmetic of the reference's BA layer
(`/root/reference/bundlenet.py`, `/root/reference/legacy/ba.py`,
`/root/reference/legacy/utils_python.py`, `/root/reference/utils.cu`).  It exists to
CHECK the HIP path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it; the product package `banet_amd` never does.

Pinning status ("parity pinned to the reference's Python, TF kernels restated"):
  * The reference has no tests / golden vectors and TensorFlow-1.x is not installable
    here, so the third-party TF kernels it calls (tf.contrib.resampler, tf.qr,
    tf.matrix_solve, conv1d, selu, ...) are restated from their published semantics in
    `oracle/tf1_shim` (numpy, eager).
  * The reference's OWN Python (`legacy/ba.py`, `legacy/utils_python.py`, `bundlenet.py`)
    is executed verbatim over that shim by `tests/golden/make_golden.py`, and this
    oracle is checked against those outputs (`tests/test_oracle_golden.py`).
  * `utils.cu` (CUDA + TF headers) cannot be compiled here; its GEMM chain is restated
    in `equation_construction` below and cross-checked against the reference's pure-TF
    twin formula (`legacy/ba.py:282-283`).

Every function takes/returns numpy arrays and works in the dtype of its inputs
(float32 = the reference's arithmetic; float64 = the "truth" twin used for tolerances).
All citations are file:line under /root/reference.
"""
import numpy as np

SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946

# legacy/ba.py:5-9 module-level flags
ANGLE_CHANGE = 0.002 * (3.14 / 180.0)
TRANSLATION_CHANGE = 0.0002
RESIDUAL_RATIO = 1.0


# --------------------------------------------------------------------------------------
# a2  grad_fixed                                  bundlenet.py:92-100 == legacy/ba.py:17-25
# --------------------------------------------------------------------------------------
def grad_fixed(img):
    """[B,H,W,C] -> [B,H,W,2C] = [gx || gy], central difference on a REFLECT-padded map
    (so the 1-px border gets exactly zero gradient)."""
    p = np.pad(img, [(0, 0), (1, 1), (1, 1), (0, 0)], mode="reflect")
    H, W = img.shape[1], img.shape[2]
    half = img.dtype.type(0.5)
    gx = half * (p[:, 1:H + 1, 2:W + 2, :] - p[:, 1:H + 1, 0:W, :])
    gy = half * (p[:, 2:H + 2, 1:W + 1, :] - p[:, 0:H, 1:W + 1, :])
    return np.concatenate([gx, gy], axis=-1)


def target_map(img):
    """conv2 = [f || gx || gy]  (legacy/ba.py:116-118, bundlenet.py:323-324)."""
    return np.concatenate([img, grad_fixed(img)], axis=-1)


# --------------------------------------------------------------------------------------
# a1  computeCoordinates                   bundlenet.py:112-120 (normalised), ba.py:27-34
# --------------------------------------------------------------------------------------
def compute_coordinates(points2d, fx, fy, ox, oy, normalize):
    """points2d [B,N,2]; fx.. [B,N]  ->  rays p [B,3,N]."""
    x = (points2d[:, :, 0] - ox) / fx
    y = (points2d[:, :, 1] - oy) / fy
    p = np.stack([x, y, np.ones_like(x)], axis=1)
    if normalize:  # tf.nn.l2_normalize(p, axis=1): x * rsqrt(max(sum(x^2), 1e-12))
        ss = np.sum(p * p, axis=1, keepdims=True)
        p = p / np.sqrt(np.maximum(ss, p.dtype.type(1e-12)))
    return p


# --------------------------------------------------------------------------------------
# a8 / a9  Jacobians                               bundlenet.py:49-74, legacy/ba.py:36-48
# --------------------------------------------------------------------------------------
def camera_jacobian(x, y, Z, fx, fy, sign):
    """x,y,Z,fx,fy [B,N] -> [B,N,2,6]; sign=+1 legacy (ba.py:47), -1 bundlenet (:60)."""
    one = x.dtype.type(1.0)
    xy = x * y
    xx = -one - x * x
    x_z = x / Z
    yy = one + y * y
    y_z = y / Z
    iz = one / Z
    zero = np.zeros_like(x)
    dx = fx[..., None] * np.stack([xy, xx, y, -iz, zero, x_z], axis=2)
    dy = fy[..., None] * np.stack([yy, -xy, -x, zero, -iz, y_z], axis=2)
    J = np.stack([dx, dy], axis=2)
    return J if sign > 0 else -J


def depth_jacobian(rx, ry, rz, x, y, Z, fx, fy):
    """[B,N] each -> jd [B,N,2]                                  bundlenet.py:63-74."""
    dx = fx * ((rx - rz * x) / Z)
    dy = fy * ((ry - rz * y) / Z)
    return np.stack([dx, dy], axis=2)


# --------------------------------------------------------------------------------------
# a3 / a5  bilinear sampling
# --------------------------------------------------------------------------------------
def _bilinear_clamped(imgs, x, y):
    """legacy/utils_python.py:61-117 / :177-232 core: weights from the UNclamped floor,
    indices clamped afterwards.  imgs [B,H,W,C], x,y [B,N] -> [B,N,C]."""
    B, H, W, C = imgs.shape
    dt = imgs.dtype.type
    x0f = np.floor(x)
    y0f = np.floor(y)
    dx = x - x0f
    dy = y - y0f
    one = dt(1.0)
    w00 = (one - dx) * (one - dy)
    w01 = dx * (one - dy)
    w10 = (one - dx) * dy
    w11 = dx * dy
    with np.errstate(invalid="ignore"):
        x0 = np.nan_to_num(x0f, nan=0.0, posinf=1e9, neginf=-1e9).astype(np.int64)
        y0 = np.nan_to_num(y0f, nan=0.0, posinf=1e9, neginf=-1e9).astype(np.int64)
    x1 = x0 + 1
    y1 = y0 + 1
    x0 = np.clip(x0, 0, W - 1)
    x1 = np.clip(x1, 0, W - 1)
    y0 = np.clip(y0, 0, H - 1)
    y1 = np.clip(y1, 0, H - 1)
    bi = np.arange(B)[:, None]
    I00 = imgs[bi, y0, x0]
    I01 = imgs[bi, y0, x1]
    I10 = imgs[bi, y1, x0]
    I11 = imgs[bi, y1, x1]
    # tf.add_n of four [.,C,1]x[.,1,1] matmuls: ((a+b)+c)+d
    return ((I00 * w00[..., None] + I01 * w01[..., None]) + I10 * w10[..., None]) + I11 * w11[..., None]


def interpolate2d(imgs, x, y):
    """legacy/utils_python.py:61-117: sample + inclusive in-image mask (NaN -> 0).
    Returns (out [B,N,C], mask [B,N,1])."""
    B, H, W, C = imgs.shape
    out = _bilinear_clamped(imgs, x, y)
    dt = imgs.dtype.type
    cx = np.clip(x, dt(0.0), dt(W - 1.0))
    cy = np.clip(y, dt(0.0), dt(H - 1.0))
    mask = np.logical_and(x == cx, y == cy).astype(imgs.dtype)[..., None]
    return out, mask


def interpolate2d2(imgs, p):
    """legacy/utils_python.py:177-232: sample only.  p [B,N,2]."""
    return _bilinear_clamped(imgs, p[:, :, 0], p[:, :, 1])


def resampler(imgs, warp):
    """tf.contrib.resampler.resampler (TF-1.x contrib/resampler/kernels/resampler_ops.cc,
    restated): bilinear with ZERO padding; a point is sampled iff
    x>-1 && y>-1 && x<W && y<H, else the output is 0.  imgs [B,H,W,C], warp [B,N,2]."""
    B, H, W, C = imgs.shape
    dt = imgs.dtype.type
    x = warp[:, :, 0]
    y = warp[:, :, 1]
    with np.errstate(invalid="ignore"):
        ok = (x > -1.0) & (y > -1.0) & (x < W) & (y < H)
    xs = np.where(ok, x, 0).astype(imgs.dtype)
    ys = np.where(ok, y, 0).astype(imgs.dtype)
    fx = np.floor(xs)
    fy = np.floor(ys)
    cx = fx + 1
    cy = fy + 1
    dx = cx - xs
    dy = cy - ys
    one = dt(1.0)
    bi = np.arange(B)[:, None]

    def tap(xi, yi):
        xi = xi.astype(np.int64)
        yi = yi.astype(np.int64)
        inside = (xi >= 0) & (yi >= 0) & (xi <= W - 1) & (yi <= H - 1)
        v = imgs[bi, np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
        return np.where(inside[..., None], v, dt(0.0))

    out = (dx * dy)[..., None] * tap(fx, fy) + ((one - dx) * (one - dy))[..., None] * tap(cx, cy) \
        + (dx * (one - dy))[..., None] * tap(fx, cy) + ((one - dx) * dy)[..., None] * tap(cx, fy)
    return np.where(ok[..., None], out, dt(0.0)).astype(imgs.dtype)


# --------------------------------------------------------------------------------------
# a15  SE(3) helpers                          bundlenet.py:17-46, legacy/ba.py:51-80
# --------------------------------------------------------------------------------------
def angle_axis_rotation(w, clamp_theta):
    """w [B,3] -> exp(w) [B,3,3].  clamp_theta=True: bundlenet.py:20 (theta>=1e-6);
    False: legacy/ba.py:63 (divides by theta, NaN at 0)."""
    dt = w.dtype.type
    theta = np.sqrt(w[:, 0] * w[:, 0] + w[:, 1] * w[:, 1] + w[:, 2] * w[:, 2])
    if clamp_theta:
        theta = np.maximum(theta, dt(1e-6))
    with np.errstate(invalid="ignore", divide="ignore"):
        wx = w[:, 0] / theta
        wy = w[:, 1] / theta
        wz = w[:, 2] / theta
    c = np.cos(theta)
    s = np.sin(theta)
    oc = dt(1.0) - c
    # the 9 stacked entries, then reshape [-1,3,3] and TRANSPOSE (bundlenet.py:28-37)
    m = np.stack([c + wx * wx * oc, wz * s + wx * wy * oc, -wy * s + wx * wz * oc,
                  wx * wy * oc - wz * s, c + wy * wy * oc, wx * s + wy * wz * oc,
                  wy * s + wx * wz * oc, -wx * s + wy * wz * oc, c + wz * wz * oc], axis=-1)
    return np.transpose(m.reshape(-1, 3, 3), (0, 2, 1))


# The reference builds the skew matrix with tf.stack(...) on axis 0 followed by
# reshape([-1,3,3]) (bundlenet.py:45, ba.py:57), which interleaves the items for B>1.  The
# oracle (like the HIP path) applies the B=1 semantics to every item by default; tests that
# replay the reference's B=2 golden outputs flip this switch to restate the literal layout.
VMATRIX_REFERENCE_BATCH_LAYOUT = False


def vmatrix(w):
    """w [B,3] -> V(w) [B,3,3]  (bundlenet.py:39-46 / ba.py:51-58, applied per item; the
    reference's stack-on-axis-0 form is only correct for B=1 -- SURVEY 2.3)."""
    dt = w.dtype.type
    wx, wy, wz = w[:, 0], w[:, 1], w[:, 2]
    theta = np.sqrt(wx * wx + wy * wy + wz * wz)
    c = np.cos(theta)
    s = np.sin(theta)
    z = np.zeros_like(wx)
    if VMATRIX_REFERENCE_BATCH_LAYOUT:
        K = np.stack([z, -wz, wy, wz, z, -wx, -wy, wx, z], axis=0).reshape(-1, 3, 3)
    else:
        K = np.stack([z, -wz, wy, wz, z, -wx, -wy, wx, z], axis=-1).reshape(-1, 3, 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        a = (dt(1.0) - c) / (theta * theta)
        b = (theta - s) / (theta * theta * theta)
    return np.eye(3, dtype=w.dtype)[None] + a[:, None, None] * K + b[:, None, None] * np.matmul(K, K)


def rotation2quaternion(R):
    """bundlenet.py:6-15."""
    dt = R.dtype.type
    diag = dt(1.0) + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    q0 = np.sqrt(diag) / dt(2.0)
    q1 = (R[:, 2, 1] - R[:, 1, 2]) / (dt(4.0) * q0)
    q2 = (R[:, 0, 2] - R[:, 2, 0]) / (dt(4.0) * q0)
    q3 = (R[:, 1, 0] - R[:, 0, 1]) / (dt(4.0) * q0)
    q = np.stack([q0, q1, q2, q3], axis=1)
    return q / np.sqrt(np.maximum(np.sum(q * q, axis=1, keepdims=True), dt(1e-12)))


# --------------------------------------------------------------------------------------
# a10 / a11 / a12  EquationConstruction (+Grad)
# --------------------------------------------------------------------------------------
def equation_construction(J, G, d):
    """utils.cu:331-414 GEMM chain, restated:  M=G^T G; (MJ); H=J^T M J summed over n;
    g=d^T G; (g J) summed over n.   J [B,N,2,P], G [B,N,C,2], d [B,N,C,1]
    -> AtA [B,P,P], Atb [B,P,1]."""
    Gt = np.swapaxes(G, -1, -2)
    M = np.matmul(Gt, G)                       # F1  [B,N,2,2]
    MJ = np.matmul(M, J)                       # F2  [B,N,2,P]
    H = np.matmul(np.swapaxes(J, -1, -2), MJ)  # F3  [B,N,P,P]
    AtA = _serial_sum(H)                       # ColumnReduceSimpleKernel utils.cu:181-198
    g = np.matmul(np.swapaxes(d, -1, -2), G)   # F4  [B,N,1,2]
    gJ = np.matmul(g, J)                       # F5  [B,N,1,P]
    Atb = _serial_sum(gJ)
    return AtA, np.swapaxes(Atb, -1, -2)


def _serial_sum(x):
    """Sum over axis 1.  utils.cu:193-196 sums rows serially in fp32; numpy's pairwise
    sum is at least as accurate, and the parity bar is 1e-4 relative, so use np.sum."""
    return np.sum(x, axis=1)


def equation_construction_tf_twin(J, G, d):
    """The pure-TF formulation the legacy path actually runs (legacy/ba.py:282-283)."""
    Gt = np.swapaxes(G, -1, -2)
    Jt = np.swapaxes(J, -1, -2)
    AtA = np.sum(np.matmul(Jt, np.matmul(np.matmul(Gt, G), J)), axis=1)
    Atb = np.sum(np.matmul(Jt, np.matmul(Gt, d)), axis=1)
    return AtA, Atb


def equation_construction_gemm(J, G, d):
    """Same quantity as `equation_construction`, arranged the way a sane CPU port would do it
    (and the way the HIP kernel does): per-pixel 2x2 M = G^T G and g = G^T d, then one
    [2N x P]^T [2N x P] GEMM per window instead of N materialised PxP products.  Used for the
    timed CPU baseline (bench.py) where the literal form needs N*P*P*4 bytes (22 GB at
    640x480, P=134)."""
    gx, gy, dd = G[..., 0], G[..., 1], d[..., 0]
    m11, m12, m22 = (gx * gx).sum(-1), (gx * gy).sum(-1), (gy * gy).sum(-1)       # [B,N]
    g1, g2 = (gx * dd).sum(-1), (gy * dd).sum(-1)
    Jx, Jy = J[:, :, 0, :], J[:, :, 1, :]                                            # [B,N,P]
    Zx = m11[..., None] * Jx + m12[..., None] * Jy
    Zy = m12[..., None] * Jx + m22[..., None] * Jy
    AtA = np.matmul(np.swapaxes(Zx, 1, 2), Jx) + np.matmul(np.swapaxes(Zy, 1, 2), Jy)
    Atb = (np.matmul(np.swapaxes(Jx, 1, 2), g1[..., None]) + np.matmul(np.swapaxes(Jy, 1, 2), g2[..., None]))
    return AtA, Atb


def equation_construction_grad(J, G, d, g0, g1):
    """utils.cu:613-690:  A=GJ; dA = 2*A*g0 + d*g1^T; dJ=G^T dA; dG=dA J^T; dd=A g1.
    g0 [B,P,P] (dL/dAtA), g1 [B,P,1] (dL/dAtb).  Note the alpha=2.0 (utils.cu:651): exact
    only for symmetric g0."""
    A = np.matmul(G, J)                                             # B1 [B,N,C,P]
    dd = np.matmul(A, g1[:, None])                                  # B2 [B,N,C,1]
    dA = J.dtype.type(2.0) * np.matmul(A, g0[:, None])              # B3
    dA = dA + np.matmul(d, np.swapaxes(g1, -1, -2)[:, None])        # B4
    dJ = np.matmul(np.swapaxes(G, -1, -2), dA)                      # B5 [B,N,2,P]
    dG = np.matmul(dA, np.swapaxes(J, -1, -2))                      # B6 [B,N,C,2]
    return dJ, dG, dd


# --------------------------------------------------------------------------------------
# a7  lambda prediction                   bundlenet.py:165-173,241-253; ba.py:266-275
# --------------------------------------------------------------------------------------
def selu(x):
    dt = x.dtype.type
    return dt(SELU_SCALE) * np.where(x > 0, x, dt(SELU_ALPHA) * (np.exp(np.minimum(x, 0)) - dt(1.0)))


def lambda_mlp(avg, weights):
    """avg [B,1,C]; weights = list of 5 (filters [Cin,Cout], biases [Cout]) -- conv1d with
    kernel 1 == per-item matvec.  selu x4, tanh (bundlenet.py:168-172).  -> [B,1,1]."""
    h = avg
    for i, (w, b) in enumerate(weights):
        h = np.matmul(h, w.astype(avg.dtype)) + b.astype(avg.dtype)
        h = selu(h) if i < 4 else np.tanh(h)
    return h


def he_normal_mlp_weights(C, seed, dtype=np.float32):
    """Five-layer C->2C->4C->2C->C->1 weights, he_normal (truncated normal, stddev
    sqrt(2/fan_in)/0.8796) and zero bias, as bundlenet.py:105-106 initialises them.
    (Random values, not a checkpoint -- the reference ships none.)"""
    rng = np.random.RandomState(seed)
    dims = [C, 2 * C, 4 * C, 2 * C, C, 1]
    ws = []
    for i in range(5):
        std = np.sqrt(2.0 / dims[i]) / 0.87962566103423978
        w = rng.standard_normal((dims[i], dims[i + 1]))
        w = np.clip(w, -2.0, 2.0) * std
        ws.append((w.astype(dtype), np.zeros(dims[i + 1], dtype)))
    return ws


# --------------------------------------------------------------------------------------
# a14  linear solves
# --------------------------------------------------------------------------------------
def solve_lu(A, b):
    """tf.matrix_solve (bundlenet.py:183,267): LU with partial pivoting (LAPACK gesv)."""
    return np.linalg.solve(A, b)


def solve_qr(A, b):
    """legacy/ba.py:292-293: q,r = tf.qr(AtA, full_matrices=True);
    motion = tf.linalg.solve(r, q^T Atb)."""
    out = np.empty_like(b)
    for i in range(A.shape[0]):
        q, r = np.linalg.qr(A[i], mode="complete")
        out[i] = np.linalg.solve(r, np.matmul(q.T, b[i]))
    return out


# --------------------------------------------------------------------------------------
# a4  warp
# --------------------------------------------------------------------------------------
def warp(R, T, p, D, fx, fy, ox, oy):
    """R [B,3,3], T [B,3,1], p [B,3,N], D [B,N,1] -> dict of [B,N] arrays
    (legacy/ba.py:239-251; bundlenet.py:209-224)."""
    Rp = np.matmul(R, p)
    RPT = Rp * np.transpose(D, (0, 2, 1)) + T
    X, Y, Z = RPT[:, 0, :], RPT[:, 1, :], RPT[:, 2, :]
    with np.errstate(invalid="ignore", divide="ignore"):
        x = X / Z
        y = Y / Z
    return dict(rx=Rp[:, 0, :], ry=Rp[:, 1, :], rz=Rp[:, 2, :], x=x, y=y, Z=Z,
                px=fx * x + ox, py=fy * y + oy)


def bundlenet_mask(px, py, H, W):
    """bundlenet.py:155,231: not any(px<0, px>W-1, py<0, py>H-1)  (NaN -> 1)."""
    dt = px.dtype.type
    with np.errstate(invalid="ignore"):
        bad = (px < 0) | (px > dt(W - 1)) | (py < 0) | (py > dt(H - 1))
    return (~bad).astype(px.dtype)


# --------------------------------------------------------------------------------------
# legacy pose-only iterations                                   legacy/ba.py:148-345
# --------------------------------------------------------------------------------------
def _legacy_residuals(conv1, conv2, w):
    """ba.py:256-264: returns diff [B,N,C,1], grad [B,N,C,2], mask [B,N,1]."""
    C = conv1.shape[2]
    s, mask = interpolate2d(conv2, w["px"], w["py"])
    diff = ((s[:, :, 0:C] - conv1) * mask)[..., None]
    grad = np.stack([s[:, :, C:2 * C] * mask, s[:, :, 2 * C:3 * C] * mask], axis=-1)
    return diff, grad, mask


def legacy_camera_iteration(conv1, conv2, fx, fy, ox, oy, p, D, R, T, use_qr=True):
    """Fixed-iteration legacy step, legacy/ba.py:148-214 (lambda = ||avg||^2, no MLP, no V,
    ratio = sum(mask)/N)."""
    N = conv1.shape[1]
    w = warp(R, T, p, D, fx, fy, ox, oy)
    diff, grad, mask = _legacy_residuals(conv1, conv2, w)
    avg = np.mean(np.abs(diff[..., 0]), axis=1, keepdims=True)             # [B,1,C]
    lam = np.sqrt(np.sum(avg * avg, axis=-1, keepdims=True)) ** conv1.dtype.type(2.0)
    J = camera_jacobian(w["x"], w["y"], w["Z"], fx, fy, +1)
    AtA, Atb = equation_construction_tf_twin(J, grad, diff)
    AtA = damp(AtA, lam, undamped_last=False)
    motion = solve_qr(AtA, Atb) if use_qr else np.matmul(np.linalg.inv(AtA), Atb)
    dr = angle_axis_rotation(motion[:, 0:3, 0], clamp_theta=False)
    Rn = np.matmul(dr, R)
    Tn = motion[:, 3:6] + np.matmul(dr, T)                                  # ba.py:213
    return Rn, Tn, np.sum(mask) / conv1.dtype.type(N)


def damp(AtA, lam, undamped_last):
    """a13: AtA += diag((diag+1e-5)*lambda); bundle variant leaves the last coefficient
    undamped (bundlenet.py:264-266).  lam [B,1,1]."""
    dt = AtA.dtype.type
    diag = np.diagonal(AtA, axis1=1, axis2=2)                              # [B,P]
    add = (diag + dt(1e-5)) * lam[:, 0, :]
    if undamped_last:
        add = add.copy()
        add[:, -1] = 0
    out = AtA.copy()
    idx = np.arange(AtA.shape[1])
    out[:, idx, idx] += add
    return out


def legacy_avg_residual(conv1, conv2, fx, fy, ox, oy, p, D, R, T):
    """ba.py:306-324 (CheckUpdate) == the per-channel scaled mean used at :268,:275."""
    N = conv1.shape[1]
    C = conv1.shape[2]
    w = warp(R, T, p, D, fx, fy, ox, oy)
    s, mask = interpolate2d(conv2, w["px"], w["py"])
    with np.errstate(divide="ignore", invalid="ignore"):
        num_valid = conv1.dtype.type(N) / np.sum(mask, axis=1, keepdims=True)   # [B,1,1]
    diff = mask * (s[:, :, 0:C] - conv1)
    avg = num_valid * np.mean(np.abs(diff), axis=1, keepdims=True)              # [B,1,C]
    return avg, num_valid


def legacy_camera_iteration2(conv1, conv2, fx, fy, ox, oy, p, D, R, T, mlp, use_qr=True):
    """One LM step with accept/reject, legacy/ba.py:226-345.  B must be 1 (the reference's
    tf.cond/tf.squeeze are scalar).  Returns R,T,update_w,update_t,ratio,(debug dict)."""
    dt = conv1.dtype.type
    N = conv1.shape[1]
    w = warp(R, T, p, D, fx, fy, ox, oy)
    diff, grad, mask = _legacy_residuals(conv1, conv2, w)
    with np.errstate(divide="ignore", invalid="ignore"):
        num_valid = dt(N) / np.sum(mask, axis=1, keepdims=True)
    avg = num_valid * np.mean(np.abs(diff[..., 0]), axis=1, keepdims=True)      # [B,1,C]
    y = lambda_mlp(avg, mlp)
    nrm = np.sqrt(np.sum(avg * avg, axis=-1, keepdims=True))
    lam = nrm ** (dt(1.0) + y)
    avg_scalar = np.mean(avg)
    J = camera_jacobian(w["x"], w["y"], w["Z"], fx, fy, +1)
    AtA0, Atb = equation_construction_tf_twin(J, grad, diff)
    AtA = damp(AtA0, lam, undamped_last=False)
    motion = solve_qr(AtA, Atb) if use_qr else np.matmul(np.linalg.inv(AtA), Atb)
    dr = angle_axis_rotation(motion[:, 0:3, 0], clamp_theta=False)
    dv = vmatrix(motion[:, 0:3, 0])
    Rn = np.matmul(dr, R)
    Tn = np.matmul(dv, motion[:, 3:6]) + np.matmul(dr, T)
    avg2, _ = legacy_avg_residual(conv1, conv2, fx, fy, ox, oy, p, D, Rn, Tn)
    avg2_scalar = np.mean(avg2)
    m = motion.reshape(-1)
    dbg = dict(AtA=AtA0, Atb=Atb, lam=lam, avg=avg, motion=motion, avg_scalar=avg_scalar,
               avg2_scalar=avg2_scalar)
    ratio = np.squeeze(num_valid)
    if avg2_scalar < dt(RESIDUAL_RATIO) * avg_scalar:
        return Rn, Tn, np.sqrt(np.sum(m[:3] * m[:3])), np.sqrt(np.sum(m[3:] * m[3:])), ratio, dbg
    return R, T, dt(0.0), dt(0.0), ratio, dbg


def legacy_track(intrinsic, layers, points, d, initR, initT, level_iters, mlps,
                 early_termination=True, use_qr=True):
    """Tracker.trackTF, legacy/ba.py:85-145.  layers: list of 3 maps coarse->fine, each
    [2,H_l,W_l,C] (item 0 = source, item 1 = target).  intrinsic [1,4,1], points [1,N,2],
    d [1,N,1].  mlps: dict level-name ('1','2','3') -> 5-layer weights.
    Returns (R, T, ratio, iters_per_level)."""
    N = points.shape[1]
    dt = points.dtype
    fx0 = np.tile(intrinsic[:, 0], (1, N))
    fy0 = np.tile(intrinsic[:, 1], (1, N))
    ox0 = np.tile(intrinsic[:, 2], (1, N))
    oy0 = np.tile(intrinsic[:, 3], (1, N))
    p = compute_coordinates(points, fx0, fy0, ox0, oy0, normalize=False)
    R, T = initR, initT
    ratio = dt.type(1.0)
    counts = []
    for level in range(1, 4):
        scale = dt.type(2 ** (3 - level))
        fx, fy, ox, oy = fx0 / scale, fy0 / scale, ox0 / scale, oy0 / scale
        pts = points / scale
        conv1 = interpolate2d2(layers[level - 1][0:1], pts)
        conv2 = target_map(layers[level - 1][1:2])
        it = 0
        if not early_termination:
            for _ in range(level_iters[level - 1]):
                R, T, ratio = legacy_camera_iteration(conv1, conv2, fx, fy, ox, oy, p, d, R, T, use_qr)
                it += 1
        else:
            uw = ut = dt.type(1.0)
            while it < level_iters[level - 1] and ANGLE_CHANGE < uw and TRANSLATION_CHANGE < ut:
                R, T, uw, ut, ratio, _ = legacy_camera_iteration2(
                    conv1, conv2, fx, fy, ox, oy, p, d, R, T, mlps[str(level)], use_qr)
                it += 1
        counts.append(it)
    return R, T, ratio, counts


# --------------------------------------------------------------------------------------
# bundlenet.py iterations                                        bundlenet.py:122-278
# --------------------------------------------------------------------------------------
def _bundle_residuals(conv1, conv2, w):
    """bundlenet.py:154-163 / :230-239: diff = (F1 - F2w)*mask."""
    H, W = conv2.shape[1], conv2.shape[2]
    C = conv1.shape[2]
    s = resampler(conv2, np.stack([w["px"], w["py"]], axis=-1))
    mask = bundlenet_mask(w["px"], w["py"], H, W)[..., None]
    diff = ((conv1 - s[:, :, 0:C]) * mask)[..., None]
    grad = np.stack([s[:, :, C:2 * C] * mask, s[:, :, 2 * C:3 * C] * mask], axis=-1)
    return diff, grad, mask


def _se3_update(sol6, R, T):
    """bundlenet.py:185-190 / :270-275."""
    wv = sol6[:, 0:3, 0]
    dr = angle_axis_rotation(wv, clamp_theta=True)
    dv = vmatrix(wv)
    return np.matmul(dr, R), np.matmul(dv, sol6[:, 3:6]) + np.matmul(dr, T)


def bundle_camera_iteration(conv1, conv2, fx, fy, ox, oy, p, D, R, T, mlp, l2_base=None):
    """BundleNet.CameraIteration, bundlenet.py:122-191 (pose only, P=6, all-diagonal
    damping; l2_regularizer_base is accepted and ignored, as in the reference)."""
    dt = conv1.dtype.type
    w = warp(R, T, p, D, fx, fy, ox, oy)
    diff, grad, mask = _bundle_residuals(conv1, conv2, w)
    avg = np.mean(np.abs(diff[..., 0]), axis=1, keepdims=True)
    y = lambda_mlp(avg, mlp)
    lam = np.sqrt(np.sum(avg * avg, axis=-1, keepdims=True)) ** (dt(2.0) + y)
    J = camera_jacobian(w["x"], w["y"], w["Z"], fx, fy, -1)
    AtA0, Atb = equation_construction(J, grad, diff)
    AtA = damp(AtA0, lam, undamped_last=False)
    motion = solve_lu(AtA, Atb)
    Rn, Tn = _se3_update(motion, R, T)
    return Rn, Tn, dict(AtA=AtA0, Atb=Atb, lam=lam, avg=avg, motion=motion)


def bundle_iteration(conv1, conv2, fx, fy, ox, oy, p, D, Bs, R, T, W, mlp, l2_base=None, eq=None):
    """BundleNet.BundleIteration, bundlenet.py:193-278.  Bs [B,N,K], W [B,K,1].
    eq: normal-equation routine (default: the literal utils.cu GEMM chain)."""
    eq = eq or equation_construction
    dt = conv1.dtype.type
    Dn = D + np.matmul(Bs, W)
    w = warp(R, T, p, Dn, fx, fy, ox, oy)
    diff, grad, mask = _bundle_residuals(conv1, conv2, w)
    avg = np.mean(np.abs(diff[..., 0]), axis=1, keepdims=True)
    y = lambda_mlp(avg, mlp)
    lam = np.sqrt(np.sum(avg * avg, axis=-1, keepdims=True)) ** (dt(2.0) + y)
    if l2_base is not None:
        lam = dt(l2_base) * lam
    Jc = camera_jacobian(w["x"], w["y"], w["Z"], fx, fy, -1)
    jd = depth_jacobian(w["rx"], w["ry"], w["rz"], w["x"], w["y"], w["Z"], fx, fy)
    Jd = np.matmul(jd[..., None], Bs[:, :, None, :])
    J = np.concatenate([Jc, Jd], axis=-1)
    AtA0, Atb = eq(J, grad, diff)
    AtA = damp(AtA0, lam, undamped_last=True)
    sol = solve_lu(AtA, Atb)
    Rn, Tn = _se3_update(sol[:, 0:6], R, T)
    Wn = W + sol[:, 6:]
    return Rn, Tn, Wn, dict(AtA=AtA0, Atb=Atb, lam=lam, avg=avg, solution=sol, mask=mask)


def bundle_window_iteration(conv1, conv2s, fx, fy, ox, oy, p, D, Bs, Rs, Ts, W, mlp, l2_base=None, eq=None):
    """Multi-frame window (F = 1 + len(conv2s) frames): NOT in the reference -- the definition fixed
    by SURVEY.md par. 8(d): keyframe 0 carries D0, the basis and W; every other frame i has its own
    (R_i, T_i) and contributes an independent BundleIteration term (bundlenet.py:206-263, the same
    functions as bundle_iteration above, unchanged); P = 6 (F-1) + K, parameter order
    [pose_1 .. pose_{F-1}, depth]; the rows of all pairs go through ONE EquationConstruction, so
    AtA is block-arrowhead (pose blocks on the diagonal, shared depth block summed).  lambda:
    bundlenet.py:241-253 with the residual averaged over all pairs; damping, solve and updates as
    bundlenet.py:264-276 (last coefficient undamped).  With one pair this IS bundle_iteration.
    conv2s: list of target maps [B,H,W,3C]; Rs [pairs][B,3,3]; Ts [pairs][B,3,1]."""
    eq = eq or equation_construction
    dt = conv1.dtype.type
    pairs = len(conv2s)
    K = Bs.shape[-1]
    Pn = 6 * pairs + K
    Dn = D + np.matmul(Bs, W)
    Js, Gs, ds, avgs, masks = [], [], [], [], []
    for i in range(pairs):
        w = warp(Rs[i], Ts[i], p, Dn, fx, fy, ox, oy)
        diff, grad, mask = _bundle_residuals(conv1, conv2s[i], w)
        avgs.append(np.mean(np.abs(diff[..., 0]), axis=1, keepdims=True))
        Jc = camera_jacobian(w["x"], w["y"], w["Z"], fx, fy, -1)
        jd = depth_jacobian(w["rx"], w["ry"], w["rz"], w["x"], w["y"], w["Z"], fx, fy)
        Jd = np.matmul(jd[..., None], Bs[:, :, None, :])
        J = np.zeros(Jc.shape[:3] + (Pn,), conv1.dtype)
        J[..., 6 * i:6 * i + 6] = Jc
        J[..., 6 * pairs:] = Jd
        Js.append(J)
        Gs.append(grad)
        ds.append(diff)
        masks.append(mask)
    avg = sum(avgs) / dt(pairs)
    y = lambda_mlp(avg, mlp)
    lam = np.sqrt(np.sum(avg * avg, axis=-1, keepdims=True)) ** (dt(2.0) + y)
    if l2_base is not None:
        lam = dt(l2_base) * lam
    AtA0, Atb = eq(np.concatenate(Js, axis=1), np.concatenate(Gs, axis=1), np.concatenate(ds, axis=1))
    AtA = damp(AtA0, lam, undamped_last=True)
    sol = solve_lu(AtA, Atb)
    Rn, Tn = [], []
    for i in range(pairs):
        r, t = _se3_update(sol[:, 6 * i:6 * i + 6], Rs[i], Ts[i])
        Rn.append(r)
        Tn.append(t)
    Wn = W + sol[:, 6 * pairs:]
    return Rn, Tn, Wn, dict(AtA=AtA0, Atb=Atb, lam=lam, avg=avg, solution=sol, mask=masks)


# --------------------------------------------------------------------------------------
# losses                                                          bundlenet.py:401-463
# --------------------------------------------------------------------------------------
def loss_r(predQ, gtQ):
    """bundlenet.py:401-404: tf.losses.cosine_distance(predQ, gtQ, axis=1) = mean_b (1 - <predQ_b, gtQ_b>)."""
    dt = predQ.dtype.type
    return np.mean(dt(1.0) - np.sum(predQ * gtQ, axis=1))


def loss_t(predT, gtT):
    """bundlenet.py:410-412 (the second `lossT` definition overrides the cosine one at :405-408)."""
    return np.mean(np.abs(predT - gtT))


def loss_f(intrisic, depth, mask, predR, predT, gtR, gtT):
    """bundlenet.py:414-463: masked mean absolute flow difference between the predicted and the ground-truth pose,
    crop-adjusted intrinsics (:441-445), unit rays of the integer pixel grid (:447-452), BOTH terms divided by the
    width (:462-463, as the reference has it), scaled by total / valid pixel count."""
    dt = depth.dtype.type
    nb, H, W = depth.shape[0], depth.shape[1], depth.shape[2]
    N = H * W
    m = mask.reshape(nb, N)
    fx, fy, ox, oy = _crop_intrinsics(intrisic, N)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    pts = np.tile(np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(depth.dtype)[None], (nb, 1, 1))
    p = compute_coordinates(pts, fx, fy, ox, oy, normalize=True)

    def flow(R, T):
        X = np.matmul(R, p) * depth.reshape(nb, 1, N) + T.reshape(nb, 3, 1)
        return fx * (X[:, 0] / X[:, 2]) + ox, fy * (X[:, 1] / X[:, 2]) + oy

    fxp, fyp = flow(predR, predT)
    fxg, fyg = flow(gtR, gtT)
    valid, total = np.sum(m), dt(N * nb)
    return (total / valid) * (np.mean(np.abs(fxp - fxg) * m) / dt(W) + np.mean(np.abs(fyp - fyg) * m) / dt(W))


def _crop_intrinsics(intrisic, N):
    """bundlenet.py:298-302 / :354-357 (crop 4 px, rescale to 320x256)."""
    dt = intrisic.dtype.type
    fx = dt(40.0) * np.tile(intrisic[:, 0], (1, N)) / dt(39.0)
    fy = dt(32.0) * np.tile(intrisic[:, 1], (1, N)) / dt(29.0)
    ox = (dt(40.0) * np.tile(intrisic[:, 2], (1, N)) / dt(39.0)) - dt(160.0 / 39.0)
    oy = (dt(32.0) * np.tile(intrisic[:, 3], (1, N)) / dt(29.0)) - dt(128.0 / 29.0)
    return fx, fy, ox, oy


def _crop_points(points):
    """bundlenet.py:285-288 / :337-340."""
    dt = points.dtype.type
    x = dt(320) * (points[..., 0:1] - dt(4)) / dt(312)
    y = dt(256) * (points[..., 1:2] - dt(4)) / dt(232)
    return np.concatenate([x, y], axis=-1)


def _swap_halves(x):
    n = x.shape[0]
    return np.concatenate([x[n // 2:n], x[0:n // 2]], axis=0)


def camera_resize(intrisic, layers, points, depths, mlps):
    """BundleNet.CameraResize, bundlenet.py:280-329: 4 levels (scale 8,4,2,1) x 1 iter."""
    _pts = _crop_points(points)
    d = resampler(depths, _pts / points.dtype.type(2))
    B = layers[-1].shape[0]
    N = points.shape[1]
    fx0, fy0, ox0, oy0 = _crop_intrinsics(intrisic, N)
    p = compute_coordinates(_pts, fx0, fy0, ox0, oy0, normalize=True)
    R = np.tile(np.eye(3, dtype=points.dtype)[None], (B, 1, 1))
    T = np.zeros((B, 3, 1), points.dtype)
    Rs, Ts = [], []
    for level in range(0, 4):
        scale = points.dtype.type(2 ** (3 - level))
        layer1 = resampler(layers[level], _pts / scale)
        layer2 = target_map(_swap_halves(layers[level]))
        R, T, _ = bundle_camera_iteration(layer1, layer2, fx0 / scale, fy0 / scale, ox0 / scale,
                                          oy0 / scale, p, d, R, T, mlps[str(level)], 1.0)
        Rs.append(R)
        Ts.append(T)
    return Rs, Ts


def bundle_resize(intrisic, layers, points, basis, init_depth, mlps, init_rotation=None,
                  init_translation=None, stop_gradient_depth=None):
    """BundleNet.BundleResize, bundlenet.py:332-399: levels 2,3 (scale 2,1) x 1 iter.
    stop_gradient_depth: the value `depths = tf.stop_gradient(init_depth)` (bundlenet.py:341) holds -- numerically
    init_depth, but a finite-difference gradient check must keep it fixed while init_depth is perturbed (default:
    init_depth itself, i.e. the forward value)."""
    dt = points.dtype
    _pts = _crop_points(points)
    d = resampler(init_depth if stop_gradient_depth is None else stop_gradient_depth, _pts / dt.type(2))
    b = resampler(basis, _pts / dt.type(2))
    B = layers[-1].shape[0]
    N = points.shape[1]
    K = basis.shape[-1]
    fx0, fy0, ox0, oy0 = _crop_intrinsics(intrisic, N)
    p = compute_coordinates(_pts, fx0, fy0, ox0, oy0, normalize=True)
    R = np.tile(np.eye(3, dtype=dt)[None], (B, 1, 1)) if init_rotation is None else init_rotation
    T = np.zeros((B, 3, 1), dt) if init_translation is None else init_translation
    W = np.zeros((B, K, 1), dt)
    Rs, Ts, Ds = [], [], []
    for level in range(2, 4):
        scale = dt.type(2 ** (3 - level))
        layer1 = resampler(layers[level], _pts / scale)
        layer2 = target_map(_swap_halves(layers[level]))
        R, T, W, _ = bundle_iteration(layer1, layer2, fx0 / scale, fy0 / scale, ox0 / scale,
                                      oy0 / scale, p, d, b, R, T, W, mlps[str(level)], 1000.0)
        Rs.append(R)
        Ts.append(T)
        Hh, Wh = init_depth.shape[1], init_depth.shape[2]
        Ds.append(init_depth + np.matmul(basis.reshape(B, -1, K), W).reshape(B, Hh, Wh, 1))
    return Rs, Ts, Ds
