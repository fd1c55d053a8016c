"""Minimal eager numpy stand-in for the TensorFlow-1.x API surface that the reference's BA
path touches  --  TEST INFRASTRUCTURE ONLY (lives under oracle/).

Purpose: TensorFlow 1.x cannot be installed in this environment, so the reference's own
Python (`legacy/ba.py`, `legacy/utils_python.py`, `bundlenet.py`) cannot run as shipped.
This package restates, from their published semantics, exactly the TF ops those files
call, as eager float32 numpy functions, so that `tests/golden/make_golden.py` can import
and execute the reference's code VERBATIM and record its outputs as golden vectors.
Nothing here is derived from TensorFlow source code; each op is the textbook definition
(see docstrings).  `load_op_library` returns the numpy restatement of `utils.cu`'s two ops
(the CUDA file itself cannot be built here).

Not a general TF emulation: graph mode, sessions, placeholders and gradients are absent.
"""
import contextlib
import types

import numpy as np

float32 = np.float32
int32 = np.int32
AUTO_REUSE = "AUTO_REUSE"


class _Shape(list):
    def as_list(self):
        return list(self)


class T(np.ndarray):
    """ndarray that answers .get_shape() like a TF1 tensor with a static shape."""

    def get_shape(self):
        return _Shape(int(s) for s in self.shape)

    # TF-1.x tensors compare by identity, so `tensor == None` is plain False
    # (bundlenet.py:364,369 rely on it); everything else stays elementwise.
    def __eq__(self, other):
        if other is None:
            return False
        return np.ndarray.__eq__(self, other)

    __hash__ = None


def _t(x, dtype=None):
    a = np.asarray(x, dtype=dtype)
    if a.dtype == np.float64 and dtype is None:
        a = a.astype(np.float32)
    return a.view(T)


def convert(x):
    return _t(x)


# ---- scopes ---------------------------------------------------------------------------
@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    yield name


_VARIABLES = {}           # name -> np.ndarray ; tests pre-populate this with chosen weights
_SCOPE_STACK = []


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _SCOPE_STACK.append(name)
    try:
        yield name
    finally:
        _SCOPE_STACK.pop()


def reset_variables():
    _VARIABLES.clear()


def set_variable(full_name, value):
    _VARIABLES[full_name] = np.asarray(value, np.float32)


def get_variable(shape=None, initializer=None, name=None, dtype=None):
    full = "/".join(_SCOPE_STACK + [name])
    if full not in _VARIABLES:
        _VARIABLES[full] = np.asarray(initializer(shape), np.float32)
    v = _VARIABLES[full]
    assert list(v.shape) == list(shape), (full, v.shape, shape)
    return _t(v)


def zeros_initializer():
    return lambda shape: np.zeros(shape, np.float32)


class _HeNormal:
    _rng = np.random.RandomState(20180925)

    def __call__(self, shape):
        fan_in = int(np.prod(shape[:-1]))
        std = np.sqrt(2.0 / fan_in) / 0.87962566103423978
        return (np.clip(self._rng.standard_normal(shape), -2, 2) * std).astype(np.float32)


keras = types.SimpleNamespace(initializers=types.SimpleNamespace(he_normal=lambda: _HeNormal()))


# ---- elementwise / shape ops ----------------------------------------------------------
def ones(shape, dtype=np.float32):
    return _t(np.ones(list(shape), dtype))


def zeros(shape, dtype=np.float32):
    return _t(np.zeros(list(shape), np.dtype(dtype)))


def eye(n, m=None, batch_shape=None):
    e = np.eye(n, m, dtype=np.float32)
    if batch_shape is not None:
        e = np.tile(e[None], list(batch_shape) + [1, 1])
    return _t(e)


def sqrt(x): return _t(np.sqrt(x))
def cos(x): return _t(np.cos(x))
def sin(x): return _t(np.sin(x))
def square(x): return _t(np.square(x))
def abs(x): return _t(np.abs(x))  # noqa: A001
def floor(x): return _t(np.floor(x))
def maximum(a, b): return _t(np.maximum(a, np.float32(b) if np.isscalar(b) else b))
def multiply(a, b): return _t(np.multiply(a, b))
def add(a, b): return _t(np.add(a, b))


def div(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return _t(np.true_divide(a, b))


def pow(x, y):  # noqa: A001
    y = np.float32(y) if np.isscalar(y) else y
    return _t(np.power(x, y))


def to_float(x): return _t(np.asarray(x).astype(np.float32))
def cast(x, dtype): return _t(np.asarray(x).astype(np.dtype(dtype)))
def identity(x): return _t(x)
def stop_gradient(x): return _t(x)
def less(a, b): return np.less(a, b)
def equal(a, b): return np.equal(a, b)
def logical_and(a, b): return np.logical_and(a, b)
def logical_not(a): return np.logical_not(a)
def reduce_all(x): return bool(np.all(x))
def reduce_any(x, axis=None, keepdims=False): return np.any(x, axis=axis, keepdims=keepdims)


def clip_by_value(x, lo, hi):
    x = np.asarray(x)
    return _t(np.clip(x, np.asarray(lo).astype(x.dtype), np.asarray(hi).astype(x.dtype)))


def range(*a):  # noqa: A001
    return _t(np.arange(*a, dtype=np.int32))


def reshape(x, shape): return _t(np.reshape(x, list(shape)))
def squeeze(x, axis=None): return _t(np.squeeze(x, axis=axis))
def expand_dims(x, axis=None, dim=None): return _t(np.expand_dims(x, axis if axis is not None else dim))
def tile(x, multiples): return _t(np.tile(x, list(multiples)))
def transpose(x, perm=None): return _t(np.transpose(x, perm))
def stack(xs, axis=0): return _t(np.stack([np.asarray(v) for v in xs], axis=axis))
def unstack(x, num=None, axis=0): return [_t(v) for v in np.moveaxis(np.asarray(x), axis, 0)]
def concat(xs, axis): return _t(np.concatenate([np.asarray(v) for v in xs], axis=axis))
def split(x, num_or_size_splits, axis=0): return [_t(v) for v in np.split(np.asarray(x), num_or_size_splits, axis=axis)]


def pad(x, paddings, mode="CONSTANT"):
    return _t(np.pad(x, paddings, mode={"REFLECT": "reflect", "CONSTANT": "constant",
                                        "SYMMETRIC": "symmetric"}[mode]))


def gather(params, indices): return _t(np.asarray(params)[np.asarray(indices)])


def add_n(xs):
    out = np.asarray(xs[0])
    for v in xs[1:]:
        out = out + np.asarray(v)
    return _t(out)


def matmul(a, b, transpose_a=False, transpose_b=False):
    a = np.asarray(a)
    b = np.asarray(b)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    return _t(np.matmul(a, b))


def _kd(keepdims, keep_dims):
    return bool(keepdims) or bool(keep_dims)


def reduce_sum(x, axis=None, keepdims=False, keep_dims=False):
    return _t(np.sum(x, axis=axis, keepdims=_kd(keepdims, keep_dims), dtype=np.asarray(x).dtype))


def reduce_mean(x, axis=None, keepdims=False, keep_dims=False):
    return _t(np.mean(x, axis=axis, keepdims=_kd(keepdims, keep_dims), dtype=np.asarray(x).dtype))


def norm(x, axis=None, keepdims=False, keep_dims=False):
    x = np.asarray(x)
    return _t(np.sqrt(np.sum(x * x, axis=axis, keepdims=_kd(keepdims, keep_dims))))


# ---- linear algebra -------------------------------------------------------------------
def matrix_diag_part(x): return _t(np.diagonal(x, axis1=-2, axis2=-1).copy())


def matrix_diag(d):
    d = np.asarray(d)
    out = np.zeros(d.shape + (d.shape[-1],), d.dtype)
    i = np.arange(d.shape[-1])
    out[..., i, i] = d
    return _t(out)


def matrix_solve(A, b): return _t(np.linalg.solve(np.asarray(A), np.asarray(b)))
def matrix_inverse(A): return _t(np.linalg.inv(np.asarray(A)))


def qr(A, full_matrices=False):
    A = np.asarray(A)
    qs, rs = [], []
    for a in A.reshape((-1,) + A.shape[-2:]):
        q, r = np.linalg.qr(a, mode="complete" if full_matrices else "reduced")
        qs.append(q)
        rs.append(r)
    q = np.stack(qs).reshape(A.shape[:-2] + qs[0].shape)
    r = np.stack(rs).reshape(A.shape[:-2] + rs[0].shape)
    return _t(q), _t(r)


linalg = types.SimpleNamespace(solve=matrix_solve)


# ---- control flow ---------------------------------------------------------------------
def cond(pred, true_fn, false_fn):
    return true_fn() if bool(np.asarray(pred)) else false_fn()


def while_loop(cond, body, loop_vars, back_prop=True, parallel_iterations=10):  # noqa: A002
    v = list(loop_vars)
    while cond(*v):
        v = list(body(*v))
    return v


# ---- nn -------------------------------------------------------------------------------
_SELU_ALPHA = 1.6732632423543772848170429916717
_SELU_SCALE = 1.0507009873554804934193349852946


def _selu(x):
    x = np.asarray(x)
    return _t(np.float32(_SELU_SCALE) * np.where(x > 0, x, np.float32(_SELU_ALPHA) * (np.exp(np.minimum(x, 0)) - np.float32(1))))


def _elu(x):
    x = np.asarray(x)
    return _t(np.where(x > 0, x, np.exp(np.minimum(x, 0)) - np.float32(1)))


def _conv1d(x, filters, stride, padding="SAME"):
    """tf.nn.conv1d; only the kernel-width-1 case the reference uses: [B,L,Cin]x[1,Cin,Cout]."""
    f = np.asarray(filters)
    assert f.shape[0] == 1 and stride == 1
    return _t(np.matmul(np.asarray(x), f[0]))


def _l2_normalize(x, axis=None, epsilon=1e-12, dim=None):
    x = np.asarray(x)
    axis = axis if axis is not None else dim
    ss = np.sum(x * x, axis=axis, keepdims=True)
    return _t(x / np.sqrt(np.maximum(ss, np.float32(epsilon))))


nn = types.SimpleNamespace(selu=_selu, elu=_elu, tanh=lambda x: _t(np.tanh(x)), conv1d=_conv1d,
                           bias_add=lambda x, b: _t(np.asarray(x) + np.asarray(b)),
                           l2_normalize=_l2_normalize)


# ---- tf.losses / tf.meshgrid (bundlenet.py:401-463) ---------------------------------------
def _cosine_distance(labels, predictions, axis=None, dim=None):
    """tf.losses.cosine_distance (TF 1.x): losses = 1 - sum(labels * predictions, axis, keepdims), reduced with the
    default Reduction.SUM_BY_NONZERO_WEIGHTS and weights = 1, i.e. the mean over the elements of `losses`."""
    axis = axis if axis is not None else dim
    l = np.float32(1.0) - np.sum(np.asarray(labels) * np.asarray(predictions), axis=axis, keepdims=True)
    return _t(np.sum(l, dtype=np.float32) / np.float32(l.size))


losses = types.SimpleNamespace(cosine_distance=_cosine_distance)


def meshgrid(x, y):
    """tf.meshgrid with the default indexing='xy'"""
    X, Y = np.meshgrid(np.asarray(x), np.asarray(y))
    return _t(X), _t(Y)


# ---- tf.contrib.resampler -------------------------------------------------------------
def _resampler(data, warp, name=None):
    """Bilinear resampling with ZERO padding outside the image (published semantics of
    tf.contrib.resampler: a point contributes iff x>-1, y>-1, x<W, y<H; taps outside the
    image read 0).  data [B,H,W,C], warp [B,N,2] (x,y) -> [B,N,C]."""
    data = np.asarray(data)
    warp = np.asarray(warp)
    B, H, W, C = data.shape
    out = np.zeros((B, warp.shape[1], C), data.dtype)
    for b in np.arange(B):
        x = warp[b, :, 0]
        y = warp[b, :, 1]
        with np.errstate(invalid="ignore"):
            ok = (x > -1.0) & (y > -1.0) & (x < W) & (y < H)
        xs = np.where(ok, x, 0).astype(data.dtype)
        ys = np.where(ok, y, 0).astype(data.dtype)
        fx = np.floor(xs).astype(np.int64)
        fy = np.floor(ys).astype(np.int64)
        cx = fx + 1
        cy = fy + 1
        dx = (cx - xs).astype(data.dtype)
        dy = (cy - ys).astype(data.dtype)

        def pt(xi, yi):
            inside = (xi >= 0) & (yi >= 0) & (xi < W) & (yi < H)
            v = data[b, np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
            return np.where(inside[:, None], v, 0).astype(data.dtype)

        o = (dx * dy)[:, None] * pt(fx, fy) + ((1 - dx) * (1 - dy))[:, None] * pt(cx, cy) \
            + (dx * (1 - dy))[:, None] * pt(fx, cy) + ((1 - dx) * dy)[:, None] * pt(cx, fy)
        out[b] = np.where(ok[:, None], o, 0)
    return _t(out)


contrib = types.SimpleNamespace(resampler=types.SimpleNamespace(resampler=_resampler))


# ---- custom-op library: numpy restatement of utils.cu ---------------------------------
def _equation_construction(jacobian, gradient, difference):
    """utils.cu:331-414: M=G^T G, MJ, J^T(MJ) -> column-sum over pixels; g=d^T G, gJ -> sum."""
    J = np.asarray(jacobian)
    G = np.asarray(gradient)
    d = np.asarray(difference)
    M = np.matmul(np.swapaxes(G, -1, -2), G)
    H = np.matmul(np.swapaxes(J, -1, -2), np.matmul(M, J))
    g = np.matmul(np.swapaxes(d, -1, -2), G)
    gJ = np.matmul(g, J)
    return _t(np.sum(H, axis=1)), _t(np.swapaxes(np.sum(gJ, axis=1), -1, -2))


def _equation_construction_grad(jacobian, gradient, difference, left_grad, right_grad):
    """utils.cu:613-690."""
    J = np.asarray(jacobian)
    G = np.asarray(gradient)
    d = np.asarray(difference)
    g0 = np.asarray(left_grad)[:, None]
    g1 = np.asarray(right_grad)[:, None]
    A = np.matmul(G, J)
    dd = np.matmul(A, g1)
    dA = np.float32(2.0) * np.matmul(A, g0) + np.matmul(d, np.swapaxes(g1, -1, -2))
    return _t(np.matmul(np.swapaxes(G, -1, -2), dA)), _t(np.matmul(dA, np.swapaxes(J, -1, -2))), _t(dd)


def _jacobian_construction(*a, **k):
    raise NotImplementedError("utils.cu defines no JacobianConstruction op (SURVEY 2.3)")


def load_op_library(path):
    return types.SimpleNamespace(equation_construction=_equation_construction,
                                 equation_construction_grad=_equation_construction_grad,
                                 jacobian_construction=_jacobian_construction)
