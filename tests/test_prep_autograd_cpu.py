"""The differentiable per-level preparation the Resize drivers switch to when something upstream needs a gradient
(banet_amd/bundlenet.py: _grad_fixed_autograd, _resampler_autograd, _depth_output): pure torch, so checked on CPU against
the oracle's restatements of grad_fixed (bundlenet.py:92-100), tf.contrib.resampler and init_depth + basis.W (:397)."""
import numpy as np
import torch

from banet_amd import bundlenet as bn
from oracle import banet_oracle as orc


def test_grad_fixed_autograd_matches_oracle_and_backpropagates():
    rng = np.random.RandomState(0)
    img = rng.standard_normal((2, 7, 9, 5)).astype(np.float32)
    x = torch.from_numpy(img).requires_grad_(True)
    g = bn._grad_fixed_autograd(x)
    np.testing.assert_array_equal(g.detach().numpy(), orc.grad_fixed(img))        # same arithmetic: bit exact
    assert float(g[:, :, 0, :5].abs().max()) == 0.0 and float(g[:, 0, :, 5:].abs().max()) == 0.0   # REFLECT rim -> 0
    tm = bn._target_map(x)
    np.testing.assert_array_equal(tm.detach().numpy(), orc.target_map(img))
    (g * torch.from_numpy(rng.standard_normal(g.shape).astype(np.float32))).sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0


def test_resampler_autograd_matches_oracle_and_has_both_gradients():
    rng = np.random.RandomState(1)
    data = rng.standard_normal((2, 6, 8, 3)).astype(np.float32)
    warp = np.stack([rng.uniform(-1.5, 8.5, (2, 40)), rng.uniform(-1.5, 6.5, (2, 40))], -1).astype(np.float32)
    d, w = torch.from_numpy(data).requires_grad_(True), torch.from_numpy(warp).requires_grad_(True)
    out = bn._resample(d, w)                                                        # grad needed -> the torch expression
    np.testing.assert_allclose(out.detach().numpy(), orc.resampler(data, warp), rtol=1e-5, atol=1e-6)
    out.square().sum().backward()
    assert float(d.grad.abs().sum()) > 0 and float(w.grad.abs().sum()) > 0


def test_depth_output_autograd():
    rng = np.random.RandomState(2)
    init = torch.from_numpy(rng.uniform(1, 3, (2, 4, 5, 1)).astype(np.float32)).requires_grad_(True)
    basis = torch.from_numpy(rng.standard_normal((2, 4, 5, 3)).astype(np.float32)).requires_grad_(True)
    W = torch.from_numpy(rng.standard_normal((2, 3, 1)).astype(np.float32)).requires_grad_(True)
    out = bn._depth_output(init, basis, W)
    want = init.detach().numpy() + np.matmul(basis.detach().numpy().reshape(2, -1, 3), W.detach().numpy()).reshape(2, 4, 5, 1)
    np.testing.assert_allclose(out.detach().numpy(), want, rtol=1e-5, atol=1e-6)
    out.sum().backward()
    assert float(init.grad.min()) == 1.0 and float(basis.grad.abs().sum()) > 0 and float(W.grad.abs().sum()) > 0
