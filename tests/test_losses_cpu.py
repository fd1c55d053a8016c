"""bundlenet.py:401-463 (lossR / lossT / lossF) and rotation2quaternion: the oracle restatement and the torch mirror
(pure tensor expressions, so they run on CPU tensors too) against the reference's OWN output
(tests/golden/golden_losses.npz, produced by executing /root/reference/bundlenet.py over oracle/tf1_shim)."""
import os

import numpy as np
import torch

import cases
from banet_amd.bundlenet import BundleNet
from oracle import banet_oracle as orc


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "golden_losses.npz"))


def test_oracle_losses_match_the_reference(golden_dir):
    c, g = cases.case_losses(), _golden(golden_dir)
    np.testing.assert_allclose(orc.loss_r(c["predQ"], c["gtQ"]), g["lossR"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(orc.loss_t(c["predT"], c["gtT"]), g["lossT"], rtol=1e-6)
    np.testing.assert_allclose(orc.loss_f(c["intr"], c["depth"], c["mask"], c["predR"], c["predT"], c["gtR"], c["gtT"]),
                               g["lossF"], rtol=2e-5)


def test_bundlenet_losses_match_the_reference(golden_dir):
    c, g = cases.case_losses(), _golden(golden_dir)
    t = {k: torch.from_numpy(v) for k, v in c.items()}
    net = BundleNet()
    net.fx = "left by a Resize call"
    np.testing.assert_allclose(float(net.lossR(t["predQ"], t["gtQ"])), g["lossR"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(float(net.lossT(t["predT"], t["gtT"])), g["lossT"], rtol=1e-6)
    lf = net.lossF(t["intr"], t["depth"], t["mask"], t["predR"], t["predT"], t["gtR"], t["gtT"])
    np.testing.assert_allclose(float(lf), g["lossF"], rtol=2e-5)
    assert net.fx == "left by a Resize call"        # lossF works on locals (bundlenet.py:441-445), it must not clobber self.fx


def test_losses_are_differentiable_and_vanish_at_the_ground_truth():
    c = cases.case_losses()
    t = {k: torch.from_numpy(v) for k, v in c.items()}
    net = BundleNet()
    assert abs(float(net.lossR(t["gtQ"], t["gtQ"]))) < 1e-6
    assert float(net.lossT(t["gtT"], t["gtT"])) == 0.0
    assert float(net.lossF(t["intr"], t["depth"], t["mask"], t["gtR"], t["gtT"], t["gtR"], t["gtT"])) == 0.0
    predT = t["predT"].clone().requires_grad_(True)
    predR = t["predR"].clone().requires_grad_(True)
    loss = net.lossF(t["intr"], t["depth"], t["mask"], predR, predT, t["gtR"], t["gtT"]) + net.lossT(predT, t["gtT"])
    loss.backward()
    assert torch.isfinite(predT.grad).all() and predT.grad.abs().sum() > 0
    assert torch.isfinite(predR.grad).all() and predR.grad.abs().sum() > 0
