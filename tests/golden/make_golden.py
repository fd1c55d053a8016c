#!/usr/bin/env python3
"""Generate golden vectors by executing the REFERENCE's own Python.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py

`/root/reference/legacy/ba.py`, `legacy/utils_python.py` and `bundlenet.py` are imported
AS THEY LIE (no copy is made) with `oracle/tf1_shim` standing in for TensorFlow-1.x (not
installable here).  The only source adjustments, applied in memory at import time, are the
two py2->py3 syntax fixes without which `bundlenet.py` cannot be parsed / sliced:
   * `print "lambda_shape",...`          (bundlenet.py:250)   -> removed
   * `nbatch/2` used as a slice index    (bundlenet.py:321,386) -> `nbatch//2`
Outputs (a few KB each) are written next to this script as golden_*.npz; the inputs are
regenerated from seeds by `cases.py`.  The .npz files are committed; this script documents
how they were made.
"""
import importlib.util
import os
import re
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle", "tf1_shim"))
sys.path.insert(0, os.path.join(REF, "legacy"))
sys.path.insert(0, HERE)

import tensorflow as tf  # noqa: E402  (the shim)
import cases  # noqa: E402


def load_reference_module(name, path, patches=()):
    src = open(path).read()
    for pat, rep in patches:
        src, n = re.subn(pat, rep, src)
        assert n > 0, (pat, path)
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    cwd = os.getcwd()
    os.chdir(os.path.dirname(path))
    try:
        exec(compile(src, path, "exec"), mod.__dict__)
    finally:
        os.chdir(cwd)
    return mod


def load_all():
    sys.modules["feat"] = types.ModuleType("feat")          # CNN (out of scope); ba.py only imports it
    ba = load_reference_module("ba", os.path.join(REF, "legacy", "ba.py"))
    bn = load_reference_module("bundlenet", os.path.join(REF, "bundlenet.py"), patches=[
        (r'(?m)^\s*print "lambda_shape".*$', "            pass"),
        (r"nbatch/2", "nbatch//2"),
    ])
    return ba, bn


def install_mlp(mlp):
    """Put the lambda-MLP weights where the reference's conv1d() will look them up
    (variable_scope(name)/name+'_filters', bundlenet.py:103-106)."""
    tf.reset_variables()
    for level, layers in mlp.items():
        for i, (w, b) in enumerate(layers):
            nm = "lambda_%s_%d" % (level, i + 1)
            tf.set_variable("%s/%s_filters" % (nm, nm), w[None])
            tf.set_variable("%s/%s_biases" % (nm, nm), b)


def A(x):
    return np.asarray(x)


def gen_legacy_ci2(ba):
    c = cases.case_legacy_ci2()
    trk = ba.Tracker.__new__(ba.Tracker)
    install_mlp(c["mlp"])
    N = c["points"].shape[1]
    intr = tf.convert(c["intr"])
    fx0, fy0 = tf.tile(intr[:, 0], [1, N]), tf.tile(intr[:, 1], [1, N])
    ox0, oy0 = tf.tile(intr[:, 2], [1, N]), tf.tile(intr[:, 3], [1, N])
    p = trk.computeCoordinates(tf.convert(c["points"]), fx0, fy0, ox0, oy0)
    s = np.float32(c["scale"])
    fx, fy, ox, oy = fx0 / s, fy0 / s, ox0 / s, oy0 / s
    conv2 = tf.concat([tf.convert(c["conv2_f"]), trk.grad_fixed(tf.convert(c["conv2_f"]))], axis=-1)
    R, T, uw, ut, ratio = trk.CameraIteration2(tf.convert(c["conv1"]), conv2, fx, fy, ox, oy, p,
                                               tf.convert(c["d"]), tf.convert(c["R"]), tf.convert(c["T"]),
                                               c["level"])
    # sub-function outputs at the same inputs
    import utils_python
    Rp = np.matmul(c["R"], A(p)) * np.transpose(c["d"], (0, 2, 1)) + c["T"]
    x, y, Z = Rp[:, 0] / Rp[:, 2], Rp[:, 1] / Rp[:, 2], Rp[:, 2]
    px, py = A(fx) * x + A(ox), A(fy) * y + A(oy)
    samp, mask = utils_python.interpolate2d(conv2, tf.convert(px), tf.convert(py))
    samp2 = utils_python.interpolate2d2(tf.convert(c["conv2_f"]), tf.convert(c["points"] / s))
    J = trk.CameraJacobianMatrix(tf.convert(x), tf.convert(y), tf.convert(Z), fx, fy)
    # fixed-iteration legacy step as well
    R1, T1, ratio1 = trk.CameraIteration(tf.convert(c["conv1"]), conv2, fx, fy, ox, oy, p, tf.convert(c["d"]),
                                         tf.convert(c["R"]), tf.convert(c["T"]))
    np.savez(os.path.join(HERE, "golden_legacy_ci2.npz"), R=A(R), T=A(T), uw=A(uw), ut=A(ut),
             ratio=A(ratio), p=A(p), conv2=A(conv2), px=px, py=py, samp=A(samp), mask=A(mask),
             samp2=A(samp2), J=A(J), R1=A(R1), T1=A(T1), ratio1=A(ratio1))
    print("legacy_ci2: uw=%g ut=%g ratio=%g" % (A(uw), A(ut), A(ratio)))


def gen_legacy_track(ba):
    c = cases.case_legacy_track()
    trk = ba.Tracker.__new__(ba.Tracker)
    install_mlp(c["mlp"])
    counts = {}
    orig = trk.CameraIteration2

    def counting(conv1, conv2, fx, fy, ox, oy, p, D, R, T, level):
        counts[level] = counts.get(level, 0) + 1
        return orig(conv1, conv2, fx, fy, ox, oy, p, D, R, T, level)

    trk.CameraIteration2 = counting
    ba.early_termination = True
    layers = [tf.convert(l) for l in c["layers"]]
    R, T, ratio = trk.trackTF(tf.convert(c["intr"]), layers, tf.convert(c["points"]), tf.convert(c["d"]),
                              tf.convert(c["R"]), tf.convert(c["T"]), c["iters"])
    its = np.array([counts.get(str(l), 0) for l in (1, 2, 3)], np.int32)
    ba.early_termination = False
    Rs, Ts, ratio_f = trk.trackTF(tf.convert(c["intr"]), layers, tf.convert(c["points"]), tf.convert(c["d"]),
                                  tf.convert(c["R"]), tf.convert(c["T"]), c["iters"])
    ba.early_termination = True
    np.savez(os.path.join(HERE, "golden_legacy_track.npz"), R=A(R), T=A(T), ratio=A(ratio), iters=its,
             Rs=np.stack([A(r) for r in Rs]), Ts=np.stack([A(t) for t in Ts]), ratio_fixed=A(ratio_f))
    print("legacy_track: iters per level", its, " |R-Rgt|=%.3e |T-Tgt|=%.3e" % (
        np.abs(A(R)[0] - c["R_gt"]).max(), np.abs(A(T)[0, :, 0] - c["T_gt"]).max()))


def gen_bundle_fns(ba, bn):
    c = cases.case_bundle_fns()
    net = bn.BundleNet()
    cv = tf.convert
    Jc = bn.CameraJacobianMatrix(cv(c["x"]), cv(c["y"]), cv(c["Z"]), cv(c["fx"]), cv(c["fy"]))
    jd = bn.DepthJacobianMatrix(cv(c["r"][0]), cv(c["r"][1]), cv(c["r"][2]), cv(c["x"]), cv(c["y"]), cv(c["Z"]),
                                cv(c["fx"]), cv(c["fy"]))
    w1 = c["w1"]
    rot1 = bn.AngleaAxisRotation(cv(w1[:, 0:1]), cv(w1[:, 1:2]), cv(w1[:, 2:3]))
    w2 = c["w2"]
    rot2 = bn.AngleaAxisRotation(cv(w2[:, 0:1]), cv(w2[:, 1:2]), cv(w2[:, 2:3]))
    V1 = bn.VMatrix(cv(w1[:, 0:1, None]), cv(w1[:, 1:2, None]), cv(w1[:, 2:3, None]))
    V2bug = bn.VMatrix(cv(w2[:, 0:1, None]), cv(w2[:, 1:2, None]), cv(w2[:, 2:3, None]))
    q = bn.rotation2quaternion(cv(c["Rm"]))
    g = net.grad_fixed(cv(c["img"]))
    B, N = c["x"].shape
    p = net.computeCoordinates(cv(c["pts"]), cv(c["fx"]), cv(c["fy"]), cv(np.full((B, N), 16.0, np.float32)),
                               cv(np.full((B, N), 10.0, np.float32)))
    trk = ba.Tracker.__new__(ba.Tracker)
    rotL = trk.AngleaAxisRotation(cv(w1[:, 0:1]), cv(w1[:, 1:2]), cv(w1[:, 2:3]))
    VL = trk.VMatrix(cv(w1[:, 0:1, None]), cv(w1[:, 1:2, None]), cv(w1[:, 2:3, None]))
    np.savez(os.path.join(HERE, "golden_bundle_fns.npz"), Jc=A(Jc), jd=A(jd), rot1=A(rot1), rot2=A(rot2),
             V1=A(V1), V2bug=A(V2bug), q=A(q), g=A(g), p=A(p), rotL=A(rotL), VL=A(VL))
    print("bundle_fns ok")


def gen_bundle_iter(bn):
    c = cases.case_bundle_iter()
    net = bn.BundleNet()
    install_mlp(c["mlp"])
    cv = tf.convert
    R, T = net.CameraIteration(cv(c["conv1"]), cv(c["conv2"]), cv(c["fx"]), cv(c["fy"]), cv(c["ox"]), cv(c["oy"]),
                               cv(c["p"]), cv(c["D"]), cv(c["R"]), cv(c["T"]), 1.0, c["level"])
    R2, T2, W2 = net.BundleIteration(cv(c["conv1"]), cv(c["conv2"]), cv(c["fx"]), cv(c["fy"]), cv(c["ox"]),
                                     cv(c["oy"]), cv(c["p"]), cv(c["D"]), cv(c["Bs"]), cv(c["R"]), cv(c["T"]),
                                     cv(c["W"]), 1000.0, c["level"])
    np.savez(os.path.join(HERE, "golden_bundle_iter.npz"), Rc=A(R), Tc=A(T), R=A(R2), T=A(T2), W=A(W2))
    print("bundle_iter ok  dW max %.3e" % np.abs(A(W2) - c["W"]).max())


def gen_resize(bn):
    c = cases.case_resize()
    net = bn.BundleNet()
    install_mlp(c["mlp"])
    cv = tf.convert
    layers = [cv(l) for l in c["layers"]]
    Rs, Ts = net.CameraResize(cv(c["intr"]), layers, cv(c["points"]), cv(c["depth"]))
    # BundleResize is seeded with CameraResize's result: from (I,0) the depth Jacobian is
    # identically ~0 and the undamped last coefficient (bundlenet.py:266) blows up.
    Rb, Tb, Db = net.BundleResize(cv(c["intr"]), layers, cv(c["points"]), cv(c["basis"]), cv(c["depth"]),
                                  init_rotation=Rs[-1], init_translation=Ts[-1])
    # depth maps are large; keep a strided sample
    np.savez(os.path.join(HERE, "golden_resize.npz"), Rs=np.stack([A(r) for r in Rs]),
             Ts=np.stack([A(t) for t in Ts]), Rb=np.stack([A(r) for r in Rb]), Tb=np.stack([A(t) for t in Tb]),
             Db=np.stack([A(d)[:, ::8, ::8, 0] for d in Db]))
    print("resize ok")


def gen_losses(bn):
    c = cases.case_losses()
    net = bn.BundleNet()
    cv = tf.convert
    lr = net.lossR(cv(c["predQ"]), cv(c["gtQ"]))
    lt = net.lossT(cv(c["predT"]), cv(c["gtT"]))          # the second definition (bundlenet.py:410-412) is the live one
    lf = net.lossF(cv(c["intr"]), cv(c["depth"]), cv(c["mask"]), cv(c["predR"]), cv(c["predT"]), cv(c["gtR"]),
                   cv(c["gtT"]))
    np.savez(os.path.join(HERE, "golden_losses.npz"), lossR=A(lr), lossT=A(lt), lossF=A(lf))
    print("losses: R %.6g  T %.6g  F %.6g" % (A(lr), A(lt), A(lf)))


if __name__ == "__main__":
    ba, bn = load_all()
    gen_legacy_ci2(ba)
    gen_legacy_track(ba)
    gen_bundle_fns(ba, bn)
    gen_bundle_iter(bn)
    gen_resize(bn)
    gen_losses(bn)
