"""Hand-derived adjoint of the dense bundle assembly  --  TEST INFRASTRUCTURE ONLY.

The product's fused backward (banet_amd/csrc/adjoint.hip, banet_amd/dense_train.py) differentiates the
reference's BundleIteration (bundlenet.py:193-278) the way the reference's TF graph does (tf.gradients through
tf.contrib.resampler, grad_fixed, the Jacobians and the registered EquationConstructionGrad, bundlenet.py:79-82,
utils.cu:465-694), but per pixel and without materialising J.  This module states that adjoint in numpy
(float64 by default) so that it can be (a) validated against finite differences of the oracle's forward on the CPU and
(b) used to check the HIP kernels number by number on the GPU.

phi = <G, AtA> + <gb, Atb> + <gavg, avg>, with (AtA, Atb, avg) = the undamped normal equations and the mean |residual|
of oracle.bundle_iteration; `assembly_adjoint` returns d phi / d (src, tgt, D0, basis, R, T, W).

Per pixel, with S = (G + G^T)/2, b = basis row, J = [Jc | jd b^T], M = G2^T G2 (2x2), g = G2^T d:
    q = S_cd b, z = S_dd b, zeta = b.z, e = gb_d.b, t = Jc q + jd zeta
    dM = (Jc S_cc + jd q^T) Jc^T + t jd^T            dg = Jc gb_c + jd e
    dJc = 2 M (Jc S_cc + jd q^T) + g gb_c^T         djd = 2 M t + g e
    db  = 2 S_cd^T u + 2 s z + r gb_d   (u = Jc^T M jd, s = jd^T M jd, r = jd^T g: the forward's per-pixel records)
"""
import numpy as np

from . import banet_oracle as orc


def _grad_fixed_adjoint(dgx, dgy):
    """adjoint of orc.grad_fixed: (d/d gx, d/d gy) [B,H,W,C] -> d/d img.  gx[x] = 0.5 (img[x+1] - img[x-1]) for
    1 <= x <= W-2 and exactly 0 on the REFLECT rim (bundlenet.py:92-100)."""
    out = np.zeros_like(dgx)
    out[:, :, 2:, :] += 0.5 * dgx[:, :, 1:-1, :]
    out[:, :, :-2, :] -= 0.5 * dgx[:, :, 1:-1, :]
    out[:, 2:, :, :] += 0.5 * dgy[:, 1:-1, :, :]
    out[:, :-2, :, :] -= 0.5 * dgy[:, 1:-1, :, :]
    return out


def forward_lean(a, tgt, R, T, W, dtype=np.float64):
    """(AtA0, Atb, avg) of oracle.bundle_iteration from the dense level inputs `a` (oracle.dense.level_inputs) and the
    raw target map, in the per-pixel block form the kernels use; returns the intermediates the adjoint needs."""
    f = lambda v: np.asarray(v, dtype)
    conv1, p, D, Bs = f(a["conv1"]), f(a["p"]), f(a["D"])[..., 0], f(a["Bs"])
    fx, fy, ox, oy = f(a["fx"]), f(a["fy"]), f(a["ox"]), f(a["oy"])
    tgt, R, T, W = f(tgt), f(R), f(T), f(W)
    nb, N, C = conv1.shape
    H, Wd = tgt.shape[1], tgt.shape[2]
    K = Bs.shape[-1]
    Dn = D + np.matmul(Bs, W)[..., 0]
    rp = np.matmul(R, p)
    rx, ry, rz = rp[:, 0], rp[:, 1], rp[:, 2]
    X, Y, Z = rx * Dn + T[:, 0], ry * Dn + T[:, 1], rz * Dn + T[:, 2]
    with np.errstate(invalid="ignore", divide="ignore"):
        x, y = X / Z, Y / Z
    px, py = fx * x + ox, fy * y + oy
    mask = orc.bundlenet_mask(px, py, H, Wd).astype(bool) & np.isfinite(px) & np.isfinite(py)
    pxs, pys = np.where(mask, px, 0.0), np.where(mask, py, 0.0)
    x0, y0 = np.floor(pxs), np.floor(pys)
    ax, ay = pxs - x0, pys - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    map3 = orc.target_map(tgt)
    bi = np.arange(nb)[:, None]

    def tap(yi, xi):
        inside = (xi >= 0) & (yi >= 0) & (xi <= Wd - 1) & (yi <= H - 1)
        v = map3[bi, np.clip(yi, 0, H - 1), np.clip(xi, 0, Wd - 1)]
        return np.where(inside[..., None], v, 0.0)

    I00, I01, I10, I11 = tap(y0, x0), tap(y0, x0 + 1), tap(y0 + 1, x0), tap(y0 + 1, x0 + 1)
    w00, w01, w10, w11 = (1 - ax) * (1 - ay), ax * (1 - ay), (1 - ax) * ay, ax * ay
    S = I00 * w00[..., None] + I01 * w01[..., None] + I10 * w10[..., None] + I11 * w11[..., None]
    m = mask[..., None].astype(dtype)
    diff = (conv1 - S[..., 0:C]) * m
    gxm, gym = S[..., C:2 * C] * m, S[..., 2 * C:] * m
    avg = np.mean(np.abs(diff), axis=1, keepdims=True)
    M11, M12, M22 = (gxm * gxm).sum(-1), (gxm * gym).sum(-1), (gym * gym).sum(-1)
    g1, g2 = (gxm * diff).sum(-1), (gym * diff).sum(-1)
    xs, ys, Zs = np.where(mask, x, 0.0), np.where(mask, y, 0.0), np.where(mask, Z, 1.0)
    Jc = orc.camera_jacobian(xs, ys, Zs, fx, fy, -1) * m[..., None]            # [B,N,2,6]
    jd = orc.depth_jacobian(rx, ry, rz, xs, ys, Zs, fx, fy) * m               # [B,N,2]
    J = np.concatenate([Jc, jd[..., None] * Bs[:, :, None, :]], axis=-1)      # [B,N,2,P]
    Mm = np.stack([np.stack([M11, M12], -1), np.stack([M12, M22], -1)], -2)   # [B,N,2,2]
    gv = np.stack([g1, g2], -1)
    AtA = np.einsum("bnip,bnij,bnjq->bpq", J, Mm, J)
    Atb = np.einsum("bnip,bni->bp", J, gv)[..., None]
    return dict(AtA=AtA, Atb=Atb, avg=avg, mask=mask, x0=x0, y0=y0, ax=ax, ay=ay, I=(I00, I01, I10, I11),
                diff=diff, gxm=gxm, gym=gym, M=(M11, M12, M22), g=(g1, g2), Jc=Jc, jd=jd, x=xs, y=ys, Z=Zs,
                rx=rx, ry=ry, rz=rz, Dn=Dn, conv1=conv1, p=p, Bs=Bs, fx=fx, fy=fy, H=H, W=Wd, C=C, K=K, N=N)


def assembly_adjoint(a, tgt, R, T, W, G, gb, gavg, dtype=np.float64):
    """-> dict(dsrc [B,N,C], dtgt [B,H,W,C], dD0 [B,N], dbasis [B,N,K], dR [B,3,3], dT [B,3,1], dW [B,K,1],
    and the per-pixel intermediates the kernel tests look at)."""
    F = forward_lean(a, tgt, R, T, W, dtype)
    f = lambda v: np.asarray(v, dtype)
    G, gb, gavg = f(G), f(gb).reshape(G.shape[0], -1), f(gavg).reshape(G.shape[0], -1)
    W = f(W)
    nb, N, C, K, H, Wd = G.shape[0], F["N"], F["C"], F["K"], F["H"], F["W"]
    S = 0.5 * (G + np.swapaxes(G, 1, 2))
    Scc, Scd, Sdd = S[:, :6, :6], S[:, :6, 6:], S[:, 6:, 6:]
    gbc, gbd = gb[:, :6], gb[:, 6:]
    Bs, mask = F["Bs"], F["mask"]
    m = mask.astype(dtype)
    q = np.einsum("bik,bnk->bni", Scd, Bs)                 # [B,N,6]
    z = np.einsum("bkl,bnl->bnk", Sdd, Bs)                 # [B,N,K]
    zeta = (z * Bs).sum(-1)
    e = np.einsum("bk,bnk->bn", gbd, Bs)
    Jc0, Jc1 = F["Jc"][:, :, 0], F["Jc"][:, :, 1]          # [B,N,6]
    jd0, jd1 = F["jd"][..., 0], F["jd"][..., 1]
    M11, M12, M22 = F["M"]
    g1, g2 = F["g"]
    JS0 = np.einsum("bni,bij->bnj", Jc0, Scc) + jd0[..., None] * q
    JS1 = np.einsum("bni,bij->bnj", Jc1, Scc) + jd1[..., None] * q
    t0 = (Jc0 * q).sum(-1) + jd0 * zeta
    t1 = (Jc1 * q).sum(-1) + jd1 * zeta
    dM11 = (JS0 * Jc0).sum(-1) + t0 * jd0
    dM12 = (JS0 * Jc1).sum(-1) + t0 * jd1
    dM22 = (JS1 * Jc1).sum(-1) + t1 * jd1
    dg1 = np.einsum("bni,bi->bn", Jc0, gbc) + jd0 * e
    dg2 = np.einsum("bni,bi->bn", Jc1, gbc) + jd1 * e
    dJ0 = 2 * (M11[..., None] * JS0 + M12[..., None] * JS1) + g1[..., None] * gbc[:, None, :]
    dJ1 = 2 * (M12[..., None] * JS0 + M22[..., None] * JS1) + g2[..., None] * gbc[:, None, :]
    djd0 = 2 * (M11 * t0 + M12 * t1) + g1 * e
    djd1 = 2 * (M12 * t0 + M22 * t1) + g2 * e
    Mjd0, Mjd1 = M11 * jd0 + M12 * jd1, M12 * jd0 + M22 * jd1
    u = Jc0 * Mjd0[..., None] + Jc1 * Mjd1[..., None]
    s = jd0 * Mjd0 + jd1 * Mjd1
    r = jd0 * g1 + jd1 * g2
    dbasis = 2 * np.einsum("bik,bni->bnk", Scd, u) + 2 * s[..., None] * z + r[..., None] * gbd[:, None, :]
    # ---- C-wide part
    diff, gxm, gym = F["diff"], F["gxm"], F["gym"]
    dgx = 2 * (dM11[..., None] * gxm + dM12[..., None] * gym) + dg1[..., None] * diff
    dgy = 2 * (dM12[..., None] * gxm + dM22[..., None] * gym) + dg2[..., None] * diff
    ddiff = dg1[..., None] * gxm + dg2[..., None] * gym + np.sign(diff) * (gavg[:, None, :] / N)
    mm = m[..., None]
    dsrc = ddiff * mm
    dS = np.concatenate([-ddiff * mm, dgx * mm, dgy * mm], axis=-1)           # adjoint of the sampled [f|gx|gy] vector
    I00, I01, I10, I11 = F["I"]
    ax, ay, x0, y0 = F["ax"], F["ay"], F["x0"], F["y0"]
    dSdax = (1 - ay)[..., None] * (I01 - I00) + ay[..., None] * (I11 - I10)
    dSday = (1 - ax)[..., None] * (I10 - I00) + ax[..., None] * (I11 - I01)
    dpx = (dS * dSdax).sum(-1)
    dpy = (dS * dSday).sum(-1)
    dmap = np.zeros((nb, H, Wd, 3 * C), dtype)
    bi = np.broadcast_to(np.arange(nb)[:, None], x0.shape)
    for (oyy, oxx, w) in ((0, 0, (1 - ax) * (1 - ay)), (0, 1, ax * (1 - ay)), (1, 0, (1 - ax) * ay), (1, 1, ax * ay)):
        yi, xi = y0 + oyy, x0 + oxx
        ok = mask & (xi >= 0) & (yi >= 0) & (xi <= Wd - 1) & (yi <= H - 1)
        np.add.at(dmap, (bi[ok], yi[ok], xi[ok]), (dS * w[..., None])[ok])
    dtgt = dmap[..., :C] + _grad_fixed_adjoint(dmap[..., C:2 * C], dmap[..., 2 * C:])
    # ---- geometry
    x, y, Z, fx, fy = F["x"], F["y"], F["Z"], F["fx"], F["fy"]
    rx, ry, rz, Dn = F["rx"], F["ry"], F["rz"], F["Dn"]
    a_, c_ = dJ0, dJ1
    dx_ = fx * dpx + fx * (-y * a_[..., 0] + 2 * x * a_[..., 1] - a_[..., 5] / Z) + fy * (y * c_[..., 1] + c_[..., 2])
    dy_ = fy * dpy + fx * (-x * a_[..., 0] - a_[..., 2]) + fy * (-2 * y * c_[..., 0] + x * c_[..., 1] - c_[..., 5] / Z)
    dZ_ = fx * (-a_[..., 3] + x * a_[..., 5]) / (Z * Z) + fy * (-c_[..., 4] + y * c_[..., 5]) / (Z * Z)
    drx = fx * djd0 / Z
    dry = fy * djd1 / Z
    drz = -(fx * x * djd0 + fy * y * djd1) / Z
    dx_ = dx_ - fx * rz * djd0 / Z
    dy_ = dy_ - fy * rz * djd1 / Z
    dZ_ = dZ_ - (jd0 * djd0 + jd1 * djd1) / Z
    dX, dY = dx_ / Z, dy_ / Z
    dZt = dZ_ - (x * dx_ + y * dy_) / Z
    dX, dY, dZt, drx, dry, drz = (v * m for v in (dX, dY, dZt, drx, dry, drz))
    drx, dry, drz = drx + dX * Dn, dry + dY * Dn, drz + dZt * Dn
    dDn = dX * rx + dY * ry + dZt * rz
    dT = np.stack([dX.sum(1), dY.sum(1), dZt.sum(1)], -1)[..., None]
    dR = np.einsum("bin,bjn->bij", np.stack([drx, dry, drz], 1), F["p"])
    dW = np.einsum("bn,bnk->bk", dDn, Bs)[..., None]
    dbasis = dbasis * mm + dDn[..., None] * W[:, None, :, 0]
    return dict(dsrc=dsrc, dtgt=dtgt, dD0=dDn, dbasis=dbasis, dR=dR, dT=dT, dW=dW, dS=dS, dmap=dmap, dpx=dpx * m, dpy=dpy * m,
                q=q, zeta=zeta, e=e, z=z, fwd=F)
