"""torch-CPU restatement of the dense BundleIteration (bundlenet.py:193-278 on every pixel of a level)  --  TEST
INFRASTRUCTURE ONLY (the product path never imports it).

Two uses:
  * `dense_assemble(..., dtype=torch.float64)`: the float64 twin that lets the 640x480 / K=128 sizes of BASELINE.json be
    checked on the GPU box in seconds (tests/test_gpu_parity.py); validated against the numpy oracle on CPU at small
    sizes (tests/test_torch_ref_cpu.py);
  * `bundle_iteration(..., dtype=torch.float32)`: the same iteration in the reference's float32 with torch's intra-op
    thread pool = all host cores -- the "honest" CPU baseline of SURVEY.md 8(d) that bench.py times next to the numpy
    port (whose elementwise work is single-threaded).  It never materialises J [B,N,2,P] (the normal equations are
    formed from the per-pixel 2x2 M = G^T G like utils.cu's result, not like its 22 GB scratch), so it is an
    OPTIMISED port: the fair comparison for a GPU number.
Citations as in oracle/banet_oracle.py (same statements, torch instead of numpy).
"""
import torch


def grad_fixed(img):
    """bundlenet.py:92-100; img [B,H,W,C]"""
    H, W = img.shape[1], img.shape[2]
    p = torch.nn.functional.pad(img.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect").permute(0, 2, 3, 1)
    gx = 0.5 * (p[:, 1:H + 1, 2:W + 2, :] - p[:, 1:H + 1, 0:W, :])
    gy = 0.5 * (p[:, 2:H + 2, 1:W + 1, :] - p[:, 0:H, 1:W + 1, :])
    return gx, gy


def prepare_level(intr, scale, src, tgt, depth, basis, normalize_rays=True, dtype=torch.float32):
    """Everything about a level that does not change between iterations (the reference builds these once per level too:
    bundlenet.py:376-385): rays, level intrinsics, the [f|gx|gy] target map as rows [B,N,3C], flat source / basis."""
    B, H, W, C = tgt.shape
    N = H * W
    dev = tgt.device
    f = lambda x: x.to(dtype)  # noqa: E731
    src, tgt, depth, intr = f(src), f(tgt), f(depth).reshape(B, N), f(intr)
    vv, uu = torch.meshgrid(torch.arange(H, dtype=dtype, device=dev), torch.arange(W, dtype=dtype, device=dev), indexing="ij")
    fx0, fy0, ox0, oy0 = [intr[:, i:i + 1] for i in range(4)]
    u, v = (uu.reshape(1, N) * scale), (vv.reshape(1, N) * scale)
    p = torch.stack([(u - ox0) / fx0, (v - oy0) / fy0, torch.ones(B, N, dtype=dtype, device=dev)], dim=1)
    if normalize_rays:
        p = p / torch.sqrt(torch.clamp((p * p).sum(1, keepdim=True), min=1e-12))
    gxm, gym = grad_fixed(tgt)
    tmap = torch.cat([tgt, gxm, gym], dim=-1).reshape(B, N, 3 * C)
    K = 0 if basis is None else basis.shape[-1]
    return dict(B=B, H=H, W=W, C=C, N=N, K=K, p=p, fx=fx0 / scale, fy=fy0 / scale, ox=ox0 / scale, oy=oy0 / scale,
                tmap=tmap, src=src.reshape(B, N, C), depth=depth, basis=f(basis).reshape(B, N, K) if K > 0 else None)


def assemble_prepared(L, R, T, Wc, bundle, stats=None, mask_override=None):
    """-> AtA [B,P,P], Atb [B,P], absres [B,C], nvalid [B] at the pose (R, T, Wc); L = prepare_level(...).
    stats: optional dict; receives "borderline" [B] = pixels whose projection lies within 4e-6 x max(W, H) pixels of the
    in-image mask's boundary (bundlenet.py:155,231): at such a state float32 and float64 arithmetic can legitimately disagree
    on the pixel's mask bit, which changes every sum by one pixel's worth (the parity gates account for it); "mask" (appended per
    call) = this evaluation's own mask [B,N] as bool.
    mask_override [B,N] (0/1): use THIS mask instead of the one the projection decides -- the parity gate evaluates the float64
    statements with the mask bits the GPU kernel decided, so that both sides solve the same system even where a pixel sits on
    the image border (taps of a pixel forced in are clamped like any other: utils_python.py:96-99)."""
    B, H, W, C, N, K = L["B"], L["H"], L["W"], L["C"], L["N"], L["K"]
    dtype = L["tmap"].dtype
    p, fx, fy, ox, oy = L["p"], L["fx"], L["fy"], L["ox"], L["oy"]
    R, T = R.to(dtype).reshape(B, 3, 3), T.to(dtype).reshape(B, 3, 1)
    D = L["depth"]
    Bs = L["basis"]
    if K > 0:
        D = D + torch.matmul(Bs, Wc.to(dtype).reshape(B, K, 1))[..., 0]
    Rp = torch.matmul(R, p)
    X = Rp * D.unsqueeze(1) + T
    x, y, Z = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2], X[:, 2]
    px, py = fx * x + ox, fy * y + oy
    mask = ((px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)).to(dtype)
    if stats is not None:
        stats.setdefault("mask", []).append(mask > 0)
        stats.setdefault("pxy", []).append((px, py))
        e = 4e-6 * max(W, H)
        inx, iny = (px >= -e) & (px <= W - 1 + e), (py >= -e) & (py <= H - 1 + e)
        near = (((px.abs() < e) | ((px - (W - 1)).abs() < e)) & iny) | (((py.abs() < e) | ((py - (H - 1)).abs() < e)) & inx)
        stats["borderline"] = stats.get("borderline", 0) + near.sum(1)
    if mask_override is not None:
        forced = mask_override.to(device=px.device).reshape(B, N) > 0
        # a pixel forced in samples at the border it sits on (its float32 projection is inside by definition)
        px = torch.where(forced & (mask == 0), torch.minimum(torch.clamp(px, min=0.0), torch.full_like(px, W - 1)), px)
        py = torch.where(forced & (mask == 0), torch.minimum(torch.clamp(py, min=0.0), torch.full_like(py, H - 1)), py)
        mask = forced.to(dtype)
    pxs = torch.where(mask > 0, px, torch.zeros_like(px))
    pys = torch.where(mask > 0, py, torch.zeros_like(py))
    x0f, y0f = torch.floor(pxs), torch.floor(pys)
    dx, dy = pxs - x0f, pys - y0f
    x0, y0 = x0f.long(), y0f.long()
    x1, y1 = (x0 + 1).clamp(0, W - 1), (y0 + 1).clamp(0, H - 1)
    x0, y0 = x0.clamp(0, W - 1), y0.clamp(0, H - 1)
    w00, w01, w10, w11 = (1 - dx) * (1 - dy), dx * (1 - dy), (1 - dx) * dy, dx * dy
    # 4 row gathers of the 3C-wide target map per window (index_select copies whole rows, multi-threaded)
    samp = torch.empty(B, N, 3 * C, dtype=dtype, device=px.device)
    for b in range(B):
        tm = L["tmap"][b]
        acc = tm.index_select(0, y0[b] * W + x0[b]) * w00[b].unsqueeze(-1)
        acc.addcmul_(tm.index_select(0, y0[b] * W + x1[b]), w01[b].unsqueeze(-1))
        acc.addcmul_(tm.index_select(0, y1[b] * W + x0[b]), w10[b].unsqueeze(-1))
        acc.addcmul_(tm.index_select(0, y1[b] * W + x1[b]), w11[b].unsqueeze(-1))
        samp[b] = acc
    mk = mask.unsqueeze(-1)
    F2w, gx, gy = samp[..., :C], samp[..., C:2 * C] * mk, samp[..., 2 * C:] * mk
    d = (F2w - L["src"]) * mk                                # legacy sign
    zero = torch.zeros_like(x)
    iz = 1.0 / Z
    Jx = fx.unsqueeze(-1) * torch.stack([x * y, -1 - x * x, y, -iz, zero, x / Z], dim=-1)
    Jy = fy.unsqueeze(-1) * torch.stack([1 + y * y, -x * y, -x, zero, -iz, y / Z], dim=-1)
    Jx, Jy = Jx * mk, Jy * mk
    if bundle:                                               # bundlenet.py:60,234: J = [-Jc | jd b], d = F1 - F2w
        d = -d
        Jx, Jy = -Jx, -Jy
        if K > 0:
            jd0 = fx * ((Rp[:, 0] - Rp[:, 2] * x) / Z) * mask
            jd1 = fy * ((Rp[:, 1] - Rp[:, 2] * y) / Z) * mask
            Jx = torch.cat([Jx, jd0.unsqueeze(-1) * Bs], dim=-1)
            Jy = torch.cat([Jy, jd1.unsqueeze(-1) * Bs], dim=-1)
    Jx = torch.nan_to_num(Jx)
    Jy = torch.nan_to_num(Jy)
    m11, m12, m22 = (gx * gx).sum(-1), (gx * gy).sum(-1), (gy * gy).sum(-1)
    g1, g2 = (gx * d).sum(-1), (gy * d).sum(-1)
    Zx = m11.unsqueeze(-1) * Jx + m12.unsqueeze(-1) * Jy
    Zy = m12.unsqueeze(-1) * Jx + m22.unsqueeze(-1) * Jy
    AtA = torch.matmul(Zx.transpose(1, 2), Jx) + torch.matmul(Zy.transpose(1, 2), Jy)
    Atb = (Jx * g1.unsqueeze(-1) + Jy * g2.unsqueeze(-1)).sum(1)
    return AtA, Atb, d.abs().sum(1), mask.sum(1)


def dense_assemble(intr, scale, src, tgt, depth, basis, R, T, Wc, bundle, normalize_rays, dtype=torch.float64):
    """-> AtA [B,P,P], Atb [B,P], absres [B,C], nvalid [B]   (P = 6 + K)."""
    L = prepare_level(intr, scale, src, tgt, depth, basis if bundle else None, normalize_rays, dtype)
    return assemble_prepared(L, R, T, Wc, bundle)


_SELU_A, _SELU_S = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946


def lambda_mlp(avg, weights):
    """bundlenet.py:168-172,245-248: five k=1 conv layers, selu x4 then tanh.  avg [B,1,C]"""
    h = avg
    for i, (w, b) in enumerate(weights):
        z = torch.matmul(h, torch.as_tensor(w, dtype=h.dtype)) + torch.as_tensor(b, dtype=h.dtype)
        h = torch.tanh(z) if i == 4 else _SELU_S * torch.where(z > 0, z, _SELU_A * (torch.exp(z) - 1))
    return h


def _rodrigues(w):
    """bundlenet.py:17-37 (theta clamped at 1e-6) and VMatrix :39-46 per item; w [B,3]"""
    th = torch.sqrt((w * w).sum(-1))
    thc = torch.clamp(th, min=1e-6)
    k = w / thc[:, None]
    c, s = torch.cos(thc)[:, None, None], torch.sin(thc)[:, None, None]
    z = torch.zeros_like(th)
    Kn = torch.stack([z, -k[:, 2], k[:, 1], k[:, 2], z, -k[:, 0], -k[:, 1], k[:, 0], z], -1).reshape(-1, 3, 3)
    eye = torch.eye(3, dtype=w.dtype, device=w.device)[None]
    Rw = c * eye + (1 - c) * k[:, :, None] * k[:, None, :] + s * Kn
    Kw = torch.stack([z, -w[:, 2], w[:, 1], w[:, 2], z, -w[:, 0], -w[:, 1], w[:, 0], z], -1).reshape(-1, 3, 3)
    a = ((1 - torch.cos(th)) / (th * th))[:, None, None]
    bq = ((th - torch.sin(th)) / (th * th * th))[:, None, None]
    V = eye + a * Kw + bq * torch.matmul(Kw, Kw)
    return Rw, V


def bundle_iteration(intr, scale, src, tgt, depth, basis, R, T, Wc, mlp, l2_base=1000.0, dtype=torch.float32, level=None):
    """One dense BundleIteration -> (R', T', W', dict(lam, solution)).  Shapes as dense_assemble; Wc [B,K,1].
    level: a prepare_level(...) result to reuse across the iterations of a level (then intr .. basis are ignored)."""
    L = level if level is not None else prepare_level(intr, scale, src, tgt, depth, basis, True, dtype)
    B, N = L["B"], L["N"]
    dtype = L["tmap"].dtype
    AtA, Atb, absres, _nv = assemble_prepared(L, R, T, Wc, True)
    avg = (absres / N).unsqueeze(1)                                              # :243
    y = lambda_mlp(avg, mlp)
    lam = torch.sqrt((avg * avg).sum(-1, keepdim=True)) ** (2.0 + y)             # :249
    lam = l2_base * lam                                                          # :252-253
    diag = torch.diagonal(AtA, dim1=1, dim2=2)
    damp = torch.cat([(diag[:, :-1] + 1e-5) * lam[:, 0], torch.zeros(B, 1, dtype=dtype)], dim=-1)   # :264-266
    sol = torch.linalg.solve(AtA + torch.diag_embed(damp), Atb.unsqueeze(-1))    # :267
    Rw, V = _rodrigues(sol[:, 0:3, 0])
    R = R.to(dtype).reshape(B, 3, 3)
    T = T.to(dtype).reshape(B, 3, 1)
    Rn = torch.matmul(Rw, R)
    Tn = torch.matmul(V, sol[:, 3:6]) + torch.matmul(Rw, T)
    Wn = Wc.to(dtype).reshape(B, -1, 1) + sol[:, 6:]
    return Rn, Tn, Wn, dict(lam=lam.reshape(-1), solution=sol)


# --------------------------------------------------------------------------------------
# multi-frame windows (restated in banet_oracle.bundle_window_iteration)
# --------------------------------------------------------------------------------------
def window_assemble(intr, scale, src, tgts, depth, basis, Rs, Ts, Wc, normalize_rays=True, dtype=torch.float64, stats=None,
                    mask_override=None):
    """Normal equations of one multi-frame window at the poses (Rs, Ts) and depth coefficients Wc, composed from the
    per-pair assemblies exactly as banet_oracle.bundle_window_iteration stacks its rows: parameter order
    [pose_1 .. pose_pairs, depth]; pose blocks on the diagonal, each pair's pose/depth cross block, the depth block and
    Atb_depth summed over the pairs (block-arrowhead).  tgts [B,pairs,H,W,C]; Rs [B,pairs,3,3]; Ts [B,pairs,3,1].
    -> AtA [B,P,P], Atb [B,P], absres [B,C] (sum over pairs and pixels), nvalid [B,pairs].  One pair at a time, so the
    full BASELINE sizes (640x480 x 4 pairs, 1280x960 x 7 pairs with K = 256) fit in memory in float64."""
    B, pairs = tgts.shape[0], tgts.shape[1]
    K = basis.shape[-1]
    C = src.shape[-1]
    P = 6 * pairs + K
    dev = src.device
    AtA = torch.zeros(B, P, P, dtype=dtype, device=dev)
    Atb = torch.zeros(B, P, dtype=dtype, device=dev)
    absres = torch.zeros(B, C, dtype=dtype, device=dev)
    nvalid = torch.zeros(B, pairs, dtype=dtype, device=dev)
    o = 6 * pairs
    for i in range(pairs):
        L = prepare_level(intr, scale, src, tgts[:, i], depth, basis, normalize_rays, dtype)
        A, b, ab, nv = assemble_prepared(L, Rs[:, i], Ts[:, i], Wc, True, stats,
                                         None if mask_override is None else mask_override[:, i])
        del L
        AtA[:, 6 * i:6 * i + 6, 6 * i:6 * i + 6] = A[:, :6, :6]
        AtA[:, 6 * i:6 * i + 6, o:] = A[:, :6, 6:]
        AtA[:, o:, 6 * i:6 * i + 6] = A[:, 6:, :6]
        AtA[:, o:, o:] += A[:, 6:, 6:]
        Atb[:, 6 * i:6 * i + 6] = b[:, :6]
        Atb[:, o:] += b[:, 6:]
        absres += ab
        nvalid[:, i] = nv
    return AtA, Atb, absres, nvalid


def window_iteration(intr, scale, src, tgts, depth, basis, Rs, Ts, Wc, mlp, l2_base=1000.0, dtype=torch.float64, mask_override=None):
    """One multi-frame BundleIteration (banet_oracle.bundle_window_iteration: lambda from the residual averaged over all
    pairs, last coefficient undamped, LU solve, per-frame SE(3) update) -> (Rs', Ts', W', dict(lam, solution, AtA, Atb))."""
    B, pairs = tgts.shape[0], tgts.shape[1]
    N = src.shape[1] * src.shape[2]
    stats = {}
    AtA, Atb, absres, nv = window_assemble(intr, scale, src, tgts, depth, basis, Rs, Ts, Wc, True, dtype, stats, mask_override)
    avg = (absres / (N * pairs)).unsqueeze(1)
    mlp = [(torch.as_tensor(w).cpu().numpy() if not hasattr(w, "numpy") else w.cpu().numpy(),
            torch.as_tensor(b).cpu().numpy() if not hasattr(b, "numpy") else b.cpu().numpy()) for w, b in mlp]
    y = lambda_mlp(avg.cpu(), mlp).to(avg.device)
    lam = l2_base * torch.sqrt((avg * avg).sum(-1, keepdim=True)) ** (2.0 + y)
    diag = torch.diagonal(AtA, dim1=1, dim2=2)
    damp = torch.cat([(diag[:, :-1] + 1e-5) * lam[:, 0], torch.zeros(B, 1, dtype=dtype, device=AtA.device)], dim=-1)
    sol = torch.linalg.solve(AtA + torch.diag_embed(damp), Atb.unsqueeze(-1))
    Rn, Tn = [], []
    for i in range(pairs):
        Rw, V = _rodrigues(sol[:, 6 * i:6 * i + 3, 0].cpu())
        Rw, V = Rw.to(AtA.device), V.to(AtA.device)
        Rn.append(torch.matmul(Rw, Rs[:, i].to(dtype)))
        Tn.append(torch.matmul(V, sol[:, 6 * i + 3:6 * i + 6]) + torch.matmul(Rw, Ts[:, i].to(dtype).reshape(B, 3, 1)))
    Wn = Wc.to(dtype).reshape(B, -1, 1) + sol[:, 6 * pairs:]
    return torch.stack(Rn, 1), torch.stack(Tn, 1), Wn, dict(lam=lam.reshape(-1), solution=sol[..., 0], AtA=AtA, Atb=Atb,
                                                             nvalid=nv.sum(1), borderline=stats["borderline"],
                                                             mask=torch.stack(stats["mask"], 1),
                                                             px=torch.stack([q[0] for q in stats["pxy"]], 1),
                                                             py=torch.stack([q[1] for q in stats["pxy"]], 1))


# --------------------------------------------------------------------------------------
# the sparse-point iteration as differentiable float64 torch statements (test yardstick for the training graphs' gradients)
# --------------------------------------------------------------------------------------
def _bilinear_sparse(conv2, px, py):
    """tf.contrib.resampler on the [f|gx|gy] map (bundlenet.py:227 / oracle.banet_oracle.resampler): four taps, a tap outside the map
    contributes zero; differentiable in the map and in (px, py).  conv2 [B,H,W,C3], px / py [B,N] -> [B,N,C3]"""
    B, H, W, C3 = conv2.shape
    x0, y0 = torch.floor(px), torch.floor(py)
    ax, ay = (px - x0).unsqueeze(-1), (py - y0).unsqueeze(-1)
    x0, y0 = x0.long(), y0.long()
    bi = torch.arange(B, device=conv2.device)[:, None]

    def tap(yi, xi):
        inside = ((xi >= 0) & (yi >= 0) & (xi <= W - 1) & (yi <= H - 1)).unsqueeze(-1)
        v = conv2[bi, yi.clamp(0, H - 1), xi.clamp(0, W - 1)]
        return torch.where(inside, v, torch.zeros_like(v))
    return (tap(y0, x0) * (1 - ax) * (1 - ay) + tap(y0, x0 + 1) * ax * (1 - ay) + tap(y0 + 1, x0) * (1 - ax) * ay +
            tap(y0 + 1, x0 + 1) * ax * ay)


def sparse_iteration(conv1, conv2, D, Bs, R, T, Wc, lw, bundle, l2, fx, fy, ox, oy, p):
    """bundlenet.py:122-278 on N sampled points -- BundleIteration (bundle = True) or the pose-only CameraIteration -- every statement
    a differentiable torch expression in the dtype of its inputs (float64 in the tests): the twin of banet_oracle.bundle_iteration /
    camera_iteration (pinned to it in tests/test_torch_ref_cpu.py), written with this module's own helpers only.
    conv1 [B,N,C], conv2 [B,H,W,3C] = [f|gx|gy], D [B,N,1], Bs [B,N,K] or None, R [B,3,3], T [B,3,1], Wc [B,K,1] or None, lw five
    (filters, biases) pairs, p [B,3,N], fx .. oy [B,N].  -> (R', T', W' or None)"""
    C, N = conv1.shape[-1], conv1.shape[1]
    dt = conv1.dtype
    fx8, fy8, ox8, oy8, p8 = fx.to(dt), fy.to(dt), ox.to(dt), oy.to(dt), p.to(dt)
    Dd = D + torch.matmul(Bs, Wc) if bundle else D                                     # :206
    Rp = torch.matmul(R, p8)
    rx, ry, rz = Rp[:, 0], Rp[:, 1], Rp[:, 2]
    RPT = Rp * Dd.transpose(1, 2) + T
    X, Y, Z = RPT[:, 0], RPT[:, 1], RPT[:, 2]
    x, y = X / Z, Y / Z
    px, py = fx8 * x + ox8, fy8 * y + oy8
    samp = _bilinear_sparse(conv2, px, py)
    Hh, Ww = conv2.shape[1], conv2.shape[2]
    m = (~((px < 0) | (px > float(Ww - 1)) | (py < 0) | (py > float(Hh - 1)))).to(dt)   # :231-233
    d = (conv1 - samp[..., :C]) * m[..., None]
    gx, gy = samp[..., C:2 * C] * m[..., None], samp[..., 2 * C:] * m[..., None]
    M11, M12, M22, g1, g2 = (gx * gx).sum(-1), (gx * gy).sum(-1), (gy * gy).sum(-1), (gx * d).sum(-1), (gy * d).sum(-1)
    avg = (d.abs().sum(dim=1) / float(N)).unsqueeze(1)
    y_ = lambda_mlp(avg, lw)
    lam = torch.sqrt((avg * avg).sum(-1, keepdim=True)) ** (2.0 + y_)
    if bundle:
        lam = l2 * lam
    # bundlenet.py:49-61 (CameraJacobianMatrix, with its leading minus), one row per residual component
    iz = 1.0 / Z
    zero = torch.zeros_like(x)
    J0 = -fx8.unsqueeze(-1) * torch.stack([x * y, -1.0 - x * x, y, -iz, zero, x * iz], dim=-1)
    J1 = -fy8.unsqueeze(-1) * torch.stack([1.0 + y * y, -(x * y), -x, zero, -iz, y * iz], dim=-1)
    MJ0 = M11.unsqueeze(-1) * J0 + M12.unsqueeze(-1) * J1
    MJ1 = M12.unsqueeze(-1) * J0 + M22.unsqueeze(-1) * J1
    Hcc = torch.matmul(J0.transpose(1, 2), MJ0) + torch.matmul(J1.transpose(1, 2), MJ1)
    bc = (J0 * g1.unsqueeze(-1) + J1 * g2.unsqueeze(-1)).sum(dim=1)
    nb = conv1.shape[0]
    if bundle:
        jd0 = fx8 * ((rx - rz * x) * iz)                                              # :63-74 (DepthJacobianMatrix)
        jd1 = fy8 * ((ry - rz * y) * iz)
        u = MJ0 * jd0.unsqueeze(-1) + MJ1 * jd1.unsqueeze(-1)
        s_ = M11 * jd0 ** 2 + 2.0 * M12 * jd0 * jd1 + M22 * jd1 ** 2
        r = jd0 * g1 + jd1 * g2
        Hcd = torch.matmul(u.transpose(1, 2), Bs)
        Hdd = torch.matmul(Bs.transpose(1, 2), Bs * s_.unsqueeze(-1))
        bd = torch.matmul(Bs.transpose(1, 2), r.unsqueeze(-1)).squeeze(-1)
        AtA = torch.cat([torch.cat([Hcc, Hcd], dim=2), torch.cat([Hcd.transpose(1, 2), Hdd], dim=2)], dim=1)
        Atb = torch.cat([bc, bd], dim=1).unsqueeze(-1)
        diag = torch.diagonal(AtA, dim1=1, dim2=2)
        damp = torch.cat([diag[:, :-1] + 1e-5, torch.zeros(nb, 1, device=diag.device, dtype=diag.dtype)], dim=-1)     # :264-266
    else:
        AtA, Atb = Hcc, bc.unsqueeze(-1)
        damp = torch.diagonal(AtA, dim1=1, dim2=2) + 1e-5                             # :181-182
    sol = torch.linalg.solve(AtA + torch.diag_embed(damp * lam.squeeze(-1)), Atb)
    Rw, V = _rodrigues(sol[:, 0:3, 0])
    return torch.matmul(Rw, R), torch.matmul(V, sol[:, 3:6]) + torch.matmul(Rw, T), (Wc + sol[:, 6:]) if bundle else None
