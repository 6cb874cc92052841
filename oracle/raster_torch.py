"""Differentiable PyTorch restatement of the Gaussian rasterizer -- the GRADIENT oracle.

TEST INFRASTRUCTURE ONLY (tests/ may import it; the product never does).  PARITY UNPINNED: see
oracle/raster_oracle.c.  The forward below is written independently of the C oracle, in plain torch
ops, so that torch.autograd differentiates it; the C oracle's analytic backward and the HIP backward
are then checked against these autograd gradients.

Three places where the published algorithm's backward is NOT the mathematical derivative of its
forward are restated here with straight-through / detach so that autograd reproduces the published
behaviour:
  * alpha = min(0.99, o*G): the gradient flows as if the min were absent;
  * the frustum clamp of the camera-space point used for the EWA Jacobian: a clamped coordinate is
    a constant (its dependence on z is dropped);
  * the early-out tests (alpha < 1/255, T < 1e-4, power > 0) are masks, not differentiated.
One regulariser of the published backward is absent here: it divides by (det^2 + 1e-7) where the true
derivative divides by det^2 (relative effect <= 1.3e-5 since det >= 0.09).

Call-site anchors: /root/reference/gaussian_renderer/__init__.py:38-53,90-98 and
/root/reference/gaussian_renderer/render_helper.py:15-28,61-63.  Covariance / SH formulas follow
/root/reference/utils/general_utils.py:110-156 and /root/reference/utils/sh_utils.py:57-112.
"""
import numpy as np
import torch

TILE = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh_rgb(deg, sh, d):
    """sh [P,M,3], d [P,3] unit directions -> [P,3] (before +0.5 / clamp)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    v = C0 * sh[:, 0]
    if deg > 0:
        v = v - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        v = (v + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        v = (v + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
             + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
             + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return v


def cov3d_from_scale_rot(s, q, mod):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (mod * s)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def rasterize_torch(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height, image_width, tanfovx,
                    tanfovy, means2D=None, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, scale_modifier=1.0, sh_degree=0):
    """Returns (color[3,H,W], radii[P] int32, depth[1,H,W], alpha[1,H,W], aux dict)."""
    dt = means3D.dtype
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    V = viewmatrix.to(dt).reshape(4, 4)     # row-vector convention: p_view = [p,1] @ V
    Pm = projmatrix.to(dt).reshape(4, 4)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], dim=1)
    t = (ph @ V)[:, :3]
    hom = ph @ Pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc_x, ndc_y = hom[:, 0] * pw, hom[:, 1] * pw
    in_front = t[:, 2] > 0.2
    tz = torch.where(in_front, t[:, 2], torch.ones_like(t[:, 2]))     # avoid NaN for culled points

    if cov3D_precomp is None:
        c6 = cov3d_from_scale_rot(scales, rotations, scale_modifier)
    else:
        c6 = cov3D_precomp
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz).detach(), t[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).reshape(P, 2, 3)
    Rw = V[:3, :3].t()                       # rotation part, R[j,k] = V[k,j]
    Mx = J @ Rw                              # [P,2,3]
    Sig = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                      dim=1).reshape(P, 3, 3)
    cov2 = Mx @ Sig @ Mx.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_ok = det != 0
    det_s = torch.where(det_ok, det, torch.ones_like(det))
    conA, conB, conC = c / det_s, -b / det_s, a / det_s
    mid = 0.5 * (a + c)
    disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    lam = torch.maximum(mid + disc, mid - disc)
    rad = torch.ceil(3.0 * torch.sqrt(lam.clamp(min=0))).detach()
    px = ((ndc_x + 1.0) * W - 1.0) * 0.5
    py = ((ndc_y + 1.0) * H - 1.0) * 0.5
    if means2D is not None:                  # dummy leaf that harvests the NDC-scaled screen-space gradient
        px = px + means2D[:, 0] * (0.5 * W)
        py = py + means2D[:, 1] * (0.5 * H)

    pxd, pyd = px.detach(), py.detach()
    rx0 = torch.clamp(((pxd - rad) / TILE).to(torch.int64), 0, gx)
    ry0 = torch.clamp(((pyd - rad) / TILE).to(torch.int64), 0, gy)
    rx1 = torch.clamp(((pxd + rad + (TILE - 1)) / TILE).to(torch.int64), 0, gx)
    ry1 = torch.clamp(((pyd + rad + (TILE - 1)) / TILE).to(torch.int64), 0, gy)
    ntile = (rx1 - rx0) * (ry1 - ry0)
    vis = in_front & det_ok & (ntile > 0)
    radii = torch.where(vis, rad, torch.zeros_like(rad)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh_rgb(sh_degree, shs, d) + 0.5, 0.0)
    depth = t[:, 2]
    opac = opacities.reshape(-1)

    # ---- binning (not differentiated): stable order by (tile, float32 depth bits, index) ----
    vis_idx = torch.nonzero(vis).reshape(-1).numpy()
    dbits = depth.detach().to(torch.float32).numpy().view(np.uint32).astype(np.uint64)
    keys, vals = [], []
    rx0n, ry0n, rx1n, ry1n = rx0.numpy(), ry0.numpy(), rx1.numpy(), ry1.numpy()
    for i in vis_idx:
        ys, xs = np.meshgrid(np.arange(ry0n[i], ry1n[i]), np.arange(rx0n[i], rx1n[i]), indexing="ij")
        tid = (ys * gx + xs).reshape(-1).astype(np.uint64)
        keys.append((tid << np.uint64(32)) | dbits[i])
        vals.append(np.full(tid.shape, i, np.int64))
    if keys:
        keys = np.concatenate(keys); vals = np.concatenate(vals)
        order = np.argsort(keys, kind="stable")
        keys, vals = keys[order], vals[order]
    else:
        keys = np.zeros(0, np.uint64); vals = np.zeros(0, np.int64)
    tile_of = (keys >> np.uint64(32)).astype(np.int64)

    bgt = bg.to(dt)
    color = torch.zeros(3, H, W, dtype=dt) + bgt[:, None, None]
    out_depth = torch.zeros(1, H, W, dtype=dt)
    out_alpha = torch.zeros(1, H, W, dtype=dt)
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    starts = np.searchsorted(tile_of, np.arange(gx * gy), side="left")
    ends = np.searchsorted(tile_of, np.arange(gx * gy), side="right")
    for tile in range(gx * gy):
        s, e = int(starts[tile]), int(ends[tile])
        if e == s:
            continue
        ids = torch.from_numpy(vals[s:e])
        x0, y0 = (tile % gx) * TILE, (tile // gx) * TILE
        x1, y1 = min(x0 + TILE, W), min(y0 + TILE, H)
        yy, xx = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
        fxp, fyp = xx.reshape(1, -1), yy.reshape(1, -1)
        dx = px[ids][:, None] - fxp
        dy = py[ids][:, None] - fyp
        power = -0.5 * (conA[ids][:, None] * dx * dx + conC[ids][:, None] * dy * dy) - conB[ids][:, None] * dx * dy
        G = torch.exp(torch.clamp(power, max=0.0))
        raw = opac[ids][:, None] * G
        alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()      # straight-through min
        valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - a_eff
        T_incl = torch.cumprod(one_m, dim=0)
        T_excl = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], dim=0)
        stop = valid & ((T_excl * one_m).detach() < 1e-4)
        done = torch.cummax(stop.to(torch.int8), dim=0).values.bool()
        contrib = valid & ~done
        w = torch.where(contrib, a_eff * T_excl, torch.zeros_like(a_eff))
        Tf = torch.prod(torch.where(contrib, one_m, torch.ones_like(one_m)), dim=0)
        cc = (w[:, None, :] * rgb[ids][:, :, None]).sum(0)              # [3, npx]
        dd = (w * depth[ids][:, None]).sum(0)
        aa = w.sum(0)
        hh, ww = y1 - y0, x1 - x0
        color[:, y0:y1, x0:x1] = cc.reshape(3, hh, ww) + Tf.reshape(1, hh, ww) * bgt[:, None, None]
        out_depth[0, y0:y1, x0:x1] = dd.reshape(hh, ww)
        out_alpha[0, y0:y1, x0:x1] = aa.reshape(hh, ww)
        final_T[y0:y1, x0:x1] = Tf.detach().reshape(hh, ww)
        pos = torch.arange(1, e - s + 1)[:, None] * contrib.to(torch.int64)
        n_contrib[y0:y1, x0:x1] = pos.max(dim=0).values.reshape(hh, ww)
    aux = dict(keys=keys, point_list=vals, final_T=final_T, n_contrib=n_contrib, xy=torch.stack([pxd, pyd], 1),
               conic=torch.stack([conA, conB, conC], 1).detach(), rgb=rgb.detach(), depth=depth.detach(), vis=vis)
    return color, radii, out_depth, out_alpha, aux
