"""Shared input builders and comparison helpers for the tests (CPU tensors; tests move them as needed)."""
import math

import numpy as np
import torch

from egogaussian_amd.scene_synth import make_scene, make_camera


def make_inputs(N, H, W, seed=0, sh_degree=0, mode="sh_cov", frame=3, scale_mul=1.0, bg=(0.1, 0.2, 0.3), dtype=torch.float32,
                opacity_shift=0.0):
    """Activated rasterizer inputs for S(N,H,W,seed).
    mode: 'sh_cov'  -> shs + cov3D_precomp   (training call, /root/reference/gaussian_renderer/__init__.py:90-98)
          'col_sr'  -> colors_precomp + scales/rotations (label call, render_helper.py:61-63)
          'sh_sr'   -> shs + scales/rotations
          'col_cov' -> colors_precomp + cov3D_precomp"""
    sc = make_scene(N, H, W, seed, sh_degree=sh_degree)
    cam = make_camera(frame, H, W)
    t = lambda a: torch.tensor(a, dtype=dtype)
    scales = torch.exp(t(sc["log_scale"])) * scale_mul
    quat = t(sc["quat"])
    d = dict(means3D=t(sc["xyz"]), opacities=torch.sigmoid(t(sc["opacity_logit"]) + opacity_shift),
             viewmatrix=cam.world_view_transform.to(dtype), projmatrix=cam.full_proj_transform.to(dtype),
             campos=cam.camera_center.to(dtype), bg=torch.tensor(bg, dtype=dtype), image_height=H, image_width=W,
             tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), sh_degree=sh_degree, scale_modifier=1.0)
    if mode in ("sh_cov", "sh_sr"):
        d["shs"] = t(sc["features"])
    else:
        rng = np.random.default_rng(seed + 77)
        d["colors_precomp"] = t(rng.uniform(0, 1, (N, 3)).astype(np.float32))
    if mode in ("sh_cov", "col_cov"):
        from egogaussian_amd.covariance import covariance_from_scaling_rotation
        d["cov3D_precomp"] = covariance_from_scaling_rotation(scales, 1.0, quat).contiguous()
    else:
        d["scales"], d["rotations"] = scales, quat
    return d


def seeded_grads(H, W, seed=5, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(3, H, W, generator=g, dtype=dtype), torch.rand(1, H, W, generator=g, dtype=dtype),
            torch.rand(1, H, W, generator=g, dtype=dtype))


def rel_err(a, b):
    """max |a-b| relative to max |b| (the 'relative fp32' figure of BASELINE.json's north_star)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def outlier_fraction(a, b, rtol):
    """fraction of elements with |a-b| > rtol * max|b|."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float((np.abs(a - b) > rtol * (np.abs(b).max() + 1e-30)).mean())


import contextlib


@contextlib.contextmanager
def tile_culling(on):
    """Run a block with the library's tile culling forced on / off (off: internal lists = the reference algorithm's)."""
    from egogaussian_amd import _C
    old = _C.set_tile_culling(on)
    try:
        yield
    finally:
        _C.set_tile_culling(old)


@contextlib.contextmanager
def sort_in_blend(on):
    """Run a block with the per-tile sort inside the forward blend's launch (on, the default) or as launches of its own."""
    from egogaussian_amd import _C
    old = _C.set_sort_in_blend(on)
    try:
        yield
    finally:
        _C.set_sort_in_blend(old)


@contextlib.contextmanager
def fused_count(on):
    """Run a block with the count pass of the bucketing inside the preprocess launch (on, the default with a placement buffer) or as its own launch."""
    from egogaussian_amd import _C
    old = _C.set_fused_count(on)
    try:
        yield
    finally:
        _C.set_fused_count(old)


def check_culled_lists(st, ranges_hip, point_list_hip, H, W):
    """With tile culling on, every tile's list must be the oracle's list with some entries removed (same order), and every
    removed (tile, splat) instance must be one the reference skips at all 256 pixels: alpha < 1/255 or power > 0
    (evaluated here in float64 from the oracle's conic / opacity / centre).  Returns (kept, dropped) instance counts."""
    import numpy as np
    ro, po = st["ranges"].astype(np.int64), st["point_list"]
    rh = ranges_hip.astype(np.int64)
    n_o, n_h = ro[:, 1] - ro[:, 0], rh[:, 1] - rh[:, 0]
    assert np.all(n_h <= n_o)
    assert int(n_h.sum()) == len(point_list_hip[:int(n_h.sum())])
    tile_o = np.repeat(np.arange(len(ro)), n_o)
    tile_h = np.repeat(np.arange(len(rh)), n_h)
    # the HIP ranges are compact and in tile order, like the oracle's
    assert np.array_equal(rh[n_h > 0, 0], (np.cumsum(n_h) - n_h)[n_h > 0])
    key_o = tile_o.astype(np.int64) * (1 << 32) + po.astype(np.int64)          # (tile, splat) is unique
    key_h = tile_h.astype(np.int64) * (1 << 32) + point_list_hip[:len(tile_h)].astype(np.int64)
    kept = np.isin(key_o, key_h)
    assert kept.sum() == len(key_h) and np.array_equal(key_o[kept], key_h), "culled list is not an ordered sub-list of the reference list"
    drop_t, drop_g = tile_o[~kept], po[~kept]
    gx = (W + 15) // 16
    co, xy = st["conic_opacity"].astype(np.float64), st["xy"].astype(np.float64)
    worst = 0.0
    for lo in range(0, len(drop_t), 200000):
        t, g = drop_t[lo:lo + 200000], drop_g[lo:lo + 200000]
        px = (t % gx)[:, None] * 16 + np.arange(16)[None, :]                     # [n,16]
        py = (t // gx)[:, None] * 16 + np.arange(16)[None, :]
        dx = xy[g, 0][:, None, None] - px[:, None, :]                            # [n,1,16]
        dy = xy[g, 1][:, None, None] - py[:, :, None]                            # [n,16,1]
        power = -0.5 * (co[g, 0][:, None, None] * dx * dx + co[g, 2][:, None, None] * dy * dy) - co[g, 1][:, None, None] * dx * dy
        alpha = np.minimum(0.99, co[g, 3][:, None, None] * np.exp(np.minimum(power, 0.0)))
        alpha = np.where((power > 0) | (px[:, None, :] >= W) | (py[:, :, None] >= H), 0.0, alpha)
        worst = max(worst, float(alpha.max()) if alpha.size else 0.0)
    assert worst < (1.0 / 255.0) * (1 + 1e-5), f"a culled instance reaches alpha {worst} >= 1/255 somewhere in its tile"
    return int(kept.sum()), int((~kept).sum())


class OracleRasterize(torch.autograd.Function):
    """The C oracle (forward + analytic backward) as a CPU autograd op, so that an oracle-side TRAINING chain (torch covariance ->
    oracle rasterizer -> torch loss -> torch Adam) can run for hundreds of steps in seconds.  Test infrastructure only.
    Inputs: means3D [N,3], opacities [N,1], shs [N,M,3], cov3D [N,6] (float32 CPU tensors) + a dict of camera / image constants."""

    @staticmethod
    def forward(ctx, means3D, opacities, shs, cov3D, const):
        from oracle.oracle import Oracle
        o = Oracle(np.float32, nthreads=const.get("nthreads", 8))
        st = o.forward(means3D=means3D, opacities=opacities, shs=shs, cov3D_precomp=cov3D, viewmatrix=const["viewmatrix"],
                       projmatrix=const["projmatrix"], campos=const["campos"], bg=const["bg"], image_height=const["H"], image_width=const["W"],
                       tanfovx=const["tanfovx"], tanfovy=const["tanfovy"], sh_degree=const.get("sh_degree", 0))
        ctx.o, ctx.st, ctx.shapes = o, st, (means3D.shape, opacities.shape, shs.shape, cov3D.shape)
        OracleRasterize.last_radii = torch.from_numpy(st["radii"].copy())          # what a trainer reads besides the image (densification)
        return torch.from_numpy(np.ascontiguousarray(st["color"], dtype=np.float32))

    @staticmethod
    def backward(ctx, g_color):
        gb = ctx.o.backward(ctx.st, g_color.contiguous(), None, None)
        OracleRasterize.last_mean2D_grad = torch.from_numpy(np.ascontiguousarray(gb["dL_dmean2D"], dtype=np.float32))   # = viewspace_points.grad
        t = lambda a, s: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(s)
        return (t(gb["dL_dmeans3D"], ctx.shapes[0]), t(gb["dL_dopacity"], ctx.shapes[1]), t(gb["dL_dsh"], ctx.shapes[2]),
                t(gb["dL_dcov3D"], ctx.shapes[3]), None)


def quantize_8bit(img):
    """What a rendered image goes through before the reference's metrics read it back: torchvision.utils.save_image to PNG
    (/root/reference/trainers/eval_metric.py:113-120) = round(255 x) clamped to [0, 255], then / 255 on load."""
    return torch.floor(img * 255.0 + 0.5).clamp(0, 255) / 255.0


# ---- threshold flips: isolate the damage instead of loosening the bar --------------------------------------------------------------
# v_exp_f32 and glibc expf differ in the last place, so a (pixel, splat) pair whose alpha sits within float32 arithmetic's reach of 1/255
# (or whose transmittance sits at 1e-4) is kept on one side and skipped on the other.  Such a pair changes ITS pixel, hence the gradients of
# the splats in that pixel's list -- and nothing else.  The parity tests therefore (1) find the pixels that differ by more than a detection
# level far below the parity bar, (2) make every one of them PROVE why (pixel_account below: the float64 re-walk of the pixel's chain
# either bounds what two float32 evaluations of that chain can differ by -- the pixel is then ordinary float noise and stays under the
# 1e-4 bar like every other pixel -- or holds a threshold-adjacent pair -- a flip; anything else fails the test), (3) collect the
# Gaussians of the oracle's tile lists of the FLIPPED pixels' tiles (a superset of the splats the pixel's chain touches), and (4) hold
# every other Gaussian's gradients and every other pixel to the north star's 1e-4; the few affected rows are bounded by a flipped pair's share.
FLIP_DETECT = 2e-6

ALPHA_WINDOW = 1e-5          # least relative half-width around 1/255 in which a pair's alpha may be kept on one side and skipped on the other
T_WINDOW = 1e-9              # least absolute half-width around the 1e-4 transmittance stop (relative 1e-5)
POWER_WINDOW = 1e-6          # least |power| below which the `power > 0` skip may go either way
EPS32 = 2.0 ** -24           # float32 unit roundoff
K_ARITH = 8.0                # roundings that separate two float32 evaluations of one exponent: ~4 per side (dx, dy, three products, two sums;
                             # this library also rounds the conic once more when it pre-scales it by log2 e), each relative to the LARGEST
                             # term of -(A dx^2 + C dy^2) / 2 - B dx dy, not to the sum: for a splat centred hundreds of pixels away the
                             # terms are in the hundreds and the exponent they cancel to is ~ -5


def pixel_account(st, y, x):
    """The chain of pixel (y, x) re-walked in FLOAT64 from the oracle's per-Gaussian state (pixel centres, conics, opacities, colours: the
    values both sides share bit for bit; oracle/raster_oracle.c egso_render_forward is the loop).  ->
      cause   : a description of the first pair, at or before the point where both sides must have stopped, at which two float32
                evaluations may take different branches -- alpha within the pair's own arithmetic error (at least ALPHA_WINDOW, relative)
                of 1/255, exponent within it of 0, T' = T (1 - alpha) within the accumulated error (at least T_WINDOW) of 1e-4 -- or None;
      noise_c : what two float32 evaluations that take the SAME branches can differ by in the pixel's colour (absolute, any channel),
      noise_T : ... and in its final transmittance.
    The error model: an exponent is off by at most K_ARITH * EPS32 * (|A dx^2| / 2 + |C dy^2| / 2 + |B dx dy| + 1) =: r_j (absolute, hence
    relative in alpha); T_j by the sum of alpha_k r_k / (1 - alpha_k) over the kept pairs in front; a contribution c alpha T by its
    relative errors added.  Worst case, first order: a pixel that is off by MORE than this with no adjacent pair is a wrong result."""
    W = st["W"]
    gx = (W + 15) // 16
    t = (y // 16) * gx + x // 16
    ids = st["point_list"][int(st["ranges"][t, 0]):int(st["ranges"][t, 1])].astype(np.int64)
    if ids.size == 0:
        return None, 0.0, 0.0
    co = st["conic_opacity"][ids].astype(np.float64)
    dx = st["xy"][ids, 0].astype(np.float64) - float(x)
    dy = st["xy"][ids, 1].astype(np.float64) - float(y)
    ta, tb, tc = 0.5 * co[:, 0] * dx * dx, co[:, 1] * dx * dy, 0.5 * co[:, 2] * dy * dy
    power = -(ta + tc) - tb
    r = K_ARITH * EPS32 * (np.abs(ta) + np.abs(tb) + np.abs(tc) + 1.0)              # per pair: |d power| = |d alpha| / alpha
    alpha = np.minimum(0.99, co[:, 3] * np.exp(np.minimum(power, 50.0)))
    near_p = np.abs(power) <= np.maximum(POWER_WINDOW, r)
    near_a = (power <= np.maximum(POWER_WINDOW, r)) & (np.abs(alpha * 255.0 - 1.0) <= np.maximum(ALPHA_WINDOW, r))
    keep = (power <= 0.0) & (alpha >= 1.0 / 255.0)
    Tp = np.cumprod(np.where(keep, 1.0 - alpha, 1.0))                 # T' after entry j if it is kept
    relT = np.cumsum(np.where(keep, (alpha * r + EPS32) / (1.0 - alpha), 0.0))      # relative error of T' after entry j
    win_T = np.maximum(T_WINDOW, 1e-4 * relT)
    near_T = keep & (np.abs(Tp - 1e-4) <= win_T)
    stop = np.nonzero(keep & (Tp < 1e-4 - win_T))[0]                   # the first entry at which BOTH sides must have stopped
    end = int(stop[0]) + 1 if stop.size else ids.size
    cause = None
    for name, m in (("alpha", near_a), ("T'", near_T), ("power", near_p)):
        j = np.nonzero(m[:end])[0]
        if j.size:
            j = int(j[0])
            cause = (f"{name} threshold: list entry {j} (Gaussian {int(ids[j])}) alpha*255 = {alpha[j] * 255.0:.9f}, power = {power[j]:.3e}, T' = {Tp[j]:.6e}, "
                     f"arithmetic reach {r[j]:.1e}, accumulated in T {relT[j]:.1e}")
            break
    # noise of a chain that takes the same branches on both sides (entries before the stop only)
    n = end if not stop.size else end - 1                              # the stopping entry itself does not contribute
    k = keep[:n]
    T_before = np.concatenate([[1.0], Tp[:-1]])[:n]
    relT_before = np.concatenate([[0.0], relT[:-1]])[:n]
    cmax = np.abs(st["rgb"][ids[:n]].astype(np.float64)).max(1) if "rgb" in st else np.ones(n)
    noise_c = float(np.sum(np.where(k, cmax * alpha[:n] * T_before * (r[:n] + relT_before + 2 * EPS32), 0.0)))
    T_fin = float(Tp[n - 1]) if n else 1.0
    rel_fin = float(relT[n - 1]) if n else 0.0
    noise_c += float(np.abs(st["bg"]).max()) * T_fin * rel_fin if "bg" in st else 0.0
    return cause, noise_c, T_fin * rel_fin


def flip_cause(st, y, x):
    """The threshold-adjacent pair of pixel (y, x), or None (see pixel_account)."""
    return pixel_account(st, y, x)[0]


def flip_pixels(color_hip, final_T_hip, st, n_contrib_hip=None, justify=True, report=None):
    """bool[H,W]: pixels whose colour, final transmittance or (reference lists only) contributor count differ from the oracle's because a
    (pixel, splat) pair went the other way at one of the compositing loop's three thresholds -- PROVEN per pixel.  Every pixel that is off
    by more than FLIP_DETECT is put through pixel_account(): it is a flip when its float64 chain holds a threshold-adjacent pair; it is
    float noise (NOT in the returned mask: it answers to the 1e-4 bar like every other pixel, and relaxes no Gaussian's bound) when it has
    no such pair but is off by no more than what two float32 evaluations of that chain can differ by; otherwise the call fails: a pixel
    that is off with neither is a wrong blend, whatever the fraction ('the flip rule excuses without proving cause').
    report: a dict that receives the counts (detected, flips, noise) and the largest noise bound used."""
    c = np.asarray(color_hip, dtype=np.float64); T = np.asarray(final_T_hip, dtype=np.float64)
    cscale = max(float(np.abs(st["color"]).max()), 1e-30)
    dc = np.abs(c - st["color"]).max(0); dT = np.abs(T - st["final_T"])
    px = (dc > FLIP_DETECT * cscale) | (dT > FLIP_DETECT)
    if n_contrib_hip is not None:
        dn = np.asarray(n_contrib_hip) != st["n_contrib"]
        px = px | dn
    if not justify or not px.any():
        if report is not None:
            report.update(detected=int(px.sum()), flips=int(px.sum()), noise=0, max_noise_bound=0.0)
        return px
    flips = np.zeros_like(px)
    bad, n_noise, worst_bound = [], 0, 0.0
    for y, x in np.argwhere(px):
        cause, noise_c, noise_T = pixel_account(st, int(y), int(x))
        if cause is not None:
            flips[y, x] = True
        elif (n_contrib_hip is None or not dn[y, x]) and dc[y, x] <= FLIP_DETECT * cscale + noise_c and dT[y, x] <= FLIP_DETECT + noise_T:
            n_noise += 1
            worst_bound = max(worst_bound, noise_c / cscale, noise_T)
        else:
            bad.append((int(y), int(x), float(dc[y, x]), float(dT[y, x]), f"float32 reach of this chain: colour {noise_c:.2e}, T {noise_T:.2e}"))
    if report is not None:
        report.update(detected=int(px.sum()), flips=int(flips.sum()), noise=n_noise, max_noise_bound=worst_bound)
    assert not bad, (f"{len(bad)} of {int(px.sum())} pixels differ from the oracle by more than the detection level ({FLIP_DETECT:g} of the image maximum) AND by more than "
                     f"two float32 evaluations of their chain can, with NO threshold-adjacent pair in the float64 chain -- neither flips nor float noise: "
                     f"(y, x, colour diff, final_T diff, reach) {bad[:8]}")
    return flips


def check_images_isolating_flips(images, st, flip_px, tol=1e-4, share=2e-2, what=""):
    """The image-sized outputs ((name, hip array, oracle array) triples) against the oracle: every pixel that is not a (justified) flipped
    pixel within `tol` of the plane's maximum -- asserted, no fraction is excused --, the flipped ones within one threshold-level
    contribution (`share`).  -> report string."""
    rep = []
    for name, hip, ora in images:
        h = np.asarray(hip, dtype=np.float64).reshape(-1, *flip_px.shape); o = np.asarray(ora, dtype=np.float64).reshape(h.shape)
        scale = float(np.abs(o).max()) + 1e-30
        err = np.abs(h - o).max(0) / scale
        e_far = float(np.where(flip_px, 0.0, err).max()) if err.size else 0.0
        e_near = float(err[flip_px].max()) if flip_px.any() else 0.0
        assert e_far < tol, f"{what} {name}: max rel err {e_far} at pixel {np.unravel_index(int(np.argmax(np.where(flip_px, 0.0, err))), err.shape)} away from every flipped pixel"
        assert e_near < share, f"{what} {name}: max rel err {e_near} on a flipped pixel"
        rep.append(f"{name} {e_far:.1e}" + (f" (flipped pixels {e_near:.1e})" if flip_px.any() else ""))
    return "; ".join(rep)


def gaussians_near_flips(st, flip_px, halo=0):
    """ids of the Gaussians in the oracle's lists of the tiles that hold a flipped pixel.  halo: pixels around a flipped pixel that count
    as flipped too -- when the upstream image gradient comes from a loss with a window (SSIM, 11x11: 5), a flipped pixel changes the
    upstream gradient of its neighbours, which may sit in the next tile."""
    H, W = flip_px.shape
    gx = (W + 15) // 16
    ys, xs = np.nonzero(flip_px)
    if halo and ys.size:
        dy, dx = np.meshgrid(np.arange(-halo, halo + 1), np.arange(-halo, halo + 1), indexing="ij")
        ys = np.clip(ys[:, None] + dy.reshape(1, -1), 0, H - 1).reshape(-1)
        xs = np.clip(xs[:, None] + dx.reshape(1, -1), 0, W - 1).reshape(-1)
    tiles = np.unique((ys // 16) * gx + xs // 16)
    if tiles.size == 0:
        return np.zeros(0, dtype=np.int64)
    rng, pl = st["ranges"], st["point_list"]
    return np.unique(np.concatenate([pl[int(rng[t, 0]):int(rng[t, 1])] for t in tiles]).astype(np.int64))


def check_grads_isolating_flips(names, hip_grads, oracle_grads, st, flip_px, tol=1e-4, share=5e-2, what="", halo=0, far_frac=0.0, far_cap=3.0, over_rows=None):
    """Every gradient array (one row per Gaussian) against the oracle: rows of Gaussians away from every flipped pixel within `tol` of
    the array's maximum (the north star's bar, asserted; "relative" is max-norm relative: |hip - oracle| over the largest |oracle| entry of
    the array), the affected rows within `share`.  -> (report string, worst unaffected error, number of affected Gaussians).
    over_rows: a dict that receives, per array name, the rows far from every flip that sit between tol and far_cap x tol (the caller then
    has to account for each of them: tests/test_gpu_bench_mode.py does so with repeated runs and the float64 oracle)."""
    near = gaussians_near_flips(st, flip_px, halo)
    rep, worst = [], 0.0
    for name, h in zip(names, hip_grads):
        ora = oracle_grads.get(name) if isinstance(oracle_grads, dict) else None
        if ora is None or h is None or (hasattr(h, "numel") and h.numel() == 0):
            continue
        hh = (h.detach().cpu().numpy() if hasattr(h, "detach") else np.asarray(h)).reshape(ora.shape).astype(np.float64)
        oo = np.asarray(ora, dtype=np.float64)
        scale = float(np.abs(oo).max()) + 1e-30
        row_err = np.abs(hh - oo).reshape(oo.shape[0], -1).max(1) / scale
        mask = np.zeros(oo.shape[0], dtype=bool); mask[near[near < oo.shape[0]]] = True
        e_far = float(row_err[~mask].max()) if (~mask).any() else 0.0
        e_near = float(row_err[mask].max()) if mask.any() else 0.0
        worst = max(worst, e_far)
        rep.append(f"{name} {e_far:.1e}" + (f" (near flips {e_near:.1e})" if mask.any() else ""))
        # Rows away from every proven flip: the bar itself, no fraction excused (far_frac = 0).  (Through round 5 one row in 100 000 was let
        # through up to 3 x the bar as "fp32 accumulation order"; round 6 traced every such row of the benched-mode test to an L1 sign tie --
        # a threshold of the LOSS, now found and proven like the compositing loop's own -- and the suite has used the allowance nowhere since.
        # A caller that passes far_frac > 0 gets the rows back in `over_rows` and has to account for each.)
        far_err = np.where(mask, 0.0, row_err)
        n_over = int((far_err >= tol).sum())
        if n_over > int(far_frac * oo.shape[0]) or (n_over and float(far_err.max()) >= far_cap * tol):
            i = int(np.argmax(far_err))
            raise AssertionError(f"{what} {name}: max rel err {e_far} on Gaussian {i} (hip {hh[i].ravel()[:4]}, oracle {oo[i].ravel()[:4]}, array max {scale:.3e}), "
                                 f"away from every flipped pixel ({int(flip_px.sum())} flipped pixels at {np.argwhere(flip_px)[:12].tolist()}, {near.size} Gaussians near them; "
                                 f"radius {int(st['radii'][i])}, centre {st['xy'][i].tolist()}); {n_over} rows over {tol:g}")
        if n_over:
            rep[-1] += f" [{n_over} row(s) of {oo.shape[0]} between {tol:g} and {far_cap * tol:g}: to be accounted for by the caller]"
            if over_rows is not None:
                over_rows[name] = np.nonzero(far_err >= tol)[0]
        assert e_near < share, f"{what} {name}: max rel err {e_near} on a Gaussian in a flipped pixel's tile list"
    return "; ".join(rep) + f"; flipped pixels {int(flip_px.sum())}, Gaussians near them {near.size}", worst, int(near.size)
