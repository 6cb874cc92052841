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
        return torch.from_numpy(np.ascontiguousarray(st["color"], dtype=np.float32))

    @staticmethod
    def backward(ctx, g_color):
        gb = ctx.o.backward(ctx.st, g_color.contiguous(), None, None)
        t = lambda a, s: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(s)
        return (t(gb["dL_dmeans3D"], ctx.shapes[0]), t(gb["dL_dopacity"], ctx.shapes[1]), t(gb["dL_dsh"], ctx.shapes[2]),
                t(gb["dL_dcov3D"], ctx.shapes[3]), None)


def quantize_8bit(img):
    """What a rendered image goes through before the reference's metrics read it back: torchvision.utils.save_image to PNG
    (/root/reference/trainers/eval_metric.py:113-120) = round(255 x) clamped to [0, 255], then / 255 on load."""
    return torch.floor(img * 255.0 + 0.5).clamp(0, 255) / 255.0
