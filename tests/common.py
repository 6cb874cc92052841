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
