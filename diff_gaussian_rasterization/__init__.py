"""Drop-in module name for the reference's import
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
(/root/reference/gaussian_renderer/__init__.py:14, /root/reference/gaussian_renderer/render_helper.py:3).
Put this repository's root on sys.path ahead of (or instead of) the CUDA submodule; the implementation is
egogaussian_amd (hand-written HIP for gfx950 behind a C ABI, no CPU fallback)."""
from egogaussian_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,
                                        _RasterizeGaussians)
from egogaussian_amd import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_RasterizeGaussians", "_C"]
