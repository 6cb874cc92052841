"""MI355X-native differentiable Gaussian rasterizer for the reference's `gaussian_renderer.render()` path.

    import egogaussian_amd
    egogaussian_amd.attach(gaussians)        # adapter.attach: fused covariance producers, raw-parameter hooks, FusedAdam
"""


def attach(gaussians, **kw):
    from .adapter import attach as _attach
    return _attach(gaussians, **kw)
