"""MI355X-native differentiable Gaussian rasterizer for the reference's `gaussian_renderer.render()` path.

    import egogaussian_amd
    egogaussian_amd.install()                # patching.install: BEFORE the reference's trainers are imported -- its loss functions, covariance
                                             # producers and optimizer run on this package's kernels, the trainers stay as they are
    egogaussian_amd.attach(gaussians)        # adapter.attach: the same for ONE model object (fused covariance producers, raw-parameter hooks, FusedAdam)
"""


def attach(gaussians, **kw):
    from .adapter import attach as _attach
    return _attach(gaussians, **kw)


def install(**kw):
    from .patching import install as _install
    return _install(**kw)


def uninstall():
    from .patching import uninstall as _uninstall
    return _uninstall()
