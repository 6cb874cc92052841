"""GPU: the reference's own render() route -- activated tensors from the model's getters into the rasterizer
(/root/reference/gaussian_renderer/__init__.py:56-98) -- with and without the provenance tags of adapter.attach() (egogaussian_amd/provenance.py).
With them the rasterizer takes the raw parameters (activations and covariance inside its preprocess kernel, gradients straight to the leaves);
image, radii and every leaf's gradient must equal the route through the activated tensors to float rounding, for the plain call, with 16 SH
coefficients (get_features is a concatenation) and for the object-rotated covariance of fine_all.py:93."""
import math

import numpy as np
import pytest
import torch

from tests.test_adapter import RefShaped, _scene

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


class Surface:
    """the attribute names the reference's render() and trainers use, and nothing else: none of this package's optional hooks is visible"""
    _ALLOWED = ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features", "get_covariance", "get_rotated_covariance", "get_is_object",
                "active_sh_degree", "max_sh_degree", "_xyz", "_is_object")

    def __init__(self, m):
        object.__setattr__(self, "_m", m)

    def __getattr__(self, name):
        if name in Surface._ALLOWED:
            return getattr(object.__getattribute__(self, "_m"), name)
        raise AttributeError(name)


@pytest.mark.parametrize("deg,rot", [(0, False), (3, False), (1, True)])
def test_reference_route_reaches_the_raw_parameter_path(deg, rot):
    import egogaussian_amd
    from egogaussian_amd import provenance
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import make_camera, Pipe
    dev = torch.device("cuda:0")
    H, W, N = 96, 128, 20000
    scene = _scene(N, H, W, seed=5, deg=deg)
    cam, bg = make_camera(9, H, W, device=dev), torch.tensor([0.1, 0.2, 0.3], device=dev)
    g = torch.Generator().manual_seed(3)
    wc, wd, wa = (torch.rand(3, H, W, generator=g).to(dev), torch.rand(1, H, W, generator=g).to(dev), torch.rand(1, H, W, generator=g).to(dev))
    th = 0.3
    R = torch.tensor([[math.cos(th), -math.sin(th), 0.0], [math.sin(th), math.cos(th), 0.0], [0.0, 0.0, 1.0]], device=dev)
    res = {}
    for tagged in (False, True):
        m = RefShaped(scene, device=dev, sh_degree=deg)
        m._is_object = (torch.arange(N, device=dev) % 3 == 0).float().reshape(N, 1)
        m.training_setup()
        egogaussian_amd.attach(m, provenance=tagged)
        n0 = provenance.substitutions
        kw = dict(rot_cov=True, accum_R=R, which_object=1, during_training=False) if rot else {}
        pkg = render(cam, Surface(m), Pipe, bg, **kw)
        assert provenance.substitutions == n0 + (1 if tagged else 0), "the tagged route did not reach (or the plain route reached) the raw-parameter path"
        loss = (pkg["render"] * wc).sum() + (pkg["depth"] * wd).sum() + (pkg["alpha"] * wa).sum()
        loss.backward()
        res[tagged] = dict(image=pkg["render"].detach().cpu().numpy(), depth=pkg["depth"].detach().cpu().numpy(), radii=pkg["radii"].cpu().numpy(),
                           means2D=pkg["viewspace_points"].grad.cpu().numpy(),
                           **{a: (getattr(m, a).grad.cpu().numpy() if getattr(m, a).grad is not None else None) for a in LEAVES})
    a, b = res[False], res[True]
    assert np.array_equal(a["radii"], b["radii"])
    rel = lambda x, y: float(np.abs(x - y).max() / (np.abs(x).max() + 1e-30))
    rep = {k: rel(a[k], b[k]) for k in ("image", "depth", "means2D") + LEAVES if a[k] is not None and a[k].size}
    print(f"\n  [deg {deg}, rotated {rot}] activated-tensor route vs raw-parameter route (max-norm relative): " + ", ".join(f"{k} {v:.1e}" for k, v in rep.items()))
    assert rep["image"] < 1e-5 and rep["depth"] < 1e-5
    for k in ("means2D",) + LEAVES:
        if k in rep:
            assert rep[k] < 1e-4, k
        else:
            assert b[k] is None or b[k].size == 0 or not np.any(b[k]), k


def test_a_touched_tensor_takes_the_route_it_was_given():
    """the tag vouches for the getter's own result only: scaled opacities (a trainer experimenting with the render inputs) are rendered as given"""
    import egogaussian_amd
    from egogaussian_amd import provenance
    from egogaussian_amd.rasterizer import GaussianRasterizer
    from egogaussian_amd.renderer import get_raster_settings
    from egogaussian_amd.scene_synth import make_camera
    dev = torch.device("cuda:0")
    H, W, N = 64, 64, 5000
    m = RefShaped(_scene(N, H, W, seed=2), device=dev); m.training_setup(); egogaussian_amd.attach(m)
    cam, bg = make_camera(3, H, W, device=dev), torch.zeros(3, device=dev)
    rast = GaussianRasterizer(get_raster_settings(cam, m, bg))
    z = torch.zeros_like(m._xyz)
    n0 = provenance.substitutions
    full = rast(means3D=m.get_xyz, means2D=z, opacities=m.get_opacity, shs=m.get_features, cov3D_precomp=m.get_covariance(1.0))[0]
    assert provenance.substitutions == n0 + 1
    half = rast(means3D=m.get_xyz, means2D=z, opacities=m.get_opacity * 0.5, shs=m.get_features, cov3D_precomp=m.get_covariance(1.0))[0]
    assert provenance.substitutions == n0 + 1                          # not substituted: the halved opacities are what was rendered
    assert float((full - half).abs().max()) > 1e-2
    with torch.no_grad():
        m._opacity.add_(1.0)                                            # the parameter moved after the getter ran: the old result is stale
    stale = rast(means3D=m.get_xyz, means2D=z, opacities=torch.sigmoid(m._opacity - 1.0), shs=m.get_features, cov3D_precomp=m.get_covariance(1.0))[0]
    assert provenance.substitutions == n0 + 1 and float((stale - full).abs().max()) < 1e-5
