"""egogaussian_amd.attach(): a reference-shaped GaussianModel pointed at the fast paths in one call (adapter.py).

The model below carries the attribute names and getters of /root/reference/scene/gaussian_model.py:125-200 (parameters, activations,
`covariance_activation`, `get_covariance`, `get_rotated_covariance`, a torch.optim.Adam over named groups); when /root/reference
is on this machine (the build container) the same checks also run on the reference's own class."""
import math
import os
import sys

import numpy as np
import pytest
import torch


class RefShaped:
    """Stand-in with the reference model's surface (gaussian_model.py:29-63,125-200)."""

    def __init__(self, scene, device="cpu", sh_degree=0):
        from egogaussian_amd.covariance import covariance_from_scaling_rotation, rotated_covariance_from_scaling_rotation
        P_ = lambda a: torch.nn.Parameter(torch.tensor(a, device=device).requires_grad_(True))
        self._xyz, self._features_dc = P_(scene["xyz"]), P_(scene["features"][:, :1].copy())
        self._features_rest = P_(scene["features"][:, 1:].copy())
        self._scaling, self._rotation, self._opacity = P_(scene["log_scale"]), P_(scene["quat"]), P_(scene["opacity_logit"])
        n = scene["xyz"].shape[0]
        self._is_object = torch.zeros((n, 1), device=device)
        self.active_sh_degree = self.max_sh_degree = sh_degree
        self.trainable_object_move = None
        self.scaling_activation, self.opacity_activation, self.rotation_activation = torch.exp, torch.sigmoid, torch.nn.functional.normalize
        self.covariance_activation = covariance_from_scaling_rotation
        self._rot_cov = rotated_covariance_from_scaling_rotation
        self.covariance_activation_w_rot = self.build_covariance_from_scaling_rotation_w_rot     # bound at __init__, gaussian_model.py:39
        self.optimizer = None

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))                    # gaussian_model.py:36-44,142-160: activations are attributes
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))

    @property
    def get_features(self):
        f = torch.cat((self._features_dc, self._features_rest), dim=1)
        if getattr(self, "_egs_tag_features", False):                                       # what patching.install() wraps around the class's property
            from egogaussian_amd.provenance import tag_features
            tag_features(f, self._features_dc, self._features_rest)
        return f
    get_is_object = property(lambda s: s._is_object)

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self._rotation)

    def build_covariance_from_scaling_rotation_w_rot(self, scaling, scaling_modifier, rotation, accum_R, which_object=None, during_training=False):
        tom = self.trainable_object_move if during_training else None
        return self._rot_cov(scaling, scaling_modifier, rotation, accum_R, self._is_object, which_object, None if tom is None else tom.rot_L)

    def get_rotated_covariance(self, accum_R, which_object, during_training, scaling_modifier=1):
        return self.covariance_activation_w_rot(self.get_scaling, scaling_modifier, self._rotation, accum_R, which_object, during_training)   # :171

    def training_setup(self):
        groups = [{"params": [self._xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [self._features_dc], "lr": 2.5e-3, "name": "f_dc"},
                  {"params": [self._features_rest], "lr": 2.5e-3 / 20.0, "name": "f_rest"}, {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
                  {"params": [self._scaling], "lr": 5e-3, "name": "scaling"}, {"params": [self._rotation], "lr": 1e-3, "name": "rotation"}]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)          # gaussian_model.py:198
        return self.optimizer


def _scene(n=300, H=48, W=80, seed=3, deg=0):
    from egogaussian_amd.scene_synth import make_scene
    sc = make_scene(n, H, W, seed=seed, sh_degree=deg)
    sc["log_scale"] += math.log(3.0)
    return sc


def _check_attached_host_side(m):
    """CPU: hooks present and inert off a HIP device, covariance paths unchanged in value, optimizer swapped with its state."""
    import egogaussian_amd
    from egogaussian_amd.optim import FusedAdam
    cov0 = m.get_covariance(1.0).detach().clone()
    old_opt = m.optimizer
    names = [g["name"] for g in old_opt.param_groups]
    p0 = old_opt.param_groups[0]["params"][0]
    old_opt.state[p0] = {"step": torch.tensor(7.0), "exp_avg": torch.full_like(p0, 0.25), "exp_avg_sq": torch.full_like(p0, 0.5)}
    assert egogaussian_amd.attach(m) is m
    for hook in ("get_raw_parameters", "get_features_split", "get_covariance_and_opacity", "get_raw_parameters_rotated", "get_rotated_covariance_and_opacity"):
        assert callable(getattr(m, hook))
    assert m.get_raw_parameters() is None and m.get_features_split() is None            # CPU tensors: the hooks step aside
    assert m.get_raw_parameters_rotated(torch.eye(3), 1, False) is None
    assert torch.allclose(m.get_covariance(1.0), cov0, rtol=1e-6, atol=1e-12)
    c, o = m.get_covariance_and_opacity(1.0)
    assert torch.allclose(c, cov0, rtol=1e-6, atol=1e-12) and torch.allclose(o, m.get_opacity)
    assert isinstance(m.optimizer, FusedAdam) and [g["name"] for g in m.optimizer.param_groups] == names
    assert [g["lr"] for g in m.optimizer.param_groups] == [g["lr"] for g in old_opt.param_groups]
    assert m.optimizer.param_groups[0]["params"][0] is p0
    st = m.optimizer.state[p0]
    assert float(st["step"]) == 7.0 and float(st["exp_avg"].mean()) == 0.25 and float(st["exp_avg_sq"].mean()) == 0.5
    assert m._egs_fused_optimizer is None
    assert egogaussian_amd.attach(m).optimizer is m.optimizer                            # idempotent
    # get_rotated_covariance() -- what the reference's render(rot_cov=True) calls -- must REACH the installed producer: it goes through
    # the attribute setup_functions() bound at __init__ (gaussian_model.py:39,171), not through the method name
    if hasattr(m, "_is_object"):
        n0 = m.covariance_activation_w_rot.calls
        before = m.get_rotated_covariance(torch.eye(3), 1, False, 1.0)
        assert m.covariance_activation_w_rot.calls == n0 + 1, "get_rotated_covariance did not run the installed producer"
        assert m.covariance_activation_w_rot is m.build_covariance_from_scaling_rotation_w_rot
        assert torch.allclose(before, cov0, rtol=1e-6, atol=1e-12)                       # identity rotation: the plain covariance


def test_attach_on_reference_shaped_model_cpu():
    import egogaussian_amd
    m = RefShaped(_scene())
    m.training_setup()
    _check_attached_host_side(m)
    with pytest.raises(TypeError):
        egogaussian_amd.attach(object())
    with pytest.raises(ValueError):
        egogaussian_amd.attach(RefShaped(_scene()), fuse_optimizer=True)


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene"), reason="the reference's Python is only present in the build container")
def test_attach_on_the_reference_class_itself_cpu():
    """The same on /root/reference/scene/gaussian_model.py's GaussianModel, imported the way tests/golden/make_golden.py does."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("pytorch3d"); mg.stub("pytorch3d.transforms", euler_angles_to_matrix=None)
    mg.stub("simple_knn"); mg.stub("simple_knn._C", distCUDA2=lambda pts: torch.full((pts.shape[0],), 1e-3))
    saved = sys.modules.get("diff_gaussian_rasterization")
    mg.install_recording_rasterizer()
    try:
        with mg.CudaToCpu():
            from scene.gaussian_model import GaussianModel
            sc = _scene()
            g = GaussianModel(0)
            P_ = lambda x: torch.nn.Parameter(torch.tensor(x, dtype=torch.float32))
            g._xyz, g._features_dc, g._features_rest = P_(sc["xyz"]), P_(sc["features"][:, :1]), P_(np.zeros((300, 0, 3), np.float32))
            g._scaling, g._rotation, g._opacity = P_(sc["log_scale"]), P_(sc["quat"]), P_(sc["opacity_logit"])
            g._label = P_(np.zeros((300, 1), np.float32))
            g._is_object = torch.zeros(300, 1); g._generation = torch.zeros(300, 1)
            groups = [{"params": [getattr(g, a)], "lr": lr, "name": n} for a, lr, n in
                      (("_xyz", 1.6e-4, "xyz"), ("_features_dc", 2.5e-3, "f_dc"), ("_features_rest", 1.25e-4, "f_rest"), ("_opacity", 0.05, "opacity"),
                       ("_scaling", 5e-3, "scaling"), ("_rotation", 1e-3, "rotation"))]
            g.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
            _check_attached_host_side(g)
            # the reference's own rotated-covariance entry point now runs through the installed producer (same values, CPU mirror)
            R = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
            g._is_object[::3] = 1.0
            n0 = g.covariance_activation_w_rot.calls
            a = g.get_rotated_covariance(R, 1, False, 1.0)
            assert g.covariance_activation_w_rot.calls == n0 + 1
            from egogaussian_amd.covariance import rotated_covariance_from_scaling_rotation
            b = rotated_covariance_from_scaling_rotation(g.get_scaling, 1.0, g._rotation, R, g._is_object, 1, None)
            assert torch.allclose(a, b)
    finally:
        if saved is not None:
            sys.modules["diff_gaussian_rasterization"] = saved
        else:
            sys.modules.pop("diff_gaussian_rasterization", None)
        for k in ("plyfile", "pytorch3d", "pytorch3d.transforms"):
            sys.modules.pop(k, None)


@pytest.mark.gpu
def test_attached_model_renders_and_trains_like_the_plain_one():
    """GPU: the attached model through this package's render() (raw parameters into the rasterizer, FusedAdam) against the same model
    left alone (PyTorch activations + covariance ops, torch.optim.Adam): same image, same parameters after three training steps; and
    the reference's own render() route -- get_covariance -> covariance_activation -- runs the HIP producer with the same values."""
    import egogaussian_amd
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import make_camera, Pipe
    from egogaussian_amd.losses import training_loss
    dev = "cuda:0"
    H, W = 48, 80
    sc = _scene(800, H, W, deg=1)
    cam, bg = make_camera(4, H, W, device=dev), torch.tensor([0.1, 0.0, 0.2], device=dev)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
    plain, fast = RefShaped(sc, dev, 1), RefShaped(sc, dev, 1)
    plain.training_setup(); fast.training_setup()
    cov_torch = plain.get_covariance(1.0)
    egogaussian_amd.attach(fast)
    assert fast.get_raw_parameters() is not None and fast.get_features_split() is not None
    # covariance_activation is the HIP producer now (off-diagonal terms cancel: the bar is relative to the largest element)
    assert float((fast.get_covariance(1.0) - cov_torch).abs().max()) <= 1e-6 * float(cov_torch.abs().max())
    imgs = []
    for m in (plain, fast):
        for it in range(3):
            out = render(cam, m, Pipe, bg)
            if it == 0:
                imgs.append(out["render"].detach().clone())
            training_loss(out["render"], gt).backward()
            m.optimizer.step(); m.optimizer.zero_grad(set_to_none=True)
    assert float((imgs[0] - imgs[1]).abs().max()) < 2e-6
    for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        pa, pb = getattr(plain, a).detach(), getattr(fast, a).detach()
        # (Adam at eps = 1e-15 turns a last-bit difference of a near-zero gradient into a full step: bound the bulk, not the worst element)
        off = ((pa - pb).abs() > 2e-4 * max(float(pa.abs().max()), 1e-6) + 1e-7).float().mean()
        assert float(off) <= 5e-3, (a, float(off))
    assert float((plain._xyz.detach() - torch.tensor(sc["xyz"], device=dev)).abs().max()) > 0    # they did move
