"""CPU: this package's host-side mirrors against fixtures captured from the reference's own Python
(tests/golden/make_golden.py is the recipe; fixtures are data only)."""
import math
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda n: np.load(os.path.join(GOLD, n), allow_pickle=False)
T = lambda a: torch.tensor(np.asarray(a))


def test_camera_matrices_match_reference():
    from egogaussian_amd.scene_synth import projection_matrix, world_to_view, SynthCamera, ZNEAR, ZFAR
    g = load("camera.npz")
    for i in range(3):
        fovx, fovy = g[f"fov{i}"]
        assert np.array_equal(world_to_view(g[f"R{i}"], g[f"T{i}"]), g[f"w2v{i}"])          # translate = 0, scale = 1
        assert np.allclose(projection_matrix(ZNEAR, ZFAR, fovx, fovy).numpy(), g[f"P{i}"], rtol=0, atol=0)
        cam = SynthCamera(g[f"w2v{i}"], 10, 10, fovx, fovy)
        assert np.array_equal(cam.world_view_transform.numpy(), g[f"wvt{i}"])
        assert np.allclose(cam.full_proj_transform.numpy(), g[f"full{i}"], rtol=1e-6, atol=1e-7)
        assert np.allclose(cam.camera_center.numpy(), g[f"center{i}"], rtol=1e-6, atol=1e-7)


def test_sh_matches_reference_and_oracle():
    from egogaussian_amd.sh import eval_sh, RGB2SH, SH2RGB
    from oracle.oracle import Oracle
    from egogaussian_amd.scene_synth import SynthCamera
    g = load("sh.npz")
    dirs, sh = T(g["dirs"]), T(g["sh"])
    for deg in range(4):
        assert np.allclose(eval_sh(deg, sh, dirs).numpy(), g[f"eval{deg}"], rtol=1e-5, atol=1e-6)
    assert np.allclose(RGB2SH(T(g["rgb"])).numpy(), g["rgb2sh"]) and np.allclose(SH2RGB(T(g["rgb"])).numpy(), g["sh2rgb"])
    # the oracle's in-"kernel" SH: place Gaussians along the fixture directions in front of an identity camera
    keep = g["dirs"][:, 2] > 0.3
    d = g["dirs"][keep].astype(np.float64)
    n = d.shape[0]
    tan = 5.0
    cam = SynthCamera(np.eye(4), 64, 64, 2 * math.atan(tan), 2 * math.atan(tan))
    for deg in range(4):
        shs = np.transpose(g["sh"][keep], (0, 2, 1)).astype(np.float64)               # [n,16,3]
        st = Oracle(np.float64).forward(means3D=4.0 * d, opacities=np.full(n, 0.5), shs=shs, scales=np.full((n, 3), 0.05),
                                        rotations=np.tile([1.0, 0, 0, 0], (n, 1)), viewmatrix=cam.world_view_transform,
                                        projmatrix=cam.full_proj_transform, campos=np.zeros(3), bg=np.zeros(3),
                                        image_height=64, image_width=64, tanfovx=tan, tanfovy=tan, sh_degree=deg,
                                        stop_after="preprocess")
        vis = st["radii"] > 0
        assert vis.sum() >= 5
        expect = np.maximum(g[f"eval{deg}"][keep] + 0.5, 0.0)
        assert np.allclose(st["rgb"][vis], expect[vis], rtol=1e-5, atol=1e-6)
        assert np.array_equal(st["clamped"][vis].astype(bool), (g[f"eval{deg}"][keep] + 0.5 < 0)[vis])


def test_losses_match_reference():
    from egogaussian_amd.losses import l1_loss, l2_loss, ssim, psnr
    g = load("losses.npz")
    a, b = T(g["a"]).requires_grad_(True), T(g["b"])
    assert np.allclose(l1_loss(a, b).item(), g["l1"], rtol=1e-6) and np.allclose(l2_loss(a, b).item(), g["l2"], rtol=1e-6)
    s = ssim(a, b)
    assert abs(s.item() - float(g["ssim"])) < 2e-6            # separable window vs the reference's 2-D window
    s.backward()
    assert np.abs(a.grad.numpy() - g["ssim_grad_a"]).max() < 1e-7 + 1e-3 * np.abs(g["ssim_grad_a"]).max()
    assert np.allclose(ssim(a[None], b[None], size_average=False).detach().numpy(), g["ssim_per_image"], atol=2e-6)
    assert np.allclose(psnr(a[None].detach(), b[None]).numpy(), g["psnr"], rtol=1e-6)


def test_covariance_matches_reference():
    from egogaussian_amd.covariance import covariance_from_scaling_rotation, rotated_covariance_from_scaling_rotation
    g = load("covariance.npz")
    ls, q, w = T(g["log_scale"]).requires_grad_(True), T(g["quat"]).requires_grad_(True), T(g["wcov"])
    cov = covariance_from_scaling_rotation(torch.exp(ls), 1.0, q)
    # (off-diagonal terms cancel: on other host CPUs -- the GPU box's -- the last bits of small elements move; the bar is relative to the largest element)
    assert np.allclose(cov.detach().numpy(), g["cov"], rtol=1e-5, atol=1e-6 * float(np.abs(g["cov"]).max()))
    (cov * w).sum().backward()
    assert np.allclose(ls.grad.numpy(), g["g_scaling"], rtol=1e-4, atol=1e-7)
    assert np.allclose(q.grad.numpy(), g["g_rotation"], rtol=1e-4, atol=1e-6)
    assert np.allclose(covariance_from_scaling_rotation(torch.exp(ls), 2.0, q).detach().numpy(), g["cov_mod2"], rtol=1e-5)
    ls.grad = None; q.grad = None
    R, is_obj = T(g["accum_R"]), T(g["is_object"])
    rc = rotated_covariance_from_scaling_rotation(torch.exp(ls), 1.0, q, R, is_obj, 1)
    assert np.allclose(rc.detach().numpy(), g["rcov"], rtol=1e-5, atol=1e-6 * float(np.abs(g["rcov"]).max()))
    (rc * w).sum().backward()
    assert np.allclose(ls.grad.numpy(), g["rg_scaling"], rtol=1e-4, atol=1e-7)
    assert np.allclose(q.grad.numpy(), g["rg_rotation"], rtol=1e-4, atol=1e-6)
    assert np.allclose(rotated_covariance_from_scaling_rotation(torch.exp(ls), 1.0, q, torch.eye(3), is_obj, 1).detach().numpy(),
                       g["rcov_identity"], rtol=1e-5, atol=1e-6 * float(np.abs(g["rcov_identity"]).max()))
    assert np.allclose(rotated_covariance_from_scaling_rotation(torch.exp(ls), 1.0, q, R, is_obj, None).detach().numpy(),
                       g["rcov_all"], rtol=1e-5, atol=1e-6 * float(np.abs(g["rcov_all"]).max()))


def _model_from_boundary(g, device="cpu"):
    from egogaussian_amd.scene_synth import SynthGaussians, SynthCamera
    scene = dict(xyz=g["xyz"], features=g["features_dc"], log_scale=g["log_scale"], quat=g["quat"],
                 opacity_logit=g["opacity_logit"])
    pc = SynthGaussians(scene, device=device)
    pc._label = torch.tensor(g["label"], device=device).requires_grad_(True)
    pc._is_object = torch.tensor(g["is_object"], device=device)
    H, W = int(g["H"]), int(g["W"])
    cam = SynthCamera(g["wvt"].T, H, W, float(g["fov"][0]), float(g["fov"][1]), device=device)
    return pc, cam


def test_boundary_arguments_match_reference_render():
    """My render()/get_render_label() hand the rasterizer exactly what the reference's do (mode 1 and mode 2)."""
    from egogaussian_amd.renderer import get_raster_settings, gaussians_to_label_rendervar
    g = load("boundary.npz")
    pc, cam = _model_from_boundary(g)
    assert np.array_equal(cam.world_view_transform.numpy(), g["m1_viewmatrix"])
    assert np.allclose(cam.full_proj_transform.numpy(), g["m1_projmatrix"], rtol=1e-6, atol=1e-7)
    assert np.allclose(cam.camera_center.numpy(), g["m1_campos"], rtol=1e-6, atol=1e-7)
    rs = get_raster_settings(cam, pc, torch.tensor(g["bg"]))
    assert np.allclose([rs.tanfovx, rs.tanfovy], g["m1_tanfov"], rtol=1e-12) and rs.sh_degree == int(g["m1_sh_degree"])
    assert rs.scale_modifier == float(g["m1_scale_modifier"]) and [rs.prefiltered, rs.debug] == list(g["m1_flags"])
    assert rs._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                          "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    # mode 1 tensors
    assert np.array_equal(pc.get_xyz.detach().numpy(), g["m1_means3D"])
    assert np.allclose(pc.get_opacity.detach().numpy(), g["m1_opacities"], rtol=1e-6)
    assert np.array_equal(pc.get_features.detach().numpy(), g["m1_shs"])
    assert np.allclose(pc.get_covariance(1.0).detach().numpy(), g["m1_cov3D_precomp"], rtol=1e-5, atol=1e-10)
    assert all(g[f"m1_{k}_rg"] for k in ("means3D", "means2D", "opacities", "shs", "cov3D_precomp")) and g["m1_absent"].all()
    # mode 2 tensors
    rv = gaussians_to_label_rendervar(pc)
    assert np.array_equal(rv["means3D"].numpy(), g["m2_means3D"]) and not rv["means3D"].requires_grad
    assert np.allclose(rv["scales"].numpy(), g["m2_scales"], rtol=1e-6)
    assert np.allclose(rv["rotations"].numpy(), g["m2_rotations"], rtol=1e-6, atol=1e-7)
    assert np.allclose(rv["opacities"].numpy(), g["m2_opacities"], rtol=1e-6)
    assert np.array_equal(rv["colors_precomp"].detach().numpy(), g["m2_colors_precomp"]) and rv["colors_precomp"].requires_grad
    assert not any(g[f"m2_{k}_rg"] for k in ("means3D", "means2D", "opacities", "scales", "rotations")) and g["m2_absent"].all()


def test_oracle_reproduces_boundary_images():
    """The C oracle, fed the captured mode-1 / mode-2 arguments, reproduces the images the reference's render() got
    back (those came from the independent torch restatement)."""
    from oracle.oracle import Oracle
    g = load("boundary.npz")
    H, W = int(g["H"]), int(g["W"])
    common = dict(viewmatrix=g["m1_viewmatrix"], projmatrix=g["m1_projmatrix"], campos=g["m1_campos"], bg=g["bg"],
                  image_height=H, image_width=W, tanfovx=float(g["m1_tanfov"][0]), tanfovy=float(g["m1_tanfov"][1]))
    st = Oracle(np.float32).forward(means3D=g["m1_means3D"], opacities=g["m1_opacities"], shs=g["m1_shs"],
                                    cov3D_precomp=g["m1_cov3D_precomp"], **common)
    assert np.array_equal(st["radii"], g["m1_radii"]) and np.array_equal(st["radii"] > 0, g["m1_visibility"])
    for name, key in (("color", "m1_render"), ("depth", "m1_depth"), ("alpha", "m1_alpha")):
        assert np.abs(st[name] - g[key]).max() < 2e-5 * max(1.0, np.abs(g[key]).max()), name
    st2 = Oracle(np.float32).forward(means3D=g["m2_means3D"], opacities=g["m2_opacities"], colors_precomp=g["m2_colors_precomp"],
                                     scales=g["m2_scales"], rotations=g["m2_rotations"], **common)
    assert np.abs(st2["color"] - g["m2_label_render"]).max() < 2e-5 * max(1.0, np.abs(g["m2_label_render"]).max())
    gr = Oracle(np.float32).backward(st2, g["wc"])
    assert np.abs(gr["dL_dcolors_precomp"].sum(1, keepdims=True) - g["m2_g_label"]).max() < 1e-4 * np.abs(g["m2_g_label"]).max()


@pytest.mark.gpu
def test_render_through_hip_matches_reference_end_to_end():
    """GPU: my render()/get_render_label() + HIP rasterizer reproduce the images and the PARAMETER gradients the
    reference's render() + GaussianModel produced (with the oracle standing in for the absent CUDA kernel)."""
    from egogaussian_amd.renderer import render, get_render_label
    from egogaussian_amd.scene_synth import Pipe
    g = load("boundary.npz")
    dev = "cuda:0"
    pc, cam = _model_from_boundary(g, dev)
    bg = torch.tensor(g["bg"], device=dev)
    out = render(cam, pc, Pipe, bg)
    wc, wd, wa = [torch.tensor(g[k], device=dev) for k in ("wc", "wd", "wa")]
    ((out["render"] * wc).sum() + (out["depth"] * wd).sum() + (out["alpha"] * wa).sum()).backward()
    close = lambda a, b, tol=1e-4: np.abs(a.detach().cpu().numpy() - b).max() <= tol * max(np.abs(b).max(), 1e-12)
    assert np.array_equal(out["radii"].cpu().numpy(), g["m1_radii"])
    assert close(out["render"], g["m1_render"]) and close(out["depth"], g["m1_depth"]) and close(out["alpha"], g["m1_alpha"])
    assert close(pc._xyz.grad, g["m1_g_xyz"]) and close(pc._features_dc.grad, g["m1_g_features_dc"])
    assert close(pc._scaling.grad, g["m1_g_scaling"]) and close(pc._rotation.grad, g["m1_g_rotation"])
    assert close(pc._opacity.grad, g["m1_g_opacity"]) and close(out["viewspace_points"].grad, g["m1_g_viewspace"])
    lab = get_render_label(cam, pc, bg)
    (lab * wc).sum().backward()
    assert close(lab, g["m2_label_render"]) and close(pc._label.grad, g["m2_g_label"])


# ---- config 4 call shape: render(..., rot_cov=True, accum_R, which_object=1) -- fixture boundary_rot.npz ---------------
def _model_from_boundary_rot(g, device="cpu", fused=True):
    from egogaussian_amd.scene_synth import SynthGaussians
    scene = dict(xyz=g["xyz"], features=g["features_dc"], log_scale=g["log_scale"], quat=g["quat"], opacity_logit=g["opacity_logit"])
    pc = SynthGaussians(scene, device=device, fused=fused)
    pc._is_object = torch.tensor(g["is_object"], device=device)           # [N,1], as the reference stores it
    return pc


def _cam_from_boundary_rot(g, f, device="cpu"):
    from egogaussian_amd.scene_synth import SynthCamera
    return SynthCamera(g[f"f{f}_wvt"].T, int(g["H"]), int(g["W"]), float(g["fov"][0]), float(g["fov"][1]), device=device)


def test_rot_cov_arguments_match_reference_render():
    """My render(rot_cov=True, accum_R, which_object=1) hands the rasterizer the covariance the reference's does
    (/root/reference/scene/gaussian_model.py:46-63 incl. the [N,1]-index quirk on Gaussian 0), and the C oracle reproduces the
    images the reference got back for it."""
    from oracle.oracle import Oracle
    g = load("boundary_rot.npz")
    pc = _model_from_boundary_rot(g)
    assert g["is_object"][0, 0] == 0 and g["is_object"].sum() > 50
    for f in range(2):
        cam = _cam_from_boundary_rot(g, f)
        k = f"f{f}_"
        assert np.array_equal(cam.world_view_transform.numpy(), g[k + "wvt"])
        assert np.allclose(cam.full_proj_transform.numpy(), g[k + "full"], rtol=1e-6, atol=1e-7)
        cov = pc.get_rotated_covariance(T(g[k + "accum_R"]), 1, False, 1.0)
        assert np.allclose(cov.detach().numpy(), g[k + "cov3D_precomp"], rtol=1e-5, atol=1e-10)
        assert not np.allclose(pc.get_covariance(1.0).detach().numpy()[0], g[k + "cov3D_precomp"][0], rtol=1e-3)   # row 0 IS rotated
        assert all(g[k + a + "_rg"] for a in ("means3D", "opacities", "shs", "cov3D_precomp")) and g[k + "absent"].all()
        H, W = int(g["H"]), int(g["W"])
        st = Oracle(np.float32).forward(means3D=g[k + "means3D"], opacities=g[k + "opacities"], shs=g[k + "shs"],
                                        cov3D_precomp=g[k + "cov3D_precomp"], viewmatrix=g[k + "wvt"], projmatrix=g[k + "full"],
                                        campos=g[k + "center"], bg=g["bg"], image_height=H, image_width=W,
                                        tanfovx=math.tan(float(g["fov"][0]) / 2), tanfovy=math.tan(float(g["fov"][1]) / 2))
        assert np.array_equal(st["radii"], g[k + "radii"])
        for name, key in (("color", "render"), ("depth", "depth"), ("alpha", "alpha")):
            assert np.abs(st[name] - g[k + key]).max() < 2e-5 * max(1.0, np.abs(g[k + key]).max()), name


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["rasterizer", "producer", "torch"])
def test_rot_cov_render_through_hip_matches_reference_end_to_end(path):
    """GPU, BASELINE.json config 4's call shape (/root/reference/trainers/fine_all.py:88-94): render(rot_cov=True, accum_R,
    which_object=1) reproduces the images and the PARAMETER gradients the reference's render() + GaussianModel produced, including the
    hand-mask gradient hook on the second frame and the [N,1]-index quirk on row 0 -- with the object rotation applied inside the
    rasterizer (egs_object_rotation, raw parameters), by the HIP covariance producer (cov3d.hip -> cov3D_precomp), or by the PyTorch
    mirror of the reference's ops."""
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import Pipe
    g = load("boundary_rot.npz")
    dev = "cuda:0"
    pc = _model_from_boundary_rot(g, dev, fused=path != "torch")
    pc.rotate_in_rasterizer = path == "rasterizer"
    bg = torch.tensor(g["bg"], device=dev)
    close = lambda a, b, tol=1e-4: np.abs(a.detach().cpu().numpy() - b).max() <= tol * max(np.abs(b).max(), 1e-12)
    for f in range(2):
        k = f"f{f}_"
        cam = _cam_from_boundary_rot(g, f, dev)
        out = render(cam, pc, Pipe, bg, rot_cov=True, accum_R=torch.tensor(g[k + "accum_R"], device=dev), which_object=1, during_training=False)
        img = out["render"]
        if bool(g[k + "masked"]):
            hand = torch.tensor(g[k + "hand"], device=dev)
            img.register_hook(lambda grad: grad * (1 - hand))
        (img * torch.tensor(g[k + "wc"], device=dev)).sum().backward()
        assert np.array_equal(out["radii"].cpu().numpy(), g[k + "radii"])
        assert close(img, g[k + "render"]) and close(out["depth"], g[k + "depth"]) and close(out["alpha"], g[k + "alpha"])
        for p, name in ((pc._xyz, "g_xyz"), (pc._features_dc, "g_features_dc"), (pc._scaling, "g_scaling"), (pc._rotation, "g_rotation"),
                        (pc._opacity, "g_opacity"), (out["viewspace_points"], "g_viewspace")):
            assert close(p.grad, g[k + name]), (f, name)
            p.grad = None


# ---- the object-pose training step, override_color, convert_SHs_python -- fixture boundary_train.npz (make_golden_training.py) ----
def _model_from_train(g, k, device="cpu", fused=True, sh_degree=0):
    from egogaussian_amd.scene_synth import SynthGaussians
    scene = dict(xyz=g[k + "xyz"], features=g[k + "features"], log_scale=g[k + "log_scale"], quat=g[k + "quat"], opacity_logit=g[k + "opacity_logit"])
    pc = SynthGaussians(scene, device=device, fused=fused, sh_degree=sh_degree)
    pc._is_object = torch.tensor(g[k + "is_object"], device=device)
    return pc


def _cam_from_train(g, k, device="cpu"):
    from egogaussian_amd.scene_synth import SynthCamera
    return SynthCamera(g[k + "wvt"].T, int(g["H"]), int(g["W"]), float(g["fov"][0]), float(g["fov"][1]), device=device)


class _TrainablePose(torch.nn.Module):
    """What a trainer keeps in `gaussians.trainable_object_move`, written from the contract this package relies on and nothing else:
    a (3, 2) parameter `obj_rotation_6d` whose columns are orthonormalised in order (third axis = their cross product) to give a
    rotation R, and `rot_L(L) = R @ L`.  The values it must reproduce are the reference's, captured in boundary_train.npz
    (pose_arg_cov3D_precomp, pose_g_rot6d)."""

    def __init__(self, six):
        super().__init__()
        self.obj_rotation_6d = torch.nn.Parameter(six.clone())

    def rot_L(self, L):
        u, v = self.obj_rotation_6d.unbind(dim=1)
        e0 = u / u.norm()
        w = v - torch.dot(e0, v) * e0
        e1 = w / w.norm()
        return torch.stack((e0, e1, torch.linalg.cross(e0, e1)), dim=1) @ L


def _object_move(g, device="cpu"):
    return _TrainablePose(torch.tensor(g["pose_rot6d"], device=device))


def test_trainable_pose_stand_in_is_a_rotation():
    """The stand-in's rot_L(I) -- the only thing adapter / scene_synth ask of a trainable pose -- is a proper rotation that spans the
    captured 6-D parameter's first column; that it is the REFERENCE's rotation is what the covariance test below pins."""
    g = load("boundary_train.npz")
    M = _object_move(g).rot_L(torch.eye(3))
    assert M.shape == (3, 3) and torch.allclose(M @ M.t(), torch.eye(3), atol=1e-6) and abs(float(torch.det(M.detach())) - 1.0) < 1e-5
    six = torch.tensor(g["pose_rot6d"])
    assert torch.allclose(M[:, 0], six[:, 0] / six[:, 0].norm(), atol=1e-6)
    L = torch.randn(7, 3, 3, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(_object_move(g).rot_L(L), M @ L)


def test_training_call_shapes_hand_the_rasterizer_what_the_reference_does():
    """CPU (PyTorch mirror): the arguments render() gives the rasterizer for the three call shapes of boundary_train.npz equal the
    reference's -- the trainable-rotation covariance (scene/gaussian_model.py:55-56 through trainable_object_move.rot_L), the
    override colours, the Python-evaluated SH colours -- and the C oracle reproduces the captured images from them."""
    from oracle.oracle import Oracle
    from egogaussian_amd.sh import eval_sh
    g = load("boundary_train.npz")
    H, W = int(g["H"]), int(g["W"])
    tan = (math.tan(float(g["fov"][0]) / 2), math.tan(float(g["fov"][1]) / 2))
    # pose: covariance with the trainable rotation on top of accum_R, [N,1]-index quirk included
    pc = _model_from_train(g, "pose_", fused=False)
    pc.trainable_object_move = _object_move(g)
    cov = pc.get_rotated_covariance(T(g["pose_accum_R"]), 1, True, 1.0)
    assert np.allclose(cov.detach().numpy(), g["pose_arg_cov3D_precomp"], rtol=2e-5, atol=1e-9)
    cov_frozen = pc.get_rotated_covariance(T(g["pose_accum_R"]), 1, False, 1.0)
    assert not np.allclose(cov_frozen.detach().numpy(), g["pose_arg_cov3D_precomp"], rtol=1e-3)      # the trainable rotation IS in play
    assert bool(g["pose_arg_cov3D_precomp_rg"]) and bool(g["pose_arg_scales_absent"]) and bool(g["pose_arg_colors_precomp_absent"])
    # override_color: passed through as colors_precomp, no SH
    assert np.array_equal(g["override_arg_colors_precomp"], g["override_override_color"]) and bool(g["override_arg_shs_absent"])
    # convert_SHs_python: clamp_min(eval_sh(deg, features^T, normalised xyz - campos) + 0.5, 0)   (gaussian_renderer/__init__.py:78-84)
    pcs = _model_from_train(g, "shs_python_", fused=False, sh_degree=2)
    cam = _cam_from_train(g, "shs_python_")
    feats = pcs.get_features
    dirs = pcs.get_xyz - cam.camera_center.repeat(feats.shape[0], 1)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    cols = torch.clamp_min(eval_sh(2, feats.transpose(1, 2).view(-1, 3, 9), dirs) + 0.5, 0.0)
    assert int(g["shs_python_arg_sh_degree"]) == 2 and bool(g["shs_python_arg_shs_absent"])
    assert np.allclose(cols.detach().numpy(), g["shs_python_arg_colors_precomp"], rtol=1e-5, atol=1e-6)
    for k, bgk in (("pose_", None), ("override_", "override_bg"), ("shs_python_", "shs_python_bg")):
        kw = dict(colors_precomp=g[k + "arg_colors_precomp"]) if not bool(g[k + "arg_colors_precomp_absent"]) else dict(shs=g[k + "arg_shs"])
        st = Oracle(np.float32).forward(means3D=g[k + "arg_means3D"], opacities=g[k + "arg_opacities"], cov3D_precomp=g[k + "arg_cov3D_precomp"],
                                        viewmatrix=g[k + "wvt"], projmatrix=g[k + "full"], campos=g[k + "center"],
                                        bg=np.zeros(3, np.float32) if bgk is None else g[bgk], image_height=H, image_width=W, tanfovx=tan[0], tanfovy=tan[1], **kw)
        assert np.array_equal(st["radii"], g[k + "radii"]), k
        for name, key in (("color", "render"), ("depth", "depth"), ("alpha", "alpha")):
            assert np.abs(st[name] - g[k + key]).max() < 2e-5 * max(1.0, np.abs(g[k + key]).max()), (k, name)


def _pose_loss(image, alpha, g, dev, fused_loss):
    """The object-pose stages' loss (/root/reference/trainers/fine_obj.py:136-149) with both hand-mask hooks."""
    from egogaussian_amd.losses import l1_loss, l2_loss, ssim
    hand, obj = torch.tensor(g["pose_hand"], device=dev), torch.tensor(g["pose_obj_mask"], device=dev)
    gt = torch.tensor(g["pose_gt"], device=dev) * obj
    lam, l1a, l2a = [float(x) for x in g["pose_lambdas"]]
    alpha.register_hook(lambda grad: grad * (1 - hand))
    if fused_loss:
        from egogaussian_amd.fused import l1_ssim_loss
        img_loss = l1_ssim_loss(image, gt, lam, grad_gate=(1 - hand)[0])          # the hook on the image, inside the fused loss's backward
    else:
        image.register_hook(lambda grad: grad * (1 - hand))
        img_loss = (1.0 - lam) * l1_loss(gt, image) + lam * (1.0 - ssim(gt, image))
    return img_loss + l1a * l1_loss(obj, alpha) + l2a * l2_loss(obj, alpha)


@pytest.mark.gpu
@pytest.mark.parametrize("fused_loss", [False, True], ids=["torch-loss", "hip-loss"])
@pytest.mark.parametrize("path", ["rasterizer", "producer", "torch"])
def test_pose_training_step_through_hip_matches_reference_end_to_end(path, fused_loss):
    """GPU: the training step of the object-pose stages -- render(rot_cov=True, accum_R, which_object=1, during_training=True) with
    gaussians.trainable_object_move set, hand-mask hooks on image AND alpha, image + L1 + L2 alpha losses -- reproduces the image,
    the alpha map, the loss, every parameter gradient and d loss / d obj_rotation_6d that the reference's render() + GaussianModel +
    ObjectMove produced.  `rasterizer`: the model prefers the in-rasterizer object rotation, which cannot return the rotation's
    gradient, so it must fall back to the covariance producer by itself; `producer`: cov3d.hip; `torch`: the PyTorch mirror."""
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import Pipe
    g = load("boundary_train.npz")
    dev = "cuda:0"
    pc = _model_from_train(g, "pose_", dev, fused=path != "torch")
    pc.rotate_in_rasterizer = path == "rasterizer"
    pc.trainable_object_move = tom = _object_move(g, dev)
    cam = _cam_from_train(g, "pose_", dev)
    out = render(cam, pc, Pipe, torch.zeros(3, device=dev), rot_cov=True, accum_R=torch.tensor(g["pose_accum_R"], device=dev), which_object=1,
                 during_training=True)
    loss = _pose_loss(out["render"], out["alpha"], g, dev, fused_loss)
    loss.backward()
    close = lambda a, b, tol=1e-4: np.abs(a.detach().cpu().numpy() - b).max() <= tol * max(np.abs(b).max(), 1e-12)
    assert np.array_equal(out["radii"].cpu().numpy(), g["pose_radii"])
    assert close(out["render"], g["pose_render"]) and close(out["alpha"], g["pose_alpha"]) and close(out["depth"], g["pose_depth"])
    assert abs(float(loss.detach()) - float(g["pose_loss"])) <= 1e-5 * abs(float(g["pose_loss"]))
    for p, name in ((pc._xyz, "g_xyz"), (pc._features_dc, "g_features_dc"), (pc._scaling, "g_scaling"), (pc._rotation, "g_rotation"),
                    (pc._opacity, "g_opacity"), (out["viewspace_points"], "g_viewspace")):
        assert close(p.grad, g["pose_" + name], 2e-4), name
    assert tom.obj_rotation_6d.grad is not None and float(np.abs(g["pose_g_rot6d"]).max()) > 0
    assert close(tom.obj_rotation_6d.grad, g["pose_g_rot6d"], 2e-4)


@pytest.mark.gpu
def test_override_color_and_python_sh_through_hip_match_reference():
    """GPU: render(override_color=c) and render() with pipe.convert_SHs_python (/root/reference/gaussian_renderer/__init__.py:75-87)
    reproduce the reference's images and gradients -- to the override colours; to the features and, through the view directions of
    the Python SH evaluation, to the positions."""
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import Pipe
    g = load("boundary_train.npz")
    dev = "cuda:0"
    close = lambda a, b, tol=1e-4: np.abs(a.detach().cpu().numpy() - b).max() <= tol * max(np.abs(b).max(), 1e-12)
    pc, cam = _model_from_train(g, "override_", dev), _cam_from_train(g, "override_", dev)
    oc = torch.tensor(g["override_override_color"], device=dev, requires_grad=True)
    out = render(cam, pc, Pipe, torch.tensor(g["override_bg"], device=dev), override_color=oc)
    (out["render"] * torch.tensor(g["override_wc"], device=dev)).sum().backward()
    assert np.array_equal(out["radii"].cpu().numpy(), g["override_radii"]) and close(out["render"], g["override_render"])
    assert close(oc.grad, g["override_g_override_color"]) and close(pc._xyz.grad, g["override_g_xyz"]) and close(pc._opacity.grad, g["override_g_opacity"])
    assert close(pc._scaling.grad, g["override_g_scaling"]) and close(pc._rotation.grad, g["override_g_rotation"])
    assert pc._features_dc.grad is None and g["override_g_features_dc"].size == 0        # the colours bypass the features, in both

    class PyPipe(Pipe):
        convert_SHs_python = True
    pc, cam = _model_from_train(g, "shs_python_", dev, sh_degree=2), _cam_from_train(g, "shs_python_", dev)
    out = render(cam, pc, PyPipe, torch.tensor(g["shs_python_bg"], device=dev))
    (out["render"] * torch.tensor(g["shs_python_wc"], device=dev)).sum().backward()
    assert np.array_equal(out["radii"].cpu().numpy(), g["shs_python_radii"]) and close(out["render"], g["shs_python_render"])
    for p, name in ((pc._xyz, "g_xyz"), (pc._features_dc, "g_features_dc"), (pc._features_rest, "g_features_rest"), (pc._scaling, "g_scaling"),
                    (pc._rotation, "g_rotation"), (pc._opacity, "g_opacity")):
        assert close(p.grad, g["shs_python_" + name], 2e-4), name
