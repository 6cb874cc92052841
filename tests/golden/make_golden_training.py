#!/usr/bin/env python
"""Generates tests/golden/boundary_train.npz by IMPORTING the reference's Python (read-only, /root/reference) in the build
container.  Three call shapes of the reference's render() that the other fixtures do not cover:

 pose   the object-pose stages' training step (/root/reference/trainers/coarse_obj_pose.py:239-260, fine_obj.py:128-151):
            gaussians.trainable_object_move = ObjectMove() with a non-identity obj_rotation_6d   (utils/geometry_utils.py:14-28)
            render(cam, gaussians, pipe, bg, rot_cov=True, accum_R=fixed_R, which_object=1, during_training=True)
            render_image.register_hook(grad * (1 - hand_mask)); render_alpha.register_hook(grad * (1 - hand_mask))
            loss = (1 - l) L1(gt * obj_mask, image) + l (1 - ssim) + l1a * L1(obj_mask, alpha) + l2a * L2(obj_mask, alpha)
        -- the covariance goes through trainable_object_move.rot_L (scene/gaussian_model.py:55-56), so the loss reaches
        obj_rotation_6d; captured: rasterizer arguments, image, alpha, every parameter gradient, obj_rotation_6d.grad.
 override   render(..., override_color=c)            (gaussian_renderer/__init__.py:75-77): colours given, gradient to c.
 shs_python render() with pipe.convert_SHs_python    (gaussian_renderer/__init__.py:78-84): SH degree 2 evaluated in Python from
            the view directions; gradient reaches the features AND, through the directions, the positions.

As in make_golden.py the rasterizer behind the calls is the oracle's differentiable torch restatement (the CUDA extension is absent
from /root/reference): the fixture pins the reference's HOST code for these call shapes.  Run: python tests/golden/make_golden_training.py
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                    # noqa: E402
from make_golden_rotcov import rot                          # noqa: E402


def main():
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("pytorch3d")
    mg.stub("pytorch3d.transforms", euler_angles_to_matrix=None)
    mg.stub("simple_knn")
    mg.stub("simple_knn._C", distCUDA2=lambda pts: torch.full((pts.shape[0],), 1e-3))
    mg.install_recording_rasterizer()
    from egogaussian_amd.scene_synth import make_scene, fov_pair
    npy = mg.npy

    with mg.CudaToCpu():
        from utils.graphics_utils import getWorld2View2, getProjectionMatrix
        from utils.geometry_utils import ObjectMove, matrix_to_rot6d
        from utils.loss_utils import l1_loss, l2_loss, ssim
        from scene.gaussian_model import GaussianModel
        from gaussian_renderer import render

        rng = np.random.default_rng(777)
        N, H, W = 400, 48, 80
        fovx, fovy = fov_pair(H, W)
        P_ = lambda x: torch.nn.Parameter(torch.tensor(x, dtype=torch.float32))

        def model(seed, sh_degree):
            sc = make_scene(N, H, W, seed=seed, sh_degree=sh_degree)
            sc["log_scale"] += math.log(3.0)
            g = GaussianModel(sh_degree)
            g.active_sh_degree = sh_degree
            g._xyz, g._features_dc = P_(sc["xyz"]), P_(sc["features"][:, :1])
            g._features_rest = P_(sc["features"][:, 1:])
            g._scaling, g._rotation, g._opacity = P_(sc["log_scale"]), P_(sc["quat"] * 0.8), P_(sc["opacity_logit"])
            g._label = P_(np.zeros((N, 1), np.float32))
            g._is_object = torch.zeros(N, 1)
            g._generation = torch.zeros(N, 1)
            return g, sc

        def camera(Rc, Tc):
            class Cam:
                pass
            cam = Cam()
            cam.image_height, cam.image_width, cam.FoVx, cam.FoVy = H, W, fovx, fovy
            cam.world_view_transform = torch.tensor(getWorld2View2(Rc, Tc, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
            proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
            cam.full_proj_transform = (cam.world_view_transform.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
            cam.camera_center = cam.world_view_transform.inverse()[3, :3]
            return cam

        class Pipe:
            convert_SHs_python = False
            compute_cov3D_python = True
            debug = False

        class PyPipe(Pipe):
            convert_SHs_python = True

        out_d = dict(N=N, H=H, W=W, fov=np.array([fovx, fovy]))
        gen = torch.Generator().manual_seed(99)
        PARAMS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")

        def record(k, g, sc, cam, pkg, extra):
            c = mg.CAPTURE[0]
            out_d.update({k + "xyz": sc["xyz"], k + "features": sc["features"], k + "log_scale": sc["log_scale"], k + "quat": npy(g._rotation),
                          k + "opacity_logit": sc["opacity_logit"], k + "is_object": npy(g._is_object),
                          k + "wvt": npy(cam.world_view_transform), k + "full": npy(cam.full_proj_transform), k + "center": npy(cam.camera_center)})
            for a in ("means3D", "opacities", "shs", "colors_precomp", "cov3D_precomp", "scales", "rotations"):
                out_d[k + "arg_" + a + "_absent"] = c[a] is None
                if c[a] is not None:
                    out_d[k + "arg_" + a] = npy(c[a]["value"]); out_d[k + "arg_" + a + "_rg"] = c[a]["requires_grad"]
            out_d[k + "arg_sh_degree"] = c["settings"]["sh_degree"]
            out_d.update({k + "render": npy(pkg["render"]), k + "depth": npy(pkg["depth"]), k + "alpha": npy(pkg["alpha"]), k + "radii": npy(pkg["radii"]),
                          k + "g_viewspace": npy(pkg["viewspace_points"].grad)})
            for p in PARAMS:
                gr = getattr(g, p).grad
                out_d[k + "g" + p] = npy(gr) if gr is not None else np.zeros(0, np.float32)
            out_d.update({k + n: npy(v) for n, v in extra.items()})

        # ---- pose: during_training=True + both hooks + image and alpha losses ----------------------------------------------
        g, sc = model(31, 0)
        is_obj = (rng.uniform(size=(N, 1)) < 0.35).astype(np.float32)
        is_obj[0, 0] = 0.0
        g._is_object = torch.tensor(is_obj)
        g.trainable_object_move = ObjectMove()
        with torch.no_grad():
            g.trainable_object_move.obj_rotation_6d.copy_(matrix_to_rot6d(torch.tensor(rot(0.15, -0.1, 0.25), dtype=torch.float32)) * 1.1)   # (not orthonormal on purpose)
        cam = camera(rot(0.02, 0.04, -0.03), np.array([0.05, 0.1, 0.2]))
        accum_R = torch.tensor(rot(-0.4, 0.3, 0.8), dtype=torch.float32)
        bg = torch.tensor([0.0, 0.0, 0.0])
        gt = torch.rand(3, H, W, generator=gen)
        hand = (torch.rand(1, H, W, generator=gen) < 0.25).float()
        obj_mask = (torch.rand(1, H, W, generator=gen) < 0.6).float()
        lam, l1a, l2a = 0.2, 0.3, 0.5                           # (arguments/__init__.py:160-163,185-188: 0.1-0.2, 0.0, 0.2-0.5; l1a non-zero here so that the term is exercised)
        mg.CAPTURE.clear()
        pkg = render(cam, g, Pipe, bg, rot_cov=True, accum_R=accum_R, which_object=1, during_training=True)
        image, alpha = pkg["render"], pkg["alpha"]
        image.register_hook(lambda grad: grad * (1 - hand))
        alpha.register_hook(lambda grad: grad * (1 - hand))
        gt_m = torch.mul(gt, obj_mask)
        loss = (1.0 - lam) * l1_loss(gt_m, image) + lam * (1.0 - ssim(gt_m, image)) + l1a * l1_loss(obj_mask, alpha) + l2a * l2_loss(obj_mask, alpha)
        loss.backward()
        tom = g.trainable_object_move
        record("pose_", g, sc, cam, pkg, dict(accum_R=accum_R, gt=gt, hand=hand, obj_mask=obj_mask, lambdas=np.array([lam, l1a, l2a]), loss=loss,
                                              rot6d=tom.obj_rotation_6d, g_rot6d=tom.obj_rotation_6d.grad,
                                              g_translation=tom.obj_translation.grad if tom.obj_translation.grad is not None else torch.zeros(3)))

        # ---- override_color -----------------------------------------------------------------------------------------------
        g, sc = model(32, 0)
        cam = camera(rot(-0.03, 0.02, 0.05), np.array([-0.1, 0.05, 0.1]))
        bg = torch.tensor([0.2, 0.1, 0.3])
        oc = torch.rand(N, 3, generator=gen).requires_grad_(True)
        wc = torch.rand(3, H, W, generator=gen)
        mg.CAPTURE.clear()
        pkg = render(cam, g, Pipe, bg, override_color=oc)
        (pkg["render"] * wc).sum().backward()
        record("override_", g, sc, cam, pkg, dict(bg=bg, wc=wc, override_color=oc, g_override_color=oc.grad))

        # ---- convert_SHs_python (degree 2) --------------------------------------------------------------------------------
        g, sc = model(33, 2)
        cam = camera(rot(0.04, -0.02, 0.01), np.array([0.0, -0.1, 0.15]))
        bg = torch.tensor([0.1, 0.2, 0.3])
        wc = torch.rand(3, H, W, generator=gen)
        mg.CAPTURE.clear()
        pkg = render(cam, g, PyPipe, bg)
        (pkg["render"] * wc).sum().backward()
        record("shs_python_", g, sc, cam, pkg, dict(bg=bg, wc=wc))
        np.savez_compressed(os.path.join(HERE, "boundary_train.npz"), **out_d)
    print("boundary_train.npz", os.path.getsize(os.path.join(HERE, "boundary_train.npz")), "bytes")


if __name__ == "__main__":
    main()
