#!/usr/bin/env python
"""Generates tests/golden/boundary_rot.npz by IMPORTING the reference's Python (read-only, /root/reference) in the build
container -- the `fine_all` call shape of BASELINE.json's config 4:

    render(cam, gaussians, pipe, bg, rot_cov=True, accum_R=fixed_R, which_object=1, during_training=False)
        /root/reference/trainers/fine_all.py:88-93, /root/reference/gaussian_renderer/__init__.py:64-66,
        /root/reference/scene/gaussian_model.py:46-63 (build_covariance_from_scaling_rotation_w_rot)

for a seeded object + background model (30 % object Gaussians, `_is_object` stored [N,1] as the reference does), two
frames with different accumulated object rotations, the hand-mask gradient hook of fine_all.py:94 on the second one.
Captured: the exact arguments the reference's render() handed to the rasterizer, the images it got back and the
gradients that reached the reference GaussianModel's parameters.  As in make_golden.py the rasterizer behind the call is
the oracle's differentiable torch restatement (the CUDA extension is absent from /root/reference), so the fixture pins
the reference's HOST code for this call shape.  Run:  python tests/golden/make_golden_rotcov.py
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                    # noqa: E402  (stubs, CudaToCpu, the recording rasterizer)


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))


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
        from scene.gaussian_model import GaussianModel
        from gaussian_renderer import render

        rng = np.random.default_rng(4242)
        N, H, W = 400, 48, 80
        sc = make_scene(N, H, W, seed=23)
        sc["log_scale"] += math.log(3.0)
        g = GaussianModel(0)
        P_ = lambda x: torch.nn.Parameter(torch.tensor(x, dtype=torch.float32))
        g._xyz, g._features_dc = P_(sc["xyz"]), P_(sc["features"][:, :1])
        g._features_rest = P_(np.zeros((N, 0, 3), np.float32))
        g._scaling, g._rotation, g._opacity = P_(sc["log_scale"]), P_(sc["quat"] * 1.3), P_(sc["opacity_logit"])
        g._label = P_(np.zeros((N, 1), np.float32))
        is_obj = (rng.uniform(size=(N, 1)) < 0.3).astype(np.float32)
        is_obj[0, 0] = 0.0                                  # Gaussian 0 is background: the [N,1]-index quirk then shows (covariance.py)
        g._is_object = torch.tensor(is_obj)
        g._generation = torch.zeros(N, 1)

        fovx, fovy = fov_pair(H, W)

        class Cam:
            pass

        class Pipe:
            convert_SHs_python = False
            compute_cov3D_python = True
            debug = False

        bg = torch.tensor([0.0, 0.0, 0.0])                  # fine_all.py:57
        gen = torch.Generator().manual_seed(77)
        out_d = dict(N=N, H=H, W=W, fov=np.array([fovx, fovy]), bg=npy(bg), xyz=sc["xyz"], features_dc=sc["features"][:, :1],
                     log_scale=sc["log_scale"], quat=npy(g._rotation), opacity_logit=sc["opacity_logit"], is_object=is_obj)
        frames = [(rot(0.0, 0.0, 0.0), np.array([0.0, 0.0, 0.0]), rot(0.2, -0.35, 0.5), False),
                  (rot(0.03, -0.05, 0.02), np.array([0.1, -0.05, 0.3]), rot(-0.6, 0.25, 1.1), True)]
        for f, (Rc, Tc, accR, masked) in enumerate(frames):
            cam = Cam()
            cam.image_height, cam.image_width, cam.FoVx, cam.FoVy = H, W, fovx, fovy
            cam.world_view_transform = torch.tensor(getWorld2View2(Rc, Tc, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
            proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
            cam.full_proj_transform = (cam.world_view_transform.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
            cam.camera_center = cam.world_view_transform.inverse()[3, :3]
            accum_R = torch.tensor(accR, dtype=torch.float32)
            wc = torch.rand(3, H, W, generator=gen)
            hand = (torch.rand(1, H, W, generator=gen) < 0.25).float()

            mg.CAPTURE.clear()
            pkg = render(cam, g, Pipe, bg, rot_cov=True, accum_R=accum_R, which_object=1, during_training=False)
            img = pkg["render"]
            if masked:
                img.register_hook(lambda grad: grad * (1 - hand))          # fine_all.py:94
            (img * wc).sum().backward()
            c = mg.CAPTURE[0]
            k = f"f{f}_"
            out_d.update({k + "wvt": npy(cam.world_view_transform), k + "full": npy(cam.full_proj_transform),
                          k + "center": npy(cam.camera_center), k + "accum_R": npy(accum_R), k + "wc": npy(wc), k + "hand": npy(hand),
                          k + "masked": masked})
            for a in ("means3D", "opacities", "shs", "cov3D_precomp"):
                out_d[k + a] = npy(c[a]["value"]); out_d[k + a + "_rg"] = c[a]["requires_grad"]
            out_d[k + "absent"] = np.array([c[a] is None for a in ("colors_precomp", "scales", "rotations")])
            out_d.update({k + "render": npy(img), k + "depth": npy(pkg["depth"]), k + "alpha": npy(pkg["alpha"]), k + "radii": npy(pkg["radii"]),
                          k + "g_xyz": npy(g._xyz.grad), k + "g_features_dc": npy(g._features_dc.grad), k + "g_scaling": npy(g._scaling.grad),
                          k + "g_rotation": npy(g._rotation.grad), k + "g_opacity": npy(g._opacity.grad),
                          k + "g_viewspace": npy(pkg["viewspace_points"].grad)})
            for p in (g._xyz, g._features_dc, g._scaling, g._rotation, g._opacity):
                p.grad = None
        np.savez_compressed(os.path.join(HERE, "boundary_rot.npz"), **out_d)
    print("boundary_rot.npz", os.path.getsize(os.path.join(HERE, "boundary_rot.npz")), "bytes")


if __name__ == "__main__":
    main()
