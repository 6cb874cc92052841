#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference's Python (read-only, /root/reference) in the build
container.  The reference cannot travel to the GPU box, so only the captured tensors are committed; this script is
the recipe.  Run:  python tests/golden/make_golden.py

What is pinned
  camera.npz       getWorld2View2 / getProjectionMatrix and the Camera transform chain  (utils/graphics_utils.py:38-71,
                   scene/cameras.py:67-70)
  covariance.npz   GaussianModel.get_covariance / get_rotated_covariance outputs and autograd gradients
                   (scene/gaussian_model.py:29-33,46-63,167-171; utils/general_utils.py:110-156)
  sh.npz           eval_sh degrees 0-3, RGB2SH, SH2RGB  (utils/sh_utils.py:57-118)
  losses.npz       l1_loss, l2_loss, ssim, psnr  (utils/loss_utils.py:57-107, utils/image_utils.py:14-19)
  boundary.npz     the exact arguments the reference's render() and get_render_label() hand to the rasterizer
                   (gaussian_renderer/__init__.py:18-107, render_helper.py:7-64) for a seeded model and camera, the
                   images the rasterizer returned, and the gradients that reached the reference GaussianModel's
                   parameters.  The rasterizer behind the reference's call here is the ORACLE's differentiable torch
                   restatement (oracle/raster_torch.py) -- the CUDA extension is absent -- so this fixture pins the
                   reference's HOST code (argument assembly, covariance, activations, autograd plumbing), not the kernel.
"""
import math
import os
import sys
import types

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


class CudaToCpu(TorchFunctionMode):
    """The reference hard-codes device='cuda' / .cuda(); reroute to CPU for the capture."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if str(kwargs.get("device", "")).startswith("cuda"):
            kwargs["device"] = "cpu"
        n = getattr(func, "__name__", "")
        if n == "cuda":
            return args[0]
        if n == "to" and len(args) > 1 and isinstance(args[1], str) and args[1].startswith("cuda"):
            args = (args[0], "cpu") + tuple(args[2:])
        return func(*args, **kwargs)


CAPTURE = []


def install_recording_rasterizer():
    """A `diff_gaussian_rasterization` whose backend is the oracle's torch restatement and which records its inputs."""
    from typing import NamedTuple
    from oracle.raster_torch import rasterize_torch

    class GaussianRasterizationSettings(NamedTuple):
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            rs = self.raster_settings
            rec = dict(settings={k: getattr(rs, k) for k in rs._fields})
            for k, v in dict(means3D=means3D, means2D=means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                             scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp).items():
                rec[k] = None if v is None else dict(value=v.detach().clone(), requires_grad=bool(v.requires_grad),
                                                      dtype=str(v.dtype), contiguous=bool(v.is_contiguous()))
            CAPTURE.append(rec)
            color, radii, depth, alpha, _ = rasterize_torch(
                means3D=means3D, means2D=means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                rotations=rotations, cov3D_precomp=cov3D_precomp, viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix,
                campos=rs.campos, bg=rs.bg, image_height=rs.image_height, image_width=rs.image_width, tanfovx=rs.tanfovx,
                tanfovy=rs.tanfovy, scale_modifier=rs.scale_modifier, sh_degree=rs.sh_degree)
            return color, radii, depth, alpha

    stub("diff_gaussian_rasterization", GaussianRasterizationSettings=GaussianRasterizationSettings,
         GaussianRasterizer=GaussianRasterizer)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def main():
    stub("plyfile", PlyData=object, PlyElement=object)
    stub("pytorch3d")
    stub("pytorch3d.transforms", euler_angles_to_matrix=None)
    stub("simple_knn")
    stub("simple_knn._C", distCUDA2=lambda pts: torch.full((pts.shape[0],), 1e-3))
    install_recording_rasterizer()
    from egogaussian_amd.scene_synth import make_scene, fov_pair

    with CudaToCpu():
        from utils.graphics_utils import getWorld2View2, getProjectionMatrix
        from utils.sh_utils import eval_sh, RGB2SH, SH2RGB
        from utils.loss_utils import l1_loss, l2_loss, ssim
        from utils.image_utils import psnr
        from scene.gaussian_model import GaussianModel
        from gaussian_renderer import render
        from gaussian_renderer.render_helper import get_render_label

        rng = np.random.default_rng(2024)

        # ---- camera ----------------------------------------------------------------------------------------
        cams = {}
        for i in range(3):
            ang = rng.normal(size=3) * 0.3
            cx, sx, cy, sy, cz, sz = [f(a) for a in ang for f in (math.cos, math.sin)]
            R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                 @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
            T = rng.normal(size=3) * 0.5
            fovx, fovy = 0.9 + 0.1 * i, 0.6 + 0.05 * i
            w2v = getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)
            P = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy)
            wvt = torch.tensor(w2v).transpose(0, 1)
            pm = P.transpose(0, 1)
            full = (wvt.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0)
            center = wvt.inverse()[3, :3]
            cams.update({f"R{i}": R, f"T{i}": T, f"fov{i}": np.array([fovx, fovy]), f"w2v{i}": w2v, f"P{i}": npy(P),
                         f"wvt{i}": npy(wvt), f"full{i}": npy(full), f"center{i}": npy(center)})
        np.savez_compressed(os.path.join(HERE, "camera.npz"), **cams)

        # ---- SH ------------------------------------------------------------------------------------------------
        dirs = torch.tensor(rng.normal(size=(64, 3)), dtype=torch.float32)
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        sh = torch.tensor(rng.normal(size=(64, 3, 16)), dtype=torch.float32)
        shd = {"dirs": npy(dirs), "sh": npy(sh)}
        for deg in range(4):
            shd[f"eval{deg}"] = npy(eval_sh(deg, sh[..., :(deg + 1) ** 2], dirs))
        rgb = torch.tensor(rng.uniform(size=(10, 3)), dtype=torch.float32)
        shd.update(rgb=npy(rgb), rgb2sh=npy(RGB2SH(rgb)), sh2rgb=npy(SH2RGB(rgb)))
        np.savez_compressed(os.path.join(HERE, "sh.npz"), **shd)

        # ---- losses ------------------------------------------------------------------------------------------
        a = torch.tensor(rng.uniform(size=(3, 40, 56)), dtype=torch.float32)
        b = (a + torch.tensor(rng.normal(scale=0.1, size=(3, 40, 56)), dtype=torch.float32)).clamp(0, 1)
        a.requires_grad_(True)
        s = ssim(a, b)
        s.backward()
        np.savez_compressed(os.path.join(HERE, "losses.npz"), a=npy(a), b=npy(b), l1=npy(l1_loss(a, b)), l2=npy(l2_loss(a, b)),
                            ssim=npy(s), ssim_grad_a=npy(a.grad), ssim_per_image=npy(ssim(a[None], b[None], size_average=False)),
                            psnr=npy(psnr(a[None], b[None])))

        # ---- model: covariance + boundary ---------------------------------------------------------------------
        N, H, W = 300, 48, 64
        sc = make_scene(N, H, W, seed=11)
        sc["log_scale"] += math.log(3.0)
        g = GaussianModel(0)
        P_ = lambda x: torch.nn.Parameter(torch.tensor(x, dtype=torch.float32))
        g._xyz, g._features_dc = P_(sc["xyz"]), P_(sc["features"][:, :1])
        g._features_rest = P_(np.zeros((N, 0, 3), np.float32))
        g._scaling, g._rotation, g._opacity = P_(sc["log_scale"]), P_(sc["quat"] * 1.7), P_(sc["opacity_logit"])
        g._label = P_(rng.normal(size=(N, 1)).astype(np.float32))
        is_obj = (rng.uniform(size=(N, 1)) < 0.3).astype(np.float32)
        g._is_object = torch.tensor(is_obj)
        g._generation = torch.zeros(N, 1)

        cov = g.get_covariance(1.0)
        wcov = torch.tensor(rng.normal(size=(N, 6)), dtype=torch.float32)
        (cov * wcov).sum().backward()
        cv = dict(log_scale=sc["log_scale"], quat=npy(g._rotation), is_object=is_obj, wcov=npy(wcov), cov=npy(cov),
                  cov_mod2=npy(g.get_covariance(2.0)), g_scaling=npy(g._scaling.grad), g_rotation=npy(g._rotation.grad))
        g._scaling.grad = None; g._rotation.grad = None
        accum_R = torch.tensor(cams["R1"], dtype=torch.float32)
        rcov = g.get_rotated_covariance(accum_R, 1, False, 1.0)
        (rcov * wcov).sum().backward()
        cv.update(accum_R=npy(accum_R), rcov=npy(rcov), rg_scaling=npy(g._scaling.grad), rg_rotation=npy(g._rotation.grad),
                  rcov_identity=npy(g.get_rotated_covariance(torch.eye(3), 1, False, 1.0)),
                  rcov_all=npy(g.get_rotated_covariance(accum_R, None, False, 1.0)))
        g._scaling.grad = None; g._rotation.grad = None
        np.savez_compressed(os.path.join(HERE, "covariance.npz"), **cv)

        class Cam:
            pass
        cam = Cam()
        fovx, fovy = fov_pair(H, W)
        cam.image_height, cam.image_width, cam.FoVx, cam.FoVy = H, W, fovx, fovy
        Rc = np.eye(3); Tc = np.array([0.1, -0.05, 0.3])
        cam.world_view_transform = torch.tensor(getWorld2View2(Rc, Tc, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        cam.full_proj_transform = (cam.world_view_transform.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        cam.camera_center = cam.world_view_transform.inverse()[3, :3]

        class Pipe:
            convert_SHs_python = False
            compute_cov3D_python = True
            debug = False
        bg = torch.tensor([0.2, 0.1, 0.3])
        gen = torch.Generator().manual_seed(7)
        wc, wd, wa = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen), torch.rand(1, H, W, generator=gen)

        CAPTURE.clear()
        out = render(cam, g, Pipe, bg)
        loss = (out["render"] * wc).sum() + (out["depth"] * wd).sum() + (out["alpha"] * wa).sum()
        loss.backward()
        c1 = CAPTURE[0]
        bd = dict(N=N, H=H, W=W, fov=np.array([fovx, fovy]), bg=npy(bg), wc=npy(wc), wd=npy(wd), wa=npy(wa),
                  xyz=sc["xyz"], features_dc=sc["features"][:, :1], log_scale=sc["log_scale"], quat=npy(g._rotation),
                  opacity_logit=sc["opacity_logit"], label=npy(g._label), is_object=is_obj,
                  wvt=npy(cam.world_view_transform), full=npy(cam.full_proj_transform), center=npy(cam.camera_center))
        for k in ("means3D", "means2D", "opacities", "shs", "cov3D_precomp"):
            bd[f"m1_{k}"] = npy(c1[k]["value"]); bd[f"m1_{k}_rg"] = c1[k]["requires_grad"]
        bd["m1_absent"] = np.array([c1[k] is None for k in ("colors_precomp", "scales", "rotations")])
        st = c1["settings"]
        bd.update(m1_tanfov=np.array([st["tanfovx"], st["tanfovy"]]), m1_sh_degree=st["sh_degree"],
                  m1_scale_modifier=st["scale_modifier"], m1_flags=np.array([st["prefiltered"], st["debug"]]),
                  m1_viewmatrix=npy(st["viewmatrix"]), m1_projmatrix=npy(st["projmatrix"]), m1_campos=npy(st["campos"]),
                  m1_render=npy(out["render"]), m1_depth=npy(out["depth"]), m1_alpha=npy(out["alpha"]), m1_radii=npy(out["radii"]),
                  m1_visibility=npy(out["visibility_filter"]), m1_g_xyz=npy(g._xyz.grad), m1_g_features_dc=npy(g._features_dc.grad),
                  m1_g_scaling=npy(g._scaling.grad), m1_g_rotation=npy(g._rotation.grad), m1_g_opacity=npy(g._opacity.grad),
                  m1_g_viewspace=npy(out["viewspace_points"].grad))
        for p in (g._xyz, g._features_dc, g._scaling, g._rotation, g._opacity):
            p.grad = None

        CAPTURE.clear()
        lab = get_render_label(cam, g, bg)
        (lab * wc).sum().backward()
        c2 = CAPTURE[0]
        for k in ("means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations"):
            bd[f"m2_{k}"] = npy(c2[k]["value"]); bd[f"m2_{k}_rg"] = c2[k]["requires_grad"]
        bd["m2_absent"] = np.array([c2[k] is None for k in ("shs", "cov3D_precomp")])
        bd.update(m2_label_render=npy(lab), m2_g_label=npy(g._label.grad),
                  m2_no_geom_grad=np.array([p.grad is None for p in (g._xyz, g._scaling, g._rotation, g._opacity)]))
        np.savez_compressed(os.path.join(HERE, "boundary.npz"), **bd)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
