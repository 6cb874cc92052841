import math, random, sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.common import OracleRasterize, quantize_8bit
from egogaussian_amd.fused import l1_ssim_loss
from egogaussian_amd.losses import training_loss, psnr
from egogaussian_amd.optim import FusedAdam
from egogaussian_amd.renderer import render
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe, N_FRAMES
dev = torch.device("cuda:0")
N, H, W, K = 10_000, 64, 64, 300
for (sx, sf, smul) in ((0.01, 0.1, 2.5), (0.03, 0.3, 2.5), (0.06, 0.5, 2.5), (0.03, 0.3, 1.5)):
    teacher = make_scene(N, H, W, seed=4); teacher["log_scale"] += math.log(smul)
    rng = np.random.default_rng(1001)
    student = {k: v.copy() for k, v in teacher.items()}
    student["xyz"] += rng.normal(0, sx, student["xyz"].shape).astype(np.float32)
    student["features"][:, :1] += rng.normal(0, sf, student["features"][:, :1].shape).astype(np.float32)
    train_frames, eval_frames = list(range(0, N_FRAMES, 25)), [12.5, 87.5, 162.5, 237.5]
    lrs = [("_xyz", 1.6e-4), ("_features_dc", 2.5e-3), ("_opacity", 0.05), ("_scaling", 5e-3), ("_rotation", 1e-3)]
    def const_of(cam):
        return dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center, bg=torch.zeros(3),
                    H=H, W=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), nthreads=8)
    def cpu_render(cam, pc):
        return OracleRasterize.apply(pc.get_xyz, pc.get_opacity, pc.get_features, pc.get_covariance(), const_of(cam))
    hist = {}
    for side in ("gpu", "cpu", "gpu2"):
        d = dev if side.startswith("gpu") else "cpu"
        bg = torch.zeros(3, device=d)
        rend = (lambda c, p: render(c, p, Pipe, bg)["render"]) if side.startswith("gpu") else cpu_render
        cams = [make_camera(k, H, W, device=d) for k in train_frames]
        ecams = [make_camera(k, H, W, device=d) for k in eval_frames]
        with torch.no_grad():
            tpc = SynthGaussians(teacher, device=d, requires_grad=False)
            gts, egts = [rend(c, tpc).clone() for c in cams], [rend(c, tpc).clone() for c in ecams]
        pc = SynthGaussians(student, device=d)
        groups = [{"params": [getattr(pc, a)], "lr": lr} for a, lr in lrs]
        opt = FusedAdam(groups, lr=0.0, eps=1e-15) if side.startswith("gpu") else torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        rnd = random.Random(0)
        h = []
        for it in range(K + 1):
            if it % 50 == 0:
                with torch.no_grad():
                    h.append(float(np.mean([psnr(rend(c, pc)[None], g[None]).item() for c, g in zip(ecams, egts)])))
            k = rnd.randrange(len(cams))
            img = rend(cams[k], pc)
            loss = l1_ssim_loss(img, gts[k], 0.2) if side.startswith("gpu") else training_loss(img, gts[k], 0.2)
            loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
        hist[side] = h
    print(f"perturb xyz {sx} f_dc {sf} scale x{smul}")
    for s, h in hist.items():
        print("  ", s, " ".join(f"{x:.3f}" for x in h))
