import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from egogaussian_amd import _C
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
from egogaussian_amd.fused import l1_ssim_loss
from egogaussian_amd.optim import FusedAdam
dev = torch.device("cuda:0"); N, H, W = 500000, 540, 960
teacher = make_scene(N, H, W, 0); student = perturb_student(teacher)
cams = [make_camera(k, H, W, device=dev) for k in range(40)]; bg = torch.zeros(3, device=dev)
with torch.no_grad():
    tpc = SynthGaussians(teacher, device=dev, requires_grad=False)
    gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
pc = SynthGaussians(student, device=dev)
opt = FusedAdam([{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                 {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}], lr=0.0, eps=1e-15)
def stats(tag):
    with torch.no_grad():
        render(cams[3], pc, Pipe, bg)
    torch.cuda.synchronize()
    ls = bench.tile_list_stats(_C, _C.stats["image_buffer"], W, H)
    print(tag, "R", _C.stats["num_rendered"], "kept", int(_C.stats["total_view"].item()), ls)
stats("step 0  ")
for it in range(400):
    k = it % 40
    out = render(cams[k], pc, Pipe, bg)
    l1_ssim_loss(out["render"], gts[k], 0.2).backward(); opt.step(); opt.zero_grad(set_to_none=True)
    if it + 1 in (25, 100, 200, 400):
        stats(f"step {it + 1:3d}")
