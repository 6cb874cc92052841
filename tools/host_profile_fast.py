import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, "/root/repo")
torch.autograd.set_multithreading_enabled(False)
from egogaussian_amd import _C
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
from egogaussian_amd.fused import l1_ssim_loss
from egogaussian_amd.optim import FusedAdam
dev = "cuda:0"; N, H, W = 500000, 540, 960
sc = make_scene(N, H, W, 0); pc = SynthGaussians(perturb_student(sc), device=dev); bg = torch.zeros(3, device=dev)
cams = [make_camera(k, H, W, device=dev) for k in range(8)]
with torch.no_grad():
    tpc = SynthGaussians(sc, device=dev, requires_grad=False)
    gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
opt = FusedAdam([{"params": [p], "lr": 1e-3} for p in (pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation)], lr=0.0, eps=1e-15, capturable=True)
guard = _C.StepGuard(torch.device(dev), deferred=True); opt.guard = guard
def step(k):
    out = render(cams[k % 8], pc, Pipe, bg, optimizer=opt, guard=guard)
    loss = l1_ssim_loss(out["render"], gts[k % 8], 0.2, raster_prologue=True)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
    return loss
for k in range(30): step(k)
torch.cuda.synchronize()
t = time.perf_counter()
for k in range(300): step(k)
th = time.perf_counter() - t
torch.cuda.synchronize()
print(f"host per step (no profiler, no sync): {1e6 * th / 300:.1f} us; wall {1e6 * (time.perf_counter() - t) / 300:.1f} us")
pr = cProfile.Profile(); pr.enable()
for k in range(300): step(k)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
