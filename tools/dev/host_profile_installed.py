"""cProfile of the reference's training loop, unchanged, with what egogaussian_amd.install() puts behind its names (bench.py
`reference_shaped_step`): where the host's ~0.7 ms per iteration go.   python tools/dev/host_profile_installed.py"""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from egogaussian_amd import patching
from egogaussian_amd.adapter import attach
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
dev = torch.device("cuda:0"); N, H, W = 500000, 540, 960
teacher = make_scene(N, H, W, 0); bg = torch.zeros(3, device=dev)
NF = int(os.environ.get("FRAMES", "8"))
cams = [make_camera(k, H, W, device=dev) for k in range(NF)]
with torch.no_grad():
    tpc = SynthGaussians(teacher, device=dev, requires_grad=False)
    gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
model = SynthGaussians(perturb_student(teacher), device=dev, fused=False)
model.training_setup(optimizer_cls=torch.optim.Adam)
l1_loss, ssim = patching.make_loss_functions()
attach(model)
model.get_covariance = lambda m=1, _g=model: _g.covariance_activation(_g.get_scaling, m, _g._rotation)
pc = bench._ReferenceSurface(model); opt = model.optimizer
hm = (torch.rand(1, H, W) < 0.1).float().to(dev)
def step(k):
    pkg = render(cams[k % NF], pc, Pipe, bg)
    img = pkg["render"]
    img.register_hook(lambda grad: grad * (1 - hm))
    loss = 0.8 * l1_loss(img, gts[k % NF]) + 0.2 * (1.0 - ssim(img, gts[k % NF]))
    loss.backward(); loss.item()
    opt.step(); opt.zero_grad(set_to_none=True)
for k in range(20): step(k)
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(200): step(k)
torch.cuda.synchronize(); print(f"wall per step {1e6 * (time.perf_counter() - t) / 200:.1f} us")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for k in range(20): step(k)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 or getattr(e, "self_device_time_total", 0) > 0]
tot = 0
rows = []
for e in prof.key_averages():
    dt = getattr(e, "self_device_time_total", 0)
    if dt > 0:
        rows.append((dt / 20, e.count / 20, e.key[:90])); tot += dt / 20
rows.sort(reverse=True)
print(f"device time per step {tot:.1f} us over {sum(r[1] for r in rows):.0f} kernels")
for r in rows[:40]: print(f"  {r[0]:8.1f} us  x{r[1]:4.1f}  {r[2]}")
pr = cProfile.Profile(); pr.enable()
for k in range(200): step(k)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
# the backward runs on autograd's device thread, which cProfile does not see: once more with autograd on the calling thread
torch.autograd.set_multithreading_enabled(False)
for k in range(20): step(k)
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(200): step(k)
torch.cuda.synchronize(); print(f"wall per step, autograd on the calling thread {1e6 * (time.perf_counter() - t) / 200:.1f} us")
pr = cProfile.Profile(); pr.enable()
for k in range(200): step(k)
pr.disable(); torch.cuda.synchronize()
print("---- autograd on the calling thread, by internal time"); pstats.Stats(pr).sort_stats("tottime").print_stats(45)
print("---- by cumulative time"); pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
from egogaussian_amd import provenance
print("provenance substitutions so far:", provenance.substitutions)
