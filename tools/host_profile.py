#!/usr/bin/env python
"""cProfile of the eager training step's host side at config C (fused host ops, one `loss.item()` per step as
/root/reference/trainers/train_static.py:112 does): where the Python / ctypes / launch time of an un-captured step goes.
   python tools/host_profile.py [steps] [--no-item]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
from egogaussian_amd.fused import l1_ssim_loss
from egogaussian_amd.optim import FusedAdam
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
item = "--no-item" not in sys.argv
if "--single-thread-autograd" in sys.argv:
    torch.autograd.set_multithreading_enabled(False)
dev = "cuda:0"; N, H, W = 500000, 540, 960
sc = make_scene(N, H, W, 0); pc = SynthGaussians(perturb_student(sc), device=dev); bg = torch.zeros(3, device=dev)
cams = [make_camera(k, H, W, device=dev) for k in range(8)]
with torch.no_grad():
    tpc = SynthGaussians(sc, device=dev, requires_grad=False)
    gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
opt = FusedAdam([{"params": [p], "lr": 1e-3} for p in (pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation)], lr=0.0, eps=1e-15)
T = {"render": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0, "item": 0.0}
def step(k, acc=True):
    t0 = time.perf_counter(); out = render(cams[k % 8], pc, Pipe, bg)
    t1 = time.perf_counter(); loss = l1_ssim_loss(out["render"], gts[k % 8], 0.2)
    t2 = time.perf_counter(); loss.backward()
    t3 = time.perf_counter(); opt.step(); opt.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    if item:
        loss.item()
    t5 = time.perf_counter()
    if acc:
        for name, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            T[name] += d
for k in range(30): step(k, False)
torch.cuda.synchronize()
t = time.perf_counter()
for k in range(steps): step(k)
torch.cuda.synchronize()
wall = time.perf_counter() - t
print(f"eager fused step: {steps / wall:.0f} it/s, {wall / steps * 1e6:.0f} us/step wall; host us/step: " + ", ".join(f"{k} {v / steps * 1e6:.0f}" for k, v in T.items()))
pr = cProfile.Profile(); pr.enable()
for k in range(steps): step(k, False)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
