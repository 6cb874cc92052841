#!/usr/bin/env python
"""List every device kernel of one training step with the PyTorch op that launched it (torch.profiler), to see which
launches around the rasterizer are avoidable.  Usage: python tools/step_ops.py [--gaussians N]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
from egogaussian_amd.fused import l1_ssim_loss
from egogaussian_amd.optim import FusedAdam

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=500_000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
H, W = 540, 960
teacher = make_scene(a.gaussians, H, W, seed=0)
pc = SynthGaussians(perturb_student(teacher), device=dev, fused=True)
cam = make_camera(0, H, W, device=dev)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    gt = render(cam, SynthGaussians(teacher, device=dev, requires_grad=False), Pipe, bg)["render"].clone()
opt = FusedAdam([{"params": [p], "lr": 1e-3} for p in (pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation)], lr=0.0, eps=1e-15)


def step():
    out = render(cam, pc, Pipe, bg)
    loss = l1_ssim_loss(out["render"], gt, 0.2)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
cpu = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
for e in evs:
    # innermost CPU op whose interval encloses the launch (matched through the correlation id when available)
    owner = ""
    for c in cpu:
        if any(k is e or getattr(k, "name", None) == e.name and getattr(k, "time_range", None) == e.time_range for k in getattr(c, "kernels", [])):
            owner = c.name
    print(f"{e.time_range.elapsed_us():8.1f} us  {e.name[:70]:70s}  <- {owner}")
print(len(evs), "device events")
