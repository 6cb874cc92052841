#!/usr/bin/env python
"""Host-side (Python + launch) time of one eager training step, by section, with the GPU kept out of the way:
every section is timed while the stream is allowed to run ahead (no synchronisation inside the loop)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
from egogaussian_amd.fused import l1_ssim_loss
from egogaussian_amd.optim import FusedAdam
dev = "cuda:0"; N, H, W = 500000, 540, 960
sc = make_scene(N, H, W, 0); pc = SynthGaussians(perturb_student(sc), device=dev); cam = make_camera(0, H, W, device=dev); bg = torch.zeros(3, device=dev)
with torch.no_grad():
    gt = render(cam, SynthGaussians(sc, device=dev, requires_grad=False), Pipe, bg)["render"].clone()
opt = FusedAdam([{"params": [p], "lr": 1e-3} for p in (pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation)], lr=0.0, eps=1e-15)
T = {"render": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0}
def step(acc):
    t0 = time.perf_counter(); out = render(cam, pc, Pipe, bg)
    t1 = time.perf_counter(); loss = l1_ssim_loss(out["render"], gt, 0.2)
    t2 = time.perf_counter(); loss.backward()
    t3 = time.perf_counter(); opt.step(); opt.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    if acc:
        T["render"] += t1 - t0; T["loss"] += t2 - t1; T["backward"] += t3 - t2; T["optimizer"] += t4 - t3
for _ in range(30): step(False)
torch.cuda.synchronize()
n = 300
t = time.perf_counter()
for _ in range(n): step(True)
host = time.perf_counter() - t
torch.cuda.synchronize()
total = time.perf_counter() - t
print(f"host enqueue {host / n * 1e6:.0f} us/step, wall {total / n * 1e6:.0f} us/step; " + ", ".join(f"{k} {v / n * 1e6:.0f}" for k, v in T.items()))
