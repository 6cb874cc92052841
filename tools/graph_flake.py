#!/usr/bin/env python
"""Stress the graph-vs-eager equivalence: repeat the test body and report per-parameter differences."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
from egogaussian_amd.fused import l1_ssim_loss
from egogaussian_amd.optim import FusedAdam
from egogaussian_amd.graph import GraphedTrainStep
DEV = torch.device("cuda", 0)
N, H, W, K, WARM = 20000, 96, 160, 6, 3
teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
student = perturb_student(teacher)
cams = [make_camera(k, H, W, device=DEV) for k in (0, 30, 60, 90)]
bg = torch.zeros(3, device=DEV)
with torch.no_grad():
    tpc = SynthGaussians(teacher, device=DEV, requires_grad=False)
    gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
groups = lambda pc: [{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                     {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}]
seq = [0] * WARM + [k % 4 for k in range(1, K + 1)]
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
def eager(capturable=False):
    pa = SynthGaussians(student, device=DEV)
    oa = FusedAdam(groups(pa), lr=0.0, eps=1e-15, capturable=capturable)
    for k in seq:
        out = render(cams[k], pa, Pipe, bg)
        l1_ssim_loss(out["render"], gts[k], 0.2).backward()
        oa.step(); oa.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    return pa, oa
ref, _ = eager()
names = ["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    if mode == "graph":
        pb = SynthGaussians(student, device=DEV)
        ob = FusedAdam(groups(pb), lr=0.0, eps=1e-15, capturable=True)
        step = GraphedTrainStep(pb, ob, bg).capture(cams[0], gts[0], warmup=WARM)
        for k in range(1, K + 1):
            step(cams[k % 4], gts[k % 4])
        torch.cuda.synchronize()
        steps = [float(ob.state[p]["step"]) for p in (pb._xyz, pb._opacity)]
    else:
        pb, ob = eager(capturable=(mode == "eager-capturable"))
        steps = [float(ob.state[p]["step"]) for p in (pb._xyz, pb._opacity)]
    row = []
    for nm, a, b in zip(names, ref.parameters(), pb.parameters()):
        if a.numel():
            d = (a.detach() - b.detach()).abs()
            row.append(f"{nm} max {float(d.max()):.2e} n>1e-4 {int((d > 1e-4).sum())}")
    print(it, steps, " | ".join(row), flush=True)
