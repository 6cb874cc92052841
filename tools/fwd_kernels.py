#!/usr/bin/env python
"""Run a few forward passes at config C (for rocprofv3 --kernel-trace --stats A/B runs of library variants).
CULL=0 in the environment turns tile culling off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import _C
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
N, H, W = int(os.environ.get("N", 500_000)), int(os.environ.get("H", 540)), int(os.environ.get("W", 960))
dev = torch.device("cuda", 0)
_C.set_tile_culling(os.environ.get("CULL", "1") != "0")
pc = SynthGaussians(make_scene(N, H, W, seed=0), device=dev, requires_grad=False)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for k in range(30):
        render(make_camera(k % 8, H, W, device=dev), pc, Pipe, bg)
torch.cuda.synchronize()
