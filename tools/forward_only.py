#!/usr/bin/env python
"""BASELINE config B: forward-only render (no grad) of S(N, H, W, seed 0), replayed back to back from a hipGraph of `per`
forwards so that the GPU stays at its working clocks (launched one by one from Python a 100k-Gaussian forward leaves the GPU
idle most of the time, and rocprofv3 then shows the same kernels 2-3x slower).
   python tools/forward_only.py [N] [H] [W] [per] [replays]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import _C
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render

N, H, W, per, replays = [int(a) for a in (sys.argv[1:6] + ["100000", "540", "960", "20", "20"][len(sys.argv) - 1:])]
dev = torch.device("cuda", 0)
pc = SynthGaussians(make_scene(N, H, W, seed=0), device=dev, requires_grad=False)
cam, bg = make_camera(0, H, W, device=dev), torch.zeros(3, device=dev)
with torch.no_grad():
    for _ in range(3):
        out = render(cam, pc, Pipe, bg)
    torch.cuda.synchronize()
    R = _C.stats["num_rendered"]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per):
            out = render(cam, pc, Pipe, bg)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / (per * replays)
npix = H * W
# algorithmic bytes of one forward (R counted on the reference's rectangles)
bytes_ = 104 * N + (96 * N + 8 * R) + 12 * R + (52 * R + 28 * npix)
print(f"S({N},{H},{W}): forward {us:.1f} us = {1e6 / us:.0f} frames/s; R = {R}; algorithmic {bytes_ / 1e6:.0f} MB -> {bytes_ / us / 1e6:.2f} TB/s")
