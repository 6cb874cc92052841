#!/usr/bin/env python
"""A miniature of /root/reference/trainers/train_static.py:67-138 on the synthetic scene, wired entirely to this package:
render -> 0.8 L1 + 0.2 (1 - SSIM) -> backward -> densification statistics -> [densify / prune, opacity reset] -> Adam,
with the steady-state iterations replayed from a hipGraph and the point cloud written as a PLY at the end.

    python examples/train_synth.py --gaussians 20000 --height 270 --width 480 --iters 600 --out /tmp/synth.ply

It shows the order of calls a trainer needs (and where a re-capture is required); it is not part of the measured path.
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import densify, ply                                          # noqa: E402
from egogaussian_amd.graph import GraphedTrainStep                                # noqa: E402
from egogaussian_amd.losses import psnr                                           # noqa: E402
from egogaussian_amd.renderer import render                                       # noqa: E402
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe, N_FRAMES   # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=20000)
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--iters", type=int, default=600)
    ap.add_argument("--densify-from", type=int, default=100)
    ap.add_argument("--densify-until", type=int, default=400)
    ap.add_argument("--densify-interval", type=int, default=100)
    ap.add_argument("--opacity-reset-interval", type=int, default=300)
    ap.add_argument("--grad-threshold", type=float, default=2e-4)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--sh-degree", type=int, default=0, help="spherical-harmonics degree of the colour model (0..3)")
    ap.add_argument("--sh-up-interval", type=int, default=0,
                    help="start at active degree 0 and raise it every this many iterations (the reference: 1000, scene/gaussian_model.py:176-178)")
    ap.add_argument("--out", default="")
    a = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    H, W = a.height, a.width
    teacher = make_scene(a.gaussians, H, W, seed=0, sh_degree=a.sh_degree)
    if a.gaussians < 100_000:
        teacher["log_scale"] += math.log(2.0)                       # few splats: make them larger so the image is covered
    cams = [make_camera(k * (N_FRAMES // a.frames), H, W, device=dev) for k in range(a.frames)]
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=dev, sh_degree=a.sh_degree, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
    pc = SynthGaussians(perturb_student(teacher), device=dev, sh_degree=a.sh_degree)
    pc.training_setup(capturable=True)
    if a.sh_up_interval > 0:
        pc.active_sh_degree = 0
    extent = 10.0

    def quality():
        with torch.no_grad():
            return float(sum(psnr(render(c, pc, Pipe, bg)["render"][None], g[None]) for c, g in zip(cams[:4], gts[:4])) / 4)

    print(f"start: {pc._xyz.shape[0]} Gaussians, PSNR {quality():.2f} dB")
    step = GraphedTrainStep(pc, pc.optimizer, bg, lambda_dssim=0.2, densify_stats=True).capture(cams[0], gts[0], warmup=2)
    t0, it = time.perf_counter(), 2
    while it < a.iters:
        k = it % a.frames
        step(cams[k], gts[k])                                        # render, loss, backward, statistics, Adam: one graph launch
        it += 1
        if a.sh_up_interval > 0 and it % a.sh_up_interval == 0 and pc.active_sh_degree < pc.max_sh_degree:
            pc.active_sh_degree += 1                                 # oneupSHdegree: a launch argument of the captured kernels -> re-capture
            step.recapture(warmup=1)
            it += 1
            print(f"iter {it}: active SH degree {pc.active_sh_degree}")
        if it <= a.densify_until and it > a.densify_from and it % a.densify_interval == 0:
            assert step.ok(), "a replayed frame outgrew the captured capacity"      # (reads a device word: synchronises)
            size_threshold = 20 if it > a.opacity_reset_interval else None
            n0, n1 = densify.densify_and_prune(pc, a.grad_threshold, 0.005, extent, size_threshold)
            if it % a.opacity_reset_interval == 0:
                densify.reset_opacity(pc)
            step.recapture(warmup=1)                                 # new parameter tensors -> new graph (one eager iteration inside)
            it += 1
            print(f"iter {it}: densify {n0} -> {n1} Gaussians")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert step.ok()
    print(f"end: {pc._xyz.shape[0]} Gaussians, PSNR {quality():.2f} dB, {a.iters / dt:.0f} it/s including densification and re-captures")
    if a.out:
        ply.save_ply(pc, a.out)
        back = ply.load_ply(SynthGaussians(teacher, device=dev, sh_degree=a.sh_degree), a.out, device=dev)
        assert torch.equal(back._xyz.detach(), pc._xyz.detach())
        print(f"wrote {a.out} ({os.path.getsize(a.out) / 1e6:.1f} MB) and read it back")
    return pc


if __name__ == "__main__":
    main()
