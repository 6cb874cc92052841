#!/usr/bin/env python
"""A miniature of /root/reference/trainers/train_static.py:67-138 on the synthetic scene, wired entirely to this package:
render -> 0.8 L1 + 0.2 (1 - SSIM) -> backward -> densification statistics -> [densify / prune, opacity reset] -> Adam,
with the steady-state iterations replayed from a hipGraph and the point cloud written as a PLY at the end.

    python examples/train_synth.py --gaussians 20000 --height 270 --width 480 --iters 600 --out /tmp/synth.ply

The model is capacity-sized (egogaussian_amd/capacity.py): densification and pruning rewrite the same arrays in place and the
number of live Gaussians is a device word, so the captured step is NOT re-captured when the model grows (`--plain` keeps the
reference's behaviour -- new tensors per densification, one re-capture each).  The step voids frames that outgrow its instance
capacity on the device and re-captures itself with more room (GraphedTrainStep(check_every=...)).

It shows the order of calls a trainer needs; it is not part of the measured path.  `--log` appends a line per report interval
(iteration, live Gaussians, it/s so far, held-out PSNR), which is how profiles/r2_train_synth_*.log were produced.
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import densify, ply                                          # noqa: E402
from egogaussian_amd.capacity import CapacityGaussians                            # noqa: E402
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
    ap.add_argument("--plain", action="store_true", help="plain model: densification replaces the tensors, the step is re-captured each time")
    ap.add_argument("--capacity-factor", type=float, default=3.0, help="rows allocated = this x the initial number of Gaussians")
    ap.add_argument("--report-every", type=int, default=0, help="print (and --log) progress every this many iterations")
    ap.add_argument("--log", default="")
    ap.add_argument("--min-opacity", type=float, default=0.005)
    ap.add_argument("--knn-init", action="store_true",
                    help="initialise the student as GaussianModel.create_from_pcd does (/root/reference/scene/gaussian_model.py:301-318): isotropic scales "
                         "from simple_knn.distCUDA2 of the (perturbed) teacher positions, identity rotations, opacity 0.1")
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
    student = perturb_student(teacher)
    if a.knn_init:
        from simple_knn._C import distCUDA2                      # the drop-in module name (egs_knn3_mean_dist2, csrc/knn.hip)
        dist2 = torch.clamp_min(distCUDA2(torch.from_numpy(student["xyz"]).float().to(dev)), 0.0000001)
        student["log_scale"] = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3).cpu().numpy()
        student["quat"][:] = 0.0; student["quat"][:, 0] = 1.0
        student["opacity_logit"][:] = math.log(0.1 / 0.9)
    if a.plain:
        pc = SynthGaussians(student, device=dev, sh_degree=a.sh_degree)
    else:
        pc = CapacityGaussians(student, int(a.gaussians * a.capacity_factor), device=dev, sh_degree=a.sh_degree)
    pc.training_setup(capturable=True)
    live = lambda: getattr(pc, "n_active", pc._xyz.shape[0])
    held = [make_camera(k * (N_FRAMES // a.frames) + 0.5 * (N_FRAMES // a.frames), H, W, device=dev) for k in range(0, a.frames, max(1, a.frames // 6))]
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=dev, sh_degree=a.sh_degree, requires_grad=False)
        held_gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in held]
        del tpc
    if a.sh_up_interval > 0:
        pc.active_sh_degree = 0
    extent = 10.0

    def quality():
        """mean PSNR over held-out views between the training cameras"""
        with torch.no_grad():
            return float(sum(psnr(render(c, pc, Pipe, bg)["render"][None], g[None]) for c, g in zip(held, held_gts)) / len(held))

    log = open(a.log, "a") if a.log else None

    def report(msg):
        print(msg, flush=True)
        if log:
            log.write(msg + "\n"); log.flush()

    report(f"# {' '.join(sys.argv)}")
    report(f"start: {live()} Gaussians, held-out PSNR {quality():.2f} dB, {'plain model' if a.plain else f'capacity {pc.capacity} rows'}")
    step = GraphedTrainStep(pc, pc.optimizer, bg, lambda_dssim=0.2, densify_stats=True, check_every=50)
    step.capture(cams[0], gts[0], warmup=2, capacity_margin=1.5, capacity_cams=cams[::max(1, a.frames // 6)])
    manual_recaptures, t_eval, next_report = 0, 0.0, a.report_every
    t0, it = time.perf_counter(), 2
    while it < a.iters:
        k = it % a.frames
        step(cams[k], gts[k])                                        # render, loss, backward, statistics, Adam: one graph launch
        it += 1
        if a.sh_up_interval > 0 and it % a.sh_up_interval == 0 and pc.active_sh_degree < pc.max_sh_degree:
            pc.active_sh_degree += 1                                 # oneupSHdegree: a launch argument of the captured kernels -> re-capture
            step.recapture(warmup=1); manual_recaptures += 1
            it += 1
            report(f"iter {it}: active SH degree {pc.active_sh_degree}")
        if it <= a.densify_until and it > a.densify_from and it % a.densify_interval == 0:
            step.check()                                             # a frame that outgrew the capacity was voided on the device; make room now
            size_threshold = 20 if it > a.opacity_reset_interval else None
            ptr = pc._xyz.data_ptr()
            n0, n1 = densify.densify_and_prune(pc, a.grad_threshold, a.min_opacity, extent, size_threshold)
            if it % a.opacity_reset_interval == 0:
                densify.reset_opacity(pc)
            if a.plain or pc._xyz.data_ptr() != ptr:                 # new parameter tensors (plain model, or the capacity had to grow) -> new graph
                step.recapture(warmup=1); manual_recaptures += 1
                it += 1
            if not a.report_every:
                report(f"iter {it}: densify {n0} -> {n1} Gaussians")
        if a.report_every and it >= next_report:
            next_report += a.report_every
            torch.cuda.synchronize()
            te = time.perf_counter()
            q = quality()
            t_eval += time.perf_counter() - te
            report(f"iter {it:6d}  live {live():8d}  {it / (time.perf_counter() - t0 - t_eval):8.1f} it/s so far  held-out PSNR {q:.3f} dB  "
                   f"re-captures {manual_recaptures + step.recaptures} (overflow {step.recaptures})  instance capacity {step.capacity}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0 - t_eval
    step.check()
    # (what the run was, for callers that assert on it: tests/test_gpu_densify.py runs the reference's full schedule through this function)
    pc.train_report = dict(gaussians_start=a.gaussians, gaussians_end=live(), psnr_end=quality(), its_per_s=a.iters / dt, recaptures=manual_recaptures + step.recaptures,
                           recaptures_after_overflow=step.recaptures, overflow_events=step.skipped_frames_seen, iterations=it, instance_capacity=step.capacity)
    report(f"end: {live()} Gaussians, held-out PSNR {quality():.2f} dB, {a.iters / dt:.0f} it/s including densification, opacity resets and "
           f"{manual_recaptures + step.recaptures} re-capture(s) ({step.recaptures} after an instance-capacity overflow, {step.skipped_frames_seen} overflow events)")
    if a.out:
        ply.save_ply(pc, a.out)
        back = ply.load_ply(SynthGaussians(teacher, device=dev, sh_degree=a.sh_degree), a.out, device=dev)
        assert torch.equal(back._xyz.detach(), pc._xyz.detach()[:live()])
        print(f"wrote {a.out} ({os.path.getsize(a.out) / 1e6:.1f} MB) and read it back")
    return pc


if __name__ == "__main__":
    main()
