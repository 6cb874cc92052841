#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its quoted configuration, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

Workload (config C of BASELINE.md): synthetic scene S(500k Gaussians, 540, 960, seed 0), the 300-frame synthetic
orbit, SH degree 0 (`--sh-degree 3` for the 16-coefficient colour model the reference ends training with; the default
single-GPU run measures that too, in a child process, and reports it as `sh_degree_3`).
One step = one full training iteration of /root/reference/trainers/train_static.py:67-138 without densification or
logging: covariance (the reference forces compute_cov3D_python, /root/reference/train.py:49; here built inside the
rasterizer's preprocess kernel from the raw parameters) -> render() forward (HIP) -> 0.8 L1 + 0.2 (1 - SSIM) ->
backward (HIP + autograd) -> Adam; replayed from one hipGraph per step (`--no-graph`: launched eagerly).
Frames are sharded round-robin over ranks (1 frame per GPU per step); ranks exchange only
scalars (loss / PSNR sums) through one RCCL all-reduce; `value` = steps of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      the dominant rasterizer stage: algorithmic bytes per launch / its mean duration, measured with HIP
                events recorded by the library on the launch stream inside the timed region;
  stages        the same for every stage (ms per launch, GB/s algorithmic);
  cpu_baseline  the C oracle (oracle/raster_oracle.c, "port") timed on this box's host cores on a bounded sample of
                forward+backward passes of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def algorithmic_bytes(stage, N, R, npix, sh_coeffs=1):
    """Bytes one launch of `stage` must move at minimum: per-unit figures x units.  R = instances the
    launch actually processes (after tile culling)."""
    return {
        "preprocess": (44 + 12 * sh_coeffs) * N + 48 * N,   # xyz 12 + log-scale 12 + quaternion 16 + opacity logit 4 + sh 12/coefficient in; record 48 out
        "tile_bucket": 2 * (16 + 32) * N + 8 * R,      # two walks over (tiles_touched, rect, depth | ellipse) per Gaussian; one pair out per instance
        "tile_sort": 8 * R + 4 * R,                    # pair in, index out; the radix passes stay in registers/LDS
        "render_forward": 4 * R + 48 * R + 28 * npix,  # id + record per instance; 7 floats per pixel out
        "render_backward": 4 * R + 48 * R + 40 * R + 32 * npix,   # + one 40-byte accumulate per instance; 8 floats/pixel in
        "preprocess_backward": 48 * N + (44 + 12 * sh_coeffs) * N + 32 * N + (68 + 12 * sh_coeffs) * N,  # accumulator + inputs + record head in; grads out (xyz, mean2D, scale, quat, sh, colour, opacity)
    }[stage]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--sh-degree", type=int, default=0, help="spherical-harmonics degree of the colours (0..3; the reference ends training at 3)")
    ap.add_argument("--no-sh3-leg", action="store_true", help="skip the extra SH-degree-3 measurement of the default single-GPU run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--op-only", action="store_true", help="time rasterizer fwd+bwd only (seeded upstream grads)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of every step from Python instead of replaying a captured hipGraph")
    ap.add_argument("--torch-host-ops", action="store_true",
                    help="build cov3D and the loss with PyTorch ops (as the reference does) instead of the fused HIP kernels")
    args = ap.parse_args()

    from egogaussian_amd import dist as egs_dist
    rank, world, local_rank = egs_dist.env_world()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU path)"
    if os.environ.get("EGS_BENCH_SHARE_DEVICE0"):               # test hook: several ranks on one GPU (RCCL refuses that; use gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    egs_dist.init(os.environ.get("EGS_BENCH_BACKEND", "nccl"), dev)        # "nccl" is RCCL on ROCm

    from egogaussian_amd import lib as egs_lib, _C
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe, N_FRAMES
    from egogaussian_amd.renderer import render
    from egogaussian_amd.losses import training_loss, psnr
    from egogaussian_amd.fused import l1_ssim_loss
    egs_lib.load()

    N, H, W = args.gaussians, args.height, args.width
    D = args.sh_degree
    teacher = make_scene(N, H, W, seed=0, sh_degree=D)
    student = perturb_student(teacher)
    bg = torch.zeros(3, device=dev)
    my_frames = egs_dist.shard_frames(N_FRAMES, rank, world)
    n_used = min(len(my_frames), args.warmup + args.steps)
    cams = [make_camera(k, H, W, device=dev) for k in my_frames[:n_used]]

    with torch.no_grad():                                      # ground truth = teacher rendered by the same path
        tpc = SynthGaussians(teacher, device=dev, sh_degree=D, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
        del tpc
    pc = SynthGaussians(student, device=dev, sh_degree=D, fused=not args.torch_host_ops)
    from egogaussian_amd.optim import FusedAdam
    use_graph = not (args.no_graph or args.torch_host_ops or args.op_only)
    Adam = (lambda g, **kw: torch.optim.Adam(g, fused=True, **kw)) if args.torch_host_ops else \
        (lambda g, **kw: FusedAdam(g, capturable=use_graph, **kw))
    opt = Adam([                                                # /root/reference/scene/gaussian_model.py:180-198 defaults
        {"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3},
        *([{"params": [pc._features_rest], "lr": 2.5e-3 / 20.0}] if D > 0 else []),
        {"params": [pc._opacity], "lr": 0.05}, {"params": [pc._scaling], "lr": 5e-3},
        {"params": [pc._rotation], "lr": 1e-3}], lr=0.0, eps=1e-15)

    def eval_psnr():
        with torch.no_grad():
            vals = [psnr(render(cams[i], pc, Pipe, bg)["render"][None], gts[i][None]).item() for i in range(min(4, n_used))]
        return float(np.mean(vals))

    g = torch.Generator().manual_seed(1234)
    up_c, up_d, up_a = [torch.rand(s, generator=g).to(dev) for s in ((3, H, W), (1, H, W), (1, H, W))]
    loss_acc = torch.zeros((), device=dev)
    r_sum = [0, 0]

    def step(i):
        k = i % n_used
        out = render(cams[k], pc, Pipe, bg)
        if args.op_only:
            loss = (out["render"] * up_c).sum() + (out["depth"] * up_d).sum() + (out["alpha"] * up_a).sum()
        elif args.torch_host_ops:
            loss = training_loss(out["render"], gts[k])
        else:
            loss = l1_ssim_loss(out["render"], gts[k], 0.2)
        loss.backward()
        if not args.op_only:
            opt.step()
        opt.zero_grad(set_to_none=True)
        loss_acc.add_(loss.detach())
        r_sum[0] += _C.stats["num_rendered"]; r_sum[1] += 1

    psnr_start = eval_psnr()
    graphed = None
    eager_step = step
    if use_graph:                                               # the whole iteration as one hipGraph (egogaussian_amd/graph.py)
        from egogaussian_amd.graph import GraphedTrainStep
        try:
            graphed = GraphedTrainStep(pc, opt, bg, 0.2).capture(cams[0], gts[0], warmup=2)
        except Exception as exc:                                # keep measuring, eagerly, rather than lose the run
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); stepping eagerly", file=sys.stderr)
            graphed, use_graph = None, False
        if graphed is not None:
            from egogaussian_amd.graph import pack_frame
            frames = [pack_frame(c, g_) for c, g_ in zip(cams, gts)]      # resident: image + camera block, one copy per replay

            def step(i):                                        # noqa: F811
                graphed(frames[i % n_used])                      # (the captured step adds its loss to graphed.loss_sum itself)
                r_sum[1] += 1
    for i in range(args.warmup):
        step(i)
    loss_acc.zero_(); r_sum[:] = [0, 0]
    if graphed is not None:
        graphed.loss_sum.zero_()
    torch.cuda.synchronize()
    egs_dist.barrier()
    egs_lib.profile_begin(max_records=min(200000, 16 * (args.steps + 8)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    egs_dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stages = egs_lib.profile_end()
    stage_timing = "HIP events recorded by the library on the launch stream inside the timed region"
    if graphed is not None:
        assert graphed.ok(), f"a replayed frame exceeded the captured capacity ({graphed.max_instances()} > {graphed.capacity})"
        # Kernels replayed from a hipGraph are not bracketed by the library's events (those are host-side records), so the
        # per-stage durations come from a second, eager timed pass over the same workload right after the replayed one.
        n_ev = min(args.steps, 50)
        loss_acc.copy_(graphed.loss_sum)
        saved = (loss_acc.clone(), list(r_sum))
        r_sum[:] = [0, 0]
        torch.cuda.synchronize()
        egs_lib.profile_begin(max_records=16 * (n_ev + 8))
        for i in range(n_ev):
            eager_step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        stages = egs_lib.profile_end()
        r_mean_eager = float(r_sum[0]) / max(r_sum[1], 1)
        loss_acc.copy_(saved[0]); r_sum[:] = [int(r_mean_eager * saved[1][1]), saved[1][1]]
        stage_timing = f"HIP events recorded by the library on the launch stream over {n_ev} eager steps of the same workload, run inside bench.py right after the graph-replayed timed region"
    psnr_end = eval_psnr()

    elapsed_max = egs_dist.reduce_scalars([elapsed], dev, "max")[0]          # max over ranks of the timed region
    sums = egs_dist.reduce_scalars([loss_acc.item(), psnr_end, psnr_start, float(r_sum[0]) / max(r_sum[1], 1)], dev, "sum")
    mean_loss = sums[0] / (world * args.steps)                                # scalar metrics only (RCCL over xGMI)
    psnr_e, psnr_s, R_mean = sums[1] / world, sums[2] / world, sums[3] / world

    if rank != 0:
        egs_dist.shutdown()
        return

    npix = H * W
    passes = _C.binning_passes(N, W, H)
    # instances that survive tile culling (what the sort and the blend kernels process): sampled on four frames
    kept, rect = 0, 0
    with torch.no_grad():
        for i in range(min(4, n_used)):
            render(cams[i], pc, Pipe, bg)
            kept += int(_C.stats["total_view"].item()); rect += _C.stats["num_rendered"]
    R_kept = R_mean * kept / max(rect, 1)
    stage_rows, dominant = {}, None
    for name, (ms, n) in stages.items():
        if n == 0:
            continue
        per = ms / n
        ab = algorithmic_bytes(name, N, R_kept, npix, (D + 1) ** 2)
        stage_rows[name] = {"ms_per_launch": round(per, 4), "launches": n, "alg_MB": round(ab / 1e6, 2),
                            "alg_GBps": round(ab / (per * 1e-3) / 1e9, 1), "frac_hbm": round(ab / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if dominant is None or per * n > stages[dominant][0]:
            dominant = name
    traffic = None
    if os.path.exists(PMC_FILE):
        try:
            pmc = json.load(open(PMC_FILE))
            ent = pmc.get(f"{N}@{W}x{H}", {}).get(dominant)
            traffic = ent["hbm_bytes_per_launch"] if ent else None
        except Exception:
            traffic = None
    valu_busy = None                                             # SQ counters of the same workload (profiles/sq_counters.json)
    sq_file = os.path.join(os.path.dirname(PMC_FILE), "sq_counters.json")
    if os.path.exists(sq_file):
        try:
            valu_busy = json.load(open(sq_file)).get(f"{N}@{W}x{H}", {}).get(dominant, {}).get("valu_busy")
        except Exception:
            valu_busy = None
    d = stage_rows[dominant]
    roofline = {"kernel": dominant, "bound": "hbm", "achieved": d["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(d["alg_GBps"] / HBM_PEAK_GBS, 5), "traffic": traffic, "valu_busy": valu_busy,
                "ms_per_launch": d["ms_per_launch"], "alg_bytes_per_launch": int(d["alg_MB"] * 1e6), "timing": stage_timing,
                "note": "blend stages are VALU-issue-bound (per pixel-splat pair work; valu_busy = measured VALUBusy of this kernel), not HBM-bound; "
                        "see `stages` for the streaming kernels"}
    op_ms = sum(ms / n for ms, n in stages.values() if n)

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import Oracle
        cores = os.cpu_count() or 1
        with torch.no_grad():
            inp = dict(means3D=pc.get_xyz.cpu(), opacities=pc.get_opacity.cpu(), shs=pc.get_features.cpu(),
                       cov3D_precomp=pc.get_covariance().cpu(), viewmatrix=cams[0].world_view_transform.cpu(),
                       projmatrix=cams[0].full_proj_transform.cpu(), campos=cams[0].camera_center.cpu(), bg=bg.cpu(),
                       image_height=H, image_width=W, tanfovx=math.tan(cams[0].FoVx / 2), tanfovy=math.tan(cams[0].FoVy / 2))
        o = Oracle(np.float32, nthreads=cores)
        n_cpu, tcpu = 0, 0.0
        while n_cpu < 4 or (tcpu < 10.0 and n_cpu < 64):                  # a bounded sample: >= 4 frames and >= 10 s of wall clock
            cam = cams[n_cpu % n_used]
            inp.update(viewmatrix=cam.world_view_transform.cpu(), projmatrix=cam.full_proj_transform.cpu(), campos=cam.camera_center.cpu())
            tc = time.perf_counter()
            st = o.forward(**inp)
            o.backward(st, up_c.cpu(), up_d.cpu(), up_a.cpu())
            tcpu += time.perf_counter() - tc
            n_cpu += 1
        cpu = {"value": round(n_cpu / tcpu, 4), "unit": "iters/s", "cores": cores, "kind": "port",
               "sample": f"{n_cpu} rasterizer forward+backward passes (op only, no loss/optimizer) over the first frames of the same "
                         f"{N}@{W}x{H} workload, oracle/raster_oracle.c with OpenMP over tiles ({tcpu:.1f} s of wall clock on {cores} cores)"}

    total_steps = world * args.steps
    out = {
        "metric": "train iters/s (fwd+bwd render) + PSNR, 500k Gaussians @ 960x540",
        "value": round(total_steps / elapsed_max, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed_max / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"S({N},{H},{W},seed0) teacher/student, 300-frame orbit, 1 frame per GPU per step; step = "
                               + ("rasterizer fwd+bwd only (seeded upstream grads on colour/depth/alpha)" if args.op_only else
                                  ("cov3D(torch) + render fwd (HIP) + 0.8 L1 + 0.2 (1-SSIM) (torch) + bwd + Adam" if args.torch_host_ops else
                                   "render fwd (HIP; activations and cov3D inside its preprocess kernel) + 0.8 L1 + 0.2 (1-SSIM) (HIP) + bwd (HIP+autograd) + Adam (HIP)")),
                   "gaussians": N, "image": [H, W], "sh_degree": D, "instances_R": int(R_mean), "instances_after_tile_culling": int(R_kept),
                   "sort_passes_max": passes,
                   "parallelism": f"frames sharded over {world} GPU(s), scalar all-reduce only",
                   "launch": "one hipGraph replay per step" if use_graph else "eager (one launch per kernel)"},
        "psnr_db": round(psnr_e, 3), "psnr_db_before": round(psnr_s, 3), "mean_loss": round(mean_loss, 6),
        "rasterizer_ms_per_step": round(op_ms, 4),
        "roofline": roofline, "stages": stage_rows, "cpu_baseline": cpu,
    }
    if world == 1 and D == 0 and not args.no_sh3_leg and not (args.op_only or args.torch_host_ops or args.no_graph):
        # The same step with the colour model the reference ends training with (max_sh_degree = 3: 16 coefficients per channel,
        # /root/reference/arguments/__init__.py), measured by this script in a child process on the same GPU and reported beside
        # the headline (which stays on the degree-0 workload the earlier rounds and the profiles were measured on).
        import subprocess
        torch.cuda.synchronize()
        try:
            leg = subprocess.run([sys.executable, os.path.abspath(__file__), "--sh-degree", "3", "--steps", str(min(args.steps, 100)), "--warmup",
                                  str(min(args.warmup, 10)), "--no-cpu-baseline", "--gaussians", str(N), "--height", str(H), "--width", str(W)],
                                 capture_output=True, text=True, timeout=600)
            j = json.loads(leg.stdout.strip().splitlines()[-1])
            out["sh_degree_3"] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                                  "psnr_db": j["psnr_db"], "stages_ms": {k: v["ms_per_launch"] for k, v in j["stages"].items()},
                                  "note": "same step, 16 SH coefficients per channel handed over as (features_dc, features_rest); Adam then "
                                          "steps 29.5 M parameters instead of 7 M"}
        except Exception as exc:                                   # the headline line must still come out
            out["sh_degree_3"] = {"error": f"{type(exc).__name__}: {exc}"}
    print(json.dumps(out))
    egs_dist.shutdown()


if __name__ == "__main__":
    main()
