#!/usr/bin/env python
"""What an N-changing step costs at config C size: densify_and_prune, then GraphedTrainStep.recapture (one eager iteration +
graph capture), against the replayed steady-state iteration.   python tools/recapture_cost.py [N] [H] [W]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import densify                                                # noqa: E402
from egogaussian_amd.graph import GraphedTrainStep                                 # noqa: E402
from egogaussian_amd.renderer import render                                        # noqa: E402
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe   # noqa: E402

N, H, W = (int(x) for x in (sys.argv[1:4] + ["500000", "540", "960"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
teacher = make_scene(N, H, W, seed=0)
cams = [make_camera(4 * k, H, W, device=dev) for k in range(8)]
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    tpc = SynthGaussians(teacher, device=dev, requires_grad=False)
    gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
pc = SynthGaussians(perturb_student(teacher), device=dev)
pc.training_setup(capturable=True)
step = GraphedTrainStep(pc, pc.optimizer, bg, densify_stats=True).capture(cams[0], gts[0], warmup=2)


def stages(tag):
    """eager per-stage HIP-event timing of the model as it is now (no optimizer step: the model stays put)"""
    from egogaussian_amd import lib
    from egogaussian_amd.fused import l1_ssim_loss
    torch.cuda.synchronize()
    lib.profile_begin(max_records=4096)
    for i in range(16):
        out = render(cams[i % 8], pc, Pipe, bg)
        l1_ssim_loss(out["render"], gts[i % 8], 0.2).backward()
        pc.optimizer.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    st = lib.profile_end()
    from egogaussian_amd import _C
    r = out["radii"].float()
    print(tag, "R", _C.stats["num_rendered"], "radii mean/max", float(r.mean()), float(r.max()), "sum (2r/16+1)^2", float(((2 * r / 16 + 1) ** 2)[r > 0].sum()),
          "scale max", float(pc.get_scaling.max()), "opacity mean", float(pc.get_opacity.mean()))
    from egogaussian_amd.renderer import get_raster_settings
    rs = get_raster_settings(cams[0], pc, bg)
    e = torch.empty(0, device=dev)
    with torch.no_grad():
        for _ in range(2):
            q = _C.rasterize_gaussians(bg, pc.get_xyz, e, pc.get_opacity, e, e, 1.0, pc.get_covariance(1.0), rs.viewmatrix, rs.projmatrix,
                                       rs.tanfovx, rs.tanfovy, H, W, pc.get_features, 0, rs.campos, False, False)
    torch.cuda.synchronize()
    v = _C.image_views(q[7], W, H)
    ln = (v["ranges"][:, 1] - v["ranges"][:, 0]).float()
    qw = v["quad_work"].float()
    rr = q[4].float()
    slots = torch.where(rr > 0, (2 * rr / 16 + 1) ** 2, torch.zeros_like(rr))
    nb = (slots.numel() + 1023) // 1024
    per_block = torch.nn.functional.pad(slots, (0, nb * 1024 - slots.numel())).view(nb, 1024).sum(1)
    print(tag, "tile list mean/p99/max", float(ln.mean()), float(ln.quantile(0.99)), float(ln.max()), "| quad_work mean/p99/max", float(qw.mean()),
          float(qw.flatten().quantile(0.99)), float(qw.max()), "| rect slots per 1024-Gaussian block mean/max", float(per_block.mean()), float(per_block.max()),
          "last 4 blocks", per_block[-4:].tolist())
    print(tag, {k: round(1e3 * ms / n, 1) for k, (ms, n) in st.items() if n})


def timed(f):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, r


ms_steps, _ = timed(lambda: [step(cams[i % 8], gts[i % 8]) for i in range(100)])
ms_r, _ = timed(lambda: step.recapture(warmup=1))
ms_steps2, _ = timed(lambda: [step(cams[i % 8], gts[i % 8]) for i in range(100)])
print(f"100 replays {ms_steps:.1f} ms | recapture alone {ms_r:.1f} ms | 100 replays {ms_steps2:.1f} ms")
stages("before densification (us):")
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    ms_steps, _ = timed(lambda: [step(cams[i % 8], gts[i % 8]) for i in range(100)])
    ms_d, (n0, n1) = timed(lambda: densify.densify_and_prune(pc, 2e-4, 0.005, 10.0, 20))
    ms_r, _ = timed(lambda: step.recapture(warmup=1))
    print(f"round {rnd}: 100 replays {ms_steps:.1f} ms | densify_and_prune {n0}->{n1}: {ms_d:.2f} ms | recapture {ms_r:.1f} ms | ok {step.ok()} | R {step.max_instances()}")
stages("after densification (us):")
