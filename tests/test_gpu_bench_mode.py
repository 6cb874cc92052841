"""GPU: the BENCHED configuration itself against the oracle, PSNR parity at a meaningful size, and the multi-rank entry point.

  * config C (500k Gaussians @ 960x540) in exactly the mode bench.py times -- raw parameters activated inside the preprocess
    kernel, tile culling on, the step replayed from a captured hipGraph -- image, radii and the five parameter gradients
    against the C oracle chained through the torch activations / covariance (north star: 1e-4 relative fp32);
  * 300 seeded training steps at config A size (10k @ 64x64), GPU chain vs oracle chain: |dPSNR| <= 0.05 dB on float images
    and after the 8-bit PNG round trip the reference's metrics go through (/root/reference/trainers/eval_metric.py:113-120,159-161);
  * bench.py --gpus 2 launched with a bare `python`: two ranks (sharing the one GPU of the test box over gloo), real
    graph-replayed steps, disjoint frame sets, equal replicas at the start, scalars reduced."""
import json
import math
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.common import rel_err, outlier_fraction, OracleRasterize, quantize_8bit

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def test_config_C_bench_mode_vs_oracle():
    from egogaussian_amd import _C
    from egogaussian_amd.covariance import covariance_from_scaling_rotation
    from egogaussian_amd.graph import GraphedTrainStep, pack_frame
    from egogaussian_amd.losses import training_loss
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from oracle.oracle import Oracle
    dev = torch.device("cuda:0")
    N, H, W = 500_000, 540, 960
    teacher = make_scene(N, H, W, seed=0)
    bg = torch.zeros(3, device=dev)
    cams = [make_camera(k, H, W, device=dev) for k in (0, 7)]
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=dev, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
    pc = SynthGaussians(perturb_student(teacher), device=dev)
    assert pc.get_raw_parameters() is not None and _C.set_tile_culling(True) in (True, False)      # the benched mode: raw parameters, culling on
    opt = FusedAdam([{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                     {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}], lr=0.0, eps=1e-15, capturable=True)
    # the benched step takes the Adam step of all five parameters inside the rasterizer backward and writes no gradient arrays;
    # here the same kernel writes them as well (AdamSink.keep_grads), so that they can be compared
    make_sink = opt.make_sink

    def keeping(**kw):
        sink = make_sink(**kw)
        sink.keep_grads = True
        return sink
    opt.make_sink = keeping
    step = GraphedTrainStep(pc, opt, bg, 0.2).capture(cams[0], gts[0], warmup=2)
    assert step.fuse_optimizer
    params = dict(xyz=pc._xyz, f_dc=pc._features_dc, opacity=pc._opacity, scaling=pc._scaling, rotation=pc._rotation)
    before = {k: v.detach().cpu().clone() for k, v in params.items()}          # the model the replayed step renders
    step(pack_frame(cams[1], gts[1]))                                          # ONE replay on another frame
    torch.cuda.synchronize()
    assert step.ok() and not step.last_frame_overflowed()
    img_hip, radii_hip = step.image.cpu().numpy(), step.radii.cpu().numpy()
    grads_hip = {k: v.grad.detach().cpu().numpy() for k, v in params.items()}
    assert all(not torch.equal(v.detach().cpu(), before[k]) for k, v in params.items())       # Adam did step

    # oracle side: torch activations + covariance (pinned by covariance.npz) -> C oracle -> torch loss (pinned by losses.npz)
    leaf = {k: v.clone().requires_grad_(True) for k, v in before.items()}
    cov = covariance_from_scaling_rotation(torch.exp(leaf["scaling"]), 1.0, leaf["rotation"])
    opac = torch.sigmoid(leaf["opacity"])
    cam = cams[1]
    o = Oracle(np.float32, nthreads=min(64, os.cpu_count() or 8))
    st = o.forward(means3D=leaf["xyz"], opacities=opac, shs=leaf["f_dc"], cov3D_precomp=cov, viewmatrix=cam.world_view_transform.cpu(),
                   projmatrix=cam.full_proj_transform.cpu(), campos=cam.camera_center.cpu(), bg=bg.cpu(), image_height=H, image_width=W,
                   tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    assert np.array_equal(radii_hip, st["radii"]), "radii differ (raw-parameter activations vs torch's differ in the last ulp at a rounding boundary?)"
    img_or = torch.tensor(st["color"], dtype=torch.float32, requires_grad=True)
    training_loss(img_or, gts[1].cpu(), 0.2).backward()
    gb = o.backward(st, img_or.grad)
    t = lambda a, like: torch.tensor(np.asarray(a, dtype=np.float32)).reshape(like.shape)
    torch.autograd.backward([cov, opac], [t(gb["dL_dcov3D"], cov), t(gb["dL_dopacity"], opac)])
    grads_or = dict(xyz=gb["dL_dmeans3D"], f_dc=gb["dL_dsh"].reshape(N, 1, 3), opacity=leaf["opacity"].grad.numpy(),
                    scaling=leaf["scaling"].grad.numpy(), rotation=leaf["rotation"].grad.numpy())
    e_img, f_img = rel_err(img_hip, st["color"]), outlier_fraction(img_hip, st["color"], TOL)
    from tests.common import flip_pixels, check_grads_isolating_flips, check_images_isolating_flips
    # (the transmittance plane of the REPLAYED frame: the captured forward's image buffer, which every replay rewrites)
    final_T_hip = _C.image_views(_C.stats["image_buffer"], W, H)["final_T"].cpu().numpy()
    rep_px = {}
    flip_px = flip_pixels(img_hip, final_T_hip, st, report=rep_px)
    # The loss has a threshold of its own: d|x - y| / dx = sign(x - y).  At a pixel-channel where the rendered value equals the ground truth to
    # float noise, the two sides' images (equal to ~1e-7) give OPPOSITE signs, and that pixel's upstream gradient differs by 2 x 0.8 / n --
    # in round 5 these rows were put down to "fp32 accumulation order", which the repeated replays below disprove (spread 1e-8).  Such a
    # pixel is found directly (the signs differ), has to prove it (both sides within 1e-6 of the ground truth there), and then counts as
    # a flipped pixel: its tile's Gaussians get the flipped pair's bound, everything else the bar.
    gt = gts[1].cpu().numpy().astype(np.float64)
    s_hip, s_or = np.sign(img_hip.astype(np.float64) - gt), np.sign(st["color"].astype(np.float64) - gt)
    tie = s_hip != s_or                                               # per pixel-channel
    sign_px = tie.any(0)
    worst_gap = float(np.maximum(np.abs(img_hip - gt), np.abs(st["color"] - gt))[tie].max()) if tie.any() else 0.0
    assert worst_gap <= 1e-6, f"the L1 term's sign differs at a pixel where an image is {worst_gap} away from the ground truth: not a tie"
    print(f"\n  config C, benched mode: R = {st['R']}, image max rel err {e_img:.2e}, pixels off by more than {2e-6:g}: {rep_px.get('detected', 0)} "
          f"({rep_px.get('flips', 0)} with a threshold-adjacent pair, {rep_px.get('noise', 0)} within their chain's float32 reach, largest {rep_px.get('max_noise_bound', 0.0):.1e}); "
          f"L1 sign ties (|image - ground truth| <= {worst_gap:.1e} on both sides, opposite signs): {int(sign_px.sum())} pixels")
    flip_px = flip_px | sign_px
    check_images_isolating_flips((("color", img_hip, st["color"]), ("final_T", final_T_hip, st["final_T"])), st, flip_px, TOL, what="config C benched mode")
    keys = list(grads_or)
    over = {}
    rep, _, _ = check_grads_isolating_flips(keys, [grads_hip[k] for k in keys], {k: np.asarray(grads_or[k]) for k in keys}, st, flip_px, TOL,
                                            what="config C benched mode", halo=10, far_frac=2e-6, far_cap=3.0, over_rows=over)     # (each side's upstream gradient comes from ITS image through the 11x11 SSIM window, forward and backward: 10 pixels)
    print("    " + rep)
    # The rows the call above let through between 1 and 10 x the bar are ACCOUNTED FOR, not excused by a comment.  The
    # claim is "fp32 accumulation order": a splat that covers thousands of pixels sums that many signed terms, and two orders of the same
    # float32 terms differ by that much.  Measured here: (i) the same replay is run three more times on the same parameters -- the atomics
    # land in another order each time -- and (ii) the float64 oracle runs the same chain.  A row is accounted for when the HIP value is no
    # further from the float64 result than the bar, than twice the float32 oracle's own distance from it, or than four times its own
    # run-to-run spread; anything else fails.
    if over:
        runs = [grads_hip]
        for _ in range(3):
            with torch.no_grad():
                for k, v in params.items():
                    v.copy_(before[k].to(dev))
            step(pack_frame(cams[1], gts[1]))
            torch.cuda.synchronize()
            runs.append({k: v.grad.detach().cpu().numpy() for k, v in params.items()})
        leaf64 = {k: v.double().clone().requires_grad_(True) for k, v in before.items()}
        cov64 = covariance_from_scaling_rotation(torch.exp(leaf64["scaling"]), 1.0, leaf64["rotation"])
        opac64 = torch.sigmoid(leaf64["opacity"])
        o64 = Oracle(np.float64, nthreads=min(64, os.cpu_count() or 8))
        st64 = o64.forward(means3D=leaf64["xyz"], opacities=opac64, shs=leaf64["f_dc"], cov3D_precomp=cov64, viewmatrix=cam.world_view_transform.cpu().double(),
                           projmatrix=cam.full_proj_transform.cpu().double(), campos=cam.camera_center.cpu().double(), bg=bg.cpu().double(), image_height=H,
                           image_width=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
        gb64 = o64.backward(st64, img_or.grad.double())            # the SAME upstream gradient as the float32 oracle's: only the rasterizer backward differs
        t64 = lambda a, like: torch.tensor(np.asarray(a, dtype=np.float64)).reshape(like.shape)
        torch.autograd.backward([cov64, opac64], [t64(gb64["dL_dcov3D"], cov64), t64(gb64["dL_dopacity"], opac64)])
        g64 = dict(xyz=gb64["dL_dmeans3D"], f_dc=gb64["dL_dsh"].reshape(N, 1, 3), opacity=leaf64["opacity"].grad.numpy(),
                   scaling=leaf64["scaling"].grad.numpy(), rotation=leaf64["rotation"].grad.numpy())
        unexplained = []
        for k, rows in over.items():
            o32 = np.asarray(grads_or[k], dtype=np.float64).reshape(N, -1); a64 = np.asarray(g64[k], dtype=np.float64).reshape(N, -1)
            scale = float(np.abs(o32).max()) + 1e-30
            stack = np.stack([np.asarray(r[k], dtype=np.float64).reshape(N, -1)[rows] for r in runs])          # [run, row, component]
            spread = (stack.max(0) - stack.min(0)).max(1) / scale
            e_h32 = np.abs(stack[0] - o32[rows]).max(1) / scale
            e_h64 = np.abs(stack[0] - a64[rows]).max(1) / scale
            e_o = np.abs(o32[rows] - a64[rows]).max(1) / scale
            for i, r in enumerate(rows):
                ok = e_h64[i] <= max(TOL, 2.0 * e_o[i], 4.0 * spread[i])
                print(f"    {k} row {int(r)} (radius {int(st['radii'][r])} px): hip vs oracle32 {e_h32[i]:.2e}, hip vs oracle64 {e_h64[i]:.2e}, oracle32 vs oracle64 {e_o[i]:.2e}, "
                      f"spread over 4 runs of the same replay {spread[i]:.2e}{'' if ok else '   <-- NOT ACCOUNTED FOR'}")
                if not ok:
                    unexplained.append((k, int(r)))
        assert not unexplained, f"rows over the 1e-4 bar that neither the float64 oracle nor the run-to-run spread accounts for: {unexplained}"


def test_training_psnr_parity_300_steps_float_and_8bit():
    """config A size, 300 seeded steps, same camera order on both sides: the product chain on the GPU (raw-parameter rasterizer, fused
    loss, FusedAdam) and the oracle chain on the CPU (torch activations + covariance, C oracle forward / analytic backward, torch
    loss, torch Adam)."""
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.losses import training_loss, psnr
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe, N_FRAMES
    dev = torch.device("cuda:0")
    N, H, W, K = 10_000, 64, 64, 300
    teacher = make_scene(N, H, W, seed=4); teacher["log_scale"] += math.log(2.5)
    # Student = teacher with xyz += N(0, 0.03^2), f_dc += N(0, 0.3^2): 23 dB at the start, 34 dB after the 300 steps and still climbing,
    # i.e. the trajectory is driven by real gradients.  (With SURVEY 8d's smaller perturbation the same 300 steps reach 49 dB, where
    # Adam at eps = 1e-15 turns last-bit differences of near-zero gradients into sign flips: two runs of the SAME GPU chain then end
    # 0.06 dB apart -- float atomics order -- which says nothing about parity.  profiles/r2_psnr_probe.txt has both regimes.)
    rng = np.random.default_rng(1001)
    student = {k: v.copy() for k, v in teacher.items()}
    student["xyz"] += rng.normal(0, 0.03, student["xyz"].shape).astype(np.float32)
    student["features"][:, :1] += rng.normal(0, 0.3, student["features"][:, :1].shape).astype(np.float32)
    train_frames, eval_frames = list(range(0, N_FRAMES, 25)), [12.5, 87.5, 162.5, 237.5]
    lrs = [("_xyz", 1.6e-4), ("_features_dc", 2.5e-3), ("_opacity", 0.05), ("_scaling", 5e-3), ("_rotation", 1e-3)]

    def const_of(cam):
        return dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center, bg=torch.zeros(3),
                    H=H, W=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), nthreads=8)

    def cpu_render(cam, pc):
        return OracleRasterize.apply(pc.get_xyz, pc.get_opacity, pc.get_features, pc.get_covariance(), const_of(cam))

    res = {}
    nt = torch.get_num_threads()
    for side in ("gpu", "cpu"):
        torch.set_num_threads(min(nt, 8) if side == "cpu" else nt)      # (10 000-row tensors on a 256-core host: thread wake-ups, not arithmetic)
        d = dev if side == "gpu" else "cpu"
        bg = torch.zeros(3, device=d)
        rend = (lambda c, p: render(c, p, Pipe, bg)["render"]) if side == "gpu" else cpu_render
        cams = [make_camera(k, H, W, device=d) for k in train_frames]
        ecams = [make_camera(k, H, W, device=d) for k in eval_frames]
        with torch.no_grad():
            tpc = SynthGaussians(teacher, device=d, requires_grad=False)
            gts, egts = [rend(c, tpc).clone() for c in cams], [rend(c, tpc).clone() for c in ecams]
        pc = SynthGaussians(student, device=d)
        groups = [{"params": [getattr(pc, a)], "lr": lr} for a, lr in lrs]
        opt = FusedAdam(groups, lr=0.0, eps=1e-15) if side == "gpu" else torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        rnd = random.Random(0)
        for it in range(K):
            k = rnd.randrange(len(cams))
            img = rend(cams[k], pc)
            loss = l1_ssim_loss(img, gts[k], 0.2) if side == "gpu" else training_loss(img, gts[k], 0.2)
            loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            imgs = [rend(c, pc) for c in ecams]
            pf = float(np.mean([psnr(i[None], g[None]).item() for i, g in zip(imgs, egts)]))
            p8 = float(np.mean([psnr(quantize_8bit(i)[None], quantize_8bit(g)[None]).item() for i, g in zip(imgs, egts)]))
            p0 = float(np.mean([psnr(rend(c, SynthGaussians(student, device=d, requires_grad=False))[None], g[None]).item() for c, g in zip(ecams, egts)]))
        res[side] = (pf, p8, p0, egts[0].cpu())
    torch.set_num_threads(nt)
    print(f"\n  PSNR on 4 held-out views after {K} steps (start {res['gpu'][2]:.2f} dB): float gpu {res['gpu'][0]:.4f} / oracle chain {res['cpu'][0]:.4f} dB; "
          f"8-bit gpu {res['gpu'][1]:.4f} / oracle chain {res['cpu'][1]:.4f} dB")
    assert rel_err(res["gpu"][3].numpy(), res["cpu"][3].numpy()) < 1e-4            # same ground truth on both sides
    assert res["gpu"][0] > res["gpu"][2] + 1.0, "training did not improve the held-out PSNR"
    assert abs(res["gpu"][0] - res["cpu"][0]) <= 0.05 and abs(res["gpu"][1] - res["cpu"][1]) <= 0.05


def test_bench_two_ranks_plain_launch():
    """`python bench.py --gpus 2` with no launcher around it: the script re-executes itself as two ranks.  The test box has one
    GPU, so both ranks share device 0 and talk over gloo (RCCL refuses two ranks on one device)."""
    env = dict(os.environ, EGS_BENCH_SHARE_DEVICE0="1", EGS_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--gaussians", "60000", "--height", "270",
           "--width", "480", "--no-cpu-baseline", "--no-sh3-leg", "--verify-ranks"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    line = [l for l in run.stdout.strip().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "weak" and j["value"] > 0
    assert j["collective_world_seen"] == 2 and j["rccl_world_seen"] is None          # (gloo here: two ranks on one device)
    r0, r1 = sorted(j["ranks"], key=lambda r: r["rank"])
    assert (r0["rank"], r1["rank"]) == (0, 1)
    assert r0["frames"] == list(range(0, 16, 2)) and r1["frames"] == list(range(1, 16, 2))       # 8 frames each (warmup + steps), disjoint
    assert not set(r0["frames"]) & set(r1["frames"])
    assert r0["param_checksum_start"] == r1["param_checksum_start"]                            # equal replicas at the start
    assert r0["graph"] and r1["graph"] and r0["overflow"]["ok"] and r1["overflow"]["ok"]       # real graph-replayed steps on both ranks
    assert r0["loss_sum"] > 0 and r1["loss_sum"] > 0 and r0["loss_sum"] != r1["loss_sum"]      # different frames -> different losses
    assert abs(j["mean_loss"] - (r0["loss_sum"] + r1["loss_sum"]) / 12) < 1e-6                 # the reduced scalar
    assert j["config"]["launch"].startswith("one hipGraph replay per")
    fa = j["fine_all_shape"]
    assert fa["n_gpus"] == 2 and fa["value"] > 0 and fa["launch"].startswith("one hipGraph") and math.isfinite(fa["psnr_db"])
    assert j["roofline"]["pairs_Q"] > 0 and j["roofline"]["visits"] > 0


def test_bench_eight_ranks_share_one_device():
    """`python bench.py --gpus 8` as the driver's 8-GPU run launches it, rehearsed on the one GPU of the test box: eight ranks on device 0
    over gloo (EGS_BENCH_SHARE_DEVICE0; RCCL refuses several ranks on one device).  Everything but RCCL itself is what the 8-GPU node
    will run: the frame sharding (rank r takes frames r, r + 8, ...), equal replicas at the start, per-rank graph-replayed steps, the
    barriers, the max-over-ranks time and the reduced scalars.  No scaling curve comes out of this -- eight ranks share one GPU."""
    env = dict(os.environ, EGS_BENCH_SHARE_DEVICE0="1", EGS_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    W8, steps, warm = 8, 4, 2
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W8), "--steps", str(steps), "--warmup", str(warm), "--gaussians", "30000", "--height", "180",
           "--width", "320", "--no-cpu-baseline", "--no-sh3-leg", "--no-fine-all-leg", "--verify-ranks", "--steps-per-replay", "2"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    j = json.loads([l for l in run.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == W8 and j["steps"] == steps and j["scaling"] == "weak" and j["value"] > 0 and j["collective"] == "gloo"
    assert j["collective_world_seen"] == W8
    ranks = sorted(j["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == list(range(W8))
    seen = set()
    for r in ranks:
        assert r["frames"] == list(range(r["rank"], W8 * (steps + warm), W8)), r           # round-robin shard, warm-up + timed frames
        assert not (seen & set(r["frames"])); seen |= set(r["frames"])
        assert r["param_checksum_start"] == ranks[0]["param_checksum_start"]                # equal replicas
        assert r["graph"] and r["overflow"]["ok"] and r["loss_sum"] > 0
        assert r["setup_s"] < 600, r                                                        # bounded set-up per rank (eight processes share one GPU here)
    assert len({r["loss_sum"] for r in ranks}) == W8                                        # different frames, different losses
    assert abs(j["mean_loss"] - sum(r["loss_sum"] for r in ranks) / (W8 * steps)) < 1e-6    # the all-reduced scalar
    assert abs(j["value"] - W8 * steps / (j["ms_per_step"] * steps * 1e-3)) < 1e-3 * j["value"]      # whole-job rate = all ranks' steps / max-over-ranks time
