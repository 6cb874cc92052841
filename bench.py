#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its quoted configuration, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
N > 1 launched plainly re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...`; launched by torchrun / the driver it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

Workload (config C of BASELINE.md): synthetic scene S(500k Gaussians, 540, 960, seed 0), the 300-frame synthetic orbit, SH
degree 0 (`--sh-degree 3` for the 16-coefficient colour model the reference ends training with; the default single-GPU
run measures that too, in a child process, and reports it as `sh_degree_3`).
One step = one full training iteration of /root/reference/trainers/train_static.py:67-138 without densification or
logging: covariance (the reference forces compute_cov3D_python, /root/reference/train.py:49; here built inside the
rasterizer's preprocess kernel from the raw parameters) -> render() forward (HIP) -> 0.8 L1 + 0.2 (1 - SSIM) ->
backward (HIP + autograd) -> Adam; replayed from one hipGraph per step (`--no-graph`: launched eagerly).
Frames are sharded round-robin over ranks (1 frame per GPU per step); ranks exchange only
scalars (loss / PSNR sums) through one RCCL all-reduce; `value` = steps of all ranks / max-over-ranks time.

Second leg, `fine_all_shape` in the same JSON line (BASELINE.json config 4, /root/reference/trainers/fine_all.py:74-101): 30 % of
the Gaussians are the object; every frame carries its own accumulated object rotation and a hand mask; the step is
render(..., rot_cov=True, accum_R=R_k, which_object=1) -> hand-mask-gated loss -> backward -> Adam, replayed from its own
hipGraph.  Same sharding, same timing rules.  `--dynamic` makes that leg the headline instead (the static one is then skipped).

The JSON line also carries
  roofline      the dominant rasterizer stage: algorithmic bytes per launch / its mean duration, measured with HIP
                events recorded by the library on the launch stream inside the timed region; pixel-splat pair throughput;
                HBM traffic and VALU counters from profiles/*.json when those were collected on THESE kernel sources;
  stages        the same for every stage (ms per launch, GB/s algorithmic);
  cpu_baseline  the C oracle (oracle/raster_oracle.c, "port") timed on this box's host cores on a bounded sample of
                forward+backward passes of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import socket
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL's device-buffer exchange needs dmabuf IPC (egogaussian_amd/dist.py); read when HSA loads

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
CLOCK_HZ, N_SIMD = 2.4e9, 1024  # same guide: 2400 MHz max clock, 256 CUs x 4 SIMDs
VALU_CYCLES_PER_WAVE_INSTR = 2.4   # tools/ubench/valu_rate.hip at 8 waves/SIMD
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")
SQ_FILE = os.path.join(ROOT, "profiles", "sq_counters.json")


def algorithmic_bytes(stage, N, R, npix, sh_coeffs=1, fused_count=False, sort_in_blend=False):
    """Bytes one launch of `stage` must move at minimum: per-unit figures x units.  R = instances the
    launch actually processes (after tile culling).  fused_count (ABI 5): the count walk of the bucketing runs inside the preprocess
    launch on what that launch holds in registers -- the `preprocess` events then cover it, `tile_bucket` is the scan and the scatter walk."""
    return {
        "preprocess": (44 + 12 * sh_coeffs) * N + 48 * N,   # xyz 12 + log-scale 12 + quaternion 16 + opacity logit 4 + sh 12/coefficient in; record 48 out
        "tile_bucket": (1 if fused_count else 2) * (16 + 32) * N + 8 * R,      # walk(s) over (tiles_touched, rect, depth | ellipse) per Gaussian; one pair out per instance
        "tile_sort": 8 * R + 4 * R,                    # pair in, index out; the radix passes stay in registers/LDS
        "render_forward": 4 * R + 48 * R + 28 * npix + (12 * R if sort_in_blend else 0),  # id + record per instance; 7 floats per pixel out (+ the per-tile sort's pair in / index out when it runs inside this launch)
        "render_backward": 4 * R + 48 * R + 40 * R + 32 * npix,   # + one 40-byte accumulate per instance; 8 floats/pixel in
        "preprocess_backward": 48 * N + (44 + 12 * sh_coeffs) * N + 32 * N + (68 + 12 * sh_coeffs) * N,  # accumulator + inputs + record head in; grads out (xyz, mean2D, scale, quat, sh, colour, opacity)
    }[stage]


def _fuses_count(N, hw):
    """Does a forward of this size run the fused preprocess + count launch (include/egs_raster.h egs_forward_fuses_count)?"""
    from egogaussian_amd import lib as egs_lib
    from egogaussian_amd import _C as egs_C
    return egs_C.forward_fuses_count(int(N), int(hw[1]), int(hw[0]))


def spawn_command(gpus, argv, port=None):
    """The command a plain `python bench.py --gpus N` re-executes itself as (one rank per GPU, rendezvous on 127.0.0.1)."""
    if port is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def stamped(path, key, current_hash):
    """Counter file -> (entry for the workload `key`, None) when it was collected on the present kernel sources, else (None, reason)."""
    if not os.path.exists(path):
        return None, f"{os.path.basename(path)} absent"
    try:
        data = json.load(open(path))
    except Exception as exc:
        return None, f"{os.path.basename(path)} unreadable ({type(exc).__name__})"
    h = data.get("_source_hash")
    if h != current_hash:
        return None, f"{os.path.basename(path)} was collected at kernel-source hash {h}, the library is now {current_hash}"
    ent = data.get(key)
    return (ent, None) if ent else (None, f"{os.path.basename(path)} has no entry for {key}")


def tile_list_stats(_C, img, W, H):
    """Lengths of the per-tile instance lists the blend kernels walk, pixel-splat pairs Q and (wave, splat) visits of the last forward."""
    iv = _C.image_views(img, W, H)
    r = iv["ranges"].long()
    ln = (r[:, 1] - r[:, 0]).float()
    return {"tile_list_len_mean": round(float(ln.mean().item()), 1), "tile_list_len_max": int(ln.max().item()),
            "pairs_Q": int(iv["quad_pairs"].long().sum().item()), "visits": int(iv["quad_visits"].long().sum().item())}


def stage_table(stages, N, R_kept, npix, sh_coeffs=1, hw=None):
    """{stage: (total ms, launches)} -> {stage: ms per launch, algorithmic MB, GB/s, fraction of the HBM roofline}"""
    rows = {}
    for name, (ms, n) in stages.items():
        if n == 0:
            continue
        per = ms / n
        ab = algorithmic_bytes(name, N, R_kept, npix, sh_coeffs, fused_count=bool(hw) and _fuses_count(N, hw), sort_in_blend="tile_sort" not in stages or not stages["tile_sort"][1])
        rows[name] = {"ms_per_launch": round(per, 4), "launches": n, "alg_MB": round(ab / 1e6, 2),
                      "alg_GBps": round(ab / (per * 1e-3) / 1e9, 1), "frac_hbm": round(ab / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return rows


def preprocess_split_leg(dev, N, H, W, sh_coeffs=1, iters=40):
    """SURVEY.md 8d holds streaming stages to >= 0.40 of the HBM roofline -- but since ABI 5 the `preprocess` launch of a forward with a
    placement buffer also walks the tile rectangles (the count pass of the bucketing), which is no streaming work.  The two are told apart
    here with the per-call flag EGS_CALL_SEPARATE_COUNT (ABI 6): the same forwards with the count pass as a launch of its own give the
    projection alone (k_preprocess: the streaming kernel the bar is about); the fused launch's duration minus that is what the walk costs
    where it now runs.  HIP events recorded by the library on the launch stream, eager forwards of S(N,H,W,seed 0), no gradient."""
    from egogaussian_amd import lib as egs_lib, _C
    from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    pc = SynthGaussians(make_scene(N, H, W, seed=0), device=dev, requires_grad=False)
    cams = [make_camera(k, H, W, device=dev) for k in range(4)]
    bg = torch.zeros(3, device=dev)
    out = {}
    for name, fused in (("fused", True), ("separate", False)):
        old = _C.set_fused_count(fused)
        try:
            with torch.no_grad():
                for k in range(6):
                    render(cams[k % 4], pc, Pipe, bg)
                torch.cuda.synchronize()
                egs_lib.profile_begin(max_records=32 * (iters + 8))
                for k in range(iters):
                    render(cams[k % 4], pc, Pipe, bg)
                torch.cuda.synchronize()
                st = egs_lib.profile_end()
        finally:
            _C.set_fused_count(old)
        out[name] = {k: ms / n for k, (ms, n) in st.items() if n}
    proj_ms, fused_ms = out["separate"]["preprocess"], out["fused"]["preprocess"]
    ab = (44 + 12 * sh_coeffs) * N + 48 * N
    del pc
    torch.cuda.empty_cache()
    return {"projection_only_us": round(proj_ms * 1e3, 2), "projection_alg_MB": round(ab / 1e6, 1),
            "projection_frac_hbm": round(ab / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "fused_projection_and_count_walk_us": round(fused_ms * 1e3, 2), "count_walk_inside_the_fused_launch_us": round((fused_ms - proj_ms) * 1e3, 2),
            "bucketing_us": {"fused (scan + scatter launches)": round(out["fused"].get("tile_bucket", 0.0) * 1e3, 2),
                             "separate (count + scan + scatter launches)": round(out["separate"].get("tile_bucket", 0.0) * 1e3, 2)},
            "how": "eager forwards with and without EGS_CALL_SEPARATE_COUNT in their flags word; HIP events recorded by the library on the launch stream"}


def config_leg(dev, N, H, W, forward_only, iters=30, log_scale_shift=0.0, scene=None, what=""):
    """One of BASELINE.json's other configurations on this GPU (parity cases elsewhere; here their timings): the rasterizer alone on
    S(N, H, W, seed 0), forward only (config 2) or forward + backward with seeded upstream gradients on colour, depth and alpha
    (config 5).  Per-stage durations from HIP events the library records around each stage on the launch stream, over `iters` eager
    calls; forward-only additionally replayed back to back from a hipGraph (frames/s).  `log_scale_shift`: ln of a factor on every
    splat's extent (the footprint legs)."""
    from egogaussian_amd import lib as egs_lib, _C
    from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    if scene is None:
        scene = make_scene(N, H, W, seed=0)
    if log_scale_shift:
        scene = dict(scene); scene["log_scale"] = scene["log_scale"] + np.float32(log_scale_shift)
    pc = SynthGaussians(scene, device=dev, requires_grad=not forward_only)
    cams = [make_camera(k, H, W, device=dev) for k in range(4)]
    bg = torch.zeros(3, device=dev)
    g = torch.Generator().manual_seed(1)
    up = [torch.rand(s_, generator=g).to(dev) for s_ in ((3, H, W), (1, H, W), (1, H, W))]

    def one(k):
        if forward_only:
            with torch.no_grad():
                return render(cams[k % 4], pc, Pipe, bg)
        out = render(cams[k % 4], pc, Pipe, bg)
        ((out["render"] * up[0]).sum() + (out["depth"] * up[1]).sum() + (out["alpha"] * up[2]).sum()).backward()
        for p_ in pc.parameters():
            p_.grad = None
        return out
    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    egs_lib.profile_begin(16 * (iters + 4))
    t0 = time.perf_counter()
    for k in range(iters):
        one(k)
    torch.cuda.synchronize()
    wall_ms = 1e3 * (time.perf_counter() - t0) / iters
    stages = egs_lib.profile_end()
    with torch.no_grad():
        render(cams[0], pc, Pipe, bg)
        torch.cuda.synchronize()
        R = int(_C.stats["num_rendered"]); R_kept = int(_C.stats["total_view"].item())
        lists = tile_list_stats(_C, _C.stats["image_buffer"], W, H)
    rows = stage_table(stages, N, R_kept, H * W, hw=(H, W))
    op_ms = sum(r["ms_per_launch"] for r in rows.values())
    alg = sum(r["alg_MB"] for r in rows.values())
    leg = {"workload": f"S({N},{H},{W},seed0){what}: rasterizer " + ("forward only, no gradient" if forward_only else
                       "forward + backward, seeded upstream gradients on colour, depth and alpha"),
           "gaussians": N, "image": [H, W], "instances_R": R, "instances_after_tile_culling": R_kept, **lists,
           "lanes_kept_per_visit": round(lists["pairs_Q"] / max(lists["visits"], 1), 2),
           "op_ms": round(op_ms, 4), "op_timing": f"sum of the per-stage means: HIP events recorded by the library on the launch stream over {iters} eager calls",
           "eager_wall_ms_per_call": round(wall_ms, 4), "alg_MB": round(alg, 1), "alg_GBps": round(alg / op_ms, 1),
           "frac_hbm": round(alg / op_ms / HBM_PEAK_GBS, 4), "stages": rows}
    if forward_only:
        per, replays = 20, 10
        with torch.no_grad():
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(per):
                    render(cams[0], pc, Pipe, bg)
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(replays):
                gr.replay()
            e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (per * replays)
        leg.update({"frames_per_s": round(1e6 / us, 1), "us_per_frame": round(us, 2),
                    "launch": f"a hipGraph of {per} forwards replayed {replays} times back to back",
                    "alg_GBps_replayed": round(alg * 1e3 / us, 1), "frac_hbm_replayed": round(alg * 1e3 / us / HBM_PEAK_GBS, 4)})
    del pc
    torch.cuda.empty_cache()
    return leg


class _ReferenceSurface:
    """A model seen through the attribute names the REFERENCE's render() and trainers use and nothing else
    (/root/reference/gaussian_renderer/__init__.py:18-107, scene/gaussian_model.py:125-200): this package's render() finds none of its
    optional hooks (raw parameters, split features, fused producers) and takes the reference's route -- get_covariance() ->
    covariance_activation, get_opacity, get_features."""
    _ALLOWED = ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features", "get_covariance", "get_rotated_covariance", "get_label",
                "get_is_object", "active_sh_degree", "max_sh_degree", "_xyz", "_is_object", "_label")

    def __init__(self, model):
        object.__setattr__(self, "_m", model)

    def __getattr__(self, name):
        if name in _ReferenceSurface._ALLOWED:
            return getattr(object.__getattribute__(self, "_m"), name)
        raise AttributeError(name)


def unchanged_trainer_legs(dev, N, H, W, steps, warmup):
    """What a reference trainer gets WITHOUT editing its loop (bench line: `reference_shaped_step`, `label_phase_shape`).
    The model is driven through the reference's attribute surface only (_ReferenceSurface), the loop is the reference's
    (/root/reference/trainers/train_static.py:67-138): render(), hand-mask gradient hook, l1_loss + ssim composed with torch scalars,
    loss.backward(), loss.item() EVERY iteration (the reference logs it), optimizer.step(), zero_grad.  `installed`: what
    egogaussian_amd.install() puts behind those names -- HIP l1_loss / ssim, HIP covariance producers, FusedAdam (patching.py; the
    reference itself is not on the GPU box, so the replacements are taken from the package directly); `import_swap_only`: PyTorch
    ops behind them (the two import names swapped and nothing else: round 4's 240 it/s).
    label_phase_shape: the object-label phase of the same loop (train_static.py:78,105-110): render() forward (its image is not part
    of that phase's loss), get_render_label() forward, channel mean, hook, BCEWithLogits against the object mask, backward -- which is
    the colours-only backward (egs_backward grad_mask == EGS_GRAD_COLORS) -- loss.item(), step."""
    import torch.nn as nn
    from egogaussian_amd import lib as egs_lib, _C, patching, losses
    from egogaussian_amd.adapter import attach
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render, get_render_label
    teacher = make_scene(N, H, W, seed=0)
    bg = torch.zeros(3, device=dev)
    n_fr = min(32, steps + warmup)
    cams = [make_camera(k, H, W, device=dev) for k in range(n_fr)]
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=dev, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
        del tpc
    gen = torch.Generator().manual_seed(11)
    hand = [(torch.rand(1, H, W, generator=gen) < 0.1).float().to(dev) for _ in range(4)]          # hand masks: 10 % of the pixels gated
    objm = [(torch.rand(1, H, W, generator=gen) < 0.3).float().to(dev) for _ in range(4)]          # object masks
    out = {}

    def timed(step_fn):
        """warm-up, `steps` timed iterations with nothing else in the loop, then a short pass with the library's per-stage HIP events on (they
        cost the host ~40 us per iteration, which an eager loop with a synchronisation per iteration cannot hide)."""
        for i in range(warmup):
            step_fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(warmup + i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        n_ev = min(steps, 24)
        egs_lib.profile_begin(max_records=32 * (n_ev + 8))
        for i in range(n_ev):
            step_fn(warmup + steps + i)
        torch.cuda.synchronize()
        st = egs_lib.profile_end()
        return el, {k: (round(ms / n, 4), n) for k, (ms, n) in st.items() if n}

    for mode in ("installed", "import_swap_only"):
        model = SynthGaussians(perturb_student(teacher), device=dev, fused=False)                 # fused=False: PyTorch ops behind every getter
        model.training_setup(optimizer_cls=torch.optim.Adam)
        torch.autograd.set_multithreading_enabled(mode != "installed")                              # what install() sets (patching.py): backward on the calling thread
        if mode == "installed":
            l1_loss, ssim = patching.make_loss_functions()
            model.covariance_activation = None
            attach(model)                                                                          # covariance producers + FusedAdam (what install() does per model)
            model.get_covariance = lambda m=1, _g=model: _g.covariance_activation(_g.get_scaling, m, _g._rotation)      # gaussian_model.py:167-168
        else:
            l1_loss, ssim = losses.l1_loss, losses.ssim
        pc = _ReferenceSurface(model)
        opt = model.optimizer

        def train_step(i):
            k = i % n_fr
            pkg = render(cams[k], pc, Pipe, bg)
            img = pkg["render"]
            hm = hand[k % 4]
            img.register_hook(lambda grad: grad * (1 - hm))
            Ll1 = l1_loss(img, gts[k])
            loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - ssim(img, gts[k]))
            loss.backward()
            loss.item()
            opt.step(); opt.zero_grad(set_to_none=True)
        from egogaussian_amd import provenance as _prov
        n_sub0 = _prov.substitutions
        el, st = timed(train_step)
        n_sub = _prov.substitutions - n_sub0
        key = "reference_shaped_step" if mode == "installed" else "import_swap_only_step"
        out[key] = {"value": round(steps / el, 2), "unit": "iters/s", "ms_per_step": round(1e3 * el / steps, 4), "steps": steps, "warmup": warmup,
                    "host_ops": ("egogaussian_amd.install(): HIP l1_loss / ssim, HIP covariance producer, FusedAdam behind the reference's names" if mode == "installed"
                                 else "PyTorch ops (only the two import names swapped)"),
                    "loop": "the reference's, unchanged: render() through the reference's attribute surface, hand-mask hook, l1_loss + ssim, backward, loss.item() "
                            "every iteration, optimizer.step(), zero_grad (/root/reference/trainers/train_static.py:67-138)",
                    "rasterizer_stage_ms": {k_: v[0] for k_, v in st.items()},
                    "raw_parameter_route": (f"{n_sub} of the leg's renders reached the rasterizer's raw-parameter path through the tagged getter results "
                                            "(egogaussian_amd/provenance.py): no activation / covariance backward launches")}
        if mode == "installed":
            # ---- label phase on the same model ----
            model._label = torch.zeros(N, 1, device=dev).requires_grad_(True)
            lab_opt = type(opt)([{"params": [model._label], "lr": 0.01, "name": "label"}], lr=0.0, eps=1e-15)
            crit = nn.BCEWithLogitsLoss()

            def label_step(i):
                k = i % n_fr
                render(cams[k], pc, Pipe, bg)                                                   # train_static.py:78 (not part of this phase's loss)
                lab = torch.mean(get_render_label(cams[k], pc, bg), dim=0, keepdim=True)
                hm = hand[k % 4]
                lab.register_hook(lambda grad: grad * (1 - hm))
                loss = crit(input=lab, target=objm[k % 4])
                loss.backward()
                loss.item()
                lab_opt.step(); lab_opt.zero_grad(set_to_none=True)
            el, st = timed(label_step)
            # the label call's backward alone, colours-only against the full backward of the same frames (HIP events on the stream)
            def bwd_ms(mask):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                tot, n = 0.0, 12
                for i in range(n + 2):
                    lab = get_render_label(cams[i % n_fr], pc, bg)
                    up = torch.ones_like(lab)
                    _C.COLORS_ONLY_BACKWARD = bool(mask)                 # False: the mask is withheld and the full backward runs
                    try:
                        ev[0].record(); lab.backward(up); ev[1].record()
                    finally:
                        _C.COLORS_ONLY_BACKWARD = True
                    torch.cuda.synchronize()
                    if i >= 2:
                        tot += ev[0].elapsed_time(ev[1])
                    model._label.grad = None
                return tot / n
            fast, full = bwd_ms(1), bwd_ms(0)
            out["label_phase_shape"] = {"value": round(steps / el, 2), "unit": "iters/s", "ms_per_step": round(1e3 * el / steps, 4), "steps": steps, "warmup": warmup,
                                        "step": "render() fwd + get_render_label() fwd + channel mean + hand-mask hook + BCEWithLogits + backward (colours only) + "
                                                "loss.item() + Adam on the labels; eager, the reference's loop",
                                        "label_backward_ms": round(fast, 4), "label_backward_full_path_ms": round(full, 4),
                                        "label_backward_ratio": round(fast / full, 3),
                                        "label_backward_kernels_ms": round(st.get("render_backward", (0, 0))[0] + st.get("preprocess_backward", (0, 0))[0], 4),
                                        "full_backward_kernels_ms": round(sum(out["reference_shaped_step"]["rasterizer_stage_ms"].get(k_, 0.0)
                                                                              for k_ in ("render_backward", "preprocess_backward")), 4),
                                        "label_backward_kernels_note": "blend + per-Gaussian kernel by the library's HIP events: k_render_backward<0> + k_colors_from_acc "
                                                                       "in this leg, k_render_backward<1> + k_preprocess_backward in reference_shaped_step (same scene, "
                                                                       "same frames; the accumulator-clearing prologue, ~7 us, is common to both)",
                                        "label_backward_note": "loss.backward() of the label render alone between two HIP events on the stream (host launch gaps included), "
                                                               "12 frames: egs_backward with grad_mask = EGS_GRAD_COLORS against the same call with the mask withheld",
                                        "rasterizer_stage_ms": {k_: v[0] for k_, v in st.items()},
                                        "reference": "/root/reference/trainers/train_static.py:78,105-110, gaussian_renderer/render_helper.py:38-64"}
        del model, pc, opt
        torch.cuda.empty_cache()
    return out


def footprint_legs(dev, train_iters=30000):
    """Heavier footprints than S(500k) (67 pairs per pixel, 918 instances per tile): the same scene with every splat three times
    larger (saturating pixels, long lists), and the model the reference's full schedule ends with on the synthetic scene -- 100k
    Gaussians initialised from simple_knn distances, densified / pruned every 100 iterations from 500 to 15 000, opacity reset every
    3 000 (examples/train_synth.py; profiles/r2_train_synth_30k.log): its arrays end with runs of clones and splits that are all on
    screen.  Rasterizer forward + backward with colour, depth and alpha gradients, per-stage timings as in config_leg."""
    out = {}
    for name, shift in (("S500k_scale_x2", math.log(2.0)), ("S500k_scale_x3", math.log(3.0))):
        try:
            out[name] = config_leg(dev, 500_000, 540, 960, False, iters=20, log_scale_shift=shift,
                                   what=f", every splat {math.exp(shift):.0f}x larger")
        except Exception as exc:
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
    try:
        scene, how = trained_scene(dev, train_iters)
        out["densified_model"] = config_leg(dev, scene["xyz"].shape[0], 540, 960, False, iters=20, scene=scene, what=how)
    except Exception as exc:
        out["densified_model"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


TRAINED_SCENE_FILE = os.path.join(ROOT, "bench_data", "trained_scene.npz")


def trained_scene(dev, train_iters=30000, regenerate=False):
    """(scene arrays, description) of a TRAINED model: what examples/train_synth.py ends with after the reference's full schedule on
    the synthetic 960x540 scene -- 100k Gaussians initialised from simple_knn distances, densified / pruned every 100 iterations from
    500 to 15 000, opacity reset every 3 000 (/root/reference/trainers/train_static.py:129-133, arguments/__init__.py:80-92).  The
    committed arrays (bench_data/trained_scene.npz, written once on an MI355X by tools/make_trained_scene.py) are used when present,
    so that every run and every test sees the same model; otherwise the schedule is run here (~20 s)."""
    if os.path.exists(TRAINED_SCENE_FILE) and not regenerate:
        z = np.load(TRAINED_SCENE_FILE)
        scene = {k: z[k] for k in ("xyz", "log_scale", "quat", "opacity_logit", "features")}
        n = scene["xyz"].shape[0]
        return scene, f" replaced by the {n}-Gaussian model examples/train_synth.py ended with after {int(z['train_iters'])} iterations of the reference's schedule (bench_data/trained_scene.npz)"
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_synth
    pc = train_synth.main(["--gaussians", "100000", "--height", "540", "--width", "960", "--iters", str(train_iters), "--knn-init",
                           "--densify-from", "500", "--densify-until", str(min(15000, train_iters)), "--densify-interval", "100",
                           "--opacity-reset-interval", "3000", "--frames", "24", "--report-every", str(max(train_iters // 6, 1))])
    n = int(getattr(pc, "n_active", pc._xyz.shape[0]))
    with torch.no_grad():
        scene = dict(xyz=pc._xyz[:n].cpu().numpy(), log_scale=pc._scaling[:n].cpu().numpy(), quat=pc._rotation[:n].cpu().numpy(),
                     opacity_logit=pc._opacity[:n].cpu().numpy(),
                     features=torch.cat((pc._features_dc[:n], pc._features_rest[:n]), 1).cpu().numpy())
    del pc
    torch.cuda.empty_cache()
    return scene, f" replaced by the {n}-Gaussian model examples/train_synth.py ends with after {train_iters} iterations of the reference's schedule (run in this process)"


def object_rotation(k, device):
    """Accumulated object rotation of frame k of the synthetic fine_all sequence (the object turns while the camera orbits)."""
    from egogaussian_amd.scene_synth import N_FRAMES
    ph = 2.0 * math.pi * (k % N_FRAMES) / N_FRAMES
    ax, ay, az = 0.35 * math.sin(ph), 0.5 * math.sin(2 * ph), 0.25 * (1 - math.cos(ph))
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
         @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    return torch.tensor(R, dtype=torch.float32, device=device)


def hand_gate(k, H, W, device):
    """1 - hand mask of frame k: a hand-sized box (a quarter of the width, a third of the height) wandering over the image."""
    from egogaussian_amd.scene_synth import N_FRAMES
    ph = 2.0 * math.pi * (k % N_FRAMES) / N_FRAMES
    cx, cy = int(W * (0.5 + 0.3 * math.cos(ph))), int(H * (0.6 + 0.25 * math.sin(2 * ph)))
    g = torch.ones((H, W), device=device)
    g[max(cy - H // 6, 0):cy + H // 6, max(cx - W // 8, 0):cx + W // 8] = 0.0
    return g


def main():
    t_process = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--sh-degree", type=int, default=0, help="spherical-harmonics degree of the colours (0..3; the reference ends training at 3)")
    ap.add_argument("--no-sh3-leg", action="store_true", help="skip the extra SH-degree-3 measurement of the default single-GPU run")
    ap.add_argument("--no-fine-all-leg", action="store_true", help="skip the fine_all-shaped (dynamic object) leg")
    ap.add_argument("--dynamic", action="store_true", help="headline = the fine_all call shape (rot_cov + accum_R + hand-mask gate) instead of the static step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--op-only", action="store_true", help="time rasterizer fwd+bwd only (seeded upstream grads)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of every step from Python instead of replaying a captured hipGraph")
    ap.add_argument("--torch-host-ops", action="store_true",
                    help="build cov3D and the loss with PyTorch ops (as the reference does) instead of the fused HIP kernels")
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="wall-clock milliseconds of forward-only renders (no training) right before the warm-up steps: clock spin-up; 0 = none")
    ap.add_argument("--steps-per-replay", type=int, default=5,
                    help="training steps captured into one hipGraph (each on its own frame); lowered to a divisor of --steps when needed; 1 = one launch per step")
    ap.add_argument("--double-buffer", action="store_true",
                    help="two captured graphs on two sets of static frames, alternated, the copy of the next replay's frames on a side stream "
                         "under the previous replay (GraphedTrainStep(double_buffer=True)).  Measured SLOWER than the one graph with the copy "
                         "between two replays (3 175 vs 3 225 it/s, three A/B pairs): off by default")
    ap.add_argument("--item-every-step", action="store_true",
                    help="with --no-graph: read loss.item() after every step, as /root/reference/trainers/train_static.py:112 does (a host "
                         "synchronisation per iteration), and report the host time spent inside render() and loss.backward()")
    ap.add_argument("--eager-fast", action="store_true",
                    help="with --no-graph: the eager loop as a trainer on this package's fast paths runs it -- the Adam step of the parameters the "
                         "render differentiates taken inside the rasterizer backward (render(optimizer=...)), the backward's preparation carried by "
                         "the loss launch, autograd on the calling thread (torch.autograd.set_multithreading_enabled(False))")
    ap.add_argument("--no-force-dist", action="store_true",
                    help="N = 1 only: do not create the one-rank RCCL group (by default the single-GPU run initialises RCCL with device_id, and its "
                         "barriers and scalar all-reduces are real collectives -- the same code path as N > 1)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the config B (100k forward-only) and config D (1M @ 1080p op-only) legs of the default run")
    ap.add_argument("--footprints", action="store_true", help="also measure the heavier-footprint legs (scale x3 scene; densified model): profiles/r3_footprint_sweep.md")
    ap.add_argument("--teacher-seed", type=int, default=0, help="seed of the teacher scene S(N,H,W,seed)")
    ap.add_argument("--unchanged-trainer-legs", action="store_true",
                    help="run ONLY the legs of the reference's unchanged loop (reference_shaped_step, import_swap_only_step, label_phase_shape) and print them "
                         "as one JSON line; the default run starts this in a child process (a fresh interpreter, as a trainer would be)")
    ap.add_argument("--verify-ranks", action="store_true", help="add per-rank frame lists, start-of-run parameter checksums and loss sums to the JSON line")
    args = ap.parse_args()

    if args.unchanged_trainer_legs:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        print(json.dumps(unchanged_trainer_legs(dev, args.gaussians, args.height, args.width, steps=min(args.steps, 200), warmup=max(args.warmup, 30))), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # launched plainly: become N ranks
        os.execv(sys.executable, spawn_command(args.gpus, sys.argv[1:]))

    from egogaussian_amd import dist as egs_dist
    rank, world, local_rank = egs_dist.env_world()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU path)"
    if os.environ.get("EGS_BENCH_SHARE_DEVICE0"):               # test hook: several ranks on one GPU (RCCL refuses that; use gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("EGS_BENCH_BACKEND", "nccl")                  # "nccl" is RCCL on ROCm
    collective_error = None
    try:
        egs_dist.init(backend, dev, force=(world == 1 and not args.no_force_dist))
    except Exception as exc:                                               # N = 1: the measurement does not depend on the group; say so and go on
        if world > 1:
            raise
        collective_error = f"{type(exc).__name__}: {exc}"
    import torch.distributed as tdist_check
    if tdist_check.is_initialized():                                       # the group that came up IS the run that was asked for
        assert tdist_check.get_world_size() == args.gpus, f"--gpus {args.gpus} but the process group has {tdist_check.get_world_size()} ranks"
        assert tdist_check.get_rank() == rank

    if args.eager_fast:
        torch.autograd.set_multithreading_enabled(False)          # backward on the calling thread: no hand-off to the device thread per step
    from egogaussian_amd import lib as egs_lib, _C
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe, N_FRAMES
    from egogaussian_amd.renderer import render
    from egogaussian_amd.losses import training_loss, psnr
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.graph import GraphedTrainStep, pack_frame
    egs_lib.load()

    N, H, W = args.gaussians, args.height, args.width
    D = args.sh_degree
    teacher = make_scene(N, H, W, seed=args.teacher_seed, sh_degree=D)
    student = perturb_student(teacher)
    is_object = (np.random.default_rng(5).uniform(size=(N, 1)) < 0.3).astype(np.float32)      # fine_all leg: 30 % object Gaussians
    bg = torch.zeros(3, device=dev)
    my_frames = egs_dist.shard_frames(N_FRAMES, rank, world)
    n_used = min(len(my_frames), args.warmup + args.steps)
    frame_ids = my_frames[:n_used]
    cams = [make_camera(k, H, W, device=dev) for k in frame_ids]
    # held-out views: eight half-frame phases spread over the orbit; no rank ever trains on them
    held_cams = [make_camera(k + 0.5, H, W, device=dev) for k in range(18, N_FRAMES, N_FRAMES // 8)][:8]
    g = torch.Generator().manual_seed(1234)
    up_c, up_d, up_a = [torch.rand(s, generator=g).to(dev) for s in ((3, H, W), (1, H, W), (1, H, W))]

    def run_leg(dynamic):
        """One measured leg: build teacher / student, capture, warm up, time args.steps steps.  -> dict of per-rank results."""
        rot = [object_rotation(k, dev) for k in frame_ids] if dynamic else None
        gates = [hand_gate(k, H, W, dev) for k in frame_ids] if dynamic else None
        held_rot = [object_rotation(k + 0.5, dev) for k in range(18, N_FRAMES, N_FRAMES // 8)][:8] if dynamic else None
        rkw = lambda R: dict(rot_cov=True, accum_R=R, which_object=1, during_training=False) if dynamic else {}

        with torch.no_grad():                                      # ground truth = teacher rendered by the same path
            tpc = SynthGaussians(teacher, device=dev, sh_degree=D, requires_grad=False)
            tpc._is_object = torch.tensor(is_object, device=dev)
            gts = [render(c, tpc, Pipe, bg, **rkw(rot[i] if dynamic else None))["render"].clone() for i, c in enumerate(cams)]
            held_gts = [render(c, tpc, Pipe, bg, **rkw(held_rot[i] if dynamic else None))["render"].clone() for i, c in enumerate(held_cams)]
            del tpc
        pc = SynthGaussians(student, device=dev, sh_degree=D, fused=not args.torch_host_ops)
        pc._is_object = torch.tensor(is_object, device=dev)
        use_graph = not (args.no_graph or args.torch_host_ops or args.op_only)
        eager_fast = bool(args.eager_fast and args.no_graph and not (args.torch_host_ops or args.op_only))
        Adam = (lambda gr, **kw: torch.optim.Adam(gr, fused=True, **kw)) if args.torch_host_ops else \
            (lambda gr, **kw: FusedAdam(gr, capturable=use_graph or eager_fast, **kw))
        opt = Adam([                                                # /root/reference/scene/gaussian_model.py:180-198 defaults
            {"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3},
            *([{"params": [pc._features_rest], "lr": 2.5e-3 / 20.0}] if D > 0 else []),
            {"params": [pc._opacity], "lr": 0.05}, {"params": [pc._scaling], "lr": 5e-3},
            {"params": [pc._rotation], "lr": 1e-3}], lr=0.0, eps=1e-15)
        with torch.no_grad():
            checksum = float(sum(p.double().sum().item() for p in (pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation)))

        def eval_psnr():
            """Mean PSNR over the held-out views (/root/reference/utils/image_utils.py:17-19)."""
            with torch.no_grad():
                vals = [psnr(render(c, pc, Pipe, bg, **rkw(held_rot[i] if dynamic else None))["render"][None], held_gts[i][None]).item()
                        for i, c in enumerate(held_cams)]
            return float(np.mean(vals))

        loss_acc = torch.zeros((), device=dev)
        r_sum = [0, 0]

        host_t = [0.0, 0.0, 0]                                      # host seconds inside render() / loss.backward(), calls
        # eager loop on the fast paths: no host wait for the instance count either -- the frame is checked at the next forward, a frame
        # that did not fit is voided on the device (egogaussian_amd/_C.py StepGuard(deferred=True))
        eguard = _C.StepGuard(dev, deferred=True) if eager_fast else None
        if eguard is not None:
            opt.guard = eguard

        def eager_step(i):
            k = i % n_used
            th0 = time.perf_counter()
            out = render(cams[k], pc, Pipe, bg, **({"optimizer": opt, "guard": eguard} if eager_fast else {}), **rkw(rot[k] if dynamic else None))
            th1 = time.perf_counter()
            if args.op_only:
                loss = (out["render"] * up_c).sum() + (out["depth"] * up_d).sum() + (out["alpha"] * up_a).sum()
            elif args.torch_host_ops:
                img = out["render"]
                if dynamic:
                    gk = gates[k]
                    img.register_hook(lambda grad: grad * gk)
                loss = training_loss(img, gts[k])
            else:
                loss = l1_ssim_loss(out["render"], gts[k], 0.2, grad_gate=gates[k] if dynamic else None, raster_prologue=eager_fast,
                                    raster_lossgrad=eager_fast)       # (no loss-backward launch: the blend computes the image gradient itself)
            th2 = time.perf_counter()
            loss.backward()
            th3 = time.perf_counter()
            if not args.op_only:
                opt.step()
            opt.zero_grad(set_to_none=True)
            if args.item_every_step:
                loss.item()                                          # the reference logs the loss every iteration: one host synchronisation
            loss_acc.add_(loss.detach())
            r_sum[0] += _C.stats["num_rendered"]; r_sum[1] += 1
            host_t[0] += th1 - th0; host_t[1] += th3 - th2; host_t[2] += 1

        psnr_start = eval_psnr()
        step, graphed = eager_step, None
        spr = 1
        if use_graph:                                               # the whole iteration as one hipGraph (egogaussian_amd/graph.py)
            spr = next(k for k in range(max(1, min(args.steps_per_replay, args.steps)), 0, -1) if args.steps % k == 0)
            try:
                graphed = GraphedTrainStep(pc, opt, bg, 0.2, dynamic=dynamic, gated=dynamic, steps_per_replay=spr, double_buffer=args.double_buffer)
                graphed.capture(cams[0], gts[0], warmup=2, accum_R=rot[0] if dynamic else None, gate=gates[0] if dynamic else None,
                                capacity_cams=cams[::max(1, n_used // 6)])
            except Exception as exc:                                # keep measuring, eagerly, rather than lose the run
                print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); stepping eagerly", file=sys.stderr)
                graphed, use_graph = None, False
            if graphed is not None:
                # resident: image + camera block (+ object rotation + gate) of every frame in ONE tensor, frame order = step order (the
                # first spr - 1 frames repeated at the end), so that the spr frames of a replay are one contiguous copy
                fl = [pack_frame(c, g_, rot[i] if dynamic else None, gates[i] if dynamic else None) for i, (c, g_) in enumerate(zip(cams, gts))]
                frames = torch.stack(fl + fl[:spr - 1])
                del fl

                def step(i):                                        # noqa: F811   (i = index of the first of spr consecutive steps)
                    k = i % n_used
                    graphed(frames[k:k + spr])                       # (the captured steps add their losses to graphed.loss_sum themselves)
                    r_sum[1] += spr
            else:
                spr = 1
        # Spin-up: forward-only renders of training views (no gradient, no parameter touched) for --spinup-ms of wall clock, so
        # that the warm-up and the timed steps do not start on a GPU that has dropped to its idle clocks during the host-side set-up
        # above.  A --steps 20 run otherwise measures the clock ramp: 0.352 ms/step against 0.326 at --steps 200 and 0.320 at 2000.
        if args.spinup_ms > 0:
            t_spin, i_spin = time.perf_counter() + args.spinup_ms * 1e-3, 0
            with torch.no_grad():
                while time.perf_counter() < t_spin:
                    render(cams[i_spin % n_used], pc, Pipe, bg, **rkw(rot[i_spin % n_used] if dynamic else None))
                    i_spin += 1
        # EXACTLY --warmup untimed steps: whole replays, the remainder (warmup % spr) launched eagerly first
        n_warm = args.warmup
        for i in range(n_warm % spr if graphed is not None else 0):
            eager_step(i)
        for i in range(n_warm % spr if graphed is not None else 0, n_warm, spr):
            step(i)
        loss_acc.zero_(); r_sum[:] = [0, 0]; host_t[:] = [0.0, 0.0, 0]
        if graphed is not None:
            graphed.loss_sum.zero_()
        # (the event pool of the stage timer is set up BEFORE the last synchronisation: nothing but the barrier sits between the
        # warm-up and the timed region, so the GPU does not idle into a lower power state right before it is timed)
        egs_lib.profile_begin(max_records=min(200000, 16 * (args.steps + 8)))
        torch.cuda.synchronize()
        egs_dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_timed_start = t0
        for i in range(0, args.steps, spr):                         # EXACTLY args.steps training steps (spr per launch)
            step(n_warm + i)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0                          # this rank's K steps are complete; the MAX over ranks is taken below
        egs_dist.barrier()
        torch.cuda.synchronize()
        loss_first = (graphed.loss_sum if graphed is not None else loss_acc).clone()      # the loss sum of the contract's region
        r_first = list(r_sum)
        # the SAME region four more times (training goes on: the workload sheds pairs as the student converges):
        # `value` stays the first -- the contract's -- and the line adds the median and the spread of the five
        reps = [elapsed]
        for rep in range(1, 5):
            t_r = time.perf_counter()
            for i in range(0, args.steps, spr):
                step(n_warm + rep * args.steps + i)
            torch.cuda.synchronize()
            reps.append(time.perf_counter() - t_r)
            egs_dist.barrier()
        (graphed.loss_sum if graphed is not None else loss_acc).copy_(loss_first); r_sum[:] = r_first
        stages = egs_lib.profile_end()
        host_us = None if (graphed is not None or not host_t[2]) else (1e6 * host_t[0] / host_t[2], 1e6 * host_t[1] / host_t[2])
        eager_overflows = None
        if eguard is not None:
            eguard.check()
            eager_overflows = int(eguard.overflows)                  # frames voided on the device (reported, not fatal: the run is still a measurement)
        stage_timing = "HIP events recorded by the library on the launch stream inside the timed region"
        overflow = None
        if graphed is not None:
            overflow = {"capacity": graphed.capacity, "max_instances": graphed.max_instances(), "ok": graphed.ok()}
            assert graphed.ok(), f"a replayed frame exceeded the captured capacity ({graphed.max_instances()} > {graphed.capacity}); its update was skipped"
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
        # instances that survive tile culling (what the sort and the blend kernels process), pixel-splat pairs Q and (wave, splat)
        # visits of the forward blend: sampled on four frames
        kept = rect = pairs = visits = 0
        list_mean, list_max = [], []
        with torch.no_grad():
            for i in range(0, n_used, max(1, n_used // 4)):
                render(cams[i], pc, Pipe, bg, **rkw(rot[i] if dynamic else None))
                kept += int(_C.stats["total_view"].item()); rect += _C.stats["num_rendered"]
                ls = tile_list_stats(_C, _C.stats["image_buffer"], W, H)
                pairs += ls["pairs_Q"]; visits += ls["visits"]
                list_mean.append(ls["tile_list_len_mean"]); list_max.append(ls["tile_list_len_max"])
        n_s = len(range(0, n_used, max(1, n_used // 4)))
        return dict(lg_in_blend=bool(graphed is not None and getattr(graphed, "loss_grad_in_blend", False)),
                    elapsed=elapsed, stages=stages, stage_timing=stage_timing, loss=loss_acc.item(), psnr_end=psnr_end, psnr_start=psnr_start,
                    R_mean=float(r_sum[0]) / max(r_sum[1], 1), kept_ratio=kept / max(rect, 1), pairs=pairs / n_s, visits=visits / n_s,
                    list_mean=float(np.mean(list_mean)), list_max=int(max(list_max)),
                    use_graph=use_graph, checksum=checksum, overflow=overflow, pc=pc, cams=cams, spr=spr, host_us=host_us, t_timed_start=t_timed_start, reps=reps,
                    eager_overflows=eager_overflows)

    def reduce_leg(r):
        """Scalars only (RCCL over xGMI): max-over-ranks time, sums of loss / PSNR / instance counts."""
        elapsed_max = egs_dist.reduce_scalars([r["elapsed"]], dev, "max")[0]
        reps_max = egs_dist.reduce_scalars(list(r["reps"]), dev, "max")
        sums = egs_dist.reduce_scalars([r["loss"], r["psnr_end"], r["psnr_start"], r["R_mean"]], dev, "sum")
        return dict(elapsed_max=elapsed_max, reps_max=reps_max, mean_loss=sums[0] / (world * args.steps), psnr=sums[1] / world, psnr_before=sums[2] / world,
                    R_mean=sums[3] / world)

    legs = {}
    if not args.dynamic:
        legs["static"] = run_leg(False)
        legs["static"]["red"] = reduce_leg(legs["static"])
    if args.dynamic or not (args.no_fine_all_leg or args.op_only or args.torch_host_ops or args.no_graph or D > 0):
        legs["dynamic"] = run_leg(True)
        legs["dynamic"]["red"] = reduce_leg(legs["dynamic"])
    head_name = "dynamic" if args.dynamic else "static"
    head, red = legs[head_name], legs[head_name]["red"]

    world_seen = egs_dist.world_seen(dev)                       # an all-reduce of ones on device tensors: the ranks the COLLECTIVE saw
    ranks_info = None
    if args.verify_ranks:
        import torch.distributed as tdist
        mine = {"rank": rank, "frames": frame_ids, "param_checksum_start": head["checksum"], "loss_sum": head["loss"], "steps": args.steps,
                "device": str(dev), "graph": head["use_graph"], "overflow": head["overflow"],
                "setup_s": round(head["t_timed_start"] - t_process, 2)}
        if world > 1:
            ranks_info = [None] * world
            tdist.all_gather_object(ranks_info, mine)
        else:
            ranks_info = [mine]

    if rank != 0:
        egs_dist.shutdown()
        return

    npix = H * W
    passes = _C.binning_passes(N, W, H)
    stages = head["stages"]
    R_mean = red["R_mean"]
    R_kept = R_mean * head["kept_ratio"]
    stage_rows, dominant = {}, None
    for name, (ms, n) in stages.items():
        if n == 0:
            continue
        per = ms / n
        ab = algorithmic_bytes(name, N, R_kept, npix, (D + 1) ** 2, fused_count=_fuses_count(N, (H, W)), sort_in_blend="tile_sort" not in stages or not stages["tile_sort"][1])
        stage_rows[name] = {"ms_per_launch": round(per, 4), "launches": n, "alg_MB": round(ab / 1e6, 2),
                            "alg_GBps": round(ab / (per * 1e-3) / 1e9, 1), "frac_hbm": round(ab / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if dominant is None or per * n > stages[dominant][0]:
            dominant = name
    # counters collected by rocprofv3 --pmc in runs of their own (tools/collect_counters.py): valid for THESE kernels only
    src_hash = egs_lib.kernel_source_hash()
    key = f"{N}@{W}x{H}"
    pmc, why_pmc = stamped(PMC_FILE, key, src_hash)
    sq, why_sq = stamped(SQ_FILE, key, src_hash)
    traffic = (pmc or {}).get(dominant, {}).get("hbm_bytes_per_launch") if pmc else None
    sq_dom = (sq or {}).get(dominant) if sq else None
    d = stage_rows[dominant]
    # the dominant kernel's duration INSIDE the replayed step, when a rocprofv3 trace of this very library is on file (the eager event
    # pass behind `stages` runs the same kernel ~8 % slower: cold caches between launches that the host spaces out)
    budget, why_budget = None, "no profiles/graph_step_budget.json"
    try:
        bj = json.load(open(os.path.join(ROOT, "profiles", "graph_step_budget.json")))
        if bj.get("_source_hash") == src_hash:
            budget = bj
        else:
            why_budget = f"profiles/graph_step_budget.json is stamped {bj.get('_source_hash')}, the library is {src_hash}"
    except Exception:
        pass
    replay_kernel = {"render_backward": "k_render_backward", "render_forward": "k_render_forward"}.get(dominant)
    replay_us = None
    if budget and replay_kernel and head["use_graph"] and key == "500000@960x540":
        replay_us = next((v for k_, v in budget["kernels_us"].items() if replay_kernel in k_), None)     # (the trace may hold mangled names)
    # Byte conventions.  `achieved` / `frac` follow SURVEY.md 8d for the dominant kernel -- the backward blend: 84 B per
    # instance (44 list re-read + 40 accumulate) + 32 B per pixel -- over the duration MEASURED IN THIS RUN (HIP events on the launch stream).
    # Named beside it, never folded in: the record as this library packs it (92 B per instance: 4 id + 48 record + 40 accumulate), and the
    # bytes of the image loss's backward that the replayed blend also does (ABI 5: 20 B per pixel-channel of maps, image, ground truth).
    t_dom = d["ms_per_launch"] * 1e-3                                 # live: this run's events
    alg_bytes = d["alg_MB"] * 1e6                                     # the stage table's convention (as packed)
    if dominant == "render_backward":
        alg_8d = 84 * R_kept + 32 * npix
    elif dominant == "render_forward":
        alg_8d = 44 * R_kept + 28 * npix
    else:
        alg_8d = alg_bytes
    lg_bytes = 20 * 3 * npix if (dominant == "render_backward" and head.get("lg_in_blend")) else 0
    replayed = None
    if replay_us:
        t_rep = replay_us * 1e-6
        replayed = {"ms_per_launch": round(replay_us * 1e-3, 5), "source": f"profiles/graph_step_budget.json: rocprofv3 --kernel-trace of the graph-replayed step on the same "
                    f"kernel sources ({budget['steps']} steps), NOT measured in this run",
                    "frac": round(alg_8d / t_rep / 1e9 / HBM_PEAK_GBS, 5),
                    **({"frac_incl_loss_gradient_bytes": round((alg_8d + lg_bytes) / t_rep / 1e9 / HBM_PEAK_GBS, 5)} if lg_bytes else {})}
    # Every launch of the REPLAYED step: duration from the rocprofv3 kernel trace on file for these kernel sources,
    # algorithmic bytes, HBM bytes from the PMC passes on file where a pass covers exactly that launch.
    step_table = None
    if budget and key == "500000@960x540":
        n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
        stride = 4
        while stride < (N + 1023) // 1024:
            stride <<= 1
        M1 = (D + 1) ** 2
        rows_def = [
            ("k_preprocess_count", "projection + count walk of the bucketing (one launch, ABI 5)", (44 + 12 * M1) * N + 48 * N + 48 * N, "preprocess"),
            ("k_table_scan", "scan of the (tile, workgroup) count table", 2 * 4 * n_tiles * stride, None),
            ("k_bin_scatter", "scatter walk of the bucketing", 48 * N + 8 * R_kept, None),
            ("k_render_forward", "per-tile sort + forward blend", 4 * R_kept + 48 * R_kept + 28 * npix + 12 * R_kept, "render_forward"),
            ("k_l1_ssim_forward", "image loss forward (value + three derivative maps)", 60 * npix, "loss"),
            ("k_render_backward", "backward blend + the image loss's gradient for its tile", 84 * R_kept + 32 * npix + 60 * npix, "render_backward"),
            ("k_preprocess_backward", "per-Gaussian backward + Adam step of five parameters", (48 + 32 + 56 + 56 + 112 + 112) * N, None),
        ]
        step_table = []
        for kname, what, ab, pmc_key in rows_def:
            us = next((v for k_, v in budget["kernels_us"].items() if kname in k_), None)
            if us is None:
                continue
            cnt = (pmc or {}).get(pmc_key, {}).get("hbm_bytes_per_launch") if (pmc and pmc_key) else None
            step_table.append({"kernel": kname, "what": what, "us": round(us, 2), "alg_MB": round(ab / 1e6, 1),
                               "frac_hbm": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "counter_MB": None if cnt is None else round(cnt / 1e6, 1), "counter_over_alg": None if cnt is None else round(cnt / ab, 2)})
    issue_frac = None
    if sq_dom and sq_dom.get("valu_wave_instructions"):
        # share of the chip's VALU issue capacity the kernel's vector instructions account for: wave-instructions x measured
        # cycles per wave-instruction / (SIMDs x clock x kernel time)
        issue_frac = round(sq_dom["valu_wave_instructions"] * VALU_CYCLES_PER_WAVE_INSTR / (N_SIMD * CLOCK_HZ * t_dom), 4)
    pair_rows = {}
    for st_name in ("render_forward", "render_backward"):
        if st_name in stage_rows:
            pair_rows[st_name] = round(head["pairs"] / (stage_rows[st_name]["ms_per_launch"] * 1e-3) / 1e9, 2)
    achieved = alg_8d / t_dom / 1e9
    roofline = {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_note": why_pmc or "profiles/pmc_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate --pmc passes, same kernel sources",
                "valu_busy": (sq_dom or {}).get("valu_busy"), "issue_frac": issue_frac,
                "counters_note": why_sq or "profiles/sq_counters.json: SQ counters of the same kernel sources",
                "kernel_source_hash": src_hash,
                "pairs_Q": int(head["pairs"]), "visits": int(head["visits"]), "lanes_kept_per_visit": round(head["pairs"] / max(head["visits"], 1), 2),
                "pairs_per_s_G": pair_rows,
                "ms_per_launch": d["ms_per_launch"], "alg_bytes_per_launch": int(alg_8d),
                "alg_bytes_convention": "SURVEY.md 8d: 84 B per instance after tile culling + 32 B per pixel (backward blend); 44 B + 28 B (forward blend)",
                "timing": head["stage_timing"],
                "as_packed": {"alg_bytes_per_launch": int(alg_bytes), "frac": round(alg_bytes / t_dom / 1e9 / HBM_PEAK_GBS, 5),
                              "note": "an instance priced as this library packs it: 4 B id + 48 B record + 40 B accumulate = 92 B"},
                **({"loss_gradient_bytes": lg_bytes, "loss_gradient_note": "the graph-replayed blend also computes dL/dimage for its tile (no loss-backward launch, ABI 5): "
                    "20 B per pixel-channel of maps, image and ground truth; NOT in `achieved` / `frac` (the live events time the eager blend, which loads dL/dimage)"} if lg_bytes else {}),
                **({"replayed": replayed} if replayed else {"replayed_note": why_budget}),
                "note": "blend stages are VALU-issue-bound (per pixel-splat pair work), not HBM-bound: pairs_Q = (pixel, splat) pairs one frame "
                        "blends, visits = (8x8-pixel wave, splat) iterations of the forward; see `stages` for the streaming kernels"}
    op_ms = sum(ms / n for ms, n in stages.values() if n)

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import Oracle
        pc, ccams = head["pc"], head["cams"]
        cores = os.cpu_count() or 1
        with torch.no_grad():
            inp = dict(means3D=pc.get_xyz.cpu(), opacities=pc.get_opacity.cpu(), shs=pc.get_features.cpu(),
                       cov3D_precomp=pc.get_covariance().cpu(), viewmatrix=ccams[0].world_view_transform.cpu(),
                       projmatrix=ccams[0].full_proj_transform.cpu(), campos=ccams[0].camera_center.cpu(), bg=bg.cpu(),
                       image_height=H, image_width=W, tanfovx=math.tan(ccams[0].FoVx / 2), tanfovy=math.tan(ccams[0].FoVy / 2))
        o = Oracle(np.float32, nthreads=cores)
        n_cpu, tcpu = 0, 0.0
        while n_cpu < 4 or (tcpu < 10.0 and n_cpu < 64):                  # a bounded sample: >= 4 frames and >= 10 s of wall clock
            cam = ccams[n_cpu % n_used]
            inp.update(viewmatrix=cam.world_view_transform.cpu(), projmatrix=cam.full_proj_transform.cpu(), campos=cam.camera_center.cpu())
            tc = time.perf_counter()
            st = o.forward(**inp)
            o.backward(st, up_c.cpu(), up_d.cpu(), up_a.cpu())
            tcpu += time.perf_counter() - tc
            n_cpu += 1
        cpu = {"value": round(n_cpu / tcpu, 4), "unit": "iters/s", "cores": cores, "kind": "port", "step": "op_only",
               "gpu_op_only": {"value": round(1e3 / op_ms, 1), "unit": "iters/s", "ms": round(op_ms, 4),
                               "what": "the same unit of work on the GPU: rasterizer forward + backward alone (sum of the per-stage HIP-event means of "
                                       "this run; `value` of the line is the full training step)"},
               "sample": f"{n_cpu} rasterizer forward+backward passes (op only, no loss/optimizer) over the first frames of the same "
                         f"{N}@{W}x{H} workload, oracle/raster_oracle.c with OpenMP over tiles ({tcpu:.1f} s of wall clock on {cores} cores)"}

    total_steps = world * args.steps
    step_text = {
        "static": ("rasterizer fwd+bwd only (seeded upstream grads on colour/depth/alpha)" if args.op_only else
                   ("cov3D(torch) + render fwd (HIP) + 0.8 L1 + 0.2 (1-SSIM) (torch) + bwd + Adam" if args.torch_host_ops else
                    "render fwd (HIP; activations and cov3D inside its preprocess kernel) + 0.8 L1 + 0.2 (1-SSIM) (HIP) + bwd (HIP+autograd) + Adam (HIP)")),
        "dynamic": "fine_all shape: object-rotated cov3D + opacity (HIP, one launch) + render(rot_cov=True, accum_R=R_k, which_object=1) fwd (HIP) + "
                   "hand-mask-gated 0.8 L1 + 0.2 (1-SSIM) (HIP) + bwd + Adam (HIP); 30 % object Gaussians"}
    out = {
        "metric": "train iters/s (fwd+bwd render) + PSNR, 500k Gaussians @ 960x540",
        "value": round(total_steps / red["elapsed_max"], 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * red["elapsed_max"] / args.steps, 4), "higher_is_better": True,
        "value_median": round(total_steps / sorted(red["reps_max"])[len(red["reps_max"]) // 2], 3),
        "value_spread": [round(total_steps / max(red["reps_max"]), 3), round(total_steps / min(red["reps_max"]), 3)],
        "value_note": "`value` is the first timed region of exactly --steps steps after --warmup (the contract); the same region is then repeated "
                      "four times while training continues: value_median / value_spread = median and [min, max] of the five",
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "collective": egs_dist.collective_name(), **({"collective_error": collective_error} if collective_error else {}),
        "rccl_world_seen": world_seen if egs_dist.collective_name() == "rccl" else None, "collective_world_seen": world_seen,
        "config": {"workload": f"S({N},{H},{W},seed{args.teacher_seed}) teacher/student, 300-frame orbit, 1 frame per GPU per step; step = " + step_text[head_name],
                   "gaussians": N, "image": [H, W], "sh_degree": D, "instances_R": int(R_mean), "instances_after_tile_culling": int(R_kept),
                   "teacher_seed": args.teacher_seed,
                   "teacher_seed_note": "SURVEY.md 8d names instance C as S(500k,540,960,seed 0) and, in its ground-truth sentence, a teacher of seed 1; "
                                        "the instance list is followed (as in rounds 1-3, so the numbers stay comparable); --teacher-seed 1 runs the other",
                   "tile_list_len_mean": round(head["list_mean"], 1), "tile_list_len_max": head["list_max"],
                   "sort_passes_max": passes,
                   "parallelism": f"frames sharded over {world} GPU(s), scalar all-reduce only",
                   "launch": (f"one hipGraph replay per {head['spr']} steps ({head['spr']} complete iterations, each on its own frame, captured back to back)"
                              if head["spr"] > 1 else "one hipGraph replay per step") if head["use_graph"] else "eager (one launch per kernel)",
                   "optimizer": "Adam steps of all five parameters inside the rasterizer backward (egs_backward_adam), no gradient arrays" if head["use_graph"] else "FusedAdam.step(), one launch",
                   "spinup_ms": args.spinup_ms,
                   "spinup": "forward-only renders (no training) for that many ms of wall clock right before the warm-up steps: GPU clocks out of idle"},
        "psnr_db": round(red["psnr"], 3), "psnr_db_before": round(red["psnr_before"], 3),
        "psnr_views": "8 held-out views at half-frame phases across the orbit (never trained on)", "mean_loss": round(red["mean_loss"], 6),
        "rasterizer_ms_per_step": round(op_ms, 4),
        **({"host_us_per_forward": round(head["host_us"][0], 1), "host_us_per_backward": round(head["host_us"][1], 1),
            "host_us_note": "host time inside render() and inside loss.backward() per step (Python, ctypes, launches, the wait for the instance count), "
                            "GPU not waited for"} if head.get("host_us") else {}),
        "roofline": roofline, "stages": stage_rows,
        **({"replayed_step_kernels": step_table,
            "replayed_step_kernels_note": f"profiles/graph_step_budget.json (rocprofv3 --kernel-trace, {budget['steps']} replayed steps, same kernel sources: {src_hash}); "
                                          "counter_MB = (2 FETCH_SIZE + WRITE_SIZE) x 1024 per launch from profiles/pmc_traffic.json where a PMC pass covers exactly that launch "
                                          "(the bucketing's scan + scatter were counted together: 'tile_bucket' there)"} if step_table else {}),
        "cpu_baseline": cpu,
        **({"eager_frames_voided": head["eager_overflows"]} if head.get("eager_overflows") is not None else {}),
        **({"valid": False, "invalid_reason": f"{head['eager_overflows']} frame(s) of the timed loop exceeded the instance capacity and were voided on the device: "
                                              "`value` counts steps that did no update"} if head.get("eager_overflows") else {}),
    }
    if head_name == "static" and "dynamic" in legs:
        dl, dr = legs["dynamic"], legs["dynamic"]["red"]
        out["fine_all_shape"] = {
            "value": round(total_steps / dr["elapsed_max"], 3), "unit": "iters/s", "ms_per_step": round(1e3 * dr["elapsed_max"] / args.steps, 4),
            "steps": args.steps, "n_gpus": world, "psnr_db": round(dr["psnr"], 3), "psnr_db_before": round(dr["psnr_before"], 3),
            "mean_loss": round(dr["mean_loss"], 6), "instances_R": int(dr["R_mean"]),
            "stages_ms": {k: round(ms / n, 4) for k, (ms, n) in dl["stages"].items() if n},
            "launch": (f"one hipGraph replay per {dl['spr']} steps" if dl["spr"] > 1 else "one hipGraph replay per step") if dl["use_graph"] else "eager",
            "step": step_text["dynamic"],
            "reference": "/root/reference/trainers/fine_all.py:74-101 (BASELINE.json config 4)"}
    if ranks_info is not None:
        out["ranks"] = ranks_info
    if world == 1 and D == 0 and not args.no_sh3_leg and not (args.op_only or args.torch_host_ops or args.no_graph or args.dynamic):
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
        # The reference's trainer loop UNCHANGED (render through the reference's attribute surface, l1_loss + ssim by name, loss.item() every
        # iteration): with what egogaussian_amd.install() puts behind those names, with PyTorch ops behind them (import swap only), and its
        # label phase.  In this process, after the headline legs.
        try:
            leg = subprocess.run([sys.executable, os.path.abspath(__file__), "--unchanged-trainer-legs", "--steps", str(min(args.steps, 200)), "--warmup", "30",
                                  "--gaussians", str(N), "--height", str(H), "--width", str(W)], capture_output=True, text=True, timeout=900)
            out.update(json.loads(leg.stdout.strip().splitlines()[-1]))
        except Exception as exc:
            out["reference_shaped_step"] = {"error": f"{type(exc).__name__}: {exc}"}
        # Between the two: every kernel launched from Python (no hipGraph), but with this package's host ops -- raw parameters into the
        # rasterizer, the fused loss, FusedAdam.step() -- and `loss.item()` read every iteration as the reference's loop does
        # (/root/reference/trainers/train_static.py:112).  What a trainer gets from `import egogaussian_amd; egogaussian_amd.attach(gaussians)`
        # plus this package's render() and loss, with its loop otherwise unchanged.
        try:
            leg = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-graph", "--item-every-step", "--eager-fast", "--steps", str(min(args.steps, 200)),
                                  "--warmup", "20", "--no-cpu-baseline", "--gaussians", str(N), "--height", str(H), "--width", str(W)],
                                 capture_output=True, text=True, timeout=600)
            j = json.loads(leg.stdout.strip().splitlines()[-1])
            out["eager_fused_step"] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                                       "host_us_per_forward": j.get("host_us_per_forward"), "host_us_per_backward": j.get("host_us_per_backward"),
                                       "rasterizer_ms_per_step": j["rasterizer_ms_per_step"], "psnr_db": j["psnr_db"],
                                       "launch": "eager: every kernel launched from Python, loss.item() after every step (one host synchronisation per iteration); "
                                                 "render(optimizer=FusedAdam): the Adam step inside the rasterizer backward; autograd on the calling thread",
                                       "step": j["config"]["workload"].split("step = ")[-1]}
        except Exception as exc:
            out["eager_fused_step"] = {"error": f"{type(exc).__name__}: {exc}"}
        # ... and the same loop for a trainer that reads the loss only now and then: with the instance count checked at the NEXT forward
        # (StepGuard(deferred=True)) nothing in the iteration waits for the GPU, so the host runs ahead and the two overlap
        try:
            leg = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-graph", "--eager-fast", "--steps", str(min(args.steps, 200)),
                                  "--warmup", "20", "--no-cpu-baseline", "--gaussians", str(N), "--height", str(H), "--width", str(W)],
                                 capture_output=True, text=True, timeout=600)
            j = json.loads(leg.stdout.strip().splitlines()[-1])
            out["eager_fused_step_no_sync"] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                                               "host_us_per_forward": j.get("host_us_per_forward"), "host_us_per_backward": j.get("host_us_per_backward"),
                                               "psnr_db": j["psnr_db"],
                                               "launch": "eager, no host synchronisation inside the loop: the forward is enqueued against the capacity hint and "
                                                         "checked at the next forward (a frame that did not fit is voided on the device)"}
        except Exception as exc:
            out["eager_fused_step_no_sync"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not args.no_config_legs and not (args.op_only or args.torch_host_ops or args.no_graph or args.dynamic or D > 0):
        # BASELINE.json configs 2 and 5 on the same GPU, in this process (the training legs' tensors are released first)
        legs.clear(); head = None
        torch.cuda.empty_cache()
        for name, kw in (("config_B_forward_only", dict(N=100_000, H=540, W=960, forward_only=True, iters=40)),
                         ("config_D_op_only", dict(N=1_000_000, H=1080, W=1920, forward_only=False, iters=20))):
            try:
                out[name] = config_leg(dev, **kw)
                out[name]["reference"] = ("/root/reference/trainers/eval_metric.py:107 (no-grad render), BASELINE.json config 2" if kw["forward_only"] else
                                          "/root/reference/gaussian_renderer/__init__.py:90-98 with depth + alpha gradients, BASELINE.json config 5")
            except Exception as exc:
                out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            out["preprocess_split"] = preprocess_split_leg(dev, N, H, W, (D + 1) ** 2)
        except Exception as exc:
            out["preprocess_split"] = {"error": f"{type(exc).__name__}: {exc}"}
        # A TRAINED scene: what the reference's trainers and its evaluation actually render (densified every 100 iterations,
        # /root/reference/trainers/train_static.py:129-133; rendered by trainers/fine_all.py:93 and trainers/eval_metric.py:107) -- a few
        # screen-filling splats among many small, faint ones.  The committed model of bench_data/ (tools/make_trained_scene.py).
        try:
            scene, how = trained_scene(dev)
            out["trained_scene_op_only"] = config_leg(dev, scene["xyz"].shape[0], 540, 960, False, iters=30, scene=scene, what=how)
            out["trained_scene_op_only"]["reference"] = "/root/reference/trainers/train_static.py:129-133 (densification), trainers/fine_all.py:93, trainers/eval_metric.py:107"
        except Exception as exc:
            out["trained_scene_op_only"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and args.footprints:
        out["footprints"] = footprint_legs(dev)
    egs_dist.shutdown()
    sys.stdout.flush()
    print(json.dumps(out), flush=True)                                  # the ONE line of stdout, and the last thing written to it


if __name__ == "__main__":
    main()
