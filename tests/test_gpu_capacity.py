"""GPU: capacity-sized models (densify / prune in place, N as a device word) and the overflow guard of replayed steps.

Reference behaviour being kept: /root/reference/scene/gaussian_model.py:565-586,678-709 (what densify_and_prune produces --
pinned by tests/golden/densify.npz) and /root/reference/trainers/train_static.py:110-138 (every optimizer.step() follows a
complete render)."""
import math

import numpy as np
import pytest
import torch

from tests.test_densify_cpu import load, state_from, case_kwargs, assert_state_equal
from tests.test_gpu_densify import Model, ATTR, DEV

pytestmark = pytest.mark.gpu


class CapModel(Model):
    """The fixture's model padded to `capacity` rows: what capacity.CapacityGaussians looks like to densify.py."""

    def __init__(self, st, capacity, percent_dense=0.01, optimizer="fused"):
        n = st["xyz"].shape[0]
        pad = lambda t: torch.cat([t, torch.full((capacity - n,) + tuple(t.shape[1:]), 7.0 if t.is_floating_point() else 7, dtype=t.dtype)])   # junk, not zeros
        super().__init__({k: pad(v) for k, v in st.items()}, percent_dense, optimizer)
        self.capacity, self.n_active = capacity, n
        self.active_count = torch.tensor([n], dtype=torch.int32, device=DEV)
        self.grown = 0

    def set_active(self, n):
        self.n_active = int(n); self.active_count.fill_(int(n))

    def grow(self, capacity):
        self.grown += 1
        raise AssertionError("the test's capacity must suffice")

    def live_state(self):
        return {k: v[:self.n_active] for k, v in self.state().items()}


@pytest.mark.parametrize("k", range(8))
def test_in_place_densify_and_prune_matches_reference_fixture(k):
    """densify_and_prune on a capacity-sized model: the live prefix afterwards is element for element what the reference's
    GaussianModel.densify_and_prune produced, while every tensor, Parameter object and optimizer-state entry kept its identity
    and address and the rows beyond the live count kept out of it."""
    from egogaussian_amd import densify
    g = load()
    m = CapModel(state_from(g, f"case{k}_in_"), 1500, float(g["percent_dense"]))
    ids = {a: (id(getattr(m, a)), getattr(m, a).data_ptr()) for a in ATTR.values()}
    mom = {a: m.optimizer.state[getattr(m, a)]["exp_avg"].data_ptr() for a in ATTR.values()}
    kw = case_kwargs(g[f"case{k}_args"])
    n0, n1 = densify.densify_and_prune(m, z=torch.tensor(g[f"case{k}_z"], device=DEV), **kw)
    assert n0 == 400 and n1 == g[f"case{k}_out_xyz"].shape[0] == m.n_active == int(m.active_count.item()) and m.grown == 0
    assert_state_equal(m.live_state(), g, f"case{k}_out_")
    for a in ATTR.values():
        p = getattr(m, a)
        assert (id(p), p.data_ptr()) == ids[a] and p.shape[0] == 1500 and m.optimizer.state[p]["exp_avg"].data_ptr() == mom[a]


def test_in_place_prune_and_reset_opacity():
    from egogaussian_amd import densify
    g = load()
    m = CapModel(state_from(g, "in_"), 900)
    p_op = m._opacity
    densify.reset_opacity(m)
    assert m._opacity is p_op and np.allclose(m._opacity.detach()[:400].cpu().numpy(), g["reset_opacity"], rtol=1e-6, atol=1e-7)
    assert float(m.optimizer.state[m._opacity]["exp_avg"].abs().sum()) == 0.0
    st0 = m.live_state()
    mask = torch.rand(900, generator=torch.Generator().manual_seed(3)) < 0.37
    n0, n1 = densify.prune_points(m, mask.to(DEV))
    assert (n0, n1) == (400, int((~mask[:400]).sum())) and m.n_active == n1
    st1 = m.live_state()
    for k_, v in st0.items():
        assert torch.equal(st1[k_], v[~mask[:400]]), k_


def _scene(n, H, W, scale=2.0):
    from egogaussian_amd.scene_synth import make_scene
    sc = make_scene(n, H, W, 0)
    sc["log_scale"] += math.log(scale)
    return sc


def test_dead_rows_are_invisible_and_untouched():
    """Rows beyond active_count -- whatever they hold, NaN included -- produce no instance, no gradient and are not stepped; the
    image and the live rows' gradients are bit-identical to the plain model's."""
    from egogaussian_amd.capacity import CapacityGaussians
    from egogaussian_amd.scene_synth import make_camera, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    H, W, n, cap = 96, 160, 5000, 8192
    sc = _scene(n, H, W)
    cam, bg = make_camera(3, H, W, device=DEV), torch.tensor([0.1, 0.2, 0.3], device=DEV)
    plain, capm = SynthGaussians(sc, device=DEV), CapacityGaussians(sc, cap, device=DEV)
    with torch.no_grad():
        for t in (capm._xyz, capm._scaling, capm._rotation, capm._opacity, capm._features_dc):
            t[n:] = float("nan")
        capm._xyz[n:n + 100] = plain._xyz[:100]; capm._scaling[n:n + 100] = 1.0          # well in view, huge: would dominate the image if live
        capm._rotation[n:n + 100] = plain._rotation[:100]; capm._opacity[n:n + 100] = 5.0; capm._features_dc[n:n + 100] = 1.0
    outs = []
    for pc in (plain, capm):
        o = render(cam, pc, Pipe, bg)
        (o["render"].sum() + 0.5 * o["alpha"].sum()).backward()
        outs.append(o)
    a, b = outs
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"][:n]) and int(b["radii"][n:].abs().sum()) == 0
    assert not bool(b["visibility_filter"][n:].any()) and b["visibility_filter"].shape[0] == cap
    for pa, pb in ((plain._xyz, capm._xyz), (plain._scaling, capm._scaling), (plain._rotation, capm._rotation), (plain._opacity, capm._opacity),
                   (plain._features_dc, capm._features_dc)):
        assert float(pb.grad[n:].abs().sum()) == 0.0 and torch.isfinite(pb.grad).all()
        assert float((pa.grad - pb.grad[:n]).abs().max()) <= 1e-5 * float(pa.grad.abs().max())      # (float atomics: order of accumulation)
    assert float(b["viewspace_points"].grad[n:].abs().sum()) == 0.0
    # the optimizer leaves dead rows alone
    opt = capm.training_setup(capturable=True)
    before = capm._xyz.detach()[n:].clone()
    for p in (capm._xyz, capm._scaling):
        p.grad = torch.ones_like(p)
    opt.step()
    assert torch.equal(torch.nan_to_num(capm._xyz.detach()[n:], nan=-1.0), torch.nan_to_num(before, nan=-1.0))
    assert float((capm._xyz.detach()[:n] - plain._xyz.detach()).abs().min()) > 0           # live rows did move


def test_captured_step_survives_densification_without_recapture():
    """A step captured once keeps replaying correctly while densify_and_prune grows and shrinks the model in place: zero
    re-captures, the replayed image equals an eager render of the current model, the training loss keeps falling."""
    from egogaussian_amd import densify
    from egogaussian_amd.capacity import CapacityGaussians
    from egogaussian_amd.graph import GraphedTrainStep
    from egogaussian_amd.scene_synth import make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    H, W, n, cap = 96, 160, 6000, 30000
    teacher = _scene(n, H, W)
    cams = [make_camera(k * 30, H, W, device=DEV) for k in range(6)]
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=DEV, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
    pc = CapacityGaussians(perturb_student(teacher), cap, device=DEV)
    pc.training_setup(capturable=True)
    step = GraphedTrainStep(pc, pc.optimizer, bg, 0.2, densify_stats=True).capture(cams[0], gts[0], warmup=2, capacity_margin=3.0)
    graph0, ptr0 = step.graph, pc._xyz.data_ptr()
    segments, sizes = [[]], [pc.n_active]
    for it in range(90):
        k = it % len(cams)
        loss = step(cams[k], gts[k])
        if it % 10 == 9:
            segments[-1].append(float(loss.item()))
        if it in (29, 59):
            assert step.ok()
            assert float(pc.denom[:pc.n_active].sum()) > 0 and float(pc.denom[pc.n_active:].sum()) == 0      # statistics: live rows only
            n0, n1 = densify.densify_and_prune(pc, 5e-5, 0.005, 10.0, None)
            assert n1 != n0 and pc.n_active == n1 <= cap
            sizes.append(n1); segments.append([])
    torch.cuda.synchronize()
    assert step.graph is graph0 and step.recaptures == 0 and pc._xyz.data_ptr() == ptr0 and step.ok()
    assert len(set(sizes)) == 3, sizes                                  # the model really changed size twice
    print(f"\n  live Gaussians {sizes} of capacity {cap}; loss every 10 steps per segment: {segments}")
    for seg in segments:                                                # (a split draws new positions: the loss jumps there, then falls again)
        assert len(seg) == 3 and seg[-1] < seg[0], segments
    with torch.no_grad():
        eager = render(cams[0], pc, Pipe, bg)["render"].clone()
    step(cams[0], gts[0])                                               # the replay renders the same model (its update comes after)
    torch.cuda.synchronize()
    assert torch.equal(step.image, eager), "the captured step does not render the densified model"


def test_overflowed_frame_leaves_parameters_moments_and_statistics_bit_unchanged():
    """A replayed frame that needs more instances than the captured capacity is clipped; its Adam launch and its fused
    densification statistics must do nothing (train_static.py:110-138: every optimizer.step() follows a complete render).
    check() then re-captures with room for it and training goes on."""
    from egogaussian_amd.graph import GraphedTrainStep
    from egogaussian_amd.scene_synth import make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd import _C
    H, W, n = 96, 160, 5000
    teacher = _scene(n, H, W, 1.5)
    cam, bg = make_camera(0, H, W, device=DEV), torch.zeros(3, device=DEV)
    with torch.no_grad():
        gt = render(cam, SynthGaussians(teacher, device=DEV, requires_grad=False), Pipe, bg)["render"].clone()
    pc = SynthGaussians(perturb_student(teacher), device=DEV)
    pc.training_setup(capturable=True)
    with torch.no_grad():
        r_now = render(cam, pc, Pipe, bg) and _C.stats["num_rendered"]
    step = GraphedTrainStep(pc, pc.optimizer, bg, 0.2, densify_stats=True).capture(cam, gt, warmup=2, capacity=int(r_now * 1.1))
    for _ in range(3):
        step(cam, gt)
    assert step.ok() and not step.last_frame_overflowed() and 0 < step.last_instance_count() <= step.capacity
    with torch.no_grad():
        pc._scaling += math.log(3.0)                                   # same tensors, several times the footprint: the next frame cannot fit
    torch.cuda.synchronize()
    params = [pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation]
    snap = lambda: [p.detach().clone() for p in params] + [pc.optimizer.state[p][k].clone() for p in params for k in ("exp_avg", "exp_avg_sq", "step")] + \
        [pc.xyz_gradient_accum.clone(), pc.denom.clone(), pc.max_radii2D.clone()]
    before = snap()
    step(cam, gt)
    torch.cuda.synchronize()
    assert step.last_frame_overflowed() and step.last_instance_count() > step.capacity and not step.ok()
    after = snap()
    for a, b in zip(before, after):
        assert torch.equal(a, b), "an overflowed frame changed training state"
    cap0 = step.capacity
    assert step.check() is False and step.recaptures == 1 and step.capacity > cap0 and step.ok()
    step(cam, gt)
    torch.cuda.synchronize()
    assert not step.last_frame_overflowed() and step.ok()
    assert not torch.equal(pc._xyz.detach(), before[0])                # and training goes on


def test_capacity_grows_when_densification_outruns_it():
    """A model whose capacity is too small for the densified count grows (new arrays, optimizer state carried over); the plan is then
    applied in place, and a captured step re-captured on the new tensors goes on training."""
    from egogaussian_amd import densify
    from egogaussian_amd.capacity import CapacityGaussians
    from egogaussian_amd.graph import GraphedTrainStep
    from egogaussian_amd.scene_synth import make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    H, W, n = 96, 160, 6000
    teacher = _scene(n, H, W)
    cam, bg = make_camera(0, H, W, device=DEV), torch.zeros(3, device=DEV)
    with torch.no_grad():
        gt = render(cam, SynthGaussians(teacher, device=DEV, requires_grad=False), Pipe, bg)["render"].clone()
    pc = CapacityGaussians(perturb_student(teacher), n + 500, device=DEV)          # room for 500 more only
    pc.training_setup(capturable=True)
    step = GraphedTrainStep(pc, pc.optimizer, bg, 0.2, densify_stats=True).capture(cam, gt, warmup=2, capacity_margin=3.0)
    for _ in range(20):
        step(cam, gt)
    torch.cuda.synchronize()
    ptr, m_before = pc._xyz.data_ptr(), pc.optimizer.state[pc._xyz]["exp_avg"][:n].clone()
    n0, n1 = densify.densify_and_prune(pc, 5e-5, 0.005, 10.0, None)
    assert n1 > n + 500 and pc.capacity >= n1 and pc.n_active == n1 and int(pc.active_count.item()) == n1
    assert pc._xyz.data_ptr() != ptr and pc._xyz.shape[0] == pc.capacity              # reallocated
    assert pc.optimizer.param_groups[0]["params"][0] is pc._xyz and "exp_avg" in pc.optimizer.state[pc._xyz]
    assert pc.optimizer.active_rows[1] == pc.capacity
    assert pc.optimizer.state[pc._xyz]["exp_avg"].shape[0] == pc.capacity and float(m_before.abs().sum()) > 0
    step.recapture(warmup=1)
    l0 = float(step(cam, gt))
    for _ in range(15):
        step(cam, gt)
    assert float(step(cam, gt)) < l0 and step.ok()


@pytest.mark.parametrize("path", ["rasterizer", "producer"])
@pytest.mark.parametrize("which_object", [1, None])
def test_capacity_model_with_rot_cov_matches_plain_model(which_object, path):
    """The fine_all call shape on a capacity-sized model: the object selection -- including the reference's [N,1]-index quirk, whose
    row-0 gradient multiplier is (number of selected Gaussians [+ 1]) or N + 1 -- counts LIVE rows only, whatever tags the dead rows
    still hold (an in-place prune leaves the vacated rows behind the live count).  Image bit-identical to the plain model's, the
    gradients of Gaussian 0 (where the multiplier lands) and of every other row equal."""
    from egogaussian_amd.capacity import CapacityGaussians
    from egogaussian_amd.scene_synth import make_camera, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    H, W, n, cap = 96, 160, 4000, 6000
    sc = _scene(n, H, W)
    cam, bg = make_camera(2, H, W, device=DEV), torch.tensor([0.05, 0.1, 0.15], device=DEV)
    tags = (torch.rand(n, 1, generator=torch.Generator().manual_seed(11)) < 0.3).float()
    th = 0.3
    accum_R = torch.tensor([[math.cos(th), -math.sin(th), 0.0], [math.sin(th), math.cos(th), 0.0], [0.0, 0.0, 1.0]], device=DEV)
    plain, capm = SynthGaussians(sc, device=DEV), CapacityGaussians(sc, cap, device=DEV)
    plain._is_object = tags.to(DEV)
    capm._is_object[:n] = tags.to(DEV)
    capm._is_object[n:] = 1.0                                      # stale tags behind the live count
    outs = []
    for pc in (plain, capm):
        pc.rotate_in_rasterizer = path == "rasterizer"
        o = render(cam, pc, Pipe, bg, rot_cov=True, accum_R=accum_R, which_object=which_object, during_training=False)
        (o["render"].sum() + 0.5 * o["alpha"].sum()).backward()
        outs.append(o)
    a, b = outs
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"][:n])
    for pa, pb in ((plain._scaling, capm._scaling), (plain._rotation, capm._rotation), (plain._xyz, capm._xyz), (plain._opacity, capm._opacity)):
        assert float(pb.grad[n:].abs().sum()) == 0.0
        scale = float(pa.grad.abs().max())
        assert float((pa.grad - pb.grad[:n]).abs().max()) <= 2e-5 * scale, (float((pa.grad - pb.grad[:n]).abs().max()), scale)
    # the multiplier really is in play on row 0 (otherwise the comparison above would prove nothing)
    assert float(plain._scaling.grad[0].abs().max()) > 0


def test_replay_after_grow_without_recapture_is_refused():
    """CapacityGaussians.grow() frees the arrays a captured step points at; replaying that step must raise, not touch freed memory."""
    from egogaussian_amd.capacity import CapacityGaussians
    from egogaussian_amd.graph import GraphedTrainStep
    from egogaussian_amd.scene_synth import make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    H, W, n = 64, 96, 2000
    teacher = _scene(n, H, W)
    cam, bg = make_camera(0, H, W, device=DEV), torch.zeros(3, device=DEV)
    with torch.no_grad():
        gt = render(cam, SynthGaussians(teacher, device=DEV, requires_grad=False), Pipe, bg)["render"].clone()
    pc = CapacityGaussians(perturb_student(teacher), n + 100, device=DEV)
    pc.training_setup(capturable=True)
    step = GraphedTrainStep(pc, pc.optimizer, bg, 0.2).capture(cam, gt, warmup=2, capacity_margin=3.0)
    step(cam, gt)
    pc.grow(2 * n)
    with pytest.raises(RuntimeError, match="recapture"):
        step(cam, gt)
    step.recapture(warmup=1)
    step(cam, gt)
    torch.cuda.synchronize()
    assert step.ok()
    # an eager optimizer step after replays is never voided by the guard word the last replay left behind
    step.guard.overflow[0] = 1
    before = pc._xyz.detach()[:n].clone()
    pc.optimizer.zero_grad(set_to_none=True)
    out = render(cam, pc, Pipe, bg)
    out["render"].sum().backward()
    pc.optimizer.step()
    torch.cuda.synchronize()
    assert not torch.equal(pc._xyz.detach()[:n], before)


def test_second_gradient_path_into_a_fused_leaf_is_refused():
    """convert_SHs_python computes the colours from the positions in Python: the positions then receive gradient through the colour
    path as well, so the rasterizer backward may not take their Adam step (optim.FusedAdam.make_sink leaves them to step())."""
    from egogaussian_amd.scene_synth import make_camera, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.optim import FusedAdam
    H, W, n = 64, 96, 2000
    pc = SynthGaussians(_scene(n, H, W), device=DEV)
    cam, bg = make_camera(0, H, W, device=DEV), torch.zeros(3, device=DEV)
    opt = FusedAdam([{"params": [pc._xyz], "lr": 1e-3}, {"params": [pc._features_dc], "lr": 1e-3}, {"params": [pc._opacity], "lr": 1e-3},
                     {"params": [pc._scaling], "lr": 1e-3}, {"params": [pc._rotation], "lr": 1e-3}], lr=0.0, eps=1e-15, capturable=True)

    class PyPipe(Pipe):
        convert_SHs_python = True
    out = render(cam, pc, PyPipe, bg, optimizer=opt)
    out["render"].sum().backward()
    assert pc._xyz.grad is not None                                 # left to step(): the colour path's share is in it
    assert pc._opacity.grad is None and pc._scaling.grad is None     # these were stepped inside the backward
    x0 = pc._xyz.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(pc._xyz.detach(), x0)
