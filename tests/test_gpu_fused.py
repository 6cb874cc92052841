"""GPU: the fused HIP ops of the 'next' rows (f-1 covariance producer, f-3 image loss) against their PyTorch versions,
which tests/test_golden_host.py pins to the reference's own outputs; plus the reference fixtures directly."""
import os

import numpy as np
import pytest
import torch

from tests.common import make_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _close(a, b, rtol=2e-5, atol=0.0):
    a = a.detach().cpu().double().numpy(); b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return np.abs(a - b).max() <= atol + rtol * max(np.abs(b).max(), 1e-30)


def test_fused_covariance_matches_torch_and_reference_fixture():
    from egogaussian_amd import fused, covariance as ref
    g = np.load(os.path.join(GOLD, "covariance.npz"))
    T = lambda a: torch.tensor(np.asarray(a), device=DEV)
    for variant in ("plain", "mod2", "rotated", "rotated_all"):
        ls, q = T(g["log_scale"]).requires_grad_(True), T(g["quat"]).requires_grad_(True)
        ls2, q2 = T(g["log_scale"]).requires_grad_(True), T(g["quat"]).requires_grad_(True)
        w, R, io = T(g["wcov"]), T(g["accum_R"]), T(g["is_object"])
        if variant == "plain":
            a, b, gold = fused.covariance_from_scaling_rotation(torch.exp(ls), 1.0, q), ref.covariance_from_scaling_rotation(torch.exp(ls2), 1.0, q2), g["cov"]
        elif variant == "mod2":
            a, b, gold = fused.covariance_from_scaling_rotation(torch.exp(ls), 2.0, q), ref.covariance_from_scaling_rotation(torch.exp(ls2), 2.0, q2), g["cov_mod2"]
        elif variant == "rotated":
            a = fused.rotated_covariance_from_scaling_rotation(torch.exp(ls), 1.0, q, R, io, 1)
            b, gold = ref.rotated_covariance_from_scaling_rotation(torch.exp(ls2), 1.0, q2, R, io, 1), g["rcov"]
        else:
            a = fused.rotated_covariance_from_scaling_rotation(torch.exp(ls), 1.0, q, R, io, None)
            b, gold = ref.rotated_covariance_from_scaling_rotation(torch.exp(ls2), 1.0, q2, R, io, None), g["rcov_all"]
        assert _close(a, b) and _close(a, gold), variant
        (a * w).sum().backward(); (b * w).sum().backward()
        assert _close(ls.grad, ls2.grad, 1e-4) and _close(q.grad, q2.grad, 1e-4), variant
        if variant == "plain":
            assert _close(ls.grad, g["g_scaling"], 1e-4) and _close(q.grad, g["g_rotation"], 1e-4)
        if variant == "rotated":
            assert _close(ls.grad, g["rg_scaling"], 1e-4) and _close(q.grad, g["rg_rotation"], 1e-4)


def test_fused_covariance_from_raw_scaling():
    """exp() folded into the kernels: same covariance and same gradient w.r.t. the RAW scaling as torch.exp + the op."""
    from egogaussian_amd import fused, covariance as ref
    gen = torch.Generator().manual_seed(5)
    N = 7000
    raw = (torch.randn(N, 3, generator=gen) * 0.7 - 3.0).to(DEV).requires_grad_(True)
    raw2 = raw.detach().clone().requires_grad_(True)
    q = torch.randn(N, 4, generator=gen).to(DEV).requires_grad_(True)
    q2 = q.detach().clone().requires_grad_(True)
    w = torch.randn(N, 6, generator=gen).to(DEV)
    a = fused.covariance_from_log_scaling(raw, 1.3, q)
    b = ref.covariance_from_scaling_rotation(torch.exp(raw2), 1.3, q2)
    assert _close(a, b)
    (a * w).sum().backward(); (b * w).sum().backward()
    assert _close(raw.grad, raw2.grad, 1e-4) and _close(q.grad, q2.grad, 1e-4)
    io = (torch.rand(N, 1, generator=gen) < 0.4).float().to(DEV)
    A = torch.linalg.qr(torch.randn(3, 3, generator=gen))[0].to(DEV)
    c = fused.rotated_covariance_from_scaling_rotation(raw.detach(), 1.0, q.detach(), A, io, 1, scaling_is_log=True)
    d = ref.rotated_covariance_from_scaling_rotation(torch.exp(raw.detach()), 1.0, q.detach(), A, io, 1)
    assert _close(c, d)


def test_fused_covariance_and_opacity_in_one_launch():
    from egogaussian_amd import fused, covariance as ref
    gen = torch.Generator().manual_seed(9)
    N = 6000
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    raw, q, o = (mk(N, 3) * 0.7 - 3.0).requires_grad_(True), mk(N, 4).requires_grad_(True), (mk(N, 1) * 2).requires_grad_(True)
    raw2, q2, o2 = [t.detach().clone().requires_grad_(True) for t in (raw, q, o)]
    w, wo = mk(N, 6), mk(N, 1)
    cov, op = fused.covariance_and_opacity(raw, 1.0, q, o)
    cov2, op2 = ref.covariance_from_scaling_rotation(torch.exp(raw2), 1.0, q2), torch.sigmoid(o2)
    assert _close(cov, cov2) and _close(op, op2, 1e-6) and op.shape == (N, 1)
    ((cov * w).sum() + (op * wo).sum()).backward(); ((cov2 * w).sum() + (op2 * wo).sum()).backward()
    assert _close(raw.grad, raw2.grad, 1e-4) and _close(q.grad, q2.grad, 1e-4) and _close(o.grad, o2.grad, 1e-5)
    # only the opacity is used downstream
    o.grad = None
    fused.covariance_and_opacity(raw, 1.0, q, o)[1].sum().backward()
    assert _close(o.grad, (op2 * (1 - op2)).detach(), 1e-5)


def test_fused_covariance_trainable_rotation_gradient():
    from egogaussian_amd import fused, covariance as ref
    gen = torch.Generator().manual_seed(3)
    N = 5000
    s = (torch.rand(N, 3, generator=gen) * 0.1 + 0.01).to(DEV)
    q = torch.randn(N, 4, generator=gen).to(DEV)
    io = (torch.rand(N, 1, generator=gen) < 0.4).float().to(DEV)
    w = torch.randn(N, 6, generator=gen).to(DEV)
    A = torch.linalg.qr(torch.randn(3, 3, generator=gen))[0].to(DEV)
    Rt = torch.linalg.qr(torch.randn(3, 3, generator=gen))[0].to(DEV).requires_grad_(True)
    Rt2 = Rt.detach().clone().requires_grad_(True)
    a = fused.rotated_covariance_from_scaling_rotation(s, 1.0, q, A, io, 1, rot_matrix=Rt)
    b = ref.rotated_covariance_from_scaling_rotation(s, 1.0, q, A, io, 1, rot_L=lambda L: torch.matmul(Rt2, L))
    assert _close(a, b)
    (a * w).sum().backward(); (b * w).sum().backward()
    assert _close(Rt.grad, Rt2.grad, 2e-4)


@pytest.mark.parametrize("C,H,W", [(3, 40, 56), (3, 37, 50), (1, 16, 16), (3, 540, 960)])
def test_fused_loss_matches_torch(C, H, W):
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.losses import training_loss
    gen = torch.Generator().manual_seed(H * W)
    a = torch.rand(C, H, W, generator=gen).to(DEV).requires_grad_(True)
    b = (a.detach() + 0.1 * torch.randn(C, H, W, generator=gen).to(DEV)).clamp(0, 1)
    a2 = a.detach().clone().requires_grad_(True)
    gate = (torch.rand(H, W, generator=gen) > 0.3).float().to(DEV)
    l1 = l1_ssim_loss(a, b, 0.2, grad_gate=gate)
    l2 = training_loss(a2, b, 0.2)
    assert abs(l1.item() - l2.item()) < 2e-6 * max(1.0, abs(l2.item()))
    (3.0 * l1).backward()
    a2.register_hook(lambda g: g)        # noqa
    (3.0 * l2).backward()
    assert _close(a.grad, a2.grad * gate[None], 2e-4)


@pytest.mark.parametrize("kind", ["black", "constant", "tiny-variance", "equal", "black-with-islands"])
def test_fused_loss_on_flat_and_small_variance_images(kind):
    """The SSIM map's 1 / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)) is a v_rcp_f32 + one Newton step here (loss.hip), an IEEE division in the
    reference (/root/reference/utils/loss_utils.py:92-102).  Where it matters most is where the denominator is smallest: flat regions
    (zero variance: the denominator is C1 C2-sized) -- and with a black background most of a rendered frame is exactly that.
    Value and gradient against the torch formulation evaluated in FLOAT64, at the north star's 1e-4 (max-norm relative)."""
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.losses import training_loss
    C, H, W = 3, 96, 160
    gen = torch.Generator().manual_seed(7)
    if kind == "black":
        a, b = torch.zeros(C, H, W), torch.zeros(C, H, W)
    elif kind == "constant":
        a, b = torch.full((C, H, W), 0.5), torch.full((C, H, W), 0.25)
    elif kind == "tiny-variance":
        a = 0.3 + 1e-4 * torch.randn(C, H, W, generator=gen); b = 0.3 + 1e-4 * torch.randn(C, H, W, generator=gen)
    elif kind == "equal":
        a = torch.rand(C, H, W, generator=gen); b = a.clone()
    else:
        a = torch.zeros(C, H, W); a[:, 20:40, 30:70] = torch.rand(C, 20, 40, generator=gen)
        b = torch.zeros(C, H, W); b[:, 22:41, 28:66] = torch.rand(C, 19, 38, generator=gen)
    x = a.to(DEV).requires_grad_(True)
    l = l1_ssim_loss(x, b.to(DEV), 0.2)
    l.backward()
    x64 = a.double().requires_grad_(True)
    l64 = training_loss(x64, b.double(), 0.2)
    l64.backward()
    assert abs(l.item() - l64.item()) <= 1e-5 * max(1.0, abs(l64.item())), (kind, l.item(), l64.item())
    g, g64 = x.grad.cpu().double(), x64.grad
    # the unit: the largest true entry, but never less than one pixel's share of the L1 term (0.8 / n) -- at the optimum ("equal", "black") the
    # true gradient is zero and both sides hold rounding noise (1e-20 in float64, 1e-11 in float32), which says nothing relative to itself
    # (sign(x - y) of the L1 term at x == y: torch gives 0 and so does the kernel; nothing to set aside)
    scale = max(float(g64.abs().max()), 0.8 / (C * H * W))
    assert float((g - g64).abs().max()) <= 1e-4 * scale, (kind, float((g - g64).abs().max()) / scale)


def test_fused_loss_matches_reference_fixture():
    from egogaussian_amd.fused import l1_ssim_loss
    g = np.load(os.path.join(GOLD, "losses.npz"))
    a = torch.tensor(g["a"], device=DEV).requires_grad_(True)
    b = torch.tensor(g["b"], device=DEV)
    ssim_only = l1_ssim_loss(a, b, 1.0)                 # lambda = 1: loss = 1 - SSIM
    assert abs((1.0 - ssim_only.item()) - float(g["ssim"])) < 3e-6
    ssim_only.backward()
    assert _close(-a.grad, g["ssim_grad_a"], 1e-3)
    l1_only = l1_ssim_loss(a.detach(), b, 0.0)
    assert abs(l1_only.item() - float(g["l1"])) < 1e-6


def test_fused_adam_matches_torch_adam():
    from egogaussian_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(0)
    shapes = [(5000, 3), (5000, 1, 3), (5000, 1), (5000, 3), (5000, 4), (7,), (1025,)]
    lrs = [1.6e-4, 2.5e-3, 0.05, 5e-3, 1e-3, 0.1, 0.01]
    pa = [torch.randn(s, generator=gen).to(DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedAdam([{"params": [p], "lr": lr, "name": f"g{i}"} for i, (p, lr) in enumerate(zip(pa, lrs))], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs)], lr=0.0, eps=1e-15)
    for it in range(12):
        for p, q in zip(pa, pb):
            g = torch.randn(p.shape, generator=gen).to(DEV) * (10.0 ** (it % 4 - 2))
            p.grad = g.clone(); q.grad = g.clone()
        oa.step(); ob.step()
    for p, q in zip(pa, pb):
        assert _close(p, q, 2e-6, 1e-7)
    st = oa.state[pa[0]]
    assert set(st) >= {"step", "exp_avg", "exp_avg_sq"} and _close(st["exp_avg"], ob.state[pb[0]]["exp_avg"], 1e-5, 1e-9)
    # state surgery as the reference's densification does it: replace the tensors, keep stepping
    keep = torch.arange(0, 5000, 2, device=DEV)
    for opt, params in ((oa, pa), (ob, pb)):
        g0 = opt.param_groups[0]
        old = g0["params"][0]
        stored = opt.state.pop(old)
        stored["exp_avg"] = stored["exp_avg"][keep].contiguous(); stored["exp_avg_sq"] = stored["exp_avg_sq"][keep].contiguous()
        new = torch.nn.Parameter(old.detach()[keep].contiguous())
        g0["params"][0] = new
        opt.state[new] = stored
        params[0] = new
    for p, q in zip(pa, pb):
        g = torch.randn(p.shape, generator=gen).to(DEV)
        p.grad = g.clone(); q.grad = g.clone()
    oa.step(); ob.step()
    assert _close(pa[0], pb[0], 2e-6, 1e-7)


@pytest.mark.parametrize("N", [5, 1000, 20000])
def test_knn_mean_dist2_matches_kdtree(N):
    """distCUDA2 replacement against scipy's exact k-d tree (k = 4 including the point itself)."""
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(N)
    pts = rng.normal(size=(N, 3)).astype(np.float32)
    pts[N // 2] = pts[0]                                     # an exact duplicate: neighbour at distance 0
    got = distCUDA2(torch.tensor(pts, device=DEV)).cpu().numpy()
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    want = (d[:, 1:] ** 2).mean(1)
    assert np.abs(got - want).max() <= 1e-5 * want.max() + 1e-7
    assert got[0] < want.mean() and np.isfinite(got).all()


@pytest.mark.parametrize("shape", ["normal", "uniform", "plane", "clusters", "line+outlier"])
def test_knn_grid_search_is_bit_identical_to_all_pairs(shape):
    """The uniform-grid search (what distCUDA2 uses from 32k points up) against the all-pairs kernel: same bits, for even and very
    uneven densities, degenerate extents, duplicates and non-finite points; and against scipy's k-d tree at 300k points."""
    import time
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(7)
    N = 60000
    if shape == "normal":
        pts = rng.normal(size=(N, 3))
    elif shape == "uniform":
        pts = rng.uniform(-3, 5, size=(N, 3))
    elif shape == "plane":
        pts = rng.uniform(-1, 1, size=(N, 3)); pts[:, 2] = 0.25                     # zero extent along z
    elif shape == "clusters":
        pts = rng.normal(size=(N, 3)) * 0.01 + rng.integers(0, 5, size=(N, 1)) * 10.0  # five dense blobs far apart
    else:
        pts = np.zeros((N, 3)); pts[:, 0] = np.linspace(0, 1, N); pts[-1] = (1e4, 1e4, 1e4)
    pts = pts.astype(np.float32)
    pts[100] = pts[200]                                                             # an exact duplicate
    pts[300] = (np.nan, 0, 0); pts[301] = (np.inf, 1, 2)                            # nobody's neighbours; their own answer is inf
    t = torch.tensor(pts, device=DEV)
    a, b = distCUDA2(t, method="pairs"), distCUDA2(t, method="grid")
    assert torch.equal(a, b), f"{int((a != b).sum())} of {N} differ"
    assert torch.isinf(b[300]) and torch.isinf(b[301]) and torch.isfinite(b[:300]).all()
    if shape == "uniform":
        from scipy.spatial import cKDTree
        big = rng.uniform(-3, 5, size=(300000, 3)).astype(np.float32)
        tb = torch.tensor(big, device=DEV)
        distCUDA2(tb); torch.cuda.synchronize()
        t0 = time.perf_counter(); got = distCUDA2(tb); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        d, _ = cKDTree(big.astype(np.float64)).query(big.astype(np.float64), k=4, workers=-1)
        want = (d[:, 1:] ** 2).mean(1)
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-5 * want.max() + 1e-7
        print(f"\n  grid search, 300k points: {dt * 1e3:.2f} ms")


def test_knn_fewer_than_four_points():
    from simple_knn._C import distCUDA2
    out = distCUDA2(torch.tensor([[0.0, 0, 0], [1.0, 0, 0]], device=DEV))
    assert torch.isinf(out).all()                            # fewer than 3 neighbours: FLT_MAX slots overflow, as upstream
    assert distCUDA2(torch.zeros((0, 3), device=DEV)).numel() == 0


def test_graphed_train_step_matches_eager():
    """A hipGraph-captured training step (covariance, render, loss, backward, Adam) replayed K times leaves the parameters
    where K eager steps leave them -- including a learning rate that is edited on the host between replays."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.graph import GraphedTrainStep
    N, H, W, K = 20000, 96, 160, 6
    teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
    student = perturb_student(teacher)
    cams = [make_camera(k, H, W, device=DEV) for k in (0, 30, 60, 90)]
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=DEV, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]

    def groups(pc):
        return [{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}]
    WARM = 3
    # eager reference: WARM steps on frame 0 (what capture() runs eagerly; the capture pass itself only records), then K steps
    pa = SynthGaussians(student, device=DEV)
    oa = FusedAdam(groups(pa), lr=0.0, eps=1e-15)
    seq = [0] * WARM + [k % 4 for k in range(1, K + 1)]
    xyz_lr = lambda it: 1.6e-4 * (0.5 ** it)                         # a schedule edited on the host every iteration, as the reference's
    for it, k in enumerate(seq):                                     # update_learning_rate does (scene/gaussian_model.py:200-206)
        if it >= WARM:
            oa.param_groups[0]["lr"] = xyz_lr(it - WARM + 1)
        out = render(cams[k], pa, Pipe, bg)
        l1_ssim_loss(out["render"], gts[k], 0.2).backward()
        oa.step(); oa.zero_grad(set_to_none=True)
    pb = SynthGaussians(student, device=DEV)
    ob = FusedAdam(groups(pb), lr=0.0, eps=1e-15, capturable=True)
    step = GraphedTrainStep(pb, ob, bg).capture(cams[0], gts[0], warmup=WARM)
    losses = []
    for k in range(1, K + 1):
        ob.param_groups[0]["lr"] = xyz_lr(k)                         # picked up by the replay (device scalar refreshed in __call__)
        losses.append(step(cams[k % 4], gts[k % 4]).clone())
    torch.cuda.synchronize()
    assert step.ok() and 0 < step.last_instance_count() <= step.capacity
    assert all(torch.isfinite(l) for l in losses) and float(losses[-1]) < float(losses[0]) * 1.5
    # The two runs differ only in the order of the floating-point gradient accumulation (atomics).  Adam with the
    # reference's eps = 1e-15 moves a parameter by ~lr whatever the size of its gradient, so the handful of parameters whose
    # gradient is at rounding level can land a fraction of lr apart; everything else must agree closely.
    for a, b in zip(pa.parameters(), pb.parameters()):
        if a.numel():
            diff = (a.detach() - b.detach()).abs()
            scale = float(a.detach().abs().max())
            assert float((diff > 2e-5 * scale + 1e-6).float().mean()) < 2e-3, "graph-replayed parameters drifted from the eager ones"
            assert float(diff.max()) < 0.02, "graph-replayed parameters drifted from the eager ones"


@pytest.mark.parametrize("gated", [False, True], ids=["plain", "hand-mask gate"])
def test_loss_gradient_inside_the_blend_matches_the_loss_backward_launch(gated):
    """ABI 5 (egs_backward_lossgrad): with raster_lossgrad=True the image loss has NO backward launch -- the rasterizer's backward blend
    computes dL/dimage for its tile from the maps the loss forward left (bit-identical per pixel: checked with the -DEGS_LG_CHECK build,
    profiles/), the loss forward carries the blend's preparation, and the deferred loss value is assembled by one wave of the blend.  Every
    parameter gradient, the loss value and the running sum must be what the separate launch gives (gradients up to the order of the
    accumulator atomics)."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    N, H, W = 20000, 135, 250                                           # (a ragged image: partial tiles on both edges)
    teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
    cam = make_camera(7, H, W, device=DEV)
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    with torch.no_grad():
        gt = render(cam, SynthGaussians(teacher, device=DEV, requires_grad=False), Pipe, bg)["render"].clone()
    gate = (torch.rand((H, W), generator=torch.Generator().manual_seed(3)) < 0.8).float().to(DEV) if gated else None
    res = []
    for mode in (False, True, True):
        pc = SynthGaussians(perturb_student(teacher), device=DEV)
        run = torch.full((1,), 0.25, device=DEV)
        out = render(cam, pc, Pipe, bg)
        loss = l1_ssim_loss(out["render"], gt, 0.2, grad_gate=gate, running_sum=run, defer_value=True, raster_prologue=True, raster_lossgrad=mode)
        loss.backward()
        torch.cuda.synchronize()
        res.append(([p.grad.clone() for p in pc.parameters() if p.grad is not None], float(loss.detach()), float(run)))
    ref = res[0]
    assert len(ref[0]) >= 5 and math.isfinite(ref[1]) and ref[1] > 0
    for got in res[1:]:
        assert got[1] == ref[1] and got[2] == ref[2] and abs(ref[2] - 0.25 - ref[1]) < 1e-6, "deferred loss value / running sum"
        assert len(got[0]) == len(ref[0])
        for a, b in zip(got[0], ref[0]):
            scale = float(b.abs().max()) + 1e-30
            assert float((a - b).abs().max()) <= 2e-5 * scale, "gradient with the loss gradient computed inside the blend"


def test_loss_gradient_in_the_blend_refuses_a_second_gradient_into_the_image():
    """with raster_lossgrad=True the tensor autograd passes from the loss to the rasterizer is uninitialised and unread.  A second
    loss term on the image, or a hook that rewrites the gradient (the reference's hand-mask hook), would be dropped silently -- the
    rasterizer's backward must refuse instead; the plain single-consumer case still runs."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    N, H, W = 3000, 48, 80
    teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
    cam = make_camera(7, H, W, device=DEV)
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    gt = torch.rand((3, H, W), device=DEV)

    def step(extra):
        pc = SynthGaussians(perturb_student(teacher), device=DEV)
        out = render(cam, pc, Pipe, bg)
        img = out["render"]
        loss = l1_ssim_loss(img, gt, 0.2, defer_value=True, raster_prologue=True, raster_lossgrad=True)
        if extra == "second term":
            loss = loss + 0.1 * img.mean()
        elif extra == "hook":
            img.register_hook(lambda g: g * 0.5)
        loss.backward()
        torch.cuda.synchronize()
        return pc

    pc = step(None)
    assert all(torch.isfinite(p.grad).all() for p in pc.parameters() if p.grad is not None)
    for extra in ("second term", "hook"):
        with pytest.raises(RuntimeError, match="another gradient contribution"):
            step(extra)


def test_loss_gradient_in_the_blend_with_a_colours_only_backward_takes_the_full_path():
    """the label-call shape (colors_precomp the only differentiable input) under raster_lossgrad=True must not reach the colours-only
    fast path, which would read the uninitialised gradient tensor: the gradient has to equal the separate loss-backward launch's."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera
    from egogaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from egogaussian_amd.fused import l1_ssim_loss
    N, H, W = 3000, 48, 80
    d = make_inputs(N, H, W, 0, 0, "col_sr", scale_mul=2.0)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=d["tanfovx"], tanfovy=d["tanfovy"], bg=d["bg"].to(DEV), scale_modifier=1.0,
                                       viewmatrix=d["viewmatrix"].to(DEV), projmatrix=d["projmatrix"].to(DEV), sh_degree=0, campos=d["campos"].to(DEV),
                                       prefiltered=False, debug=False)
    gt = torch.rand((3, H, W), generator=torch.Generator().manual_seed(2)).to(DEV)
    res = []
    for mode in (False, True):
        cols = d["colors_precomp"].to(DEV).requires_grad_(True)
        img, _, _, _ = GaussianRasterizer(rs)(means3D=d["means3D"].to(DEV), means2D=torch.zeros((N, 3), device=DEV), opacities=d["opacities"].to(DEV),
                                              colors_precomp=cols, scales=d["scales"].to(DEV), rotations=d["rotations"].to(DEV))
        l1_ssim_loss(img, gt, 0.2, defer_value=True, raster_prologue=True, raster_lossgrad=mode).backward()
        torch.cuda.synchronize()
        res.append(cols.grad.clone())
    scale = float(res[0].abs().max())
    assert scale > 0 and torch.isfinite(res[1]).all()
    assert float((res[1] - res[0]).abs().max()) <= 2e-5 * scale


def test_loss_gradient_in_the_blend_on_a_frame_with_no_instance():
    """A camera that sees nothing: R == 0, no blend launch -- the deferred loss value must still be assembled (egs_launch_loss_finish) and
    every gradient is zero, as with the separate loss-backward launch."""
    from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    N, H, W = 2000, 64, 96
    scene = make_scene(N, H, W, 0)
    cam = make_camera(0, H, W, device=DEV)
    bg = torch.tensor([0.3, 0.1, 0.2], device=DEV)
    gt = torch.rand((3, H, W), generator=torch.Generator().manual_seed(1)).to(DEV)
    vals = []
    for mode in (False, True):
        pc = SynthGaussians(scene, device=DEV)
        with torch.no_grad():
            pc._xyz += 1.0e4                                             # far outside every frustum
        out = render(cam, pc, Pipe, bg)
        assert int(out["radii"].max()) == 0
        run = torch.zeros(1, device=DEV)
        loss = l1_ssim_loss(out["render"], gt, 0.2, running_sum=run, defer_value=True, raster_prologue=True, raster_lossgrad=mode)
        loss.backward()
        torch.cuda.synchronize()
        assert all(float(p.grad.abs().max()) == 0.0 for p in pc.parameters() if p.grad is not None)
        vals.append((float(loss.detach()), float(run)))
    assert vals[0] == vals[1] and vals[0][0] > 0 and vals[0][0] == vals[0][1]


def test_graphed_step_with_and_without_loss_gradient_in_the_blend():
    """GraphedTrainStep(loss_grad_in_blend=False) keeps the loss-backward launch; both captured steps train alike."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.graph import GraphedTrainStep
    N, H, W, K = 20000, 96, 160, 8
    teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
    student = perturb_student(teacher)
    cams = [make_camera(k, H, W, device=DEV) for k in (0, 30, 60, 90)]
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=DEV, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
    groups = lambda pc: [{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                         {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}]
    runs = []
    for in_blend in (False, True):
        pc = SynthGaussians(student, device=DEV)
        opt = FusedAdam(groups(pc), lr=0.0, eps=1e-15, capturable=True)
        step = GraphedTrainStep(pc, opt, bg, loss_grad_in_blend=in_blend).capture(cams[0], gts[0], warmup=2)
        losses = [float(step(cams[k % 4], gts[k % 4]).clone()) for k in range(1, K + 1)]
        torch.cuda.synchronize()
        assert step.ok()
        runs.append((losses, [p.detach().clone() for p in pc.parameters()], float(opt.state[pc._xyz]["step"])))
    (la, pa, sa), (lb, pb, sb) = runs
    assert sa == sb == 2 + K
    assert all(abs(x - y) <= 2e-4 * abs(x) + 1e-7 for x, y in zip(la, lb)), (la, lb)
    for a, b in zip(pa, pb):
        if a.numel():
            diff = (a - b).abs(); scale = float(a.abs().max())
            assert float((diff > 2e-5 * scale + 1e-6).float().mean()) < 2e-3 and float(diff.max()) < 0.02


def test_graphed_fine_all_step_matches_eager():
    """The `fine_all` call shape captured into a graph -- render(rot_cov=True, accum_R=<static, refreshed per call>, which_object=1),
    hand-mask gate as a static input -- replayed on changing (camera, image, accum_R, gate) leaves the parameters where the same steps
    launched eagerly leave them (/root/reference/trainers/fine_all.py:88-101)."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.graph import GraphedTrainStep, pack_frame
    N, H, W, K = 15000, 96, 160, 5
    teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
    student = perturb_student(teacher)
    gen = torch.Generator().manual_seed(9)
    is_obj = (torch.rand(N, 1, generator=gen) < 0.3).float().to(DEV)
    cams = [make_camera(k, H, W, device=DEV) for k in (0, 40, 80, 120)]
    bg = torch.zeros(3, device=DEV)

    def rot(a):
        c, s_ = math.cos(a), math.sin(a)
        return torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], device=DEV)
    Rs = [rot(0.2 * k) for k in range(4)]
    gates = [(torch.rand(H, W, generator=gen) > 0.2).float().to(DEV) for _ in range(4)]
    kw = lambda k: dict(rot_cov=True, accum_R=Rs[k], which_object=1, during_training=False)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=DEV, requires_grad=False); tpc._is_object = is_obj
        gts = [render(c, tpc, Pipe, bg, **kw(k))["render"].clone() for k, c in enumerate(cams)]
    groups = lambda pc: [{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                         {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}]
    WARM = 2
    pa = SynthGaussians(student, device=DEV); pa._is_object = is_obj
    oa = FusedAdam(groups(pa), lr=0.0, eps=1e-15)
    for k in [0] * WARM + [i % 4 for i in range(1, K + 1)]:
        out = render(cams[k], pa, Pipe, bg, **kw(k))
        l1_ssim_loss(out["render"], gts[k], 0.2, grad_gate=gates[k]).backward()
        oa.step(); oa.zero_grad(set_to_none=True)
    pb = SynthGaussians(student, device=DEV); pb._is_object = is_obj
    ob = FusedAdam(groups(pb), lr=0.0, eps=1e-15, capturable=True)
    step = GraphedTrainStep(pb, ob, bg, dynamic=True, gated=True).capture(cams[0], gts[0], warmup=WARM, accum_R=Rs[0], gate=gates[0])
    for i in range(1, K + 1):
        k = i % 4
        if i % 2:
            step(cams[k], gts[k], accum_R=Rs[k], gate=gates[k])                           # four separate copies
        else:
            step(pack_frame(cams[k], gts[k], Rs[k], gates[k]))                            # one packed copy
    torch.cuda.synchronize()
    assert step.ok()
    for a, b in zip(pa.parameters(), pb.parameters()):
        if a.numel():
            diff = (a.detach() - b.detach()).abs()
            assert float((diff > 2e-5 * float(a.detach().abs().max()) + 1e-6).float().mean()) < 2e-3 and float(diff.max()) < 0.02
    # the gate really gates: a zero gate leaves no gradient at all
    ob.zero_grad(set_to_none=True)                                       # (the replays left their gradients in .grad)
    out = render(cams[1], pb, Pipe, bg, **kw(1))
    l1_ssim_loss(out["render"], gts[1], 0.2, grad_gate=torch.zeros(H, W, device=DEV)).backward()
    assert all(float(p.grad.abs().max()) == 0.0 for p in (pb._xyz, pb._features_dc, pb._opacity, pb._scaling, pb._rotation))


def test_several_steps_per_replay_match_single_step_replays():
    """GraphedTrainStep(steps_per_replay=3): one launch = three complete iterations on three frames; six such steps must leave the
    parameters where six single-step replays leave them (up to the order of the float atomics), with every loss recorded."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.graph import GraphedTrainStep, pack_frame
    N, H, W = 20000, 96, 160
    teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
    student = perturb_student(teacher)
    cams = [make_camera(k, H, W, device=DEV) for k in (0, 30, 60, 90, 120, 150)]
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=DEV, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
    frames = [pack_frame(c, g) for c, g in zip(cams, gts)]
    groups = lambda pc: [{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                         {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}]
    res = {}
    for spr in (1, 3):
        pc = SynthGaussians(student, device=DEV)
        opt = FusedAdam(groups(pc), lr=0.0, eps=1e-15, capturable=True)
        step = GraphedTrainStep(pc, opt, bg, steps_per_replay=spr).capture(cams[0], gts[0], warmup=2)
        losses = []
        if spr == 1:
            for f in frames:
                losses.append(float(step(f)))
        else:
            for lo in (0, 3):
                step(torch.stack(frames[lo:lo + 3]) if lo == 0 else frames[lo:lo + 3])      # one stacked copy / a list of frames
                losses += [float(l) for l in step.losses]
        torch.cuda.synchronize()
        assert step.ok() and len(losses) == 6
        res[spr] = (losses, [p.detach().clone() for p in pc.parameters()], float(step.loss_sum), float(opt.state[pc._xyz]["step"]))
    assert res[1][3] == res[3][3] == 8.0                                  # 2 eager + 6 replayed Adam steps on both sides
    assert np.allclose(res[1][0], res[3][0], rtol=2e-4) and abs(res[1][2] - res[3][2]) < 1e-4 * abs(res[1][2])
    for a, b in zip(res[1][1], res[3][1]):
        if a.numel():
            diff = (a - b).abs()
            assert float((diff > 2e-5 * float(a.abs().max()) + 1e-6).float().mean()) < 2e-3 and float(diff.max()) < 0.02


def test_l1_ssim_deferred_value_and_running_sum():
    """defer_value=True moves the assembly of the scalar into the backward kernel (one launch less in a replayed training step):
    same value, same gradient; running_sum receives the value exactly once per loss, in either mode."""
    from egogaussian_amd.fused import l1_ssim_loss
    g = torch.Generator().manual_seed(5)
    img0, gt = torch.rand(3, 71, 129, generator=g).to(DEV), torch.rand(3, 71, 129, generator=g).to(DEV)
    res = []
    for defer in (False, True):
        img = img0.clone().requires_grad_(True)
        acc = torch.full((), 10.0, device=DEV)
        loss = l1_ssim_loss(img, gt, 0.2, running_sum=acc, defer_value=defer)
        if not defer:
            assert abs(float(acc) - 10.0 - float(loss.detach())) < 1e-6        # available (and summed) right after the forward
        loss.backward()
        res.append((float(loss.detach()), float(acc), img.grad.clone()))
    (l0, a0, g0), (l1, a1, g1) = res
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l0)) and abs(a0 - a1) < 1e-5 and abs(a1 - 10.0 - l1) < 1e-5
    assert torch.equal(g0, g1)


def test_object_rotation_inside_the_rasterizer_matches_the_covariance_path():
    """render(rot_cov=True, accum_R, which_object=1) with the covariance of the object's rows rotated INSIDE the preprocess kernels
    (egs_object_rotation) against the same call through the fused covariance producer (cov3D_precomp): identical covariance arithmetic,
    so radii and images are equal bit for bit; parameter gradients equal up to the order of the backward's float atomics -- including
    row 0, which the reference's [N,1]-index quirk rotates and whose gradient it multiplies (covariance.py)."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    N, H, W = 12000, 96, 160
    teacher = make_scene(N, H, W, 3); teacher["log_scale"] += math.log(2.0)
    student = perturb_student(teacher)
    gen = torch.Generator().manual_seed(4)
    is_obj = (torch.rand(N, 1, generator=gen) < 0.3).float().to(DEV)
    is_obj[0, 0] = 0.0                                                   # row 0 is background: only the quirk moves it
    cam, bg = make_camera(30, H, W, device=DEV), torch.zeros(3, device=DEV)
    c, s_ = math.cos(0.7), math.sin(0.7)
    R = torch.tensor([[c, -s_, 0.0], [s_, c * 0.9, -0.3], [0.1, 0.3, 0.95]], device=DEV)     # (any 3x3: nothing requires a rotation)
    gt = torch.rand(3, H, W, generator=gen).to(DEV)
    outs = []
    for inside in (True, False):
        pc = SynthGaussians(student, device=DEV); pc._is_object = is_obj
        pc.rotate_in_rasterizer = inside
        out = render(cam, pc, Pipe, bg, rot_cov=True, accum_R=R, which_object=1, during_training=False)
        l1_ssim_loss(out["render"], gt, 0.2).backward()
        torch.cuda.synchronize()
        outs.append((out, pc))
    (oa, pa), (ob, pb) = outs
    assert torch.equal(oa["radii"], ob["radii"]) and int((oa["radii"] > 0).sum()) > 1000
    for k in ("render", "depth", "alpha"):
        assert torch.equal(oa[k], ob[k]), k
    for name in ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"):
        ga, gb = getattr(pa, name).grad, getattr(pb, name).grad
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-9, name
    assert float(pa._scaling.grad[0].abs().max()) > 0                    # row 0 saw the rotation (and the multiplier) on both paths
    # and a rotation that is being trained keeps the covariance path (its gradient needs it)
    pc = SynthGaussians(student, device=DEV); pc._is_object = is_obj
    assert pc.get_raw_parameters_rotated(R.clone().requires_grad_(True), 1, False) is None


@pytest.mark.gpu
def test_double_buffered_graph_step_matches_single_buffered():
    """GraphedTrainStep(double_buffer=True): two captures of the same iterations on two sets of static frames, alternated, the copy of
    the next call's frames on a side stream.  Ten calls of three iterations each must leave the parameters where the single-buffered
    step leaves them (same frames in the same order; only the accumulation order of the atomics differs), count every step, and
    report each call's losses."""
    import math
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.graph import GraphedTrainStep, pack_frame
    N, H, W, S, CALLS = 20000, 96, 160, 3, 10
    teacher = make_scene(N, H, W, 0); teacher["log_scale"] += math.log(2.0)
    student = perturb_student(teacher)
    cams = [make_camera(k, H, W, device=DEV) for k in range(0, 120, 10)]
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=DEV, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
    frames = torch.stack([pack_frame(c, g_) for c, g_ in zip(cams, gts)])
    res = {}
    for db in (False, True, "fresh", "fresh_event"):
        pc = SynthGaussians(student, device=DEV)
        opt = FusedAdam([{"params": [pc._xyz], "lr": 1.6e-4}, {"params": [pc._features_dc], "lr": 2.5e-3}, {"params": [pc._opacity], "lr": 0.05},
                         {"params": [pc._scaling], "lr": 5e-3}, {"params": [pc._rotation], "lr": 1e-3}], lr=0.0, eps=1e-15, capturable=True)
        step = GraphedTrainStep(pc, opt, bg, steps_per_replay=S, double_buffer=bool(db)).capture(cams[0], gts[0], warmup=2)
        losses = []
        producer = torch.cuda.Stream()
        for c in range(CALLS):
            k = (c * S) % (len(cams) - S + 1)
            if db == "fresh":
                # the frames are WRITTEN on the current stream right before the call and dropped right after it -- the side-stream
                # copy must wait for the write and keep the allocation alive (a filler of the same size is allocated at once)
                fr = torch.empty_like(frames[k:k + S]); fr.fill_(float("nan")); fr.copy_(frames[k:k + S])
                step(fr); del fr
                junk = torch.full_like(frames[k:k + S], float("nan")); del junk
            elif db == "fresh_event":
                producer.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(producer):
                    fr = torch.empty_like(frames[k:k + S]); fr.fill_(float("nan")); fr.copy_(frames[k:k + S])
                    ev = torch.cuda.Event(); ev.record(producer)
                step(fr, ready=ev); fr.record_stream(torch.cuda.current_stream()); del fr
            else:
                step(frames[k:k + S])
            losses.append([l.clone() for l in step.losses])
        torch.cuda.synchronize()
        assert step.ok()
        res[db] = (pc, opt, [[float(l) for l in ls] for ls in losses], float(step.loss_sum))
    (pa, oa, la, sa) = res[False]
    for mode in (True, "fresh", "fresh_event"):
        (pb, ob, lb, sb) = res[mode]
        assert float(oa.state[pa._xyz]["step"]) == float(ob.state[pb._xyz]["step"]) == 2 + S * CALLS
        for x, y in zip(la, lb):
            assert all(abs(u - v) <= 2e-3 * abs(u) for u, v in zip(x, y)), (mode, x, y)
        assert abs(sa - sb) <= 2e-3 * abs(sa)
        for a in ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"):
            u, v = getattr(pa, a).detach(), getattr(pb, a).detach()
            assert float((u - v).abs().mean()) <= 2e-4 * float(u.abs().mean()) + 1e-7, (mode, a)
