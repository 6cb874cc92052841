"""GPU: the optimizer inside the rasterizer backward (include/egs_raster.h egs_backward_adam, optim.FusedAdam.make_sink).

What it replaces in the reference: loss.backward() hands per-parameter gradients to autograd and optimizer.step() reads them back
(/root/reference/trainers/train_static.py:97,137; torch.optim.Adam(l, lr=0.0, eps=1e-15), scene/gaussian_model.py:180-198).
The fused path must take exactly the step the stand-alone kernel takes from the same gradients -- checked bit for bit, with the
gradients of the owned leaves written as well (the C ABI allows both) so that the stand-alone step can be replayed on copies."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LEAVES = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation")


def _groups(pc):
    g = [{"params": [pc._xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [pc._features_dc], "lr": 2.5e-3, "name": "f_dc"},
         {"params": [pc._opacity], "lr": 0.05, "name": "opacity"}, {"params": [pc._scaling], "lr": 5e-3, "name": "scaling"},
         {"params": [pc._rotation], "lr": 1e-3, "name": "rotation"}]
    if pc._features_rest.numel():
        g.append({"params": [pc._features_rest], "lr": 2.5e-3 / 20, "name": "f_rest"})      # gaussian_model.py:189
    return g


def _scene(N=12000, H=96, W=160, seed=0, sh_degree=0):
    from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    teacher = make_scene(N, H, W, seed, sh_degree=sh_degree); teacher["log_scale"] += math.log(2.0)
    cams = [make_camera(k, H, W, device=DEV) for k in (0, 40, 80, 120)]
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=DEV, sh_degree=sh_degree, requires_grad=False)
        gts = [render(c, tpc, Pipe, bg)["render"].clone() for c in cams]
    return perturb_student(teacher), cams, gts, bg


@pytest.mark.parametrize("sh_degree", [0, 3])
def test_fused_step_is_the_standalone_step_bit_for_bit(sh_degree):
    """(sh_degree 3: features_dc / features_rest handed over split, 16 coefficients -- those two and the positions are then stepped by
    the spherical-harmonics launch, the other three by the preprocess backward.)
    Four iterations; in each the backward steps the five leaves itself AND writes their gradients; a twin optimizer on copies of
    (parameter, exp_avg, exp_avg_sq) taken before the backward steps with those gradients through the stand-alone kernel: parameters,
    both moments and state["step"] must be equal bit for bit.  The fifth iteration goes the other way round -- the optimizer that was
    fused so far takes a stand-alone step (its own step counters are re-seeded from state["step"]) -- and must again agree."""
    from egogaussian_amd.scene_synth import SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    import egogaussian_amd.optim as optim
    student, cams, gts, bg = _scene(sh_degree=sh_degree)
    pa = SynthGaussians(student, device=DEV, sh_degree=sh_degree)
    oa = FusedAdam(_groups(pa), lr=0.0, eps=1e-15, capturable=True)
    pb = SynthGaussians(student, device=DEV, sh_degree=sh_degree)
    ob = FusedAdam(_groups(pb), lr=0.0, eps=1e-15, capturable=True)
    LEAVES = globals()["LEAVES"] + (("_features_rest",) if sh_degree else ())
    real_make = oa.make_sink

    def keeping(**kw):
        sink = real_make(**kw)
        sink.keep_grads = True
        keeping.last = sink
        return sink
    oa.make_sink = keeping
    for it in range(5):
        k = it % 4
        fused = it < 4
        for a in LEAVES:                                            # the twin starts every iteration from the same bits
            with torch.no_grad():
                getattr(pb, a).copy_(getattr(pa, a))
        out = render(cams[k], pa, Pipe, bg, optimizer=oa if fused else None)
        loss = l1_ssim_loss(out["render"], gts[k], 0.2)
        # capture the gradients the backward writes: hook on the rasterizer's Function is not needed -- with keep_grads the
        # backward returns them to autograd as usual AND has already stepped the leaves
        before = {a: getattr(pa, a).detach().clone() for a in LEAVES}
        loss.backward()
        if fused:
            assert keeping.last.owned == ({0, 1, 2, 3, 4, 5} if sh_degree else {0, 1, 2, 3, 4})
            for a in LEAVES:
                g = getattr(pa, a).grad
                assert g is not None and float(g.abs().max()) > 0            # (keep_grads) the gradient arrays were written too
                assert not torch.equal(before[a], getattr(pa, a).detach()), f"{a}: the backward did not step it"
                getattr(pb, a).grad = g.clone()
            oa.step()                                                # the gradients are there (keep_grads), yet nothing may be stepped twice
            oa.zero_grad(set_to_none=True)
        else:
            for a in LEAVES:
                getattr(pb, a).grad = getattr(pa, a).grad.clone()
            oa.step(); oa.zero_grad(set_to_none=True)
        ob.step(); ob.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        for a in LEAVES:
            x, y = getattr(pa, a), getattr(pb, a)
            assert torch.equal(x.detach(), y.detach()), f"iteration {it}: {a} differs"
            for key in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(oa.state[x][key], ob.state[y][key]), f"iteration {it}: {a} {key} differs"
            assert float(oa.state[x]["step"]) == float(ob.state[y]["step"]) == it + 1


def test_fused_leaves_have_no_gradient_arrays_and_step_once():
    """Default use (no keep_grads): the owned leaves end the backward with .grad None, took exactly one step, and the screen-space
    gradient (densification statistic input) is still delivered."""
    from egogaussian_amd.scene_synth import SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    student, cams, gts, bg = _scene(N=6000)
    pc = SynthGaussians(student, device=DEV)
    opt = FusedAdam(_groups(pc), lr=0.0, eps=1e-15, capturable=True)
    before = {a: getattr(pc, a).detach().clone() for a in LEAVES}
    out = render(cams[0], pc, Pipe, bg, optimizer=opt)
    l1_ssim_loss(out["render"], gts[0], 0.2).backward()
    opt.step()
    torch.cuda.synchronize()
    assert out["viewspace_points"].grad is not None and float(out["viewspace_points"].grad.abs().max()) > 0
    for a in LEAVES:
        p = getattr(pc, a)
        assert p.grad is None and float(opt.state[p]["step"]) == 1.0
        moved = (p.detach() != before[a]).any(dim=tuple(range(1, p.dim())))
        assert bool(moved[out["visibility_filter"]].any())
    # an evaluation render under no_grad builds no sink and steps nothing
    snap = pc._xyz.detach().clone()
    with torch.no_grad():
        render(cams[1], pc, Pipe, bg, optimizer=opt)
    torch.cuda.synchronize()
    assert torch.equal(snap, pc._xyz.detach()) and float(opt.state[pc._xyz]["step"]) == 1.0


def test_alternating_fused_and_plain_steps_keep_the_step_count():
    """fused backward (sink built), plain step(), fused backward (sink found in the cache), plain step(), ...: the plain steps must use
    the step number the fused ones advanced (bias correction), state["step"] never goes backwards, and the parameters follow
    torch.optim.Adam fed with the same gradients (the fused backward writes them too: keep_grads)."""
    from egogaussian_amd.scene_synth import SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    student, cams, gts, bg = _scene(N=6000)
    pa = SynthGaussians(student, device=DEV)
    oa = FusedAdam(_groups(pa), lr=0.0, eps=1e-15, capturable=True)
    pb = SynthGaussians(student, device=DEV)
    ob = torch.optim.Adam(_groups(pb), lr=0.0, eps=1e-15)
    real_make, made = oa.make_sink, []

    def keeping(**kw):
        sink = real_make(**kw)
        sink.keep_grads = True
        made.append(sink)
        return sink
    oa.make_sink = keeping
    for it in range(6):
        fused = it % 2 == 0
        out = render(cams[it % 4], pa, Pipe, bg, optimizer=oa if fused else None)
        l1_ssim_loss(out["render"], gts[it % 4], 0.2).backward()
        for a in LEAVES:
            getattr(pb, a).grad = getattr(pa, a).grad.clone()
        oa.step(); oa.zero_grad(set_to_none=True)
        ob.step(); ob.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        for a in LEAVES:
            x, y = getattr(pa, a), getattr(pb, a)
            assert float(oa.state[x]["step"]) == it + 1 == float(ob.state[y]["step"]), (it, a, float(oa.state[x]["step"]))
            err = float((x.detach() - y.detach()).abs().max())
            assert err <= 1e-5 * max(1.0, float(y.detach().abs().max())), f"iteration {it}: {a} off torch Adam by {err}"
    assert len(made) == 3 and made[1] is made[0] and made[2] is made[0]          # the later fused steps DID take the cached sink


def test_two_fused_consumers_in_one_iteration_are_refused():
    """The sole-consumer precondition of egs_backward_adam: a loss that reaches the parameters through TWO renders, both with
    optimizer=, would step every leaf twice with partial gradients -- the second backward raises."""
    from egogaussian_amd.scene_synth import SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    student, cams, gts, bg = _scene(N=4000)
    pc = SynthGaussians(student, device=DEV)
    opt = FusedAdam(_groups(pc), lr=0.0, eps=1e-15, capturable=True)
    a = render(cams[0], pc, Pipe, bg, optimizer=opt)
    b = render(cams[1], pc, Pipe, bg, optimizer=opt)
    loss = l1_ssim_loss(a["render"], gts[0], 0.2) + l1_ssim_loss(b["render"], gts[1], 0.2)
    with pytest.raises(RuntimeError, match="second rasterizer backward"):
        loss.backward()
    torch.cuda.synchronize()
    # the refusal comes BEFORE the second launch -- exactly one step was taken (by the first of the two backwards to run)
    assert all(float(opt.state[p]["step"]) == 1.0 for g in opt.param_groups for p in g["params"])
    opt.zero_grad(set_to_none=True)                                   # a new iteration (no step() in between: zero_grad() is enough)
    out = render(cams[2], pc, Pipe, bg, optimizer=opt)
    l1_ssim_loss(out["render"], gts[2], 0.2).backward()               # an iteration that ends without step() ...
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    assert all(float(opt.state[p]["step"]) == 2.0 for g in opt.param_groups for p in g["params"])
    # one fused consumer per iteration, iteration after iteration, is the supported use
    for k in range(2):
        out = render(cams[k], pc, Pipe, bg, optimizer=opt)
        l1_ssim_loss(out["render"], gts[k], 0.2).backward()
        opt.step()
    torch.cuda.synchronize()


def test_graph_step_with_and_without_fused_optimizer_agree():
    """GraphedTrainStep(fuse_optimizer=True / False) on the same frames: same step counts, no gradient arrays for the leaves in the
    fused graph, and parameters that differ from the unfused run's no more than two unfused runs differ from each other (the backward's
    float atomics land in a different order on every run, and Adam at eps = 1e-15 amplifies last-bit differences of near-zero gradients;
    the bit-for-bit statement is test_fused_step_is_the_standalone_step_bit_for_bit)."""
    from egogaussian_amd.scene_synth import SynthGaussians
    from egogaussian_amd.optim import FusedAdam
    from egogaussian_amd.graph import GraphedTrainStep
    student, cams, gts, bg = _scene(N=15000)
    res = []
    for fuse in (False, False, True):
        pc = SynthGaussians(student, device=DEV)
        opt = FusedAdam(_groups(pc), lr=0.0, eps=1e-15, capturable=True)
        step = GraphedTrainStep(pc, opt, bg, 0.2, fuse_optimizer=fuse, densify_stats=False).capture(cams[0], gts[0], warmup=2)
        for i in range(1, 9):
            step(cams[i % 4], gts[i % 4])
        torch.cuda.synchronize()
        assert step.ok()
        for a in LEAVES:
            assert float(opt.state[getattr(pc, a)]["step"]) == 10.0
            assert (getattr(pc, a).grad is None) == fuse
        res.append(pc)

    def off(pa, pb, a):                                               # (fraction of entries apart by more than the tolerance, largest difference)
        x, y = getattr(pa, a).detach(), getattr(pb, a).detach()
        diff = (x - y).abs()
        return float((diff > 2e-5 * float(x.abs().max()) + 1e-6).float().mean()), float(diff.max())
    for a in LEAVES:
        noise_frac, noise_max = off(res[0], res[1], a)                # unfused vs unfused: the run-to-run noise
        frac, mx = off(res[0], res[2], a)
        assert frac <= max(3.0 * noise_frac, 2e-3) and mx <= max(3.0 * noise_max, 0.02), (a, frac, mx, noise_frac, noise_max)


def test_partial_sinks_follow_the_library_conditions():
    """cov3D_precomp given (the `fine_all` call shape through the covariance producer): scaling / rotation / opacity reach the
    rasterizer as activations, so only xyz and features_dc are owned and the other three are stepped by optimizer.step() from their
    gradient arrays; the same call with the object rotation inside the rasterizer: all five are owned; colours precomputed: only xyz
    can be owned."""
    from egogaussian_amd.scene_synth import SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    import egogaussian_amd.lib as lib
    student, cams, gts, bg = _scene(N=6000)
    pc = SynthGaussians(student, device=DEV)
    pc._is_object = (torch.rand(6000, 1, generator=torch.Generator().manual_seed(3)) < 0.3).float().to(DEV)
    opt = FusedAdam(_groups(pc), lr=0.0, eps=1e-15, capturable=True)
    seen = []
    real = opt.make_sink
    opt.make_sink = lambda **kw: seen.append(real(**kw)) or seen[-1]
    pc.rotate_in_rasterizer = False                                  # (the producer path: cov3D_precomp + activated opacity)
    out = render(cams[0], pc, Pipe, bg, rot_cov=True, accum_R=torch.eye(3, device=DEV), which_object=1, during_training=False, optimizer=opt)
    l1_ssim_loss(out["render"], gts[0], 0.2).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    assert seen[-1].owned == {lib.SINK_MEANS3D, lib.SINK_SH}
    for a in LEAVES:
        assert float(opt.state[getattr(pc, a)]["step"]) == 1.0, a
    pc.rotate_in_rasterizer = True
    out = render(cams[0], pc, Pipe, bg, rot_cov=True, accum_R=torch.eye(3, device=DEV), which_object=1, during_training=False, optimizer=opt)
    l1_ssim_loss(out["render"], gts[0], 0.2).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    assert seen[-1].owned == {0, 1, 2, 3, 4} and all(getattr(pc, a).grad is None for a in LEAVES)
    for a in LEAVES:
        assert float(opt.state[getattr(pc, a)]["step"]) == 2.0, a
    for a in LEAVES:                                                 # (the counts below start from here)
        opt.state[getattr(pc, a)]["step"].fill_(1.0)
    out = render(cams[1], pc, Pipe, bg, override_color=torch.rand(pc.get_xyz.shape[0], 3, device=DEV), optimizer=opt)
    l1_ssim_loss(out["render"], gts[1], 0.2).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    assert lib.SINK_SH not in seen[-1].owned and lib.SINK_MEANS3D in seen[-1].owned
    assert float(opt.state[pc._xyz]["step"]) == 2.0 and float(opt.state[pc._features_dc]["step"]) == 1.0


def test_c_abi_rejects_inconsistent_sinks():
    """egs_backward_adam argument checking: a leaf whose `param` is not the array passed as that input, a scales leaf together with
    cov3D_precomp, a missing moment array."""
    import ctypes as C
    import egogaussian_amd.lib as lib
    L = lib.load()
    P, H, W = 64, 32, 32
    z = lambda *s: torch.zeros(s, device=DEV)
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    means, sh, scales, rots, cov = z(P, 3), z(P, 1, 3), z(P, 3), z(P, 4), z(P, 6)
    m, v, lr, st, coef = z(P, 3), z(P, 3), z(1), z(1), z(10)
    radii = torch.zeros(P, dtype=torch.int32, device=DEV)
    geom = torch.zeros(L.egs_geom_bytes(P), dtype=torch.uint8, device=DEV)
    img = torch.zeros(L.egs_image_bytes(W, H), dtype=torch.uint8, device=DEV)
    scratch = torch.zeros(L.egs_backward_scratch_bytes(P), dtype=torch.uint8, device=DEV)
    cam = z(16); pos = z(3); bg = z(3); gcol = z(3, H, W)

    def call(sink, use_cov=False, scales_in=scales):
        return L.egs_backward_adam(P, 0, 1, 0, vp(bg), vp(means), vp(sh), None, None, None if use_cov else vp(scales_in), 1.0,
                                   None if use_cov else vp(rots), vp(cov) if use_cov else None, 0, vp(cam), vp(cam), vp(pos), W, H, 1.0, 1.0,
                                   vp(radii), vp(geom), None, vp(img), vp(gcol), None, None, vp(z(P, 3)), vp(z(P, 3)), vp(z(P, 1)), vp(z(P, 3)),
                                   vp(z(P, 6)) if use_cov else None, vp(z(P, 1, 3)), None, None if use_cov else vp(z(P, 3)),
                                   None if use_cov else vp(z(P, 4)), None, None, None, None, C.byref(sink), 0, None, 0, vp(scratch), None, 0)

    def sink_for(leaf, param, moments=True):
        s = lib.AdamSink()
        f = s.leaf[leaf]
        f.param, f.lr, f.step = param.data_ptr(), lr.data_ptr(), st.data_ptr()
        if moments:
            f.exp_avg, f.exp_avg_sq = m.data_ptr(), v.data_ptr()
        s.beta1, s.beta2, s.eps, s.coef = 0.9, 0.999, 1e-15, coef.data_ptr()
        return s
    ERR_ARG, ERR_MODE = -1, -2
    assert call(sink_for(lib.SINK_SCALES, scales)) == 0                              # R = 0: nothing rendered, a plain zero-gradient step
    torch.cuda.synchronize()
    assert float(st) == 1.0
    assert call(sink_for(lib.SINK_SCALES, z(P, 3))) == ERR_ARG                       # not the array passed as `scales`
    assert call(sink_for(lib.SINK_SCALES, scales), use_cov=True) == ERR_MODE
    assert call(sink_for(lib.SINK_MEANS3D, means, moments=False)) == ERR_ARG
    s = sink_for(lib.SINK_MEANS3D, means); s.coef = None
    assert call(s) == ERR_ARG


@pytest.mark.parametrize("H,W", [(96, 160), (540, 960), (1080, 1920), (2160, 3840)])
def test_loss_backward_carrying_the_raster_prologue(H, W):
    """(sizes: a band of 8 / 255 / 1 020 / 4 050 tiles per XCD -- the balanced small-band path, its limit, the snake-order path)
    l1_ssim_loss(raster_prologue=True): the loss's backward launch prepares the rasterizer's backward (tile order, cleared
    accumulator, optimizer bookkeeping) and that backward starts at its blend kernel.  Same image gradient bit for bit, same
    parameter gradients up to the order of the float atomics, every tile handed to exactly one workgroup, one Adam step counted."""
    from egogaussian_amd import _C
    from egogaussian_amd.scene_synth import SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    from egogaussian_amd.fused import l1_ssim_loss
    from egogaussian_amd.optim import FusedAdam
    student, cams, gts, bg = _scene(N=8000, H=H, W=W)
    grads, orders = [], []
    for carried in (False, True):
        pc = SynthGaussians(student, device=DEV)
        out = render(cams[1], pc, Pipe, bg)
        img_buf = _C.stats["image_buffer"]
        out["render"].retain_grad()
        l1_ssim_loss(out["render"], gts[1], 0.2, raster_prologue=carried).backward()
        torch.cuda.synchronize()
        grads.append({a: getattr(pc, a).grad.clone() for a in LEAVES} | {"image": out["render"].grad.clone(), "screen": out["viewspace_points"].grad.clone()})
        nt = ((W + 15) // 16) * ((H + 15) // 16)
        lay = _C._lib.ImageLayout(); _C._lib.load().egs_get_image_layout(W, H, __import__("ctypes").byref(lay))
        words = img_buf[lay.tile_order:lay.tile_order + 4 * _C._lib.load().egs_order_words(W, H)].view(torch.int32).cpu().numpy().astype("uint32")
        tiles = [int(w & 0xffff) if (w & 0x01000000) else int(w) for w in words if w != 0xffffffff]
        orders.append(sorted(tiles))
    assert orders[0] == orders[1] == list(range(nt))
    assert torch.equal(grads[0]["image"], grads[1]["image"])
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, k
    # with a fused optimizer: the bookkeeping rides in the loss launch, one step is counted and taken
    pc = SynthGaussians(student, device=DEV)
    opt = FusedAdam(_groups(pc), lr=0.0, eps=1e-15, capturable=True)
    before = pc._xyz.detach().clone()
    for it in range(2):
        out = render(cams[it], pc, Pipe, bg, optimizer=opt)
        l1_ssim_loss(out["render"], gts[it], 0.2, raster_prologue=True, defer_value=True).backward()
        opt.step()
    torch.cuda.synchronize()
    assert all(float(opt.state[getattr(pc, a)]["step"]) == 2.0 and getattr(pc, a).grad is None for a in LEAVES)
    assert not torch.equal(before, pc._xyz.detach())


@pytest.mark.parametrize("H,W", [(96, 160), (1080, 1920), (2160, 3840)])
def test_forward_placement_buffer_never_changes_results(H, W):
    """The persistent placement buffer (include/egs_raster.h): whatever its cost and tile-order words hold -- zeros, the previous frame's
    costs, random words -- every tile is blended exactly once and the outputs are bit-identical; after a forward it holds that forward's
    per-quadrant costs.  ABI 5: the sums region behind them (the fused count pass's chunk sums) is zero between frames -- the chain that
    used it clears it -- and is the one part a caller must not scribble on."""
    from egogaussian_amd import _C
    from egogaussian_amd.scene_synth import SynthGaussians, Pipe
    from egogaussian_amd.renderer import render
    import egogaussian_amd.lib as lib
    student, cams, gts, bg = _scene(N=8000, H=H, W=W)
    pc = SynthGaussians(student, device=DEV, requires_grad=False)
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    place = _C.placement_buffer(torch.device(DEV), W, H)
    n_free = ((nt * 4 + lib.load().egs_order_words(W, H)) * 4 + 255) // 256 * 256 // 4      # words in front of the sums region
    words = place.view(torch.int32)[:n_free]
    sums = place.view(torch.int32)[n_free:]
    assert sums.numel() >= 8 * nt
    outs = []
    gen = torch.Generator().manual_seed(5)
    for fill in ("zeros", "previous", "random", "huge"):
        if fill == "zeros":
            words.zero_()
        elif fill == "random":
            words.copy_(torch.randint(-2**31, 2**31 - 1, words.shape, generator=gen, dtype=torch.int64).to(torch.int32).to(DEV))
        elif fill == "huge":
            words.fill_(-1)
        with torch.no_grad():
            out = render(cams[2], pc, Pipe, bg)
        torch.cuda.synchronize()
        outs.append((out["render"].clone(), out["depth"].clone(), out["alpha"].clone()))
        order = words[nt * 4:nt * 4 + lib.load().egs_order_words(W, H)].cpu().numpy().astype("uint32")
        tiles = sorted(int(w & 0xffff) if (w & 0x01000000) else int(w) for w in order if w != 0xffffffff)
        assert tiles == list(range(nt)), fill
        cost = words[:nt * 4].cpu().numpy()
        visits = _C.image_views(_C.stats["image_buffer"], W, H)["quad_visits"].cpu().numpy().reshape(-1)
        assert (cost >= 10 * visits).all() and cost.sum() > 0
        assert not bool(sums.any()), "the chain left chunk sums behind"
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    assert lib.load().egs_placement_bytes(W, H) == place.numel()
