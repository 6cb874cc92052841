"""Densification bookkeeping on the device, with the reference's method names and
semantics so that a trainer can call these instead of the GaussianModel methods:

    reference (scene/gaussian_model.py)                        here
    -----------------------------------------------------     --------------------------------------------------------
    gaussians.max_radii2D[vis] = max(...)  (train_static:125)  add_densification_stats(gaussians, viewspace, vis, radii)
    gaussians.add_densification_stats(viewspace, vis)  :735        (one kernel for both; no index tensors)
    gaussians.densify_and_prune(...)                   :678     densify_and_prune(gaussians, ...)
    gaussians.prune_points(mask)                       :536     prune_points(gaussians, mask)
    gaussians.reset_opacity()                          :484     reset_opacity(gaussians)

`gaussians` is duck-typed: the reference's GaussianModel or egogaussian_amd.scene_synth.SynthGaussians -- anything with
`_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation, _label` (parameters registered one per group, named
"xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "label", in `optimizer`), `_generation`, `_is_object`,
`xyz_gradient_accum`, `denom`, `max_radii2D`, `percent_dense`.  As in the reference every call leaves NEW nn.Parameter
objects in the model and in the optimizer, with the Adam moments carried over for surviving Gaussians and zero for new ones.

The selection rules and the element order are the reference's, evaluated by HIP kernels (csrc/densify.hip): one pass
computes, for every Gaussian, which of {itself, a clone, two split children} survive; scans turn that into a plan
(source index + kind per new Gaussian); one gather per array builds the new model.  The reference materialises ~40
intermediate masked copies and concatenations per call and synchronises with the host at every boolean index; here the
host reads four counters once.  HIP device tensors only (no CPU path; the CPU restatement is oracle/densify_torch.py, pinned
by a fixture captured from the reference).
"""
import ctypes as C

import torch
from torch import nn

from . import lib as _lib
from . import _hip as _launch

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "label")
_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
         "rotation": "_rotation", "label": "_label"}


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return _launch.stream_of(torch.device("cuda", torch.cuda.current_device()))


def _hip(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the densification kernels have no CPU path")
    return t


def add_densification_stats(gaussians, viewspace_point_tensor, update_filter, radii=None):
    """xyz_gradient_accum[f] += |viewspace.grad[f, :2]|, denom[f] += 1 and, when `radii` is given, max_radii2D[f] =
    max(max_radii2D[f], radii[f]) for the visible Gaussians f -- in place, one kernel."""
    L = _lib.load()
    g = _hip(viewspace_point_tensor.grad, "viewspace_point_tensor.grad").float().contiguous()
    P = g.shape[0]
    vis = None
    if update_filter is not None:
        vis = update_filter if update_filter.dtype in (torch.bool, torch.uint8) else update_filter != 0
        vis = vis.contiguous().view(torch.uint8)
    r = None if radii is None else radii.to(torch.int32).contiguous()
    acc, den = gaussians.xyz_gradient_accum, gaussians.denom
    mr = gaussians.max_radii2D if radii is not None else None
    for t in (acc, den) + ((mr,) if mr is not None else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("densification statistics must be contiguous float32 HIP tensors")
    with torch.cuda.device(g.device):
        _lib.check(L.egs_densify_stats(P, _p(g), _p(vis), _p(r), _p(acc), _p(den), _p(mr), _stream()))


class _Plan:
    """Device-side description of the new model: src[k], kind[k] for its k-th Gaussian."""

    def __init__(self, P, dev):
        L = _lib.load()
        self.P, self.dev = P, dev
        self.scratch = torch.empty(L.egs_densify_plan_scratch_bytes(P), dtype=torch.uint8, device=dev)
        self.src = torch.empty(3 * P, dtype=torch.int32, device=dev)
        self.kind = torch.empty(3 * P, dtype=torch.uint8, device=dev)
        self.split_rank = torch.empty(P, dtype=torch.int32, device=dev)
        self.totals = torch.zeros(4, dtype=torch.int64, device=dev)

    def finish(self):
        n_orig, n_clone, n_child, n_split = [int(v) for v in self.totals.tolist()]          # the one host read
        self.n_split, self.n_new = n_split, n_orig + n_clone + 2 * n_child
        self.counts = (n_orig, n_clone, n_child)
        return self

    def rows(self, t, zero_new=False):
        """New float array: row k = t[src[k]]; zero_new: zeros for every row that is not a kept original (Adam moments)."""
        L = _lib.load()
        src = t.detach().float().contiguous()
        D = int(src.numel() // max(src.shape[0], 1))
        out = torch.empty((self.n_new,) + tuple(src.shape[1:]), dtype=torch.float32, device=self.dev)
        with torch.cuda.device(self.dev):
            _lib.check(L.egs_gather_rows_f32(self.n_new, D, _p(self.src), _p(self.kind), int(zero_new), _p(src), _p(out), _stream()))
        return out

    def ints(self, t, clone_value=None):
        L = _lib.load()
        orig_dtype = t.dtype
        src = t.detach().to(torch.int32).contiguous()
        out = torch.empty((self.n_new,) + tuple(src.shape[1:]), dtype=torch.int32, device=self.dev)
        with torch.cuda.device(self.dev):
            _lib.check(L.egs_gather_i32(self.n_new, _p(self.src), _p(self.kind), int(clone_value is not None),
                                        0 if clone_value is None else int(clone_value), _p(src.view(-1)), _p(out.view(-1)), _stream()))
        return out.to(orig_dtype)


def _apply_in_place(gaussians, plan, reset_stats, z=None, curr_gen=None):
    """Capacity-sized model (capacity.CapacityGaussians): the new model is gathered from the live prefix into temporaries and
    copied back INTO the same arrays; tensors, Parameter objects and optimizer state entries keep their identity and address.
    Returns False (nothing changed) when the new count does not fit the capacity -- the caller grows the model first."""
    L = _lib.load()
    n_old, n_new = gaussians.n_active, plan.n_new
    if n_new > gaussians.capacity:
        return False
    live = lambda t: t.detach()[:n_old]
    old = {k: live(getattr(gaussians, _ATTR[k])) for k in GROUPS}
    new = {k: plan.rows(old[k]) for k in GROUPS}
    if plan.n_split:
        with torch.cuda.device(plan.dev):
            _lib.check(L.egs_split_children(plan.n_new, _p(plan.src), _p(plan.kind), _p(plan.split_rank), plan.n_split, _p(z),
                                            _p(old["xyz"].float().contiguous()), _p(old["scaling"].float().contiguous()),
                                            _p(old["rotation"].float().contiguous()), _p(new["xyz"]), _p(new["scaling"]), _stream()))
    opt = getattr(gaussians, "optimizer", None)
    with torch.no_grad():
        if opt is not None:
            for group in opt.param_groups:
                name = group.get("name")
                if name not in new:
                    continue
                stored = opt.state.get(group["params"][0], None)
                if stored is not None and "exp_avg" in stored:
                    for key in ("exp_avg", "exp_avg_sq"):
                        stored[key][:n_new].copy_(plan.rows(stored[key][:n_old], zero_new=True))
        for k in GROUPS:
            getattr(gaussians, _ATTR[k]).detach()[:n_new].copy_(new[k])
        gaussians._generation[:n_new].copy_(plan.ints(gaussians._generation[:n_old], clone_value=curr_gen))
        gaussians._is_object[:n_new].copy_(plan.ints(gaussians._is_object[:n_old]))
        if n_new < n_old:                                       # vacated rows hold no Gaussian: their tags must not be counted by anyone
            gaussians._generation[n_new:n_old].zero_()
            gaussians._is_object[n_new:n_old].zero_()
        for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = getattr(gaussians, name)
            if reset_stats:
                t.zero_()
            else:
                t[:n_new].copy_(plan.rows(t[:n_old]))
    gaussians.set_active(n_new)
    return True


def _apply(gaussians, plan, reset_stats, z=None, curr_gen=None, during_training=True):
    L = _lib.load()
    old = {k: getattr(gaussians, _ATTR[k]) for k in GROUPS}
    new = {k: plan.rows(old[k]) for k in GROUPS}
    if plan.n_split:
        with torch.cuda.device(plan.dev):
            _lib.check(L.egs_split_children(plan.n_new, _p(plan.src), _p(plan.kind), _p(plan.split_rank), plan.n_split, _p(z),
                                            _p(old["xyz"].detach().float().contiguous()), _p(old["scaling"].detach().float().contiguous()),
                                            _p(old["rotation"].detach().float().contiguous()), _p(new["xyz"]), _p(new["scaling"]), _stream()))
    opt = getattr(gaussians, "optimizer", None) if during_training else None
    if opt is not None:
        for group in opt.param_groups:
            name = group.get("name")
            if name not in new:                              # object pose groups etc. are left alone (gaussian_model.py:515, :250)
                continue
            p_old = group["params"][0]
            stored = opt.state.get(p_old, None)
            p_new = nn.Parameter(new[name].requires_grad_(True))
            if stored is not None:
                stored["exp_avg"] = plan.rows(stored["exp_avg"], zero_new=True)
                stored["exp_avg_sq"] = plan.rows(stored["exp_avg_sq"], zero_new=True)
                del opt.state[p_old]
                opt.state[p_new] = stored
            group["params"][0] = p_new
            new[name] = p_new
    for k in GROUPS:
        v = new[k]
        if opt is None and during_training:
            v = nn.Parameter(v.requires_grad_(True))
        setattr(gaussians, _ATTR[k], v)
    gaussians._generation = plan.ints(gaussians._generation, clone_value=curr_gen)
    gaussians._is_object = plan.ints(gaussians._is_object)
    if during_training:
        if reset_stats:
            gaussians.xyz_gradient_accum = torch.zeros((plan.n_new, 1), device=plan.dev)
            gaussians.denom = torch.zeros((plan.n_new, 1), device=plan.dev)
            gaussians.max_radii2D = torch.zeros((plan.n_new,), device=plan.dev)
        else:
            gaussians.xyz_gradient_accum = plan.rows(gaussians.xyz_gradient_accum)
            gaussians.denom = plan.rows(gaussians.denom)
            gaussians.max_radii2D = plan.rows(gaussians.max_radii2D)


def densify_and_prune(gaussians, max_grad, min_opacity, extent, max_screen_size, clone=True, split=True, curr_gen=None,
                      prune_prev_gen=True, split_prev_gen=True, which_object=None, z=None, generator=None):
    """Same arguments as GaussianModel.densify_and_prune (:678).  `z` ([2 * n_split, 3] standard-normal draws, first children
    first -- or a callable rows -> such a tensor) makes the split deterministic; by default it is drawn here with `generator`.  Returns (n_before, n_after)."""
    if not split_prev_gen and split:
        raise NotImplementedError("split_prev_gen=False: the reference raises here (curr_gen lands in densify_and_split's N slot, "
                                  "gaussian_model.py:698 vs :588); there is no behaviour to reproduce")
    if not prune_prev_gen and curr_gen is None:
        raise ValueError("prune_prev_gen=False needs curr_gen")
    L = _lib.load()
    xyz = _hip(gaussians._xyz, "_xyz")
    in_place = getattr(gaussians, "capacity", None) is not None      # capacity.CapacityGaussians: plan on the live prefix, write back in place
    P, dev = (gaussians.n_active if in_place else xyz.shape[0]), xyz.device
    if P == 0:
        return 0, 0
    f = lambda t: t.detach()[:P].float().contiguous()
    i32 = lambda t: t.detach()[:P].to(torch.int32).contiguous()
    plan = _Plan(P, dev)
    mss = 0.0 if not max_screen_size else float(max_screen_size)
    with torch.cuda.device(dev):
        _lib.check(L.egs_densify_plan(
            P, _p(f(gaussians.xyz_gradient_accum)), _p(f(gaussians.denom)), _p(f(gaussians._scaling)), _p(f(gaussians._opacity)),
            _p(f(gaussians.max_radii2D)), _p(i32(gaussians._generation)), _p(i32(gaussians._is_object)), float(max_grad),
            float(min_opacity), float(gaussians.percent_dense), float(extent), mss, int(bool(clone)), int(bool(split)),
            int(curr_gen is not None), 0 if curr_gen is None else int(curr_gen), int(bool(prune_prev_gen)),
            int(which_object is not None), 0 if which_object is None else int(which_object), _p(plan.scratch), _p(plan.src),
            _p(plan.kind), _p(plan.split_rank), _p(plan.totals), _stream()))
    plan.finish()
    if plan.n_split:
        if z is None:
            z = torch.randn((2 * plan.n_split, 3), device=dev, generator=generator)
        elif callable(z):
            z = z(2 * plan.n_split).to(dev)                          # z(rows) -> [rows, 3]: the caller's draws, sized once the plan is known
        z = _hip(z, "z").float().contiguous()
        if tuple(z.shape) != (2 * plan.n_split, 3):
            raise ValueError(f"z must be [{2 * plan.n_split}, 3] (two children per split Gaussian), got {tuple(z.shape)}")
    if in_place:
        if not _apply_in_place(gaussians, plan, reset_stats=bool(clone or split), z=z, curr_gen=curr_gen):
            gaussians.grow(max(int(plan.n_new * 1.5), plan.n_new + 1024))        # new arrays: a captured step must be captured again
            if not _apply_in_place(gaussians, plan, reset_stats=bool(clone or split), z=z, curr_gen=curr_gen):
                raise RuntimeError(f"densify_and_prune: {plan.n_new} Gaussians do not fit the capacity {gaussians.capacity} after grow()")
        return P, plan.n_new
    _apply(gaussians, plan, reset_stats=bool(clone or split), z=z, curr_gen=curr_gen)
    return P, plan.n_new


def prune_points(gaussians, mask, during_training=True):
    """GaussianModel.prune_points (:536): drop the Gaussians where `mask` is True, carrying the optimizer moments along."""
    L = _lib.load()
    xyz = _hip(gaussians._xyz, "_xyz")
    in_place = getattr(gaussians, "capacity", None) is not None and during_training
    P, dev = (gaussians.n_active if in_place else xyz.shape[0]), xyz.device
    if P == 0:
        return 0, 0
    m = (mask if mask.dtype in (torch.bool, torch.uint8) else mask != 0)[:P].contiguous().view(torch.uint8)
    plan = _Plan(P, dev)
    with torch.cuda.device(dev):
        _lib.check(L.egs_prune_plan(P, _p(m), _p(plan.scratch), _p(plan.src), _p(plan.kind), _p(plan.split_rank), _p(plan.totals), _stream()))
    plan.finish()
    if in_place:
        _apply_in_place(gaussians, plan, reset_stats=False)
        return P, plan.n_new
    _apply(gaussians, plan, reset_stats=False, during_training=during_training)
    return P, plan.n_new


def reset_opacity(gaussians):
    """GaussianModel.reset_opacity (:484): opacity <- inverse_sigmoid(min(opacity, 0.01)), Adam moments of that group zeroed."""
    o = torch.minimum(torch.sigmoid(gaussians._opacity.detach()), torch.full_like(gaussians._opacity, 0.01))
    new = torch.log(o / (1 - o))
    opt = gaussians.optimizer
    if getattr(gaussians, "capacity", None) is not None:            # capacity-sized model: same tensors, new contents
        with torch.no_grad():
            gaussians._opacity.copy_(new)
            stored = opt.state.get(gaussians._opacity, None) if opt is not None else None
            if stored is not None and "exp_avg" in stored:
                stored["exp_avg"].zero_(); stored["exp_avg_sq"].zero_()
        return
    for group in opt.param_groups:
        if group.get("name") == "opacity":
            stored = opt.state.get(group["params"][0], None)
            p_new = nn.Parameter(new.requires_grad_(True))
            if stored is not None:
                stored["exp_avg"] = torch.zeros_like(new); stored["exp_avg_sq"] = torch.zeros_like(new)
                del opt.state[group["params"][0]]
                opt.state[p_new] = stored
            group["params"][0] = p_new
            gaussians._opacity = p_new
