"""FusedAdam: torch.optim.Adam semantics with every parameter of every group stepped by ONE HIP kernel.

Drop-in for the optimizer the reference builds in GaussianModel.training_setup
(/root/reference/scene/gaussian_model.py:180-198: `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one named group per
parameter).  The per-parameter state uses torch's own keys ("step", "exp_avg", "exp_avg_sq"), so the reference's
densification code, which rewrites optimizer.state entries directly (gaussian_model.py:506-560), keeps working.
"""
import ctypes as C

import torch
from torch.utils.weak import WeakIdKeyDictionary

from . import lib as _lib
from . import _hip



def _touched(p):
    """A kernel of this library wrote parameter `p` in place through its raw pointer: tell autograd (the tensor's version counter is what
    saved-tensor checks, and provenance.py's "raw parameters unchanged since the activation" test, go by)."""
    torch.autograd.graph.increment_version(p)

class AdamSink:
    """The per-Gaussian leaves of a FusedAdam(capturable=True) that ONE rasterizer backward steps itself (include/egs_raster.h,
    egs_backward_adam): built by FusedAdam.make_sink() for the tensors of one render call, consumed by that call's backward.
    `owned` holds the EGS_SINK_* ids; `struct` is the egs_adam_sink handed to the library; `ptrs` the leaves' data pointers."""

    def __init__(self, struct, owned, ptrs, keep, opt=None, params=()):
        self.struct, self.owned, self.ptrs, self._keep = struct, owned, ptrs, keep
        self.split16 = False              # True: features_dc / features_rest / positions are stepped by the spherical-harmonics launch
        self._opt, self._params = opt, tuple(params)
        self._aux = [opt._aux_of(p) for p in self._params] if opt is not None else []      # the per-parameter derived-state dicts (no weak lookup per step)
        self.keep_grads = False           # True: the backward also writes (and returns) the owned leaves' gradients -- inspection / tests

    def mark_stepped(self):
        """Called once the backward that carries this sink has been enqueued: the next optimizer.step() leaves the owned leaves
        alone even if a gradient reached them (keep_grads)."""
        if self._opt is not None:
            for p in self._params:
                self._opt._sunk[p] = bool(self.keep_grads)
                _touched(p)
            # k_adam's own per-workgroup step counters do not follow a step taken here (only state["step"] advances): the next
            # plain step() of this parameter re-seeds them.  Set on EVERY fused backward -- a cached sink (make_sink hit) after a
            # plain step() cleared the flag would otherwise leave the counters one behind.
            for a in self._aux:
                a["counter_stale"] = True

    def check(self, means3D, scales, rotations, sh, sh_rest, own_cov, colors):
        """Called by _C.rasterize_gaussians_backward with the arrays it is about to pass, BEFORE anything is enqueued: the owned leaves
        must be those arrays, and none of them may have taken a fused step already in this iteration (the sole-consumer precondition of
        egs_backward_adam, include/egs_raster.h) -- raising here leaves parameters, moments and step counts untouched."""
        if self._opt is not None:
            for p in self._params:
                if p in self._opt._sunk:
                    raise RuntimeError("FusedAdam: a parameter would take its Adam step inside a second rasterizer backward since the last "
                                       "optimizer.step() / zero_grad() (two renders with optimizer= feed one loss, or an iteration ran backward "
                                       "without step() or zero_grad()); render all but one of them without optimizer=, or call "
                                       "optimizer.step() / zero_grad() between them.  Nothing was enqueued: the state is unchanged")
        got = {_lib.SINK_MEANS3D: means3D, _lib.SINK_SCALES: scales, _lib.SINK_ROTATIONS: rotations, _lib.SINK_SH: sh, _lib.SINK_SH_REST: sh_rest}
        for leaf, t in got.items():
            if leaf in self.owned and (t is None or t.data_ptr() != self.ptrs[leaf]):
                raise RuntimeError("AdamSink: the backward received a different array than the leaf the sink was built for")
        return self.owned


class FusedAdam(torch.optim.Optimizer):
    """capturable=True keeps the step count and every group's learning rate in device scalars that the kernel reads (and, for
    the count, advances), so
    `step()` can be captured into a hipGraph and replayed while `param_groups[i]["lr"]` keeps being edited on the host
    (call `sync_lr()` before a replay to push the edits)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = capturable
        # capturable variant, per PARAMETER OBJECT (entries die with the parameter, e.g. when densification replaces it):
        #   "lr": float32[1] device copy of the group's learning rate; "counter": (int32[workgroups] device step counters of k_adam,
        #   workgroups, numel)
        self._aux = WeakIdKeyDictionary()
        self.guard = None                 # _C.StepGuard of a captured step: its overflow word makes step() a no-op for a clipped frame
        self.active_rows = None           # (int32[1] device tensor, capacity rows): only the live rows of a capacity-sized model are stepped
        self._coef = {}                   # device -> float32[12] scratch of the fused path (make_sink)
        self._arrays = {}                 # (device, betas, eps) -> cached ctypes pointer arrays of the eager step()
        self._sinks = {}                  # render call shape (tensor addresses) -> (AdamSink, [(parameter, its exp_avg)]) built for it
        # parameters a rasterizer backward stepped since the last step() -> whether that backward also wrote their gradient
        # (keep_grads); weak keys: an entry dies with a parameter that densification replaced, its id cannot be reused by another
        self._sunk = WeakIdKeyDictionary()

    def zero_grad(self, set_to_none=True):
        """torch's zero_grad walks foreach / profiler machinery (~27 us for six parameters); dropping the gradients is a loop."""
        self._sunk = WeakIdKeyDictionary()     # a new iteration: whatever the last backward stepped by itself is history (an iteration may
                                               # legitimately end without step(): a skipped one, the last one, one that steps another optimizer)
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def load_state_dict(self, state_dict):
        """The device-side step counters and learning rates are derived state: dropped here and re-seeded from the loaded
        state["step"] / param_groups at the next step()."""
        super().load_state_dict(state_dict)
        self._aux = WeakIdKeyDictionary()
        self._sinks = {}; self._arrays = {}

    def _aux_of(self, p):
        a = self._aux.get(p)
        if a is None:
            a = self._aux[p] = {}
        return a

    def _capturable_state(self, p, group):
        """(exp_avg, exp_avg_sq, step float32[1] on the device, lr float32[1] on the device) of parameter p, created on first use."""
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if not (torch.is_tensor(st.get("step")) and st["step"].is_cuda):
            st["step"] = torch.full((1,), float(st.get("step", 0)), device=p.device)
        aux = self._aux_of(p)
        lr_t = aux.get("lr")
        if lr_t is None:
            lr_t = aux["lr"] = torch.full((1,), float(group["lr"]), device=p.device)
            aux["lr_host"] = float(group["lr"])
        return st["exp_avg"], st["exp_avg_sq"], st["step"], lr_t

    def make_sink(self, means3D=None, opacities=None, scales=None, rotations=None, sh=None, sh_rest=None, cov3D_given=False,
                  colors_given=False, colors_need_grad=False):
        """An AdamSink for the tensors ONE rasterizer call is about to receive, or None when none of them can be fused.  A tensor
        is fused when it IS a parameter of this optimizer (the raw leaf, not an activation of it) that requires grad, and the
        library's conditions hold (include/egs_raster.h).  The caller vouches that this rasterizer call is the ONLY consumer of
        those leaves in the backward to come: their gradient is consumed in place, `p.grad` stays None and step() skips them
        (step() raises if a gradient reached such a leaf through a second path after all).  The one second path render() itself
        can create is excluded here: colours computed in Python from the positions (`convert_SHs_python`, an `override_color` that
        depends on xyz) arrive as `colors_precomp` that requires grad (`colors_need_grad`) -- the positions are then left to step().
        All leaves of one sink share betas / eps (one param-group configuration), as the reference's optimizer does."""
        if not self.capturable:
            return None
        # The sink of one render call shape is the same object every iteration while the tensors stay (their addresses, requires_grad
        # and the optimizer's state tensors): built once, found again by the addresses -- densification replaces the tensors and so
        # the key.  (Building it walks the parameter groups and fills a ctypes structure: ~25 us per eager step.)
        ck = tuple((0, False) if t is None else (t.data_ptr(), bool(t.requires_grad)) for t in (means3D, opacities, scales, rotations, sh, sh_rest)) + \
            (bool(cov3D_given), bool(colors_given), bool(colors_need_grad), None if means3D is None else means3D.shape[0],
             None if self.active_rows is None else self.active_rows[0].data_ptr())
        hit = self._sinks.get(ck)
        if hit is not None and all(self.state.get(p, {}).get("exp_avg") is m for p, m in hit[1]):
            if not torch.cuda.is_current_stream_capturing():
                self.sync_lr()
            return hit[0]
        by_ptr = {}
        for group in self.param_groups:
            for p in group["params"]:
                by_ptr[p.data_ptr()] = (p, group)
        P = None if means3D is None else means3D.shape[0]
        has_rest = sh_rest is not None and sh_rest.numel() != 0
        sh_dc = sh is not None and sh.numel() != 0 and sh.dim() == 3 and sh.shape[1] == 1
        sh_single = sh_dc and not has_rest                             # one coefficient: finished by the preprocess backward
        # split harmonics with 16 coefficients, 16-byte aligned: finished (with the positions) by the M = 16 spherical-harmonics launch
        sh_split16 = sh_dc and has_rest and sh_rest.dim() == 3 and sh_rest.shape[1] == 15 and sh_rest.is_contiguous() and \
            sh.data_ptr() % 16 == 0 and sh_rest.data_ptr() % 16 == 0
        colour_ok = not colors_given and (sh_single or sh_split16)
        want = {_lib.SINK_MEANS3D: (means3D, 3, (colors_given and not colors_need_grad) or sh_single or sh_split16), _lib.SINK_OPACITY: (opacities, 1, True),
                _lib.SINK_SCALES: (scales, 3, not cov3D_given), _lib.SINK_ROTATIONS: (rotations, 4, not cov3D_given),
                _lib.SINK_SH: (sh, 3, colour_ok), _lib.SINK_SH_REST: (sh_rest if has_rest else None, 45, colour_ok and sh_split16)}
        struct, owned, ptrs, keep, cfg, params = _lib.AdamSink(), set(), {}, [], None, []
        for leaf, (t, rf, allowed) in want.items():
            if t is None or not allowed or t.numel() == 0 or not t.requires_grad:
                continue
            ent = by_ptr.get(t.data_ptr())
            if ent is None:
                continue
            p, group = ent
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] == P and p.numel() == P * rf):
                continue
            this_cfg = (tuple(group["betas"]), float(group["eps"]))
            if cfg is None:
                cfg = this_cfg
            elif cfg != this_cfg:
                continue                                              # another betas / eps: left to step()
            m, v, step, lr_t = self._capturable_state(p, group)
            if not (m.is_contiguous() and v.is_contiguous()):
                continue
            f = struct.leaf[leaf]
            f.param, f.exp_avg, f.exp_avg_sq, f.lr, f.step = p.data_ptr(), m.data_ptr(), v.data_ptr(), lr_t.data_ptr(), step.data_ptr()
            owned.add(leaf); ptrs[leaf] = p.data_ptr(); keep += [p, m, v, lr_t, step]; params.append(p)
            self._aux_of(p)["counter_stale"] = True                   # k_adam's own step counters no longer follow state["step"]
        if sh_split16 and (_lib.SINK_SH in owned) != (_lib.SINK_SH_REST in owned):      # the two colour blocks go together or not at all
            for leaf in (_lib.SINK_SH, _lib.SINK_SH_REST):
                if leaf in owned:
                    owned.discard(leaf); ptrs.pop(leaf)
                    f = struct.leaf[leaf]; f.param = f.exp_avg = f.exp_avg_sq = f.lr = f.step = None
        if not owned:
            return None
        dev = means3D.device
        coef = self._coef.get(dev)
        if coef is None:
            coef = self._coef[dev] = torch.zeros(12, device=dev)
        struct.beta1, struct.beta2, struct.eps, struct.coef = float(cfg[0][0]), float(cfg[0][1]), float(cfg[1]), coef.data_ptr()
        if self.active_rows is not None and P == self.active_rows[1]:
            struct.active_rows = self.active_rows[0].data_ptr()
            keep.append(self.active_rows[0])
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        sink = AdamSink(struct, owned, ptrs, keep + [coef], self, params)
        sink.split16 = bool(sh_split16)
        if len(self._sinks) >= 8:
            self._sinks.clear()
        self._sinks[ck] = (sink, [(p, self.state[p]["exp_avg"]) for p in params])
        return sink

    def sync_lr(self):
        for group in self.param_groups:
            lr = float(group["lr"])
            for p in group["params"]:
                a = self._aux_of(p)
                t = a.get("lr")
                if t is None:
                    a["lr"] = torch.full((1,), lr, device=p.device)
                elif lr != a.get("lr_host"):
                    t.fill_(lr)
                a["lr_host"] = lr

    @torch.no_grad()
    def _step_capturable(self):
        L = _lib.load()
        sunk = self._sunk
        if len(sunk) and all(p.grad is None for group in self.param_groups for p in group["params"]):
            # every parameter that had work was stepped inside the rasterizer backward (the common eager-fast iteration): nothing to
            # launch, nothing to look up
            self._sunk = WeakIdKeyDictionary()
            return
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        by_cfg = {}
        sunk, self._sunk = self._sunk, WeakIdKeyDictionary()
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p in sunk:                                        # a rasterizer backward took this step already (make_sink)
                    if p.grad is not None and not sunk[p]:
                        raise RuntimeError("FusedAdam: a parameter whose Adam step was taken inside the rasterizer backward also received a gradient "
                                           "through another path of the loss; that gradient would be lost.  Render without optimizer= (or "
                                           "GraphedTrainStep(fuse_optimizer=False)) when the rasterizer is not the parameter's only consumer")
                    continue
                if p.grad is None:
                    continue
                self._capturable_state(p, group)
                st = self.state[p]
                aux = self._aux_of(p)
                lr_t = aux["lr"]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                # the kernel keeps the step number in one word per workgroup of this tensor and advances them itself;
                # (re)seeded here from state["step"] when the tensor is new (densification replaced it, or a state dict was loaded)
                G = int(L.egs_adam_workgroups(p.numel()))
                ent = aux.get("counter")
                if ent is None or ent[1] != G or ent[2] != p.numel() or ent[0].device != p.device:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("FusedAdam(capturable=True): take one eager step() before capturing (device counters are created then)")
                    ent = aux["counter"] = (torch.full((max(G, 1),), int(round(float(st["step"]))), dtype=torch.int32, device=p.device), G, p.numel())
                    aux["counter_stale"] = False
                if aux.get("counter_stale"):
                    # steps taken inside a rasterizer backward (make_sink) advanced state["step"] only: re-seed the words on the device
                    ent[0].copy_(st["step"].round().to(torch.int32).expand(ent[0].shape))
                    aux["counter_stale"] = False
                rf = 0
                if self.active_rows is not None and p.dim() >= 1 and p.shape[0] == self.active_rows[1] and p.shape[0] > 0:
                    rf = p.numel() // p.shape[0]                  # a per-Gaussian array of the capacity-sized model
                by_cfg.setdefault((p.device, group["betas"], group["eps"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"], st["step"], lr_t, ent[0], rf))
        for (dev, betas, eps), items in by_cfg.items():
            n = len(items)
            arr = lambda k: (C.c_void_p * n)(*[t[k].data_ptr() for t in items])
            NN = (C.c_int64 * n)(*[t[0].numel() for t in items])
            RF = (C.c_int32 * n)(*[t[7] for t in items])
            # the guard's overflow word is written by CAPTURED forwards only (egs_forward_enqueue); an eager step() follows an eager
            # render, which is never clipped, and must not be voided by what the last replay left in that word
            # (... unless the eager render itself ran without the host wait: a StepGuard(deferred=True) frame can be clipped)
            honour = self.guard is not None and (torch.cuda.is_current_stream_capturing() or getattr(self.guard, "deferred", False))
            skip = C.c_void_p(self.guard.overflow.data_ptr()) if honour else None
            rows = None if self.active_rows is None else C.c_void_p(self.active_rows[0].data_ptr())
            with _hip.device_ctx(dev):                               # the kernel itself advances the counters and writes st["step"]
                _lib.check(L.egs_adam_step_capturable(n, arr(0), arr(1), arr(2), arr(3), NN, arr(4), arr(5), arr(6), float(betas[0]),
                                                      float(betas[1]), float(eps), skip, rows, RF if rows is not None else None,
                                                      _hip.stream_of(dev)))
            for t in items:
                _touched(t[0])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.capturable:
            self._step_capturable()
            return loss
        L = _lib.load()
        by_cfg = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("FusedAdam: dense float32 parameters on a HIP device only")
                st = self.state[p]
                if len(st) == 0 or "exp_avg" not in st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if "step" not in st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                if torch.is_tensor(st["step"]):
                    st["step"] += 1                                   # in place: no new host tensor per parameter and step
                else:
                    st["step"] = st["step"] + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not (p.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    raise RuntimeError("FusedAdam: parameters and optimizer state must be contiguous")
                key = (p.device, group["betas"], group["eps"])
                by_cfg.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), int(st["step"])))
        for (dev, betas, eps), items in by_cfg.items():
            n = len(items)
            # the parameter / moment pointer arrays change only when the tensors do (densification): kept between steps
            # (the element count is part of the signature: clone / split / prune between two steps can hand a differently sized
            # tensor the addresses of an old one)
            sig = tuple((t[0].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[0].numel()) for t in items)
            cached = self._arrays.get((dev, betas, eps))
            if cached is None or cached[0] != sig:
                cached = self._arrays[(dev, betas, eps)] = (sig, (C.c_void_p * n)(*[t[0].data_ptr() for t in items]),
                                                            (C.c_void_p * n)(*[t[2].data_ptr() for t in items]),
                                                            (C.c_void_p * n)(*[t[3].data_ptr() for t in items]),
                                                            (C.c_int64 * n)(*[t[0].numel() for t in items]))
            _, PP, MM, VV, NN = cached
            GG = (C.c_void_p * n)(*[t[1].data_ptr() for t in items])
            LR = (C.c_float * n)(*[t[4] for t in items])
            ST = (C.c_int64 * n)(*[t[5] for t in items])
            with _hip.device_ctx(dev):
                _lib.check(L.egs_adam_step(n, PP, GG, MM, VV, NN, LR, ST, float(betas[0]), float(betas[1]), float(eps),
                                           _hip.stream_of(dev)))
            for t in items:
                _touched(t[0])
        return loss
