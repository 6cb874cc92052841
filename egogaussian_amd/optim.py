"""FusedAdam: torch.optim.Adam semantics with every parameter of every group stepped by ONE HIP kernel.

Drop-in for the optimizer the reference builds in GaussianModel.training_setup
(/root/reference/scene/gaussian_model.py:180-198: `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one named group per
parameter).  The per-parameter state uses torch's own keys ("step", "exp_avg", "exp_avg_sq"), so the reference's
densification code, which rewrites optimizer.state entries directly (gaussian_model.py:506-560), keeps working.
"""
import ctypes as C

import torch

from . import lib as _lib


class FusedAdam(torch.optim.Optimizer):
    """capturable=True keeps the step count and every group's learning rate in device scalars that the kernel reads (and, for
    the count, advances), so
    `step()` can be captured into a hipGraph and replayed while `param_groups[i]["lr"]` keeps being edited on the host
    (call `sync_lr()` before a replay to push the edits)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = capturable
        self._lr_dev = {}
        self._counter = {}                # (group index, index in group) -> (int32[workgroups] device counters, workgroups, numel); see k_adam

    def sync_lr(self):
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                t = self._lr_dev.get((gi, id(p)))
                if t is None:
                    self._lr_dev[(gi, id(p))] = torch.full((1,), float(group["lr"]), device=p.device)
                elif float(group["lr"]) != getattr(t, "_host_value", None):
                    t.fill_(float(group["lr"]))
                self._lr_dev[(gi, id(p))]._host_value = float(group["lr"])

    @torch.no_grad()
    def _step_capturable(self):
        L = _lib.load()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        by_cfg = {}
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not (torch.is_tensor(st.get("step")) and st["step"].is_cuda):
                    st["step"] = torch.full((1,), float(st.get("step", 0)), device=p.device)
                lr_t = self._lr_dev.get((gi, id(p)))
                if lr_t is None:
                    lr_t = self._lr_dev[(gi, id(p))] = torch.full((1,), float(group["lr"]), device=p.device)
                    lr_t._host_value = float(group["lr"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                # the kernel keeps the step number in one word per workgroup of this tensor and advances them itself;
                # (re)seeded here from state["step"] when the tensor is new or its size changed (densification)
                key, G = (gi, group["params"].index(p) if len(group["params"]) > 1 else 0), int(L.egs_adam_workgroups(p.numel()))
                ent = self._counter.get(key)
                if ent is None or ent[1] != G or ent[2] != p.numel() or ent[0].device != p.device:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("FusedAdam(capturable=True): take one eager step() before capturing (device counters are created then)")
                    ent = (torch.full((max(G, 1),), int(round(float(st["step"]))), dtype=torch.int32, device=p.device), G, p.numel())
                    self._counter[key] = ent
                by_cfg.setdefault((p.device, group["betas"], group["eps"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"], st["step"], lr_t, ent[0]))
        for (dev, betas, eps), items in by_cfg.items():
            n = len(items)
            arr = lambda k: (C.c_void_p * n)(*[t[k].data_ptr() for t in items])
            NN = (C.c_int64 * n)(*[t[0].numel() for t in items])
            with torch.cuda.device(dev):                             # the kernel itself advances the counters and writes st["step"]
                _lib.check(L.egs_adam_step_capturable(n, arr(0), arr(1), arr(2), arr(3), NN, arr(4), arr(5), arr(6), float(betas[0]),
                                                      float(betas[1]), float(eps), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

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
            PP = (C.c_void_p * n)(*[t[0].data_ptr() for t in items])
            GG = (C.c_void_p * n)(*[t[1].data_ptr() for t in items])
            MM = (C.c_void_p * n)(*[t[2].data_ptr() for t in items])
            VV = (C.c_void_p * n)(*[t[3].data_ptr() for t in items])
            NN = (C.c_int64 * n)(*[t[0].numel() for t in items])
            LR = (C.c_float * n)(*[t[4] for t in items])
            ST = (C.c_int64 * n)(*[t[5] for t in items])
            with torch.cuda.device(dev):
                _lib.check(L.egs_adam_step(n, PP, GG, MM, VV, NN, LR, ST, float(betas[0]), float(betas[1]), float(eps),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return loss
