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

    def load_state_dict(self, state_dict):
        """The device-side step counters and learning rates are derived state: dropped here and re-seeded from the loaded
        state["step"] / param_groups at the next step()."""
        super().load_state_dict(state_dict)
        self._aux = WeakIdKeyDictionary()

    def _aux_of(self, p):
        a = self._aux.get(p)
        if a is None:
            a = self._aux[p] = {}
        return a

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
                aux = self._aux_of(p)
                lr_t = aux.get("lr")
                if lr_t is None:
                    lr_t = aux["lr"] = torch.full((1,), float(group["lr"]), device=p.device)
                    aux["lr_host"] = float(group["lr"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                # the kernel keeps the step number in one word per workgroup of this tensor and advances them itself;
                # (re)seeded here from state["step"] when the tensor is new (densification replaced it, or a state dict was loaded)
                G = int(L.egs_adam_workgroups(p.numel()))
                ent = aux.get("counter")
                if ent is None or ent[1] != G or ent[2] != p.numel() or ent[0].device != p.device:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("FusedAdam(capturable=True): take one eager step() before capturing (device counters are created then)")
                    ent = aux["counter"] = (torch.full((max(G, 1),), int(round(float(st["step"]))), dtype=torch.int32, device=p.device), G, p.numel())
                rf = 0
                if self.active_rows is not None and p.dim() >= 1 and p.shape[0] == self.active_rows[1] and p.shape[0] > 0:
                    rf = p.numel() // p.shape[0]                  # a per-Gaussian array of the capacity-sized model
                by_cfg.setdefault((p.device, group["betas"], group["eps"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"], st["step"], lr_t, ent[0], rf))
        for (dev, betas, eps), items in by_cfg.items():
            n = len(items)
            arr = lambda k: (C.c_void_p * n)(*[t[k].data_ptr() for t in items])
            NN = (C.c_int64 * n)(*[t[0].numel() for t in items])
            RF = (C.c_int32 * n)(*[t[7] for t in items])
            skip = None if self.guard is None else C.c_void_p(self.guard.overflow.data_ptr())
            rows = None if self.active_rows is None else C.c_void_p(self.active_rows[0].data_ptr())
            with torch.cuda.device(dev):                             # the kernel itself advances the counters and writes st["step"]
                _lib.check(L.egs_adam_step_capturable(n, arr(0), arr(1), arr(2), arr(3), NN, arr(4), arr(5), arr(6), float(betas[0]),
                                                      float(betas[1]), float(eps), skip, rows, RF if rows is not None else None,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))

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
