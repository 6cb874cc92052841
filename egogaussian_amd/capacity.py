"""Capacity-sized Gaussian model: densify / prune without reallocation.

The reference's densification (/root/reference/scene/gaussian_model.py:565-586 densification_postfix, :678-709
densify_and_prune) concatenates new rows onto every parameter, builds new nn.Parameter objects and rewrites the optimizer
state around them -- every 100 iterations the tensors behind a training step change address and size.  For a step that is
captured once into a hipGraph and replayed (graph.py) that means a re-capture each time.

Here every per-Gaussian array (parameters, Adam moments, densification statistics, generation / is_object tags) is
allocated ONCE with `capacity` rows.  The number of live Gaussians is a host int (`n_active`) mirrored in a device word
(`active_count`, int32[1]) that the kernels read (include/egs_raster.h, "capacity-sized models"):
  * the rasterizer's preprocess kernel treats rows >= active_count as culled (radii 0, no instances, zero gradients);
  * FusedAdam(capturable) steps only the live rows (`optimizer.active_rows`);
  * densify.densify_and_prune / prune_points / reset_opacity, given such a model, plan on the live prefix, gather into
    temporaries and copy the result back INTO the same arrays, then update the two counts.
P, every launch size, every buffer layout and every tensor address stay fixed, so a captured step survives densification
with no re-capture as long as the new count fits the capacity (otherwise `grow()` reallocates, and the step must be
captured again).  The getters return the full capacity-sized tensors: outputs of render() (`radii`, `visibility_filter`,
`viewspace_points.grad`) have `capacity` rows, dead rows being zero / False.
"""
import numpy as np
import torch

from .scene_synth import SynthGaussians

_PAD = {"xyz": 0.0, "log_scale": 0.0, "opacity_logit": 0.0, "features": 0.0}


def _pad_rows(a, rows, quat=False):
    out = np.zeros((rows,) + a.shape[1:], dtype=a.dtype)
    if quat:
        out[:, 0] = 1.0
    out[:a.shape[0]] = a
    return out


class CapacityGaussians(SynthGaussians):
    """SynthGaussians whose arrays have `capacity` rows, the first `n_active` of them live."""

    def __init__(self, scene, capacity, device="cuda", sh_degree=0, requires_grad=True, fused=True):
        n = scene["xyz"].shape[0]
        if capacity < n:
            raise ValueError(f"capacity {capacity} < {n} Gaussians")
        padded = {k: _pad_rows(v, capacity, quat=(k == "quat")) for k, v in scene.items()}
        super().__init__(padded, device=device, sh_degree=sh_degree, requires_grad=requires_grad, fused=fused)
        self.capacity = int(capacity)
        self.n_active = int(n)
        self.active_count = torch.tensor([n], dtype=torch.int32, device=device)

    def set_active(self, n):
        """New live row count (host int + the device word the kernels read; one tiny fill, outside any capture)."""
        if not 0 <= n <= self.capacity:
            raise ValueError(f"{n} live rows do not fit the capacity {self.capacity}")
        self.n_active = int(n)
        self.active_count.fill_(int(n))

    def training_setup(self, optimizer_cls=None, percent_dense=0.01, **kw):
        opt = super().training_setup(optimizer_cls, percent_dense, **kw)
        if hasattr(opt, "active_rows"):
            opt.active_rows = (self.active_count, self.capacity)
        return opt

    def live(self, t):
        """The live prefix of a per-Gaussian array (a view)."""
        return t[:self.n_active]

    def grow(self, capacity):
        """Reallocate every array with a larger capacity (a captured step must be captured again afterwards).  Optimizer state
        is carried over.  Returns the new capacity."""
        from torch import nn
        if capacity <= self.capacity:
            return self.capacity
        dev = self._xyz.device

        def bigger(t, quat=False):
            out = torch.zeros((capacity,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            if quat:
                out[:, 0] = 1
            out[:t.shape[0]] = t.detach()
            return out

        names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
                 "rotation": "_rotation", "label": "_label"}
        new = {k: bigger(getattr(self, a), quat=(k == "rotation")) for k, a in names.items()}
        opt = self.optimizer
        if opt is not None:
            for group in opt.param_groups:
                name = group.get("name")
                if name not in new:
                    continue
                p_old = group["params"][0]
                st = opt.state.pop(p_old, None)
                p_new = nn.Parameter(new[name].requires_grad_(True))
                if st is not None:
                    st["exp_avg"], st["exp_avg_sq"] = bigger(st["exp_avg"]), bigger(st["exp_avg_sq"])
                    opt.state[p_new] = st
                group["params"][0] = p_new
                new[name] = p_new
        for k, a in names.items():
            v = new[k]
            setattr(self, a, v if isinstance(v, torch.nn.Parameter) or not getattr(self, a).requires_grad else v.requires_grad_(True))
        for a in ("_is_object", "_generation", "max_radii2D", "xyz_gradient_accum", "denom"):
            setattr(self, a, bigger(getattr(self, a)))
        self.capacity = int(capacity)
        self.model_version = getattr(self, "model_version", 0) + 1     # the arrays a captured step points at are gone: graph.py checks this
        if opt is not None and hasattr(opt, "active_rows"):
            opt.active_rows = (self.active_count, self.capacity)
        return self.capacity
