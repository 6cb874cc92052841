"""attach(): one call that points a reference-shaped GaussianModel at this package's fast paths.

The reference's model (/root/reference/scene/gaussian_model.py:125-200) activates its parameters with PyTorch ops on every render
(`get_scaling` exp, `get_rotation` normalize, `get_opacity` sigmoid, :36-44), builds the covariance with eight more
(`build_covariance_from_scaling_rotation`, :29-33 / `..._w_rot`, :46-63), concatenates the colour coefficients (`get_features`,
:157-160) and steps `torch.optim.Adam` over six parameter groups (:180-198).  Swapping the two import lines already runs the HIP
rasterizer under those ops (the `reference_shaped_step` of bench.py: 0.3 ms of GPU work inside 3.3 ms of host time); this call
removes the ops themselves, without editing the model's class:

  for the reference's OWN render() (gaussian_renderer/__init__.py:64-71 calls `pc.get_covariance` / `pc.get_rotated_covariance`):
    * `gaussians.covariance_activation`                            -> fused.covariance_from_scaling_rotation   (one launch each way)
    * `gaussians.covariance_activation_w_rot` (what `get_rotated_covariance` calls, gaussian_model.py:39,171) and
      `gaussians.build_covariance_from_scaling_rotation_w_rot`     -> the fused object-rotated producer, same arguments
  for this package's render() (egogaussian_amd.renderer.render, same signature), which looks for optional hooks on the model:
    * `get_raw_parameters()`, `get_features_split()`                raw parameters straight into the rasterizer: no activation, covariance
                                                                    or concatenation launches at all
    * `get_covariance_and_opacity()`, `get_raw_parameters_rotated()`, `get_rotated_covariance_and_opacity()`
  optimizer:
    * `gaussians.optimizer` (torch.optim.Adam)                      -> optim.FusedAdam over the same groups, state carried over: one launch
                                                                    per step; the reference's densification keeps editing
                                                                    `optimizer.state` / `param_groups` as before (torch's state keys)

Everything reads the model's attributes at call time, so densification (which replaces the Parameters) needs no re-attach of the
hooks; call `attach` again after `training_setup()` / `restore()` built a new torch optimizer.
"""
import torch

from . import fused
from .optim import FusedAdam

_PARAMS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _on_hip(g):
    return g._xyz.is_cuda


def _trainable_rotation(g, during_training):
    """3x3 matrix of gaussians.trainable_object_move (utils/geometry_utils.py ObjectMove) when it takes part, else None.  rot_L is
    linear: rot_L(I) IS the matrix, gradient to obj_rotation_6d included -- no assumption about the object's other methods."""
    tom = getattr(g, "trainable_object_move", None) if during_training else None
    if tom is None:
        return None
    return tom.rot_L(torch.eye(3, device=g._xyz.device))


def _selection(g, which_object):
    n_live = getattr(g, "n_active", None)
    io = g.get_is_object if hasattr(g, "get_is_object") else g._is_object
    key = (which_object, io.data_ptr(), io._version, tuple(io.shape), n_live)
    cache = g.__dict__.setdefault("_egs_selection", {})
    if cache.get("key") != key:
        cache["key"], cache["value"] = key, fused.object_selection(io, which_object, g._xyz.shape[0], n_live)
    return cache["value"]


def attach(gaussians, optimizer=True, capturable=False, fuse_optimizer=False, provenance=True):
    """Install the hooks described in the module docstring on `gaussians` (any object with the reference's attribute names:
    _xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity, _is_object, [trainable_object_move], [optimizer]).
    optimizer=True      replace a torch.optim.Adam in `gaussians.optimizer` by FusedAdam (same groups, state carried over);
    capturable=True     ... as FusedAdam(capturable=True): step counts and learning rates on the device (graph.GraphedTrainStep);
    fuse_optimizer=True this package's render() hands that optimizer to the rasterizer, whose backward then takes the Adam step of
                        the parameters it differentiates (no gradient arrays).  ONLY for trainers whose loss reaches the model
                        through that one render -- e.g. not while the entropy term of train_static.py:97-102 is active; FusedAdam
                        raises if a second gradient path shows up.  Needs capturable=True.
    provenance=True     the activated tensors the model's getters return remember their raw parameters, so that the rasterizer can take
                        those instead when the reference's own render() hands it the activated ones (see below); False: plain tensors.
    Returns `gaussians`."""
    g = gaussians
    missing = [a for a in _PARAMS if not hasattr(g, a)]
    if missing:
        raise TypeError(f"attach: not a reference-shaped Gaussian model (no {', '.join(missing)})")
    if fuse_optimizer and not capturable:
        raise ValueError("fuse_optimizer=True needs capturable=True (the in-backward Adam step reads its step count and learning rates on the device)")

    # ---- the reference's own render(): activations that remember where their result came from, covariance producers ----
    # The reference's render() hands the rasterizer ACTIVATED tensors (get_opacity, get_covariance(get_scaling, ., _rotation), get_features:
    # gaussian_renderer/__init__.py:56-82).  Each getter result made here carries `_egs_origin` = which raw parameter(s) it was computed from
    # (a Python attribute of that very tensor object).  When the rasterizer is handed exactly these objects, unmodified, with the raw
    # parameters unmodified since (provenance.py checks versions), it takes the RAW parameters instead -- activations and covariance inside its
    # preprocess kernel, gradients straight to the leaves -- and the activation / covariance / concatenation backward launches and their
    # autograd nodes never run.  Anyone else who reads the tensors gets ordinary tensors with ordinary autograd history.
    from . import provenance as _prov
    if provenance:
        g.scaling_activation = _prov.tagging_activation(torch.exp, "scaling")                  # gaussian_model.py:36 (torch.exp)
        g.opacity_activation = _prov.tagging_activation(torch.sigmoid, "opacity")              # :40 (torch.sigmoid)
        # :44 (normalize: three launches).  Nothing substitutes a "rotation" tag; the point is the memo -- the label render asks for
        # get_rotation every iteration of a phase that does not move the rotations (render_helper.py:38-54, train_static.py:105-109)
        g.rotation_activation = _prov.tagging_activation(torch.nn.functional.normalize, "rotation")
        g._egs_tag_features = True                                                             # get_features (:157-160) is a property of the class: patching.install wraps it

    def covariance_activation(scaling, scaling_modifier, rotation):
        if scaling.is_cuda:
            cov = fused.covariance_from_scaling_rotation(scaling, scaling_modifier, rotation)
        else:
            from .covariance import covariance_from_scaling_rotation
            cov = covariance_from_scaling_rotation(scaling, scaling_modifier, rotation)
        if provenance:
            _prov.tag_covariance(cov, scaling, scaling_modifier, rotation)
        return cov

    def build_covariance_w_rot(scaling, scaling_modifier, rotation, accum_R, which_object=None, during_training=False):
        build_covariance_w_rot.calls += 1
        if scaling.is_cuda:
            trot = _trainable_rotation(g, during_training)
            cov = fused.rotated_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation, accum_R, g._is_object, which_object,
                                                                 trot, selection=_selection(g, which_object))
            if provenance and trot is None and not (accum_R is not None and accum_R.requires_grad):
                # (a rotation that is being trained needs the covariance path: its gradient; same rule as get_raw_parameters_rotated below)
                M = torch.eye(3, device=g._xyz.device) if accum_R is None else accum_R.to(g._xyz.device, torch.float32)
                sel, mult = _selection(g, which_object)
                _prov.tag_covariance(cov, scaling, scaling_modifier, rotation, object_rotation=(M, sel, mult))
            return cov
        from .covariance import rotated_covariance_from_scaling_rotation
        tom = getattr(g, "trainable_object_move", None) if during_training else None
        return rotated_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation, accum_R, g._is_object, which_object,
                                                        None if tom is None else tom.rot_L)
    build_covariance_w_rot.calls = 0                                # how often the installed producer ran (tests spy on it)
    g.covariance_activation = covariance_activation
    # get_rotated_covariance (gaussian_model.py:170-171) calls `covariance_activation_w_rot`, the bound method setup_functions()
    # captured at __init__ (gaussian_model.py:39) -- that attribute is the one render(rot_cov=True) reaches; the method name is
    # replaced too for callers that go to it directly.
    g.covariance_activation_w_rot = build_covariance_w_rot
    g.build_covariance_from_scaling_rotation_w_rot = build_covariance_w_rot

    # ---- this package's render(): raw parameters straight into the rasterizer ----
    g.get_raw_parameters = lambda: (g._scaling, g._rotation, g._opacity) if _on_hip(g) else None
    g.get_features_split = lambda: (g._features_dc, g._features_rest) if _on_hip(g) else None

    def get_covariance_and_opacity(scaling_modifier=1):
        if _on_hip(g):
            return fused.covariance_and_opacity(g._scaling, scaling_modifier, g._rotation, g._opacity)
        return g.get_covariance(scaling_modifier), g.get_opacity

    def get_raw_parameters_rotated(accum_R, which_object, during_training):
        if not _on_hip(g) or _trainable_rotation(g, during_training) is not None:
            return None                                            # a rotation that is being trained needs the covariance path (its gradient)
        if accum_R is not None and accum_R.requires_grad:
            return None
        # the reference's default is a CPU eye(3) and it moves accum_R itself (gaussian_model.py:54-58): tolerate a CPU matrix
        M = torch.eye(3, device=g._xyz.device) if accum_R is None else accum_R.to(g._xyz.device, torch.float32)
        sel, mult = _selection(g, which_object)
        return g._scaling, g._rotation, g._opacity, (M, sel, mult)

    def get_rotated_covariance_and_opacity(accum_R, which_object, during_training, scaling_modifier=1):
        if _on_hip(g):
            return fused.rotated_covariance_from_scaling_rotation(g._scaling, scaling_modifier, g._rotation, accum_R, g._is_object, which_object,
                                                                  _trainable_rotation(g, during_training), scaling_is_log=True,
                                                                  selection=_selection(g, which_object), opacity_raw=g._opacity)
        return g.get_rotated_covariance(accum_R, which_object, during_training, scaling_modifier), g.get_opacity
    g.get_covariance_and_opacity = get_covariance_and_opacity
    g.get_raw_parameters_rotated = get_raw_parameters_rotated
    g.get_rotated_covariance_and_opacity = get_rotated_covariance_and_opacity

    # ---- optimizer ----
    old = getattr(g, "optimizer", None)
    if optimizer and old is not None and not isinstance(old, FusedAdam):
        if not isinstance(old, torch.optim.Adam):
            raise TypeError(f"attach: gaussians.optimizer is {type(old).__name__}; only torch.optim.Adam is replaced")
        for group in old.param_groups:
            if group.get("amsgrad") or group.get("weight_decay") or group.get("maximize"):
                raise ValueError("attach: FusedAdam implements Adam with weight_decay = 0, amsgrad = False, maximize = False (the reference's settings)")
        keep = ("params", "lr", "betas", "eps", "name")
        new = FusedAdam([{k: v for k, v in group.items() if k in keep} for group in old.param_groups], lr=0.0, eps=1e-15, capturable=capturable)
        for group in old.param_groups:
            for p in group["params"]:
                st = old.state.get(p)
                if st:
                    new.state[p] = dict(st)                        # torch's keys: "step", "exp_avg", "exp_avg_sq"
        g.optimizer = new
    g._egs_fused_optimizer = g.optimizer if (fuse_optimizer and isinstance(getattr(g, "optimizer", None), FusedAdam)) else None
    return g
