"""Python surface of the drop-in `diff_gaussian_rasterization` package.

The reference constructs these objects at /root/reference/gaussian_renderer/__init__.py:38-53 and
/root/reference/gaussian_renderer/render_helper.py:15-28,61 and calls the module at
/root/reference/gaussian_renderer/__init__.py:90-98 and render_helper.py:63 (SURVEY.md section 8a rows
a-2, a-3, a-12):

    GaussianRasterizationSettings(image_height=, image_width=, tanfovx=, tanfovy=, bg=, scale_modifier=,
                                  viewmatrix=, projmatrix=, sh_degree=, campos=, prefiltered=, debug=)
    GaussianRasterizer(raster_settings)(means3D=, means2D=, opacities=, shs=|colors_precomp=,
                                        scales=+rotations= | cov3D_precomp=) -> (color, radii, depth, alpha)

`means2D` is never read; it only receives the screen-space gradient (NDC-scaled), which the reference's
densification statistic consumes (/root/reference/scene/gaussian_model.py:735-740).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C
from . import provenance as _provenance


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                activation_flags=0, sh_rest=None, densify_stats=None, active_count=None, guard=None, optimizer=None,
                object_rotation=None, color_only=False):
        rs = raster_settings
        if guard is not None and getattr(guard, "deferred", False) and any(ctx.needs_input_grad) and not getattr(optimizer, "capturable", False) \
                and not getattr(guard, "_warned", False):
            # a deferred frame that turns out clipped is voided ON THE DEVICE -- by the statistics and the Adam step that read the overflow
            # word, i.e. a FusedAdam(capturable=True) taking its step inside this backward or launched with this guard.  Any other optimizer
            # would apply the clipped frame's gradient unnoticed.
            import warnings
            guard._warned = True
            warnings.warn("StepGuard(deferred=True) without a FusedAdam(capturable=True) behind it: a frame that exceeds the instance capacity is "
                          "only counted (guard.overflows), its clipped gradient is NOT voided -- pass optimizer=FusedAdam(..., capturable=True) "
                          "with guard set on it, or check guard.overflows before optimizer.step()", RuntimeWarning)
        ctx.object_rotation = object_rotation     # (M, selected, multiplier): constants of the loss (include/egs_raster.h egs_object_rotation)
        # optimizer (extension): a FusedAdam(capturable=True) whose leaves among THIS call's inputs take their step inside the backward
        ctx.sink = None if (optimizer is None or not any(ctx.needs_input_grad)) else optimizer.make_sink(
            means3D=means3D, opacities=opacities, scales=scales, rotations=rotations, sh=sh, sh_rest=sh_rest,
            cov3D_given=cov3Ds_precomp.numel() != 0, colors_given=colors_precomp.numel() != 0,
            colors_need_grad=bool(colors_precomp.numel() != 0 and ctx.needs_input_grad[3]))
        if sh_rest is None:
            sh_rest = torch.empty(0, device=means3D.device, dtype=torch.float32)
        num_rendered, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
            rs.campos, rs.prefiltered, rs.debug, activation_flags, sh_rest, active_count, guard, object_rotation, color_only)
        ctx.guard = guard
        ctx.egs_raster_node = True                  # fused.l1_ssim_loss(raster_prologue=True) recognises its input's grad_fn by this
        ctx.prologue_scratch = None
        ctx.loss_grad = None                        # fused.l1_ssim_loss(raster_lossgrad=True): (lib.LossGrad, tensors it points to) -- this backward's blend then computes dL/dcolour itself
        ctx.raster_settings = rs
        ctx.activation_flags = int(activation_flags)
        ctx.densify_stats = densify_stats           # (tensors updated in place by the backward; not autograd inputs)
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img,
                              alpha, sh_rest)
        # radii > 0, straight from the preprocess kernel (a bool view of the geometry buffer: read-only for the caller).  Returned
        # as a fifth output instead of through any module-level state, so concurrent renders cannot see each other's mask.
        visible = _C.visible_view(geom, means3D.shape[0])
        ctx.mark_non_differentiable(radii, visible)
        ctx.set_materialize_grads(False)        # unused depth/alpha outputs arrive as None in backward, not as zero images
        return color, radii, depth, alpha, visible

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha, grad_visible=None):
        rs = ctx.raster_settings
        try:
            colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img, alpha, sh_rest = ctx.saved_tensors
        except RuntimeError as exc:
            # A leaf this backward differentiates was written in place since the forward saved it.  With an optimizer fused into rasterizer
            # backwards that is the second of two such backwards in one iteration (the first one's Adam step moved the leaf; FusedAdam tells
            # autograd about its writes, optim._touched) -- say so in this library's words as well; autograd's own message follows.
            if ctx.sink is not None and "modified by an inplace operation" in str(exc):
                raise RuntimeError("FusedAdam: a parameter would take its Adam step inside a second rasterizer backward since the last "
                                   "optimizer.step() / zero_grad() (two renders with optimizer= feed one loss, or an iteration ran backward "
                                   "without step() or zero_grad()); render all but one of them without optimizer=, or call "
                                   "optimizer.step() / zero_grad() between them.  Nothing was enqueued by this backward: the state is as the "
                                   f"first one left it.  [autograd: {exc}]") from exc
            raise
        H, W = int(rs.image_height), int(rs.image_width)
        dev = means3D.device
        if grad_color is None:
            grad_color = torch.zeros((3, H, W), device=dev)
        if grad_depth is None:
            grad_depth = torch.empty(0, device=dev)
        if grad_alpha is None:
            grad_alpha = torch.empty(0, device=dev)
        if ctx.loss_grad is not None:
            # the loss in front handed over NO gradient tensor (fused.l1_ssim_loss(raster_lossgrad=True): the blend computes dL/dcolour itself and
            # `grad_color` is uninitialised memory nobody may read).  That is only the whole gradient of the image when nothing else contributed
            # to it: a second loss term on the image, a hook that rewrote the gradient, or accumulation would arrive here as ANOTHER tensor (or
            # a modified one) -- and would be dropped.  Refuse instead.
            ptr, version = ctx.loss_grad[2]
            if grad_color.data_ptr() != ptr or grad_color._version != version:
                ctx.loss_grad = None
                raise RuntimeError("l1_ssim_loss(raster_lossgrad=True): the rendered image has another gradient contribution (a second consumer, or a "
                                   "hook that changed the gradient) -- the in-blend loss gradient would drop it.  Use raster_lossgrad=False "
                                   "(GraphedTrainStep(loss_grad_in_blend=False)), or express a per-pixel mask as grad_gate=")
        split = sh_rest.numel() != 0
        # autograd's view of who reads which gradient, for the library (include/egs_raster.h EGS_GRAD_*): inputs order of forward()
        need = ctx.needs_input_grad
        grad_mask = ((_C.GRAD_MEANS3D if need[0] else 0) | (_C.GRAD_MEANS2D if need[1] else 0) | (_C.GRAD_SH if (need[2] or need[10]) else 0) |
                     (_C.GRAD_COLORS if need[3] else 0) | (_C.GRAD_OPACITY if need[4] else 0) | (_C.GRAD_SCALES if need[5] else 0) |
                     (_C.GRAD_ROTATIONS if need[6] else 0) | (_C.GRAD_COV3D if need[7] else 0))
        grads = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color, grad_depth, grad_alpha, sh, rs.sh_degree, rs.campos, geom,
            ctx.num_rendered, binning, img, alpha, rs.debug, ctx.activation_flags, sh_rest if split else None, ctx.densify_stats, ctx.guard,
            ctx.sink, ctx.prologue_scratch, ctx.object_rotation, grad_mask, loss_grad=None if ctx.loss_grad is None else ctx.loss_grad[0])
        ctx.prologue_scratch = None
        ctx.loss_grad = None
        (g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rots) = grads[:8]
        none_if_absent = lambda g, x: g if (g is not None and x.numel() != 0) else None
        return (g_means3D, g_means2D, none_if_absent(g_sh, sh), none_if_absent(g_colors, colors_precomp),
                g_opac, none_if_absent(g_scales, scales),
                none_if_absent(g_rots, rotations), none_if_absent(g_cov3D, cov3Ds_precomp), None, None,
                grads[8] if split else None, None, None, None, None, None, None)


def backward_prologue_of(node):
    """For the backward node of a rasterizer call that has not run yet: the egs_backward_prologue describing what that backward needs
    prepared (include/egs_raster.h), or None.  The scratch buffer it names is allocated here and handed to the node, whose backward
    then runs with prologue_done = 1.  Called by the image loss's backward, which carries the preparation in its own launch."""
    from . import lib as _lib
    import ctypes as C
    try:
        saved = node.saved_tensors
    except RuntimeError:
        return None                                   # already freed: that backward ran before
    means3D, geom, img = saved[1], saved[7], saved[9]
    P = means3D.shape[0]
    sink = node.sink
    if P == 0 or node.prologue_scratch is not None:
        return None
    rs = node.raster_settings
    scratch = torch.empty((_lib.load().egs_backward_scratch_bytes(P),), device=means3D.device, dtype=torch.uint8)
    side = _lib.BackwardPrologue()
    side.P, side.width, side.height = P, int(rs.image_width), int(rs.image_height)
    side.image_buffer, side.scratch, side.geom_buffer = img.data_ptr(), scratch.data_ptr(), geom.data_ptr()
    if sink is not None:
        side.sink = C.pointer(sink.struct)
    if node.guard is not None:
        side.skip_flag = node.guard.overflow.data_ptr()
    node.prologue_scratch = scratch
    return side


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, activation_flags=0, sh_rest=None, densify_stats=None, active_count=None, guard=None, optimizer=None,
                        object_rotation=None, color_only=False):
    """-> (color, radii, depth, alpha, visible); upstream's function returns the first four, `visible` (bool[P] = radii > 0) is an
    extension GaussianRasterizer keeps for render()."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, activation_flags, sh_rest, densify_stats, active_count, guard, optimizer,
                                     object_rotation, color_only)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        self.visible = None             # bool[P] = radii > 0 of this module's most recent forward (an extension; see forward)

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, raw_parameters=False, densify_stats=None, active_count=None, guard=None, optimizer=None,
                object_rotation=None, color_only=False):
        """Same call as upstream's.  raw_parameters=True (an extension): `scales`, `rotations` and `opacities` are the model's RAW
        parameters (log-scales, unnormalised quaternions, opacity logits); the activations run inside the preprocess kernel and
        the gradients come back w.r.t. the raw tensors (include/egs_raster.h, EGS_ACT_*).
        `shs` may also be the pair (features_dc [P,1,3], features_rest [P,M-1,3]) -- the two parameters the reference's model keeps
        (/root/reference/scene/gaussian_model.py:157-160) -- which spares the torch.cat of get_features and the split of its gradient."""
        # densify_stats (an extension): (xyz_gradient_accum, denom, max_radii2D or None) -- the backward updates the trainer's
        # densification statistics in place, in the kernel that produces the screen-space gradient (include/egs_raster.h)
        # active_count (an extension): int32[1] device tensor = live rows of a capacity-sized model; guard: a _C.StepGuard
        # object_rotation (an extension): (M [3,3], selected [P] or None, row-0 gradient multiplier) with `scales` + `rotations`: the
        # covariance of the selected rows is (M R S)(M R S)^T, built inside the rasterizer -- the reference's render(rot_cov=True,
        # accum_R, which_object) without a covariance tensor; M is a constant of the loss (include/egs_raster.h egs_object_rotation)
        # optimizer (an extension): a FusedAdam(capturable=True).  Every input of this call that IS one of its parameters takes its
        # Adam step inside the backward (its .grad stays None and optimizer.step() skips it) -- only valid when this call is the
        # sole consumer of those parameters in the backward pass (optim.FusedAdam.make_sink)
        # color_only (an extension): depth and alpha come back as None and the blend leaves their sums and planes out (a training step
        # whose loss reads the colour image only)
        # After the call `self.visible` holds radii > 0 as a bool view the preprocess kernel wrote (no compare launch); it aliases
        # state saved for the backward and, under hipGraph replay, follows every replay -- clone it to keep or edit it.
        # The reference's own render() hands over ACTIVATED tensors (opacities = get_opacity, cov3D_precomp = get_covariance(...), shs =
        # get_features).  When those are the untouched results of getters installed by adapter.attach() (provenance.py), the raw parameters
        # they were computed from are rasterized instead: same image and gradients to float rounding, without the activation / covariance /
        # concatenation backward launches and their autograd nodes.
        if cov3D_precomp is not None and scales is None and rotations is None and not raw_parameters and object_rotation is None:
            sub = _provenance.substitute(opacities, cov3D_precomp, shs, self.raster_settings.scale_modifier)
            if sub is not None:
                scales, rotations, opacities, shs, object_rotation = sub["scales"], sub["rotations"], sub["opacities"], sub["shs"], sub["object_rotation"]
                cov3D_precomp, raw_parameters = None, True
                _provenance.substitutions += 1
        shs_rest = None
        if isinstance(shs, (tuple, list)):
            shs, shs_rest = shs
            if shs_rest.shape[1] == 0:
                shs_rest = None
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("GaussianRasterizer: pass exactly one of `shs` or `colors_precomp`")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("GaussianRasterizer: pass exactly one of (`scales`, `rotations`) or `cov3D_precomp`")
        empty = torch.empty(0, device=means3D.device, dtype=torch.float32)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        if (raw_parameters or object_rotation is not None) and cov3D_precomp.numel() != 0:
            raise Exception("GaussianRasterizer: raw_parameters / object_rotation need `scales` and `rotations`, not `cov3D_precomp`")
        color, radii, depth, alpha, self.visible = rasterize_gaussians(
            means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, self.raster_settings,
            _C.ACT_RAW_PARAMETERS if raw_parameters else 0, shs_rest, densify_stats, active_count, guard, optimizer, object_rotation, color_only)
        return color, radii, depth, alpha
