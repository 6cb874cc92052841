"""Host-side mirror of the reference's render boundary: render() and get_render_label().

Same names, arguments, argument meaning and return dict as /root/reference/gaussian_renderer/__init__.py:18-107
and /root/reference/gaussian_renderer/render_helper.py:7-64, written
against this package's rasterizer so a trainer can import it in place of `gaussian_renderer`.  Tensors are
created on the point cloud's own device (the reference hard-codes "cuda").
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def get_raster_settings(viewpoint_camera, pc, bg_color, scaling_modifier=1.0):
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)


_zero_points = {}


def _screenspace_leaf(xyz):
    """Fresh autograd leaf of zeros shaped like `xyz` whose only job is to collect d loss / d mean2D in `.grad`
    (/root/reference/gaussian_renderer/__init__.py:27-31 allocates and zero-fills one per call).  Nothing ever writes
    its values, so every call's leaf aliases one cached block of zeros per (device, shape): no fill kernel per render."""
    key = (xyz.device, tuple(xyz.shape), xyz.dtype)
    base = _zero_points.get(key)
    if base is None:
        _zero_points.clear()                                  # one live shape at a time (the count changes at densification)
        base = _zero_points[key] = torch.zeros_like(xyz, requires_grad=False)
    return base.detach().requires_grad_(True)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, rot_cov=False,
           accum_R=None, which_object=None, during_training=False, fused_densify_stats=False, guard=None, optimizer=None, color_only=False):
    """Extensions (defaults = the reference's behaviour):
    fused_densify_stats  the backward of this render also updates pc.xyz_gradient_accum, pc.denom and pc.max_radii2D in place
                         (the trainer then skips add_densification_stats / the max_radii2D update for this iteration);
    guard                a _C.StepGuard: device words a captured training step uses to void an overflowed frame (graph.py);
    optimizer            a FusedAdam(capturable=True): the model parameters this render hands to the rasterizer as they are stored
                         (xyz, and with the raw-parameter hooks scaling / rotation / opacity / features_dc) take their Adam step
                         inside the rasterizer's backward -- valid when this render is their only use in the iteration's loss;
    color_only           "depth" and "alpha" of the result are None and the blend does not compute them (GraphedTrainStep: the loss reads
                         the colour image only);
    a model with an `active_count` attribute (int32[1] device tensor; capacity.CapacityGaussians) renders only its live rows.
    `visibility_filter` is radii > 0 as written by the preprocess kernel: a fresh tensor in eager calls; while a hipGraph is being
    captured it is a VIEW of the rasterizer's saved state that follows every replay (no launch) -- clone it to keep or edit it."""
    xyz = pc.get_xyz
    if optimizer is None:
        optimizer = getattr(pc, "_egs_fused_optimizer", None)      # adapter.attach(..., fuse_optimizer=True)
    screenspace_points = _screenspace_leaf(xyz)
    rasterizer = GaussianRasterizer(raster_settings=get_raster_settings(viewpoint_camera, pc, bg_color, scaling_modifier))

    scales = rotations = cov3D_precomp = opacity = None
    raw = False
    object_rotation = None
    if pipe.compute_cov3D_python:
        rotated_raw = pc.get_raw_parameters_rotated(accum_R, which_object, during_training) \
            if (rot_cov and getattr(pc, "get_raw_parameters_rotated", None) is not None) else None
        if rotated_raw is not None:
            # optional hook: the RAW parameters plus (M, selected rows, row-0 multiplier) -- the rasterizer builds the object-rotated
            # covariance itself (rasterizer.py object_rotation): no covariance tensor, no producer launches either way
            scales, rotations, opacity, object_rotation = rotated_raw
            raw = True
        elif rot_cov:
            if getattr(pc, "get_rotated_covariance_and_opacity", None) is not None:
                # optional fused producer: rotated covariance and activated opacity from the raw parameters in one launch
                cov3D_precomp, opacity = pc.get_rotated_covariance_and_opacity(accum_R, which_object, during_training, scaling_modifier)
            else:
                cov3D_precomp = pc.get_rotated_covariance(accum_R, which_object, during_training, scaling_modifier)
        elif getattr(pc, "get_raw_parameters", None) is not None and pc.get_raw_parameters() is not None:
            # optional hook: hand the RAW parameters to the rasterizer, which applies exp / normalize / sigmoid itself and builds
            # the covariance in its preprocess kernel -- no activation or covariance launches at all (rasterizer.py)
            scales, rotations, opacity = pc.get_raw_parameters()
            raw = True
        elif getattr(pc, "get_covariance_and_opacity", None) is not None:
            # optional fused producer (fused.covariance_and_opacity): covariance and activated opacity from one launch
            cov3D_precomp, opacity = pc.get_covariance_and_opacity(scaling_modifier)
        else:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif pipe.convert_SHs_python:
        from .sh import eval_sh
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dirs = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dirs) + 0.5, 0.0)
    elif getattr(pc, "get_features_split", None) is not None and pc.get_features_split() is not None:
        shs = pc.get_features_split()             # optional hook: (features_dc, features_rest) as they are stored -- no torch.cat per render
    else:
        shs = pc.get_features

    image, radii, depth, alpha = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs,
                                            colors_precomp=colors_precomp, opacities=pc.get_opacity if opacity is None else opacity, scales=scales,
                                            rotations=rotations, cov3D_precomp=cov3D_precomp, **({"raw_parameters": True} if raw else {}),
                                            **({"densify_stats": (pc.xyz_gradient_accum, pc.denom, pc.max_radii2D)} if fused_densify_stats else {}),
                                            **({"active_count": pc.active_count} if getattr(pc, "active_count", None) is not None else {}),
                                            **({"guard": guard} if guard is not None else {}),
                                            **({"optimizer": optimizer} if optimizer is not None else {}),
                                            **({"object_rotation": object_rotation} if object_rotation is not None else {}),
                                            **({"color_only": True} if color_only else {}))
    visible = rasterizer.visible                           # radii > 0 from the preprocess kernel of THIS call (returned, not shared state)
    if visible is None:
        visible = radii > 0
    elif not torch.cuda.is_current_stream_capturing():
        visible = visible.clone()                          # eager: the caller owns it, as with the reference's `radii > 0`
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": visible, "radii": radii,
            "depth": depth, "alpha": alpha}


def get_pts_label_as_rgb(gaussians_label):
    return gaussians_label.reshape(-1, 1).float().expand(-1, 3).contiguous()


def gaussians_to_label_rendervar(gaussians):
    xyz = gaussians.get_xyz.detach()
    return {"means3D": xyz, "colors_precomp": get_pts_label_as_rgb(gaussians.get_label),
            "rotations": gaussians.get_rotation.detach(), "opacities": gaussians.get_opacity.detach(),
            "scales": gaussians.get_scaling.detach(), "means2D": _screenspace_leaf(xyz).detach()}      # (the reference: zeros_like(xyz) + 0, two launches; nothing reads the values)


def get_render_label(viewpoint_camera, pc, bg_color):
    """Per-Gaussian scalar label rendered as a grey colour through a fresh rasterizer (object segmentation)."""
    renderer = GaussianRasterizer(get_raster_settings(viewpoint_camera, pc, bg_color))
    label, _, _, _ = renderer(**gaussians_to_label_rendervar(pc))
    return label
