"""Autograd wrappers of the fused HIP ops on the "next" rows of SURVEY.md section 8f, bound through the same C ABI:

  covariance_from_scaling_rotation / rotated_covariance_from_scaling_rotation   (row f-1)
      drop-in for the reference's covariance activations (/root/reference/scene/gaussian_model.py:29-33,46-63).  A
      reference GaussianModel can be pointed at them after construction:
          gaussians.covariance_activation = egogaussian_amd.fused.covariance_from_scaling_rotation
  l1_ssim_loss   (row f-3)
      (1 - lambda) * l1_loss + lambda * (1 - ssim), /root/reference/trainers/train_static.py:92-95.

HIP device tensors only.  Their oracles are the PyTorch versions in covariance.py / losses.py, which are pinned by
fixtures captured from the reference (tests/golden/covariance.npz, losses.npz).
"""
import ctypes as C

import torch

from . import lib as _lib
from . import _hip


def _p(t):
    return None if t is None else t.data_ptr()


_stream = _hip.stream_of           # (device) -> c_void_p of its current stream


def _need_hip(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: fused HIP op has no CPU path (use egogaussian_amd.covariance / losses)")
    return t.float().contiguous()


class _Cov3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, rotation, M, selected, modifier, row0_mult, log_scaling=False, opacity_raw=None):
        L = _lib.load()
        scaling, rotation = _need_hip(scaling, "scaling"), _need_hip(rotation, "rotation")
        N = scaling.shape[0]
        Mc = None if M is None else _need_hip(M, "M").reshape(9)
        sel = None if selected is None else selected.to(torch.uint8).contiguous()
        cov = torch.empty((N, 6), device=scaling.device, dtype=torch.float32)
        o_raw = None if opacity_raw is None else _need_hip(opacity_raw, "opacity_raw")
        opacity = None if o_raw is None else torch.empty_like(o_raw)
        with _hip.device_ctx(scaling.device):
            _lib.check(L.egs_cov3d_forward(N, _p(scaling), int(bool(log_scaling)), float(modifier), _p(rotation), _p(Mc), _p(sel), _p(cov),
                                           _p(o_raw), _p(opacity), _stream(scaling.device)))
        empty = torch.empty(0, device=scaling.device)
        ctx.save_for_backward(scaling, rotation, Mc if Mc is not None else empty, sel if sel is not None else empty,
                              opacity if opacity is not None else empty)
        mult_dev = row0_mult if torch.is_tensor(row0_mult) else None          # device float[1]: no host read of the selection count
        ctx.mult_dev = None if mult_dev is None else mult_dev.detach().float().reshape(1).contiguous()
        ctx.modifier, ctx.row0_mult, ctx.has_M, ctx.has_sel = float(modifier), (1.0 if mult_dev is not None else float(row0_mult)), M is not None, selected is not None
        ctx.log_scaling, ctx.has_opacity = int(bool(log_scaling)), opacity is not None
        return cov if opacity is None else (cov, opacity)

    @staticmethod
    def backward(ctx, dcov, dopacity=None):
        L = _lib.load()
        scaling, rotation, Mc, sel, opacity = ctx.saved_tensors
        Mc = Mc if ctx.has_M else None
        sel = sel if ctx.has_sel else None
        N = scaling.shape[0]
        dcov = torch.zeros((N, 6), device=scaling.device) if dcov is None else dcov.float().contiguous()
        ds, dr = torch.empty_like(scaling), torch.empty_like(rotation)
        dM = torch.empty(9, device=scaling.device) if (ctx.has_M and ctx.needs_input_grad[2]) else None
        dM_scratch = torch.empty(L.egs_cov3d_dm_scratch_floats(N), device=scaling.device) if dM is not None else None
        o = do = do_raw = None
        if ctx.has_opacity:
            o = opacity
            do = torch.zeros_like(o) if dopacity is None else dopacity.float().contiguous()
            do_raw = torch.empty_like(o)
        with _hip.device_ctx(scaling.device):
            _lib.check(L.egs_cov3d_backward(N, _p(scaling), ctx.log_scaling, ctx.modifier, _p(rotation), _p(Mc), _p(sel), ctx.row0_mult,
                                            _p(ctx.mult_dev), _p(dcov), _p(ds), _p(dr), _p(dM), _p(dM_scratch), _p(o), _p(do), _p(do_raw), _stream(scaling.device)))
        return ds, dr, (None if dM is None else dM.view(3, 3)), None, None, None, None, do_raw


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    return _Cov3D.apply(scaling, rotation, None, None, scaling_modifier, 1.0)


def covariance_from_log_scaling(log_scaling, scaling_modifier, rotation):
    """Same, fed with the RAW scaling parameters (GaussianModel._scaling): exp() and its derivative run inside the kernels,
    which saves the two elementwise launches of `get_scaling` per step.  For a reference GaussianModel:
        gaussians.get_covariance = lambda m=1: fused.covariance_from_log_scaling(gaussians._scaling, m, gaussians._rotation)"""
    return _Cov3D.apply(log_scaling, rotation, None, None, scaling_modifier, 1.0, True)


def covariance_and_opacity(log_scaling, scaling_modifier, rotation, opacity_raw):
    """(cov3D [N,6], opacity [N,1]) from the raw parameters in ONE launch each way: covariance_from_log_scaling plus the opacity
    activation sigmoid (gaussian_model.py:40) and its derivative.  render() uses it when the model offers
    `get_covariance_and_opacity(scaling_modifier)`:
        gaussians.get_covariance_and_opacity = lambda m=1: fused.covariance_and_opacity(gaussians._scaling, m, gaussians._rotation, gaussians._opacity)"""
    return _Cov3D.apply(log_scaling, rotation, None, None, scaling_modifier, 1.0, True, opacity_raw)


def object_selection(is_object, which_object, n, n_live=None):
    """(selected uint8[N] or None, row-0 gradient multiplier: python float or float32[1] device tensor) for the rows the reference's
    build_covariance_from_scaling_rotation_w_rot rotates -- including its [N,1]-index quirk (covariance.py): Gaussian 0 is rotated
    too whenever any Gaussian is selected, and its gradient is multiplied by (count + [0 selected]).  Evaluated on the device, no
    host read.  The result depends only on (is_object, which_object, n_live): callers may keep it across steps.
    n_live: a capacity-sized model's live row count -- rows beyond it are not Gaussians: never selected, never counted (the
    reference's model has exactly n_live rows)."""
    sel, mult = None, 1.0
    if n_live is not None and n_live < n:
        n_eff = int(n_live)
    else:
        n_eff = n
    if which_object is not None and is_object is not None:
        sel = (is_object.reshape(-1) == which_object)
        if n_eff < n:
            sel = sel.clone(); sel[n_eff:] = False
        if is_object.dim() == 2 and n_eff > 0:
            cnt = sel.sum()
            mult = (cnt + sel[0]).to(torch.float32).reshape(1)
            first = torch.logical_or(sel[0:1], (cnt > 0).reshape(1))
            sel = sel.clone(); sel[0:1] = first
        sel = sel.to(torch.uint8).contiguous()
    elif is_object is not None and is_object.dim() == 2 and n_eff > 0:
        mult = float(n_eff + 1)
    return sel, mult


def rotated_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation, accum_R, is_object=None, which_object=None,
                                             rot_matrix=None, scaling_is_log=False, selection=None, opacity_raw=None):
    """`rot_matrix` (3x3, may require grad) is the trainable object rotation applied on top of accum_R during training
    (trainable_object_move.rot_L in the reference).  Keeps the reference's [N,1]-index quirk, see covariance.py.
    selection: a cached object_selection(is_object, which_object, N); opacity_raw: also return sigmoid(opacity_raw) from the
    same launch -> (cov3D, opacity)."""
    dev = scaling.device
    if accum_R is None:
        accum_R = torch.eye(3, device=dev)
    M = accum_R.to(dev).float()
    if rot_matrix is not None:
        M = rot_matrix @ M
    sel, mult = selection if selection is not None else object_selection(is_object, which_object, scaling.shape[0])
    return _Cov3D.apply(scaling, rotation, M, sel, scaling_modifier, mult, scaling_is_log, opacity_raw)


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, lambda_dssim, gate, running_sum=None, defer_value=False, raster_node=None, lossgrad=False):
        L = _lib.load()
        img, gt = _need_hip(img, "image"), _need_hip(gt, "gt")
        assert img.dim() == 3 and img.shape == gt.shape
        Cc, H, W = img.shape
        dev = img.device
        partial = torch.empty(L.egs_l1_ssim_partial_count(Cc, H, W), device=dev)
        maps = torch.empty((3, Cc, H, W), device=dev)
        loss = torch.empty((), device=dev)
        side = None
        if lossgrad and raster_node is not None and Cc == 3:
            # raster_lossgrad: the rasterizer's backward blend will compute this loss's image gradient itself; what the loss BACKWARD launch
            # used to carry for it (tile order, cleared accumulator, optimizer bookkeeping) rides in this forward launch instead
            from .rasterizer import backward_prologue_of
            side = backward_prologue_of(raster_node)
        with _hip.device_ctx(dev):
            if side is not None:
                _lib.check(L.egs_l1_ssim_forward_ex(Cc, H, W, _p(img), _p(gt), float(lambda_dssim), _p(partial), _p(maps[0]), _p(maps[1]),
                                                    _p(maps[2]), None if defer_value else _p(loss), None if defer_value else _p(running_sum),
                                                    C.byref(side), _stream(dev)))
            else:
                _lib.check(L.egs_l1_ssim_forward(Cc, H, W, _p(img), _p(gt), float(lambda_dssim), _p(partial), _p(maps[0]), _p(maps[1]),
                                                 _p(maps[2]), None if defer_value else _p(loss), None if defer_value else _p(running_sum), _stream(dev)))
        ctx.lossgrad = side is not None
        ctx.save_for_backward(img, gt, maps, gate if gate is not None else torch.empty(0))
        ctx.lam, ctx.has_gate = float(lambda_dssim), gate is not None
        ctx.deferred = (partial, loss, running_sum) if defer_value else None
        ctx.raster_node = raster_node
        return loss

    @staticmethod
    def backward(ctx, g):
        L = _lib.load()
        img, gt, maps, gate = ctx.saved_tensors
        gate = gate.float().contiguous() if ctx.has_gate else None
        Cc, H, W = img.shape
        g = g.reshape(1)
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.float().contiguous()
        dimg = torch.empty_like(img)
        if ctx.lossgrad:
            # no launch here: the rasterizer backward that follows computes dL/dimage inside its blend kernel (include/egs_raster.h
            # egs_backward_lossgrad) -- bit-identical to what this launch would have written.  `dimg` goes back uninitialised and unread.
            lg = _lib.LossGrad()
            lg.image, lg.gt, lg.dm_dmu1, lg.dm_dexx, lg.dm_dexy = img.data_ptr(), gt.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr()
            lg.gate = gate.data_ptr() if gate is not None else None
            lg.upstream_grad, lg.lambda_dssim = g.data_ptr(), ctx.lam
            d = ctx.deferred
            if d:
                lg.deferred_partial_sums, lg.deferred_loss = d[0].data_ptr(), d[1].data_ptr()
                lg.loss_running_sum = d[2].data_ptr() if d[2] is not None else None
            # (the third element is how the rasterizer's backward recognises THIS tensor: anything autograd put between the two nodes -- a second
            # consumer of the image whose gradient was added, a hook that returned another tensor -- arrives as a different one and is refused
            # there instead of being silently dropped)
            ctx.raster_node.loss_grad = (lg, (img, gt, maps, gate, g, d), (dimg.data_ptr(), dimg._version))
            return dimg, None, None, None, None, None, None, None
        side = None
        if ctx.raster_node is not None:
            from .rasterizer import backward_prologue_of
            side = backward_prologue_of(ctx.raster_node)
        with _hip.device_ctx(img.device):
            d = ctx.deferred
            _lib.check(L.egs_l1_ssim_backward_ex(Cc, H, W, _p(img), _p(gt), ctx.lam, _p(g), _p(gate), _p(maps[0]), _p(maps[1]),
                                                 _p(maps[2]), _p(dimg), _p(d[0]) if d else None, _p(d[1]) if d else None,
                                                 _p(d[2]) if d else None, C.byref(side) if side is not None else None, _stream(img.device)))
        return dimg, None, None, None, None, None, None, None


class _L1SSIMPair(torch.autograd.Function):
    """(mean|img - gt|, mean SSIM(img, gt)) from one forward launch; one backward launch for both upstream scalars
    (include/egs_raster.h egs_l1_ssim_pair_forward / _backward)."""

    @staticmethod
    def forward(ctx, img, gt, raster_node=None):
        L = _lib.load()
        img, gt = _need_hip(img, "image"), _need_hip(gt, "gt")
        assert img.dim() == 3 and img.shape == gt.shape
        Cc, H, W = img.shape
        dev = img.device
        partial = torch.empty(L.egs_l1_ssim_partial_count(Cc, H, W), device=dev)
        maps = torch.empty((3, Cc, H, W), device=dev)
        vals = torch.empty(2, device=dev)
        ctx.raster_node = raster_node
        with _hip.device_ctx(dev):
            _lib.check(L.egs_l1_ssim_pair_forward(Cc, H, W, _p(img), _p(gt), _p(partial), _p(maps[0]), _p(maps[1]), _p(maps[2]), _p(vals[0:1]),
                                                  _p(vals[1:2]), _stream(dev)))
        ctx.save_for_backward(img, gt, maps)
        return vals[0], vals[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        L = _lib.load()
        img, gt, maps = ctx.saved_tensors
        Cc, H, W = img.shape
        ups = torch.zeros(2, device=img.device) if (g_l1 is None or g_ssim is None) else None
        up = lambda g, k: (ups[k:k + 1] if g is None else g.reshape(1).float().contiguous())
        u1, u2 = up(g_l1, 0), up(g_ssim, 1)
        dimg = torch.empty_like(img)
        side = None
        if ctx.raster_node is not None:
            from .rasterizer import backward_prologue_of
            side = backward_prologue_of(ctx.raster_node)
        with _hip.device_ctx(img.device):
            _lib.check(L.egs_l1_ssim_pair_backward(Cc, H, W, _p(img), _p(gt), _p(u1), _p(u2), None, _p(maps[0]), _p(maps[1]), _p(maps[2]), _p(dimg),
                                                   C.byref(side) if side is not None else None, _stream(img.device)))
        return dimg, None, None


def l1_and_ssim(image, gt, raster_prologue=True):
    """-> (mean |image - gt|, mean SSIM(image, gt)) as two differentiable scalars from ONE HIP launch each way: the reference's
    l1_loss(x, gt) and ssim(x, gt) (/root/reference/utils/loss_utils.py:57-58,79-107) when a loop combines them itself.
    raster_prologue: when `image` is the rasterizer's output itself, the backward launch also carries the preparation of the rasterizer's
    backward (as l1_ssim_loss(raster_prologue=True)): one launch less per iteration, results unchanged (a gradient hook on the image,
    the reference's hand mask, sits between the two backwards and is unaffected)."""
    node = None
    if raster_prologue:
        fn = image.grad_fn
        if fn is not None and getattr(fn, "egs_raster_node", False):
            node = fn
    return _L1SSIMPair.apply(image, gt, node)


def l1_ssim_loss(image, gt, lambda_dssim=0.2, grad_gate=None, running_sum=None, defer_value=False, raster_prologue=False, raster_lossgrad=False):
    """(1 - lambda) * mean|image - gt| + lambda * (1 - SSIM(image, gt)).  `grad_gate` [H,W] multiplies d loss / d image
    per pixel (the reference's `render_image.register_hook(lambda grad: grad * (1 - hand_mask))`).
    running_sum: optional device scalar the loss value is also added to (logging without a launch or a host read per iteration).
    defer_value=True: the returned tensor receives its value during backward() instead of right away -- one launch less per
    iteration, for steps whose loss is only read after the backward (graph.GraphedTrainStep).
    raster_prologue=True: see below -- one launch less per iteration when `image` is the rasterizer's output itself.
    raster_lossgrad=True (with raster_prologue, three channels, and a loss.backward() that is SURE to follow -- the optimizer bookkeeping of a
    fused Adam rides in the FORWARD launch then): this loss has no backward launch at all; the rasterizer's backward blend computes the image
    gradient from the maps this forward leaves, bit-identical to the launch it replaces (include/egs_raster.h egs_backward_lossgrad).  The
    gradient tensor autograd hands from this loss to the rasterizer is uninitialised memory: hooks on `image` must not read it (use grad_gate)."""
    if running_sum is not None:
        running_sum = _need_hip(running_sum, "running_sum")
    node = None
    if raster_prologue:
        # `image` straight from the rasterizer: this loss's backward launch also carries the preparation of the rasterizer's backward
        # (tile order, cleared accumulator, fused-optimizer bookkeeping; include/egs_raster.h egs_l1_ssim_backward_ex) in extra
        # workgroups, which otherwise is a launch of its own right after this one.  Results are the same either way.
        fn = image.grad_fn
        if fn is not None and getattr(fn, "egs_raster_node", False):
            node = fn
    return _L1SSIM.apply(image, gt, lambda_dssim, grad_gate, running_sum, defer_value, node, bool(raster_lossgrad and node is not None))
