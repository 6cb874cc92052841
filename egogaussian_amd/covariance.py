"""3D covariance producers (PyTorch, autograd) -- the step right before the rasterizer on every training
call, because the reference forces pipe.compute_cov3D_python = True (/root/reference/train.py:49).

Restates /root/reference/utils/general_utils.py:110-156 (build_rotation, build_scaling_rotation,
strip_symmetric) and /root/reference/scene/gaussian_model.py:29-33,46-63 (covariance activation and its
object-rotated variant) as batched tensor expressions.
"""
import torch


def rotation_matrices(q):
    """[N,4] quaternions (w,x,y,z), normalised here -> [N,3,3]."""
    q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
    r, x, y, z = q.unbind(dim=1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)


def scaling_rotation(s, q):
    """L = R(q) diag(s)  [N,3,3]."""
    return rotation_matrices(q) * s[:, None, :]


def strip_symmetric(S):
    """[N,3,3] symmetric -> [N,6] in the order (00,01,02,11,12,22) the rasterizer expects."""
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def _outer_gram(L):
    """L L^T for [N,3,3] without a batched GEMM: on ROCm a 500k-batch of 3x3 products runs for milliseconds in
    rocBLAS, while the broadcast product + reduction is two elementwise kernels."""
    return (L[:, :, None, :] * L[:, None, :, :]).sum(-1)


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    L = scaling_rotation(scaling_modifier * scaling, rotation)
    return strip_symmetric(_outer_gram(L))


class _ScaleRowGrad(torch.autograd.Function):
    """Identity in the forward; multiplies the gradient of row n by mult[n] in the backward."""

    @staticmethod
    def forward(ctx, x, mult):
        ctx.save_for_backward(mult)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (mult,) = ctx.saved_tensors
        return g * mult.view(-1, *([1] * (g.dim() - 1))), None


def rotated_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation, accum_R, is_object=None,
                                             which_object=None, rot_L=None):
    """Object Gaussians' L is left-multiplied by accum_R (and by the trainable rotation `rot_L`, a callable,
    during training) before Sigma = L L^T -- /root/reference/scene/gaussian_model.py:46-63."""
    L = scaling_rotation(scaling_modifier * scaling, rotation)
    if accum_R is None:
        accum_R = torch.eye(3, device=L.device, dtype=L.dtype)
    accum_R = accum_R.to(L.device)
    if which_object is not None and is_object is not None:
        sel = (is_object.reshape(-1) == which_object)
        if is_object.dim() == 2 and sel.numel() > 0:
            # Reference quirk kept for parity: with is_object of shape [N,1] (how the reference stores it,
            # scene/gaussian_model.py:327,478) `nonzero(...).squeeze()` is an [M,2] tensor of (row, 0) pairs, and
            # indexing L with it also selects row 0 -- so Gaussian 0 is rotated whenever any Gaussian is selected.
            # Row 0 then occurs once per selected Gaussian in that index list; the reference gathers it that many
            # times and index_put's the (identical) results back, so autograd hands row 0 the gradient M times
            # (M + 1 if Gaussian 0 is itself selected).  Reproduced exactly.
            count0 = sel.sum() + sel[:1].sum()
            mult = torch.cat([torch.clamp(count0, min=1).to(L.dtype).reshape(1),
                              torch.ones(sel.numel() - 1, dtype=L.dtype, device=L.device)])
            sel = torch.cat([sel[:1] | sel.any(), sel[1:]])
        else:
            mult = None
    else:
        n = L.shape[0]
        sel = torch.ones(n, dtype=torch.bool, device=L.device)
        mult = None
        if is_object is not None and is_object.dim() == 2 and n > 0:          # same quirk with the all-ones mask
            mult = torch.cat([torch.full((1,), float(n + 1), dtype=L.dtype, device=L.device),
                              torch.ones(n - 1, dtype=L.dtype, device=L.device)])
    moved = (accum_R.to(L.dtype)[None, :, :, None] * L[:, None, :, :]).sum(2)        # accum_R @ L, per Gaussian
    if rot_L is not None:
        moved = rot_L(moved)
    if mult is not None:
        moved = _ScaleRowGrad.apply(moved, mult)
    L = torch.where(sel[:, None, None], moved, L)
    return strip_symmetric(_outer_gram(L))
