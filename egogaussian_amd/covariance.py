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
    else:
        sel = torch.ones(L.shape[0], dtype=torch.bool, device=L.device)
    moved = (accum_R.to(L.dtype)[None, :, :, None] * L[:, None, :, :]).sum(2)        # accum_R @ L, per Gaussian
    if rot_L is not None:
        moved = rot_L(moved)
    L = torch.where(sel[:, None, None], moved, L)
    return strip_symmetric(_outer_gram(L))
