"""`simple_knn._C`: distCUDA2(points[N,3]) -> float32[N], the mean squared distance to the 3 nearest neighbours.
Backed by egs_knn3_mean_dist2 in libegs_raster.so (egogaussian_amd/csrc/knn.hip).  HIP device tensors only."""
import ctypes as C

import torch

from egogaussian_amd import lib as _lib


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError(f"distCUDA2: points are on {points.device}; the HIP implementation has no CPU fallback")
    pts = points.detach().float().contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("distCUDA2: expected points of shape [N, 3]")
    out = torch.empty((pts.shape[0],), device=pts.device, dtype=torch.float32)
    with torch.cuda.device(pts.device):
        _lib.check(_lib.load().egs_knn3_mean_dist2(pts.shape[0], C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out
