"""`simple_knn._C`: distCUDA2(points[N,3]) -> float32[N], the mean squared distance to the 3 nearest neighbours.
Backed by libegs_raster.so (egogaussian_amd/csrc/knn.hip): the all-pairs kernel egs_knn3_mean_dist2 for small clouds, the uniform-grid
search egs_knn3_grid from GRID_FROM points up (same values bit for bit; O(N) instead of O(N^2)).  HIP device tensors only."""
import ctypes as C

import torch

from egogaussian_amd import lib as _lib


GRID_FROM = 32768


def distCUDA2(points, method=None):
    """method: None (by size), "pairs" or "grid"."""
    if not points.is_cuda:
        raise RuntimeError(f"distCUDA2: points are on {points.device}; the HIP implementation has no CPU fallback")
    pts = points.detach().float().contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("distCUDA2: expected points of shape [N, 3]")
    out = torch.empty((pts.shape[0],), device=pts.device, dtype=torch.float32)
    L, n = _lib.load(), pts.shape[0]
    grid = (n >= GRID_FROM) if method is None else (method == "grid")
    with torch.cuda.device(pts.device):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if grid and n > 0:
            scratch = torch.empty(L.egs_knn3_grid_scratch_bytes(n), dtype=torch.uint8, device=pts.device)
            _lib.check(L.egs_knn3_grid(n, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), stream))
        else:
            _lib.check(L.egs_knn3_mean_dist2(n, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), stream))
    return out
