#!/usr/bin/env python
"""Time of the fused loss kernels alone at 3x540x960, launched back to back (A/B of library variants: EGS_RASTER_LIB=...).
Prints microseconds per forward launch and per backward launch (HIP events around 300 launches each)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import lib as _lib
L = _lib.load()
H, W = int(os.environ.get("H", 540)), int(os.environ.get("W", 960))
g = torch.Generator().manual_seed(0)
a = torch.rand(3, H, W, generator=g).cuda(); b = torch.rand(3, H, W, generator=g).cuda()
partial = torch.empty(L.egs_l1_ssim_partial_count(3, H, W), device="cuda"); maps = torch.empty(3, 3, H, W, device="cuda")
dimg = torch.empty_like(a); one = torch.ones(1, device="cuda"); loss = torch.empty(1, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fwd = lambda: L.egs_l1_ssim_forward(3, H, W, p(a), p(b), 0.2, p(partial), p(maps[0]), p(maps[1]), p(maps[2]), None, None, s)
bwd = lambda: L.egs_l1_ssim_backward(3, H, W, p(a), p(b), 0.2, p(one), None, p(maps[0]), p(maps[1]), p(maps[2]), p(dimg), p(partial), p(loss), None, s)
out = []
for f in (fwd, bwd):
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(300): f()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 300 * 1e3)
print(f"{os.environ.get('TAG', '')}: forward {out[0]:.2f} us, backward {out[1]:.2f} us per launch")
