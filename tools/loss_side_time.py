#!/usr/bin/env python
"""Time of the loss backward launch WITH the rasterizer backward's preparation carried (egs_l1_ssim_backward_ex: tile ordering, accumulator
clearing) against the bare launch, at config C, on the buffers of a real forward.  EGS_RASTER_LIB selects an A/B build
(-DEGS_ABL_SIDE=1: no zeroing, =2: no ordering -- timing only)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import lib as _lib, _C
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
L = _lib.load()
N, H, W = 500000, 540, 960
dev = torch.device("cuda:0")
pc = SynthGaussians(make_scene(N, H, W, 0), device=dev, requires_grad=False)
with torch.no_grad():
    out = render(make_camera(0, H, W, device=dev), pc, Pipe, torch.zeros(3, device=dev))
img_buf = _C.stats["image_buffer"]
g = torch.Generator().manual_seed(0)
a = out["render"].contiguous(); b = torch.rand(3, H, W, generator=g).to(dev)
partial = torch.empty(L.egs_l1_ssim_partial_count(3, H, W), device=dev); maps = torch.empty(3, 3, H, W, device=dev)
dimg = torch.empty_like(a); one = torch.ones(1, device=dev); loss = torch.empty(1, device=dev)
scratch = torch.empty(L.egs_backward_scratch_bytes(N), dtype=torch.uint8, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.egs_l1_ssim_forward(3, H, W, p(a), p(b), 0.2, p(partial), p(maps[0]), p(maps[1]), p(maps[2]), None, None, s)
side = _lib.BackwardPrologue(); side.P, side.width, side.height = N, W, H
side.image_buffer, side.scratch = img_buf.data_ptr(), scratch.data_ptr()
if os.environ.get("TICK", "1") == "1":            # the fused optimizer's bookkeeping job (egs_adam_tick) rides along as in the captured step
    sink = _lib.AdamSink()
    keep = []
    for leaf, rf in ((0, 3), (1, 1), (2, 3), (3, 4), (4, 3)):
        prm, m, v, lr, st = [torch.zeros(N * rf, device=dev) for _ in range(3)] + [torch.full((1,), 1e-3, device=dev), torch.zeros(1, device=dev)]
        f = sink.leaf[leaf]; f.param, f.exp_avg, f.exp_avg_sq, f.lr, f.step = prm.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(), st.data_ptr()
        keep += [prm, m, v, lr, st]
    coef = torch.zeros(12, device=dev); keep.append(coef)
    sink.beta1, sink.beta2, sink.eps, sink.coef = 0.9, 0.999, 1e-15, coef.data_ptr()
    side.sink = C.pointer(sink)
bare = lambda: L.egs_l1_ssim_backward_ex(3, H, W, p(a), p(b), 0.2, p(one), None, p(maps[0]), p(maps[1]), p(maps[2]), p(dimg), p(partial), p(loss), None, None, s)
carried = lambda: L.egs_l1_ssim_backward_ex(3, H, W, p(a), p(b), 0.2, p(one), None, p(maps[0]), p(maps[1]), p(maps[2]), p(dimg), p(partial), p(loss), None, C.byref(side), s)
res = []
for f in (bare, carried):
    for _ in range(20): assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(300): f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 300 * 1e3)
print(f"{os.environ.get('TAG', '')}: loss backward bare {res[0]:.2f} us, carrying the rasterizer backward's preparation {res[1]:.2f} us")
