#!/usr/bin/env python
"""Bit-for-bit check of the loss gradient computed inside the backward blend against k_l1_ssim_backward's, per pixel, on the device.
Needs a library built with -DEGS_LG_CHECK (make OBJDIR=... LIB=... EXTRA=-DEGS_LG_CHECK; EGS_RASTER_LIB=<that library>): the blend then
computes the gradient (egs_debug_set_lossgrad) AND loads the one the loss backward wrote, and counts the pixels whose bits differ."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egogaussian_amd import lib
from egogaussian_amd.scene_synth import make_scene, make_camera, perturb_student, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
from egogaussian_amd.fused import l1_ssim_loss
L = lib.load()
L.egs_debug_set_lossgrad.restype, L.egs_debug_set_lossgrad.argtypes = C.c_int, [C.c_void_p] * 7 + [C.c_float]
L.egs_debug_lg_mismatches.restype = C.c_uint
dev = "cuda:0"
total = 0
for (N, H, W, gated) in ((500000, 540, 960, False), (20000, 135, 250, True), (3000, 33, 47, False), (100000, 540, 960, True)):
    teacher = make_scene(N, H, W, 0)
    pc = SynthGaussians(perturb_student(teacher), device=dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    with torch.no_grad():
        tpc = SynthGaussians(teacher, device=dev, requires_grad=False)
    gate = (torch.rand((H, W)) < 0.8).float().to(dev) if gated else None
    one = torch.ones(1, device=dev)
    for k in range(4):
        cam = make_camera(k * 17, H, W, device=dev)
        with torch.no_grad():
            gt = render(cam, tpc, Pipe, bg)["render"].clone()
        out = render(cam, pc, Pipe, bg)
        loss = l1_ssim_loss(out["render"], gt, 0.2, grad_gate=gate)
        img, gt_s, maps, gate_s = loss.grad_fn.saved_tensors
        L.egs_debug_set_lossgrad(img.data_ptr(), gt_s.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(),
                                 gate.data_ptr() if gated else None, one.data_ptr(), 0.2)
        loss.backward(gradient=one.reshape(()))
        torch.cuda.synchronize()
        L.egs_debug_set_lossgrad(None, None, None, None, None, None, None, 0.2)
    n = L.egs_debug_lg_mismatches()
    print(f"{N} Gaussians @ {W}x{H}{' gated' if gated else ''}: mismatching pixels so far {n} (4 frames)")
    total = n
print("LG mismatching pixels:", total)
