#!/usr/bin/env python
"""Phase times of k_bin_emit / k_bin_partition workgroups from an -DEGS_BIN_TIMING build (thread 0 stamps s_memtime at phase boundaries).
   make -C egogaussian_amd/csrc OBJDIR=/tmp/objT LIB=$PWD/egogaussian_amd/libegs_timing.so EXTRA=-DEGS_BIN_TIMING
   EGS_RASTER_LIB=egogaussian_amd/libegs_timing.so python tools/dev/emit_phases.py [N] [H] [W]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from egogaussian_amd import lib, _C
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
N, H, W = [int(a) for a in (sys.argv[1:4] + ["500000", "540", "960"][len(sys.argv) - 1:])]
dev = "cuda:0"
pc = SynthGaussians(make_scene(N, H, W, 0), device=dev, requires_grad=False)
cam = make_camera(0, H, W, device=dev); bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for _ in range(5):
        render(cam, pc, Pipe, bg)
torch.cuda.synchronize()
L = lib.load()
for name, n_wg, fn, labels in (("k_bin_emit", (N + 511) // 512, "egs_debug_emit_stamps", ["loads issued+landed (set-up)", "barrier", "walk", "barrier", "flush"]),
                               ("k_bin_partition", None, "egs_debug_part_stamps", ["pass 1 (count)", "reduce barrier", "scan + pass 2 (place)", "barrier", "write-out"])):
    cap = 8192 if name == "k_bin_emit" else 2048
    buf = np.zeros(cap * 8, np.uint64)
    f = getattr(L, fn); f.restype = C.c_int; f.argtypes = [C.c_void_p]
    assert f(buf.ctypes.data) == 0
    st = buf.reshape(cap, 8)[:, :6].astype(np.int64)
    live = st[:, 5] > 0
    st = st[live]
    if n_wg: st = st[:n_wg]
    t0 = st[:, 0].min()
    ticks = lambda x: float(x)                        # raw s_memtime ticks (calibrate: launch span in ticks vs the kernel's duration under rocprofv3)
    print(f"{name}: {st.shape[0]} workgroups; launch span {us(st[:, 5].max() - t0):.1f} ticks; starts: median {us(np.median(st[:, 0]) - t0):.1f} ticks, max {us(st[:, 0].max() - t0):.1f} ticks")
    for k, lab in enumerate(labels):
        d = st[:, k + 1] - st[:, k]
        print(f"   {lab:32s} mean {us(d.mean()):6.2f} ticks   p90 {us(np.percentile(d, 90)):6.2f}   max {us(d.max()):6.2f}")
    life = st[:, 5] - st[:, 0]
    print(f"   {'lifetime':32s} mean {us(life.mean()):6.2f} ticks   p90 {us(np.percentile(life, 90)):6.2f}   max {us(life.max()):6.2f}")
