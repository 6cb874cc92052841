#!/usr/bin/env python
"""Per-row phase times of one wave of the loss forward kernel, from an EGS_LOSS_TIMING build (s_memtime stamps; 100 MHz ticks):
    make -C egogaussian_amd/csrc OBJDIR=/tmp/lt LIB=$PWD/build_ab/libegs_lt.so EXTRA=-DEGS_LOSS_TIMING=300
    EGS_RASTER_LIB=$PWD/build_ab/libegs_lt.so python tools/loss_rows.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import lib as _lib
L = _lib.load()
H, W = 540, 960
g = torch.Generator().manual_seed(0)
a = torch.rand(3, H, W, generator=g).cuda(); b = torch.rand(3, H, W, generator=g).cuda()
partial = torch.empty(L.egs_l1_ssim_partial_count(3, H, W), device="cuda"); maps = torch.empty(3, 3, H, W, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    L.egs_l1_ssim_forward(3, H, W, p(a), p(b), 0.2, p(partial), p(maps[0]), p(maps[1]), p(maps[2]), None, None, s)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
out = (C.c_ulonglong * 256)()
raw.egs_debug_loss_stamps(out)
t = np.array(out[:], dtype=np.int64).reshape(64, 4)
t0 = t[0, 0]
for r in range(27):
    row = t[r]
    if row[0] == 0: continue
    d = [int(row[k] - row[0]) if row[k] else -1 for k in range(4)]
    nxt = int(t[r + 1, 0] - row[0]) if t[r + 1, 0] else -1
    print(f"row {r:2d}: start {int(row[0] - t0):6d}  vblur done {d[1]:5d}  hblur done {d[2]:5d}  end {d[3]:5d}  next row starts {nxt:5d}   (10 ns ticks)")
