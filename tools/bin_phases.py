#!/usr/bin/env python
"""Phase times of k_bin_count's workgroups at config C from an EGS_BIN_TIMING build (s_memtime stamps by every wave):
    make -C egogaussian_amd/csrc OBJDIR=/tmp/bt LIB=$PWD/build_ab/libegs_bt.so EXTRA=-DEGS_BIN_TIMING
    EGS_RASTER_LIB=$PWD/build_ab/libegs_bt.so python tools/bin_phases.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import _C, lib as _lib
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render
N, H, W = 500_000, 540, 960
dev = torch.device("cuda", 0)
pc = SynthGaussians(make_scene(N, H, W, seed=0), device=dev, requires_grad=False)
cam, bg = make_camera(0, H, W, device=dev), torch.zeros(3, device=dev)
with torch.no_grad():
    for _ in range(3):
        render(cam, pc, Pipe, bg)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
out = (C.c_ulonglong * (512 * 16 * 6))()
raw.egs_debug_bin_stamps(out)
t = np.array(out[:], dtype=np.int64).reshape(512, 16, 6)
t = t[(t[:, :, 0] > 0).all(1)]                                     # workgroups that ran
t0 = t[:, :, 0].min()
names = ["set-up (zero hist, load + scan 64 Gaussians, park in LDS)", "wait at the barrier", "deal units (slot walk, cull test, LDS atomic)",
         "wait at the barrier", "table column + chunk sums"]
print(f"{len(t)} workgroups x 16 waves; kernel span {(t[:, :, 5].max() - t0)} cycles; workgroup start spread {t[:, :, 0].min(1).max() - t0}")
for k, nm in enumerate(names):
    d = (t[:, :, k + 1] - t[:, :, k]).astype(float)
    print(f"  {nm:62s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}  max {d.max():8.0f} cycles")
life = (t[:, :, 5].max(1) - t[:, :, 0].min(1)).astype(float)
print(f"  workgroup lifetime mean {life.mean():.0f} p90 {np.percentile(life, 90):.0f} max {life.max():.0f} cycles; last workgroup ends at {t[:, :, 5].max() - t0}")
u = (C.c_ulonglong * 48)()
raw.egs_debug_bin_unit_stamps(u)
u = np.array(u[:], dtype=np.int64).reshape(8, 6)
print("units of wave 0 of workgroup 7: cycles for  owner search | rectangle + slot -> tile | cull test | LDS atomic | (to next unit)")
for r in range(8):
    if u[r, 0] and u[r, 4]:
        nxt = u[r + 1, 0] - u[r, 4] if r + 1 < 8 and u[r + 1, 0] else -1
        print("   ", [int(u[r, k + 1] - u[r, k]) for k in range(4)], int(nxt))
st = (C.c_ulonglong * (2048 * 8))()
raw.egs_debug_sort_stamps(st)
st = np.array(st[:], dtype=np.int64).reshape(2048, 8)
st = st[(st[:, :7] > 0).all(1)]
print(f"k_tile_sort, wave 0 of {len(st)} tiles that took the one-pass path (cycles): ")
for k, nm in enumerate(["ranges + load keys", "min / max over the workgroup (2 barriers)", "rank pass (LDS atomics)", "barrier + digit bases", "scatter to LDS + barrier",
                        "in-bucket comparison + point_list stores"]):
    d = (st[:, k + 1] - st[:, k]).astype(float)
    print(f"  {nm:46s} mean {d.mean():7.0f}  p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f}  max {d.max():7.0f}")
print(f"  total mean {(st[:, 6] - st[:, 0]).mean():.0f} max {(st[:, 6] - st[:, 0]).max()}")
