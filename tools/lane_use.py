#!/usr/bin/env python
"""Lane utilisation of the forward blend: (wave, splat) visits and kept (pixel, splat) pairs at config C.
Needs two instrumented libraries:
  make -C egogaussian_amd/csrc OBJDIR=/tmp/m1 LIB=/tmp/m1/libegs.so EXTRA=-DEGS_MEASURE=1
  make -C egogaussian_amd/csrc OBJDIR=/tmp/m2 LIB=/tmp/m2/libegs.so EXTRA=-DEGS_MEASURE=2
and is run once per library:  EGS_RASTER_LIB=<lib> python tools/lane_use.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import _C
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render

N, H, W = int(os.environ.get("N", 500_000)), int(os.environ.get("H", 540)), int(os.environ.get("W", 960))
dev = torch.device("cuda", 0)
if os.environ.get("SCENE"):                            # e.g. SCENE=bench_data/trained_scene.npz (the densified model)
    import numpy as _np
    _z = _np.load(os.environ["SCENE"])
    _scene = {k: _z[k] for k in ("xyz", "log_scale", "quat", "opacity_logit", "features")}
    N = _scene["xyz"].shape[0]
else:
    _scene = make_scene(N, H, W, seed=0)
pc = SynthGaussians(_scene, device=dev, requires_grad=False)
from egogaussian_amd.renderer import get_raster_settings
cam, bg = make_camera(0, H, W, device=dev), torch.zeros(3, device=dev)
rs = get_raster_settings(cam, pc, bg)
e = torch.empty(0, device=dev)
with torch.no_grad():
    for _ in range(3):                                  # the second call runs at an adequate capacity, the third is placed by the second's costs
        r = _C.rasterize_gaussians(bg, pc.get_xyz, e, pc.get_opacity, e, e, 1.0, pc.get_covariance(1.0), rs.viewmatrix,
                                   rs.projmatrix, rs.tanfovx, rs.tanfovy, H, W, pc.get_features, 0, rs.campos, False, False)
torch.cuda.synchronize()
v = _C.image_views(r[7], W, H)
if os.environ.get("QUADS"):                             # per-quadrant statistics of the product build for offline analysis
    import numpy as _np
    _np.savez(os.environ["QUADS"], ranges=v["ranges"].cpu().numpy(), quad_pairs=v["quad_pairs"].cpu().numpy(), quad_visits=v["quad_visits"].cpu().numpy(), quad_work=v["quad_work"].cpu().numpy(),
              n_contrib=v["n_contrib"].cpu().numpy())
print("R", r[0], "sum(quad_work)", int(v["quad_work"].long().sum()), "sum(n_contrib)", int(v["n_contrib"].long().sum()))
if os.environ.get("BACKWARD"):                          # EGS_MEASURE=4 build: the backward logs its timeline into n_contrib
    gen = torch.Generator().manual_seed(1)
    gc = torch.rand((3, H, W), generator=gen).to(dev)
    v["n_contrib"][:, :6][::8] = v["n_contrib"][:, :6][::8]          # (no-op; keeps the view alive)
    _C.rasterize_gaussians_backward(bg, pc.get_xyz, r[4], e, e, e, 1.0, pc.get_covariance(1.0), rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                    rs.tanfovy, gc, e, e, pc.get_features, 0, rs.campos, r[5], r[0], r[6], r[7], r[3], False)
    torch.cuda.synchronize()
if os.environ.get("TIMELINE"):
    import numpy as np
    nc = v["n_contrib"].cpu().numpy().astype(np.int64)
    gy, gx = (H + 15) // 16, (W + 15) // 16
    rows = []
    for ty in range(gy):
        for tx in range(gx):
            for q in range(4):
                y, x = ty * 16 + (q >> 1) * 8, tx * 16 + (q & 1) * 8
                if y >= H or x + 3 >= W:
                    continue
                t0, t1, hw, n = nc[y, x], nc[y, x + 1], nc[y, x + 2], nc[y, x + 3]
                rows.append((t0 & 0xffffffff, t1 & 0xffffffff, hw, n, ty * gx + tx, q, nc[y, x + 4], nc[y, x + 5]))
    a = np.array(rows, dtype=np.int64)
    t0, t1 = a[:, 0] - a[:, 0].min(), a[:, 1] - a[:, 0].min()
    dur = t1 - t0
    print("waves", len(a), "kernel span (10 ns ticks)", t1.max(), "start spread", t0.max(), "wave duration mean/median/max", dur.mean(), np.median(dur), dur.max())
    for pct in (50, 75, 90, 95, 99, 100):
        print(f"  {pct}% of waves finished by tick", np.percentile(t1, pct))
    simd = (a[:, 2] >> 16) * 100000 + ((a[:, 2] >> 8) & 0xff) * 16 + ((a[:, 2] >> 4) & 3)     # xcc, (se,sh,cu), simd
    ids, inv = np.unique(simd, return_inverse=True)
    busy = np.zeros(len(ids)); endt = np.zeros(len(ids)); cnt = np.zeros(len(ids))
    np.add.at(busy, inv, dur); np.maximum.at(endt, inv, t1); np.add.at(cnt, inv, 1)
    print("SIMDs", len(ids), "waves/SIMD min/mean/max", cnt.min(), cnt.mean(), cnt.max())
    print("SIMD end time min/mean/max", endt.min(), endt.mean(), endt.max(), " sum of wave durations per SIMD min/mean/max", busy.min(), busy.mean(), busy.max())
    vis_simd = np.zeros(len(ids)); np.add.at(vis_simd, inv, a[:, 6])
    cu = simd // 4
    cids, cinv = np.unique(cu, return_inverse=True)
    vis_cu = np.zeros(len(cids)); np.add.at(vis_cu, cinv, a[:, 6])
    print("blended splats per SIMD min/mean/max", vis_simd.min(), vis_simd.mean(), vis_simd.max(), " per CU min/mean/max", vis_cu.min(), vis_cu.mean(), vis_cu.max(),
          " corr(SIMD visits, SIMD end)", np.corrcoef(vis_simd, endt)[0, 1])
    print("corr(list length n, wave duration)", np.corrcoef(a[:, 3], dur)[0, 1])
    visits, depth = a[:, 6].astype(float), a[:, 7].astype(float)
    batches = np.ceil(np.minimum(depth + 64, a[:, 3]) / 64.0)
    X = np.stack([np.ones(len(a)), batches, visits], 1)
    coef, *_ = np.linalg.lstsq(X, dur.astype(float), rcond=None)
    print("duration ~ %.1f + %.2f * batches + %.2f * visits (ticks of 10 ns); mean batches %.1f visits %.1f; r2 %.3f" % (
        coef[0], coef[1], coef[2], batches.mean(), visits.mean(), 1 - ((X @ coef - dur) ** 2).sum() / ((dur - dur.mean()) ** 2).sum()))
    np.save(os.environ["TIMELINE"], a)
