#!/usr/bin/env python
"""(uses the oracle: lives under tests/)  Iterations of a quadrant-wave of the blend kernels under different pixel-to-lane work lists, on
the trained scene (default) or config C (`C`): today's mapping (one splat per 64-lane iteration, work list = splats that reach the 8x8
quadrant) against per-sub-block work lists (four 4x4 blocks, or two 8x4 / 4x8 blocks, each walking ITS OWN splats: iterations per batch
of 64 list entries = the longest of the sub-lists), with sub-block masks from the exact alpha >= 1/255 footprint or from its bounding box."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from egogaussian_amd.scene_synth import make_camera, SynthGaussians, make_scene

H, W = 540, 960
if len(sys.argv) > 1 and sys.argv[1] == "C":
    scene = make_scene(500000, H, W, seed=0)
else:
    z = np.load(os.path.join(ROOT, "bench_data", "trained_scene.npz"))
    scene = {k: z[k] for k in ("xyz", "log_scale", "quat", "opacity_logit", "features")}
pc = SynthGaussians(scene, device="cpu", requires_grad=False, fused=False)
cam = make_camera(0, H, W, device="cpu")
o = Oracle(np.float32, nthreads=8)
st = o.forward(means3D=pc.get_xyz, opacities=pc.get_opacity, shs=pc.get_features, cov3D_precomp=pc.get_covariance(),
               viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
               bg=np.zeros(3, np.float32), image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
ro, po = st["ranges"].astype(np.int64), st["point_list"]
co, xy = st["conic_opacity"].astype(np.float32), st["xy"].astype(np.float32)
ncontrib = st["n_contrib"]
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = rng.choice(len(ro), size=240, replace=False)
tot = {k: 0 for k in ("entries", "kept", "cur", "s4_exact", "s4_bbox", "s2h_exact", "s2v_exact", "s4_exact_queue", "pairs", "cur_fwdlen")}
for t in tiles:
    g_ = po[ro[t, 0]:ro[t, 1]]
    if len(g_) == 0:
        continue
    px = (t % gx) * 16 + np.arange(16); py = (t // gx) * 16 + np.arange(16)
    dx = xy[g_, 0][:, None, None] - px[None, None, :]
    dy = xy[g_, 1][:, None, None] - py[None, :, None]
    power = -0.5 * (co[g_, 0][:, None, None] * dx * dx + co[g_, 2][:, None, None] * dy * dy) - co[g_, 1][:, None, None] * dx * dy
    alpha = np.minimum(0.99, co[g_, 3][:, None, None] * np.exp(np.minimum(power, 0.0)))
    m = (alpha >= 1.0 / 255.0) & (power <= 0) & (px[None, None, :] < W) & (py[None, :, None] < H)      # [n, y, x]
    keep = m.any(axis=(1, 2))
    tot["entries"] += len(g_); tot["kept"] += int(keep.sum()); tot["pairs"] += int(m.sum())
    m = m[keep]
    # the backward walks the list up to the wave's last contributor; approximate with the tile's n_contrib maximum over the kept list
    sb = m.reshape(len(m), 4, 4, 4, 4).any(axis=(2, 4))                       # [n, sy, sx] 4x4-pixel blocks
    rows, cols = m.any(axis=2), m.any(axis=1)                                  # bounding box of the footprint
    r0, r1 = rows.argmax(1), 15 - rows[:, ::-1].argmax(1); c0, c1 = cols.argmax(1), 15 - cols[:, ::-1].argmax(1)
    bb = np.zeros_like(sb)
    for sy in range(4):
        for sx in range(4):
            bb[:, sy, sx] = (r0 <= 4 * sy + 3) & (r1 >= 4 * sy) & (c0 <= 4 * sx + 3) & (c1 >= 4 * sx)
    for q in range(4):
        qy, qx = q >> 1, q & 1
        e = sb[:, 2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].reshape(len(m), 4)    # the quadrant's four 4x4 blocks
        b = bb[:, 2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].reshape(len(m), 4)
        hit = e.any(1)
        tot["cur"] += int(hit.sum())
        tot["s4_exact_queue"] += int(e.sum(0).max())
        for lo in range(0, len(m), 64):
            eb, bbch = e[lo:lo + 64], (b[lo:lo + 64] & hit[lo:lo + 64, None])
            tot["s4_exact"] += int(eb.sum(0).max())
            tot["s4_bbox"] += int(bbch.sum(0).max())
            tot["s2h_exact"] += int(max((eb[:, 0] | eb[:, 1]).sum(), (eb[:, 2] | eb[:, 3]).sum()))     # two 4(rows) x 8 blocks
            tot["s2v_exact"] += int(max((eb[:, 0] | eb[:, 2]).sum(), (eb[:, 1] | eb[:, 3]).sum()))     # two 8 x 4(cols) blocks
print({k: v for k, v in tot.items()})
c = tot["cur"]
print(f"kept {tot['kept'] / tot['entries']:.3f} of rect entries; visits per kept entry and quadrant {c / (4 * tot['kept']):.3f}; lanes kept {tot['pairs'] / c:.1f}")
for k in ("s4_exact", "s4_bbox", "s2h_exact", "s2v_exact", "s4_exact_queue"):
    print(f"{k:16s} iterations {tot[k] / c:.3f} of today's   -> lanes per iteration {tot['pairs'] / tot[k]:.1f}")
