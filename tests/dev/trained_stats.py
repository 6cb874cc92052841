#!/usr/bin/env python
"""(uses the oracle: lives under tests/)  CPU study of the trained scene (bench_data/trained_scene.npz): how many tiles the splats
cover, which Gaussians the backward's atomics hit most, how they sit in memory."""
import math, os, sys
import numpy as np
import torch
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
               bg=np.zeros(3, np.float32), image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
               stop_after="duplicate")
print(st.keys())
tt = st["tiles_touched"].astype(np.int64)
rad = st["radii"]
print("P", len(tt), "visible", (rad > 0).sum(), "R", tt.sum())
print("tiles per visible Gaussian pctl 50/90/99/99.9/max", np.percentile(tt[tt > 0], [50, 90, 99, 99.9]), tt.max())
print("radius pctl", np.percentile(rad[rad > 0], [50, 90, 99, 99.9]), rad.max())
op = 1 / (1 + np.exp(-scene["opacity_logit"].reshape(-1)))
print("opacity pctl 10/50/90", np.percentile(op[rad > 0], [10, 50, 90]))
big = np.argsort(-tt)[:20]
print("biggest:", [(int(i), int(tt[i]), int(rad[i]), round(float(op[i]), 3)) for i in big])
# hot candidates: alpha >= 1/255 box (from the conic and the opacity, as k_preprocess derives it) covering >= 256 tiles
co = st["conic_opacity"].astype(np.float64); xyc = st["xy"].astype(np.float64)
A, B, Cc, o = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
vis = (rad > 0) & (255 * o >= 0.999)
tau2 = 2 * (np.log(np.maximum(255 * o, 1.0)) + 0.01)
detq = A * Cc - B * B
with np.errstate(all="ignore"):
    ex = np.sqrt(tau2 * Cc / detq) * 1.002 + 0.01; ey = np.sqrt(tau2 * A / detq) * 1.002 + 0.01
x0 = np.maximum(np.floor(xyc[:, 0] - ex), 0); x1 = np.minimum(np.ceil(xyc[:, 0] + ex), W - 1)
y0 = np.maximum(np.floor(xyc[:, 1] - ey), 0); y1 = np.minimum(np.ceil(xyc[:, 1] + ey), H - 1)
ok = vis & (x0 <= x1) & (y0 <= y1)
bt = np.where(ok, (x1 // 16 - x0 // 16 + 1) * (y1 // 16 - y0 // 16 + 1), 0)
print("box tiles total", bt.sum(), "vs rect", tt.sum(), "; Gaussians with an empty box among the visible:", int(((rad > 0) & ~ok).sum()))
for thr in (64, 128, 256, 512, 1024):
    hot = bt >= thr
    blocks = np.bincount(np.nonzero(hot)[0] // 256, minlength=(len(bt) + 255) // 256)
    print(f"box >= {thr} tiles: {hot.sum()} Gaussians, {bt[hot].sum()} box tiles ({bt[hot].sum() / bt.sum():.2f} of all); per 256-block max {blocks.max()}, blocks with > 3: {(blocks > 3).sum()}, > 7: {(blocks > 7).sum()}")
