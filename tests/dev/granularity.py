import math, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.common import make_inputs
from oracle.oracle import Oracle
N, H, W = 500000, 540, 960
d = make_inputs(N, H, W, 0, 0, "sh_cov", frame=0, bg=(0, 0, 0))
t = time.time()
o = Oracle(np.float32, nthreads=8)
st = o.forward(**d)
print("oracle forward", time.time() - t, "s; R", st["R"])
ro, po = st["ranges"].astype(np.int64), st["point_list"]
n_o = ro[:, 1] - ro[:, 0]
tile_o = np.repeat(np.arange(len(ro)), n_o)
gx = (W + 15) // 16
co, xy = st["conic_opacity"].astype(np.float32), st["xy"].astype(np.float32)
rng = np.random.default_rng(0)
sel = rng.choice(len(po), size=400000, replace=False)        # sample of instances
tt, gg = tile_o[sel], po[sel]
shapes = {"16x16": (16, 16), "8x8": (8, 8), "8x4(r x c)": (8, 4), "4x8": (4, 8), "4x4": (4, 4), "2x32->(2,16)": (2, 16), "4x16": (4, 16), "16x4": (16, 4)}
visits = {k: 0 for k in shapes}; pairs = 0
for lo in range(0, len(tt), 50000):
    t_, g_ = tt[lo:lo + 50000], gg[lo:lo + 50000]
    px = (t_ % gx)[:, None] * 16 + np.arange(16)[None, :]
    py = (t_ // gx)[:, None] * 16 + np.arange(16)[None, :]
    dx = xy[g_, 0][:, None, None] - px[:, None, :]
    dy = xy[g_, 1][:, None, None] - py[:, :, None]
    power = -0.5 * (co[g_, 0][:, None, None] * dx * dx + co[g_, 2][:, None, None] * dy * dy) - co[g_, 1][:, None, None] * dx * dy
    alpha = np.minimum(0.99, co[g_, 3][:, None, None] * np.exp(np.minimum(power, 0.0)))
    m = (alpha >= 1.0 / 255.0) & (power <= 0) & (px[:, None, :] < W) & (py[:, :, None] < H)      # [n,16(y),16(x)]
    pairs += int(m.sum())
    for k, (bh, bw) in shapes.items():
        mb = m.reshape(len(t_), 16 // bh, bh, 16 // bw, bw).any(axis=(2, 4))
        visits[k] += int(mb.sum())
print("sampled instances", len(tt), "pairs", pairs, "pairs per instance", pairs / len(tt))
for k, (bh, bw) in shapes.items():
    v = visits[k]
    print(f"{k:14s} lanes {bh*bw:4d}  visits/instance {v/len(tt):6.3f}  lane-slots/instance {v*bh*bw/len(tt):8.2f}  utilisation {pairs/(v*bh*bw):.3f}")
