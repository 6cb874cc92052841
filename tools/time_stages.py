#!/usr/bin/env python
"""Per-stage HIP-event timing of the rasterizer alone (forward + backward) on a synthetic scene.
   python tools/time_stages.py [N] [H] [W] [iters]     (EGS_RASTER_LIB=path selects an A/B build of the library;
   SH_DEGREE=d for more colour coefficients, SH_SPLIT=0 to hand them over concatenated as the reference's get_features does,
   SCALE_MUL=f multiplies every splat's extent, SCENE=file.npz replaces the scene)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egogaussian_amd import lib, _C
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render

N, H, W, iters = [int(a) for a in (sys.argv[1:5] + ["500000", "540", "960", "30"][len(sys.argv) - 1:])]
dev = "cuda:0"
D = int(os.environ.get("SH_DEGREE", "0"))
scene = make_scene(N, H, W, 0, sh_degree=D)
if os.environ.get("SCENE"):                            # e.g. SCENE=bench_data/trained_scene.npz (the densified model)
    import numpy as np
    z = np.load(os.environ["SCENE"])
    scene = {k: z[k] for k in ("xyz", "log_scale", "quat", "opacity_logit", "features")}
    N = scene["xyz"].shape[0]
if os.environ.get("SCALE_MUL"):
    import numpy as np
    scene["log_scale"] = scene["log_scale"] + np.float32(math.log(float(os.environ["SCALE_MUL"])))
pc = SynthGaussians(scene, device=dev, sh_degree=D)
if os.environ.get("SH_SPLIT", "1") == "0":
    pc.get_features_split = lambda: None
vis = render(make_camera(0, H, W, device=dev), pc, Pipe, torch.zeros(3, device=dev))["visibility_filter"]
print("visible fraction", float(vis.float().mean()))
cams = [make_camera(k, H, W, device=dev) for k in range(8)]
bg = torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(1)
up = [torch.rand(s, generator=g).to(dev) for s in ((3, H, W), (1, H, W), (1, H, W))]
def step(k):
    out = render(cams[k % 8], pc, Pipe, bg)
    ((out["render"] * up[0]).sum() + (out["depth"] * up[1]).sum() + (out["alpha"] * up[2]).sum()).backward()
for k in range(5): step(k)
torch.cuda.synchronize()
lib.profile_begin(16 * (iters + 4))
for k in range(iters): step(k)
torch.cuda.synchronize()
st = lib.profile_end()
tot = 0.0
for name, (ms, n) in st.items():
    if n:
        print(f"{name:22s} {1e3 * ms / n:9.1f} us  x{n}"); tot += ms / n
print(f"{'total':22s} {1e3 * tot:9.1f} us   R={_C.stats['num_rendered']}")
