#!/usr/bin/env python
"""Backward blend per instantiation: MODE 1 (colour gradients only: the training loss) and MODE 0 (the label call: only colors_precomp
carries gradient), HIP-event stage times of the rasterizer alone.  EGS_RASTER_LIB / SCENE as tools/time_stages.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egogaussian_amd import lib
from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
from egogaussian_amd.renderer import render, get_render_label
N, H, W, iters = 500000, 540, 960, 40
dev = "cuda:0"
scene = make_scene(N, H, W, 0)
if os.environ.get("SCENE"):
    import numpy as np
    z = np.load(os.environ["SCENE"])
    scene = {k: z[k] for k in ("xyz", "log_scale", "quat", "opacity_logit", "features")}
pc = SynthGaussians(scene, device=dev)
cams = [make_camera(k, H, W, device=dev) for k in range(8)]
bg = torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(1)
up = torch.rand((3, H, W), generator=g).to(dev)
def step1(k):
    (render(cams[k % 8], pc, Pipe, bg)["render"] * up).sum().backward()
up_d, up_a = torch.rand((1, H, W), generator=g).to(dev), torch.rand((1, H, W), generator=g).to(dev)
def step2(k):
    out = render(cams[k % 8], pc, Pipe, bg)
    ((out["render"] * up).sum() + (out["depth"] * up_d).sum() + (out["alpha"] * up_a).sum()).backward()
def step0(k):
    (get_render_label(cams[k % 8], pc, bg) * up).sum().backward()
for name, fn in (("mode1", step1), ("mode2", step2), ("mode0", step0), ("mode1", step1), ("mode2", step2)):
    try:
        for k in range(5): fn(k)
    except Exception as e:
        print(name, "skipped:", repr(e)[:200]); continue
    torch.cuda.synchronize()
    lib.profile_begin(16 * (iters + 4))
    for k in range(iters): fn(k)
    torch.cuda.synchronize()
    st = lib.profile_end()
    print(name, " ".join(f"{n}={1e3 * ms / c:.1f}us" for n, (ms, c) in st.items() if c and n.startswith("render")))
