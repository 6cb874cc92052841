#!/usr/bin/env python
"""(uses the oracle)  Replays the random draws of tests/fuzz_parity.py for a seed until the draw whose tag contains `needle`, runs that one
case and prints what differs from the oracle (flipped pixels, the worst gradient entries).  EGS_RASTER_LIB selects an A/B library.
    python tests/dev/fuzz_repro.py <seed> "<needle>" """
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.common import make_inputs, seeded_grads, rel_err, outlier_fraction, tile_culling
from tests.test_gpu_parity import hip_forward, hip_backward, oracle_forward, TOL
from egogaussian_amd import _C
seed, needle = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
for it in range(200000):
    N = int(rng.choice([1, 2, 63, 64, 65, 300, 1023, 1025, 2500, 7000, 20000, 70000]))
    H, W = int(rng.integers(1, 300)), int(rng.integers(1, 420))
    mode = str(rng.choice(["sh_cov", "sh_sr", "col_sr", "col_cov"]))
    deg = int(rng.integers(0, 4)) if mode.startswith("sh") else 0
    active = int(rng.integers(0, deg + 1))
    smul = float(rng.choice([0.5, 1.0, 2.0, 4.0, 8.0]))
    frame = int(rng.integers(0, 300))
    cull = bool(rng.integers(0, 2))
    split = bool(rng.integers(0, 2)) and deg > 0
    tag = f"N={N} {W}x{H} {mode} M={(deg + 1) ** 2} active={active} scale x{smul} frame {frame} culling {'on' if cull else 'off'}{' split-SH' if split else ''}"
    s_in, osh = int(rng.integers(0, 1000)), float(rng.choice([0.0, 2.0, -2.0]))
    if needle not in tag:
        continue
    print("draw", it, tag, "scene seed", s_in, "opacity shift", osh)
    d = make_inputs(N, H, W, s_in, deg, mode, frame=frame, scale_mul=smul, opacity_shift=osh)
    d["sh_degree"] = active
    o, st = oracle_forward(d)
    o64 = __import__("oracle.oracle", fromlist=["Oracle"]).Oracle(np.float64, nthreads=8)
    d64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
    st64 = o64.forward(**d64)
    grads = seeded_grads(H, W, 7)
    with tile_culling(cull):
        g, out = hip_forward(d, dev)
        hb = hip_backward(g, out, grads, dev)
    torch.cuda.synchronize()
    iv = _C.image_views(out[7], W, H)
    col = out[1].cpu().numpy()
    flip = (np.abs(col - st["color"]) > TOL * np.abs(st["color"]).max()).any(0) | (np.abs(iv["final_T"].cpu().numpy() - st["final_T"]) > TOL)
    print("R", out[0], "pixels off (colour / final_T):", int(flip.sum()), np.argwhere(flip)[:5].tolist())
    gb = o.backward(st, *grads)
    gb64 = o64.backward(st64, *[x.double() for x in grads])
    for name, h in zip(["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"], hb):
        ora = gb.get(name)
        if ora is None or h.numel() == 0:
            continue
        hh = h.cpu().numpy().reshape(ora.shape)
        err = np.abs(hh - ora); sc = np.abs(ora).max()
        i = np.unravel_index(err.argmax(), err.shape)
        print(f"  {name}: max rel {err.max() / sc:.3e} at {i}: hip {hh[i]:.6e} oracle32 {ora[i]:.6e} oracle64 {gb64[name].reshape(ora.shape)[i]:.6e}; "
              f"oracle32 vs oracle64 max rel {np.abs(ora - gb64[name].reshape(ora.shape)).max() / sc:.3e}; hip vs oracle64 {np.abs(hh - gb64[name].reshape(ora.shape)).max() / sc:.3e}")
    break
