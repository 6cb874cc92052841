#!/usr/bin/env python
"""Bucketing debug driver: one forward of a parity case, lists against the oracle.
   python tests/dev/bin_debug.py N H W seed scale_mul cull [debug]        (EGS_SUPER_LOG=lsx,lsy forces the super-tile shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.common import make_inputs, tile_culling, check_culled_lists
from tests.test_gpu_parity import hip_forward, oracle_forward
from egogaussian_amd import _C
N, H, W, seed = [int(a) for a in sys.argv[1:5]]
smul, cull = float(sys.argv[5]), int(sys.argv[6])
debug = len(sys.argv) > 7 and sys.argv[7] == "1"
dev = torch.device("cuda:0")
d = make_inputs(N, H, W, seed, 0, "sh_cov", scale_mul=smul)
o, st = oracle_forward(d)
print("oracle R", st["R"], flush=True)
if os.environ.get("EGS_DEBUG_CAP"):
    _C.set_capacity_hint(int(os.environ["EGS_DEBUG_CAP"]), dev)
for rep in range(2):
    with tile_culling(bool(cull)):
        g, out = hip_forward(d, dev, debug=debug)
    torch.cuda.synchronize()
    print("rep", rep, "forward done R", out[0], "capacity", _C.stats["capacity"], "retries", _C.stats["retries"], flush=True)
R = out[0]
bv = _C.binning_views(out[6], N, R, W, H, _C.stats["capacity"]); iv = _C.image_views(out[7], W, H)
pl = bv["point_list"].cpu().numpy().view(np.uint32); rng = iv["ranges"].cpu().numpy().view(np.uint32)
print("total_view", int(_C.stats["total_view"].item()))
if cull:
    kept, dropped = check_culled_lists(st, rng, pl, H, W)
    print("culled lists ok: kept", kept, "dropped", dropped)
else:
    print("ranges equal", np.array_equal(rng, st["ranges"]), "point_list equal", np.array_equal(pl, st["point_list"]))
    if not np.array_equal(rng, st["ranges"]):
        bad = np.nonzero((rng != st["ranges"]).any(1))[0]
        print("first bad tiles", bad[:10], rng[bad[:5]], st["ranges"][bad[:5]])
    elif not np.array_equal(pl, st["point_list"]):
        bad = np.nonzero(pl != st["point_list"])[0]
        print("mismatching positions", len(bad), bad[:10], pl[bad[:10]], st["point_list"][bad[:10]])
print("image max abs diff", float(np.abs(out[1].cpu().numpy() - st["color"]).max()))
