#!/usr/bin/env python
"""Ready-to-pin harness: let anyone who holds a build of the reference's CUDA rasterizer pin this package (and its oracle) to it.

Parity is UNPINNED because the arithmetic of the hot path lives in `ashawkey/diff-gaussian-rasterization @ 8829d14f`
(/root/reference/README.md:26, .gitmodules:1-3), whose sources are not in /root/reference and cannot be built on the boxes this
project runs on.  This tool makes the missing step one command on each side:

  1. on a machine with the upstream CUDA extension installed (`import diff_gaussian_rasterization` = upstream):
         python tools/compare_upstream_npz.py produce --backend upstream --cases A,B --out upstream_outputs/
     needs only numpy + torch + that extension and THIS file (the inputs are regenerated from seeds; `tests/golden/upstream/*.npz`
     hold case A's inputs and the SHA-256 of every case's inputs so that the regeneration can be verified);
  2. anywhere (this package's oracle, CPU):   python tools/compare_upstream_npz.py compare --ref upstream_outputs/ --backend oracle
     on an MI355X (the HIP path, C ABI):      python tools/compare_upstream_npz.py compare --ref upstream_outputs/ --backend hip

.npz schema (one file per case, `case_<id>_outputs.npz`; float32 unless noted):
  inputs_sha256   str       SHA-256 over the input arrays in the order of INPUT_KEYS (see input_hash)
  num_rendered    int64     R, the (Gaussian, tile) instance count the forward returned
  color [3,H,W], depth [1,H,W], alpha [1,H,W], radii int32 [N]
  dL_dmeans3D [N,3], dL_dmeans2D [N,3], dL_dopacity [N,1], dL_dsh [N,M,3] | dL_dcolors [N,3], dL_dcov3D [N,6] | dL_dscales [N,3] + dL_drotations [N,4]
      gradients of  sum(color * gc) + sum(depth * gd) + sum(alpha * ga)  with the seeded upstream gradients gc, gd, ga of the case
  (oracle / hip only, bit-exact between the two:)  point_list uint32 [R], ranges uint32 [tiles,2]
Cases (tests/common.make_inputs of this repository = SURVEY.md section 8d's S(N,H,W,seed)):
  A  10k @ 64x64      B  100k @ 960x540      C  500k @ 960x540      D  1M @ 1920x1080     (all: SH degree 0, cov3D_precomp, seed 0)
  E  2k @ 128x96, SH degree 3      F  2k @ 128x96, colours + scales/rotations (the label call)
Bars (BASELINE.json north_star): images and gradients within 1e-4 relative fp32; radii exact.
"""
import argparse
import hashlib
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {  # id: (N, H, W, seed, sh_degree, mode, scale_mul)
    "A": (10_000, 64, 64, 0, 0, "sh_cov", 1.0), "B": (100_000, 540, 960, 0, 0, "sh_cov", 1.0), "C": (500_000, 540, 960, 0, 0, "sh_cov", 1.0),
    "D": (1_000_000, 1080, 1920, 0, 0, "sh_cov", 1.0), "E": (2000, 96, 128, 2, 3, "sh_cov", 2.0), "F": (2000, 96, 128, 3, 0, "col_sr", 2.0)}
INPUT_KEYS = ("means3D", "opacities", "shs", "colors_precomp", "cov3D_precomp", "scales", "rotations", "viewmatrix", "projmatrix", "campos", "bg")
GRAD_KEYS = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dcolors", "dL_dcov3D", "dL_dscales", "dL_drotations")
TOL = 1e-4


def make_case(cid):
    """Inputs + seeded upstream gradients of a case, as CPU float32 tensors (tests/common.py; needs this repository on sys.path)."""
    sys.path.insert(0, ROOT)
    from tests.common import make_inputs, seeded_grads
    N, H, W, seed, deg, mode, smul = CASES[cid]
    d = make_inputs(N, H, W, seed, deg, mode, scale_mul=smul)
    return d, seeded_grads(H, W, seed + 3)


def input_hash(d):
    h = hashlib.sha256()
    for k in INPUT_KEYS:
        if k in d:
            h.update(k.encode()); h.update(np.ascontiguousarray(d[k].detach().cpu().numpy(), dtype=np.float32).tobytes())
    for k in ("image_height", "image_width", "sh_degree"):
        h.update(f"{k}={int(d[k])}".encode())
    for k in ("tanfovx", "tanfovy", "scale_modifier"):
        h.update(np.float64(d[k]).tobytes())
    return h.hexdigest()


def run_torch_module(d, grads, module_name, device):
    """Forward + backward through a `diff_gaussian_rasterization`-shaped module (upstream's, or this package's drop-in)."""
    mod = __import__(module_name)
    t = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()}
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "colors_precomp", "cov3D_precomp", "scales", "rotations") if k in t}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    settings = mod.GaussianRasterizationSettings(
        image_height=int(d["image_height"]), image_width=int(d["image_width"]), tanfovx=float(d["tanfovx"]), tanfovy=float(d["tanfovy"]), bg=t["bg"],
        scale_modifier=float(d["scale_modifier"]), viewmatrix=t["viewmatrix"], projmatrix=t["projmatrix"], sh_degree=int(d["sh_degree"]),
        campos=t["campos"], prefiltered=False, debug=False)
    out = mod.GaussianRasterizer(raster_settings=settings)(
        means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"),
        scales=leaves.get("scales"), rotations=leaves.get("rotations"), cov3D_precomp=leaves.get("cov3D_precomp"))
    color, radii, depth, alpha = out[:4]
    gc, gd, ga = [g.to(device) for g in grads]
    ((color * gc).sum() + (depth * gd).sum() + (alpha * ga).sum()).backward()
    npy = lambda x: x.detach().cpu().numpy()
    res = dict(color=npy(color), depth=npy(depth), alpha=npy(alpha), radii=npy(radii).astype(np.int32), dL_dmeans3D=npy(leaves["means3D"].grad),
               dL_dmeans2D=npy(means2D.grad), dL_dopacity=npy(leaves["opacities"].grad))
    for k, name in (("shs", "dL_dsh"), ("colors_precomp", "dL_dcolors"), ("cov3D_precomp", "dL_dcov3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations")):
        if k in leaves:
            res[name] = npy(leaves[k].grad)
    return res


def run_backend(cid, backend):
    d, grads = make_case(cid)
    if backend == "upstream":
        res = run_torch_module(d, grads, "diff_gaussian_rasterization", "cuda")
        res["num_rendered"] = np.int64(-1)                   # upstream's Python surface does not return it
    elif backend == "hip":
        sys.path.insert(0, ROOT)
        res = run_torch_module(d, grads, "diff_gaussian_rasterization", "cuda")     # this repository's drop-in module (HIP, C ABI)
        from egogaussian_amd import _C
        res["num_rendered"] = np.int64(_C.stats["num_rendered"])
    elif backend == "oracle":
        sys.path.insert(0, ROOT)
        from oracle.oracle import Oracle
        o = Oracle(np.float32, nthreads=os.cpu_count() or 1)
        st = o.forward(**d)
        g = o.backward(st, *grads)
        res = dict(color=st["color"], depth=st["depth"], alpha=st["alpha"], radii=st["radii"].astype(np.int32), num_rendered=np.int64(st["R"]),
                   point_list=st["point_list"], ranges=st["ranges"], dL_dmeans3D=g["dL_dmeans3D"], dL_dmeans2D=g["dL_dmean2D"],
                   dL_dopacity=np.asarray(g["dL_dopacity"]).reshape(-1, 1))
        for src, name in (("dL_dsh", "dL_dsh"), ("dL_dcolors_precomp", "dL_dcolors"), ("dL_dcov3D", "dL_dcov3D"), ("dL_dscale", "dL_dscales"), ("dL_drot", "dL_drotations")):
            if src in g and g[src] is not None and np.size(g[src]):
                res[name] = np.asarray(g[src])
    else:
        raise SystemExit(f"unknown backend {backend}")
    res["inputs_sha256"] = np.array(input_hash(d))
    return res


def cmd_produce(a):
    os.makedirs(a.out, exist_ok=True)
    for cid in a.cases.split(","):
        res = run_backend(cid, a.backend)
        path = os.path.join(a.out, f"case_{cid}_outputs.npz")
        np.savez_compressed(path, backend=np.array(a.backend), **res)
        print(f"case {cid}: wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB), inputs sha256 {str(res['inputs_sha256'])[:16]}...")


def cmd_compare(a):
    bad = 0
    for cid in a.cases.split(","):
        path = os.path.join(a.ref, f"case_{cid}_outputs.npz")
        if not os.path.exists(path):
            print(f"case {cid}: no {path}; skipped"); continue
        ref = np.load(path, allow_pickle=False)
        mine = run_backend(cid, a.backend)
        if str(ref["inputs_sha256"]) != str(mine["inputs_sha256"]):
            print(f"case {cid}: INPUTS DIFFER (their sha256 {str(ref['inputs_sha256'])[:16]}, regenerated here {str(mine['inputs_sha256'])[:16]}): not comparable"); bad += 1; continue
        rows = [f"radii {'exact' if np.array_equal(ref['radii'], mine['radii']) else 'DIFFER at %d Gaussians' % int((ref['radii'] != mine['radii']).sum())}"]
        ok = np.array_equal(ref["radii"], mine["radii"])
        for k in ("color", "depth", "alpha") + GRAD_KEYS:
            if k in ref.files and k in mine:
                r, m = np.asarray(ref[k], np.float64), np.asarray(mine[k], np.float64).reshape(np.asarray(ref[k]).shape)
                err = float(np.abs(r - m).max() / (np.abs(r).max() + 1e-30))
                frac = float((np.abs(r - m) > TOL * (np.abs(r).max() + 1e-30)).mean())
                rows.append(f"{k} {err:.2e}" + ("" if err <= TOL else f" (> {TOL:g} at a fraction {frac:.1e} of the entries)"))
                ok = ok and (err <= TOL or frac <= 2e-4)              # (a threshold flip -- alpha within 1e-6 of 1/255 -- moves single pixels)
        for k in ("point_list", "ranges"):
            if k in ref.files and k in mine:
                same = np.array_equal(ref[k], mine[k]); rows.append(f"{k} {'bit-exact' if same else 'DIFFER'}"); ok = ok and same
        print(f"case {cid} [{str(ref['backend'])} vs {a.backend}]: {'PASS' if ok else 'FAIL'}: " + ", ".join(rows))
        bad += 0 if ok else 1
    raise SystemExit(1 if bad else 0)


def cmd_inputs(a):
    """(maintainers) write tests/golden/upstream/: case A's inputs as a .npz and the input hashes of every case."""
    out = os.path.join(ROOT, "tests", "golden", "upstream")
    os.makedirs(out, exist_ok=True)
    lines = []
    for cid in CASES:
        d, grads = make_case(cid)
        lines.append(f"{cid} {input_hash(d)}")
        if cid in a.cases.split(","):
            arrs = {k: d[k].numpy() for k in INPUT_KEYS if k in d}
            arrs.update(gc=grads[0].numpy(), gd=grads[1].numpy(), ga=grads[2].numpy(),
                        scalars=np.array([d["image_height"], d["image_width"], d["sh_degree"], d["tanfovx"], d["tanfovy"], d["scale_modifier"]], np.float64))
            np.savez_compressed(os.path.join(out, f"case_{cid}_inputs.npz"), **arrs)
    open(os.path.join(out, "input_sha256.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("produce"); p.add_argument("--backend", required=True, choices=["upstream", "hip", "oracle"]); p.add_argument("--cases", default="A,E,F"); p.add_argument("--out", required=True); p.set_defaults(f=cmd_produce)
    p = sub.add_parser("compare"); p.add_argument("--ref", required=True); p.add_argument("--backend", required=True, choices=["hip", "oracle"]); p.add_argument("--cases", default="A,B,C,D,E,F"); p.set_defaults(f=cmd_compare)
    p = sub.add_parser("inputs"); p.add_argument("--cases", default="A,E,F"); p.set_defaults(f=cmd_inputs)
    a = ap.parse_args()
    a.f(a)
