#!/usr/bin/env python
"""Randomised parity sweep: HIP path vs the C oracle on random (N, image size, colour mode, SH degree, coefficient count,
splat size, camera) draws -- lists bit-exact with tile culling off, an ordered sub-list with it on, images and gradients
within the test tolerances (a handful of entries may sit on the other side of an alpha threshold: the two implementations
round alpha differently in the last place).  Complements tests/test_gpu_parity.py (fixed cases); run it for as long as you like:
    python tests/fuzz_parity.py [seconds] [seed]
(lives under tests/ because it uses the oracle, which is test infrastructure; tests/test_gpu_sweep.py runs a fixed-seed slice
of it -- 320 draws -- under pytest -m gpu)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.common import flip_pixels, check_grads_isolating_flips, check_images_isolating_flips, make_inputs, seeded_grads, rel_err, outlier_fraction, tile_culling, check_culled_lists   # noqa: E402
from tests.test_gpu_parity import hip_forward, hip_backward, oracle_forward, TOL                                     # noqa: E402
from egogaussian_amd import _C                                                                                       # noqa: E402


def run_draws(seed=0, budget_s=None, n_draws=None, dev=None, verbose=False, keep_going=False):
    """Random draws until `n_draws` are done or `budget_s` seconds have passed.  -> (cases run, worst max-relative gradient errors)."""
    rng = np.random.default_rng(int(seed))
    dev = torch.device("cuda:0") if dev is None else dev
    t_end, n_cases, worst = (time.time() + budget_s) if budget_s else None, 0, {}
    while (n_draws is None or n_cases < n_draws) and (t_end is None or time.time() < t_end):
        N = int(rng.choice([1, 2, 63, 64, 65, 300, 1023, 1025, 2500, 7000, 20000, 70000]))
        H, W = int(rng.integers(1, 300)), int(rng.integers(1, 420))
        mode = str(rng.choice(["sh_cov", "sh_sr", "col_sr", "col_cov"]))
        deg = int(rng.integers(0, 4)) if mode.startswith("sh") else 0
        active = int(rng.integers(0, deg + 1))
        smul = float(rng.choice([0.5, 1.0, 2.0, 4.0, 8.0]))
        frame = int(rng.integers(0, 300))
        cull = bool(rng.integers(0, 2))
        split = bool(rng.integers(0, 2)) and deg > 0                      # hand the coefficients over as (dc, rest)
        tag = f"N={N} {W}x{H} {mode} M={(deg + 1) ** 2} active={active} scale x{smul} frame {frame} culling {'on' if cull else 'off'}{' split-SH' if split else ''}"
        if verbose:
            print(tag, flush=True)
        d = make_inputs(N, H, W, int(rng.integers(0, 1000)), deg, mode, frame=frame, scale_mul=smul, opacity_shift=float(rng.choice([0.0, 2.0, -2.0])))
        d["sh_degree"] = active
        try:
            strict = _one_draw(d, N, H, W, cull, split, active, tag, dev, worst, keep_going, n_cases)
        except AssertionError as err:
            if not keep_going:
                raise
            worst.setdefault("_failed_draws", []).append(f"draw {n_cases}: {tag}: {str(err)[:420]}")     # (a long run reports them all; pytest's slice stops at the first)
            strict = False
        n_cases += 1
        worst["_strict_draws"] = worst.get("_strict_draws", 0) + (1 if strict else 0)
    return n_cases, worst


def _one_draw(d, N, H, W, cull, split, active, tag, dev, worst, keep_going, n_cases):
    """One draw of run_draws against the oracle; -> whether it met 1e-4 everywhere outright (no threshold flip)."""
    if True:
        o, st = oracle_forward(d)
        with tile_culling(cull):
            if split:
                g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
                e0 = torch.empty(0, device=dev)
                dc, rest = g["shs"][:, :1].contiguous(), g["shs"][:, 1:].contiguous()
                out = _C.rasterize_gaussians(g["bg"], g["means3D"], e0, g["opacities"], g.get("scales", e0), g.get("rotations", e0), 1.0,
                                             g.get("cov3D_precomp", e0), g["viewmatrix"], g["projmatrix"], g["tanfovx"], g["tanfovy"], H, W, dc, active,
                                             g["campos"], False, False, 0, rest)
            else:
                g, out = hip_forward(d, dev)
            R, color, depth, alpha, radii, geom, binning, img = out
            torch.cuda.synchronize()
            assert R == st["R"], tag
            assert np.array_equal(radii.cpu().numpy(), st["radii"]), tag
            if R:
                bv = _C.binning_views(binning, N, R, W, H, _C.stats["capacity"]); iv = _C.image_views(img, W, H)
                pl = bv["point_list"].cpu().numpy().view(np.uint32); rngs = iv["ranges"].cpu().numpy().view(np.uint32)
                if cull:
                    check_culled_lists(st, rngs, pl, H, W)
                else:
                    assert np.array_equal(pl, st["point_list"]) and np.array_equal(rngs, st["ranges"]), tag
            # pixels on the other side of an alpha / transmittance threshold than the oracle's (v_exp_f32 vs glibc expf in the last place),
            # found with a threshold far below the parity bar AND proven: the float64 re-walk of such a pixel's chain must hold a threshold-adjacent
            # pair, else flip_pixels fails the draw (tests/common.py flip_cause); every other pixel of every plane is held to 1e-4
            flip_px = np.zeros((H, W), dtype=bool)
            if R:
                flip_px = flip_pixels(color.cpu().numpy(), iv["final_T"].cpu().numpy(), st, None if cull else iv["n_contrib"].cpu().numpy().view(np.uint32))
            check_images_isolating_flips((("colour", color.cpu().numpy(), st["color"]), ("depth", depth.cpu().numpy(), st["depth"]), ("alpha", alpha.cpu().numpy(), st["alpha"])),
                                         st, flip_px, TOL, what=tag)
            flips = int(flip_px.sum())
            strict = flips == 0                                  # this draw meets the 1e-4 bar outright: no outlier anywhere (no threshold flip)
            grads = seeded_grads(H, W, 7)
            if split:
                gc, gd, ga = [x.to(dev) for x in grads]
                full = _C.rasterize_gaussians_backward(g["bg"], g["means3D"], radii, e0, g.get("scales", e0), g.get("rotations", e0), 1.0,
                                                       g.get("cov3D_precomp", e0), g["viewmatrix"], g["projmatrix"], g["tanfovx"], g["tanfovy"], gc, gd, ga,
                                                       dc, active, g["campos"], geom, R, binning, img, alpha, False, 0, rest)
                hb = list(full[:8]); hb[5] = torch.cat((full[5], full[8]), dim=1)
            else:
                hb = hip_backward(g, out, grads, dev)
            torch.cuda.synchronize()
        gb = o.backward(st, *grads)
        names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"]
        for name, h in zip(names, hb):
            ora = gb.get(name)
            if ora is None or h.numel() == 0:
                continue
            hh = h.cpu().numpy().reshape(ora.shape)
            assert np.isfinite(hh).all() == np.isfinite(ora).all(), f"{tag}: {name} finiteness"
            e = rel_err(hh, ora)
            worst[name] = max(worst.get(name, 0.0), e)
            strict = strict and e <= TOL
        # Gaussians away from every flipped pixel: the north star's 1e-4, whatever the frame; the ones in a flipped pixel's tile list: a
        # flipped pair moves the gradients of ITS splat by that pair's whole share -- for a faint splat that reaches three or four pixels a
        # few per cent of its dL/dopacity (seed 4242, draw 3215: one flipped pixel, 2.3 %; tests/dev/fuzz_repro.py replays a draw)
        if np.isfinite(st["color"]).all() and all(np.isfinite(v).all() for v in gb.values() if v is not None):
            try:
                _, far, _ = check_grads_isolating_flips(names, hb, gb, st, flip_px, TOL, share=5e-2, what=tag)
            except AssertionError as first:
                # The float32 oracle adds a splat's thousands of pixel terms in whatever order its OpenMP threads reach them: on a splat that
                # covers the whole frame its own sum moves by 1-4e-4 of the array's maximum from run to run (seed 20260930, draw 227: Gaussian
                # 67346, radius 377 px on a 222x82 image -- oracle32 1548.96 / 1548.40 in two runs, oracle64 1548.242, HIP 1548.242), and the HIP
                # path's float32 sums (wave reductions + atomics) carry the same kind of noise.  The rows that missed the bar against the float32
                # oracle -- and only those -- are arbitrated by the float64 oracle of the same draw: within the bar, or within 3 x the bar
                # (far_cap) for a splat of 36 px radius or more (>= 4 000 pixel terms).  Any row that misses THAT fails the draw.
                from oracle.oracle import Oracle
                from tests.common import gaussians_near_flips
                o64 = Oracle(np.float64, nthreads=8)
                d64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
                st64 = o64.forward(**d64)
                gb64 = o64.backward(st64, *[x.double() for x in grads])
                near = gaussians_near_flips(st, flip_px, 0)
                far = 0.0
                for name, h in zip(names, hb):
                    o32, o64g = gb.get(name), gb64.get(name)
                    if o32 is None or h is None or h.numel() == 0:
                        continue
                    hh = h.detach().cpu().numpy().reshape(o32.shape).astype(np.float64)
                    a32, a64 = np.asarray(o32, dtype=np.float64), np.asarray(o64g, dtype=np.float64).reshape(o32.shape)
                    sc = float(np.abs(a32).max()) + 1e-30
                    e32 = np.abs(hh - a32).reshape(a32.shape[0], -1).max(1) / sc
                    e64 = np.abs(hh - a64).reshape(a32.shape[0], -1).max(1) / sc
                    e32[near[near < a32.shape[0]]] = 0.0                 # (rows near a flipped pixel have their own bound, checked above)
                    rows = np.nonzero(e32 >= TOL)[0]
                    # frame-filling splats: 3 x the bar, or -- when the float32 ORACLE itself is further than that from the float64 one (seed 777123,
                    # draw 129: radius 300 px on 412x270, 111 k pixel terms: oracle32 18145.8, HIP 18149.1, oracle64 18155.9) -- no worse than
                    # 1.5 x the float32 oracle's own distance from the float64 result
                    o_err = np.abs(a32 - a64).reshape(a32.shape[0], -1).max(1) / sc
                    # ... and, whatever the radius, where the float32 ORACLE is itself further than the bar from the float64 one the row is held to
                    # 1.5 x that distance (seed 9001, draw 10962: ONE Gaussian of 15 px radius, symmetric footprint -- sum |gd dx| is 7 000 x the
                    # sum: oracle32 0.55225 with one thread, 0.55207 with eight, HIP 0.55220, the float32 terms summed in float64 0.55215,
                    # oracle64 0.55254: every float32 evaluation is 5-8e-4 off; tests/test_gpu_offscreen.py holds that draw with all three distances)
                    bar = np.where(np.asarray(st["radii"])[rows] >= 36, np.maximum(3.0 * TOL, 1.5 * o_err[rows]), np.maximum(TOL, 1.5 * o_err[rows]))
                    if (e64[rows] >= bar).any():
                        if not keep_going:
                            raise first
                        worst.setdefault("_failed_draws", []).append(f"draw {n_cases}: {str(first)[:420]}")     # (a long run reports them all; pytest's slice stops here)
                        break
                    far = max(far, float(np.where(e32 >= TOL, 0.0, e32).max()))
                else:
                    worst["_arbitrated_by_f64_oracle"] = worst.get("_arbitrated_by_f64_oracle", 0) + 1
            worst["_far_from_flips"] = max(worst.get("_far_from_flips", 0.0), far)
        return strict


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    n_cases, worst = run_draws(int(sys.argv[2]) if len(sys.argv) > 2 else 0, budget_s=budget, keep_going=True)
    failed = worst.pop("_failed_draws", [])
    strict = worst.pop("_strict_draws", 0)
    arb = worst.pop("_arbitrated_by_f64_oracle", 0)
    for f in failed:
        print("MISSED THE BAR (also against the float64 oracle):", f)
    print(f"{n_cases} random cases run in {budget:.0f} s, {len(failed)} of them missed the gradient bar (listed above), {strict} of them with every image and gradient within {TOL:g} outright (no threshold flip), "
          f"{arb} after the float64 oracle arbitrated a row the float32 oracle's own accumulation noise had put over the bar; "
          "worst max-relative gradient errors: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    sys.exit(1 if failed else 0)                   # (a run with missed draws must not exit 0)
