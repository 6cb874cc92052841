"""GPU: the family of inputs behind the one parity miss on record (profiles/r5_fuzz_parity_seed777123.txt, draw 3869): splats of 300 px
radius and more whose CENTRE lies 200 px and more outside a thin image, so that every pixel they touch sits hundreds of pixels from the
centre.  The per-splat sums of the backward blend then run over terms with |dx| in the hundreds, and any formulation that cancels after
summing instead of per pixel shows it here first.

Every gradient row of every draw is held to the float64 oracle: within 1e-4 (max-norm relative: of the largest |entry| of the array), or --
where the float32 ORACLE itself is further than that from the float64 one -- within twice the float32 oracle's own distance.  Rows in the
tile list of a proven threshold flip are set aside as everywhere else (tests/common.py).  The three distances are printed per draw."""
import numpy as np
import pytest
import torch

from tests.common import make_inputs, seeded_grads, tile_culling, flip_pixels, gaussians_near_flips, check_images_isolating_flips
from tests.test_gpu_parity import hip_forward, hip_backward, oracle_forward, TOL

pytestmark = pytest.mark.gpu

# (N, H, W, scene seed, SH degree, mode, frame, scale multiplier, opacity shift): each holds at least one splat of radius >= 300 px whose
# centre is >= 200 px outside the image (found with the oracle's preprocess; asserted below)
DRAWS = [
    (70000, 105, 3, 21, 1, "sh_sr", 195, 4.0, 2.0),          # seed 777123, draw 3869 of tests/fuzz_parity.py: the recorded miss
    (20000, 105, 3, 0, 1, "sh_sr", 195, 8.0, 0.0), (20000, 105, 3, 5, 1, "sh_sr", 195, 8.0, 2.0), (20000, 105, 3, 1, 0, "sh_cov", 120, 8.0, 0.0),
    (20000, 200, 4, 0, 1, "sh_sr", 195, 8.0, 0.0), (20000, 200, 4, 1, 0, "col_sr", 195, 8.0, 2.0), (20000, 200, 4, 2, 1, "sh_sr", 195, 8.0, -2.0),
    (20000, 200, 4, 3, 0, "sh_cov", 120, 8.0, 0.0), (20000, 200, 4, 4, 1, "sh_sr", 195, 8.0, 2.0), (20000, 200, 4, 5, 0, "col_cov", 195, 8.0, 0.0),
    (20000, 200, 4, 0, 1, "sh_sr", 120, 4.0, 2.0), (20000, 200, 4, 3, 1, "sh_sr", 195, 4.0, 0.0),
    (20000, 2, 300, 0, 1, "sh_sr", 40, 8.0, 0.0), (20000, 2, 300, 4, 0, "sh_cov", 120, 8.0, 2.0),
    (20000, 150, 6, 0, 1, "sh_sr", 195, 8.0, 0.0), (20000, 150, 6, 3, 0, "col_sr", 120, 8.0, 2.0), (20000, 150, 6, 4, 1, "sh_sr", 195, 8.0, 2.0),
    (20000, 150, 6, 5, 1, "sh_sr", 195, 8.0, -2.0),
    (20000, 300, 3, 0, 1, "sh_sr", 195, 8.0, 0.0), (20000, 300, 3, 1, 1, "sh_sr", 40, 8.0, 2.0), (20000, 300, 3, 2, 0, "sh_cov", 40, 8.0, 0.0),
    (20000, 300, 3, 3, 1, "sh_sr", 120, 8.0, 2.0), (20000, 300, 3, 4, 0, "col_sr", 40, 8.0, 0.0), (20000, 300, 3, 5, 1, "sh_sr", 40, 8.0, 2.0),
    (20000, 300, 3, 0, 1, "sh_sr", 195, 4.0, 2.0), (20000, 300, 3, 4, 1, "sh_sr", 120, 4.0, 0.0),
]
NAMES = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"]


@pytest.mark.parametrize("N,H,W,seed,deg,mode,frame,smul,oshift", DRAWS)
def test_large_offscreen_centred_splats_vs_float64_oracle(N, H, W, seed, deg, mode, frame, smul, oshift):
    from oracle.oracle import Oracle
    from egogaussian_amd import _C
    dev = torch.device("cuda:0")
    d = make_inputs(N, H, W, seed, deg, mode, frame=frame, scale_mul=smul, opacity_shift=oshift)
    o, st = oracle_forward(d)
    xy, r = st["xy"], st["radii"]
    outside = np.maximum(np.maximum(np.maximum(-xy[:, 0], xy[:, 0] - (W - 1)), np.maximum(-xy[:, 1], xy[:, 1] - (H - 1))), 0.0)
    family = np.nonzero((r >= 300) & (outside >= 200) & (st["tiles_touched"] > 0))[0]
    assert family.size >= 1, "the draw holds no splat of the family"
    o64 = Oracle(np.float64, nthreads=8)
    st64 = o64.forward(**{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()})
    grads = seeded_grads(H, W, 7)
    gb, gb64 = o.backward(st, *grads), o64.backward(st64, *[g.double() for g in grads])
    with tile_culling(False):
        g, out = hip_forward(d, dev)
        hb = hip_backward(g, out, grads, dev)
    torch.cuda.synchronize()
    assert out[0] == st["R"] and np.array_equal(out[4].cpu().numpy(), st["radii"])
    iv = _C.image_views(out[7], W, H)
    flip_px = flip_pixels(out[1].cpu().numpy(), iv["final_T"].cpu().numpy(), st, iv["n_contrib"].cpu().numpy().view(np.uint32))
    check_images_isolating_flips((("color", out[1].cpu().numpy(), st["color"]), ("depth", out[2].cpu().numpy(), st["depth"]), ("alpha", out[3].cpu().numpy(), st["alpha"])),
                                 st, flip_px, TOL)
    near = gaussians_near_flips(st, flip_px, 0)
    rep, bad = [], []
    for name, h in zip(NAMES, hb):
        o32 = gb.get(name)
        if o32 is None or h.numel() == 0:
            continue
        a32 = np.asarray(o32, dtype=np.float64).reshape(N, -1); a64 = np.asarray(gb64[name], dtype=np.float64).reshape(N, -1)
        hh = h.cpu().numpy().astype(np.float64).reshape(N, -1)
        scale = float(np.abs(a64).max()) + 1e-30
        e_h64, e_h32, e_o = (np.abs(hh - a64).max(1) / scale, np.abs(hh - a32).max(1) / scale, np.abs(a32 - a64).max(1) / scale)
        e_h64[near] = 0.0                                            # (their own bound: check_grads_isolating_flips in the other parity files)
        miss = np.nonzero(e_h64 > np.maximum(TOL, 2.0 * e_o))[0]
        i = int(np.argmax(e_h64))
        f = int(family[np.argmax(e_h64[family])])
        rep.append(f"{name}: worst row {i} hip-f64 {e_h64[i]:.1e} (oracle32-f64 {e_o[i]:.1e}, hip-oracle32 {e_h32[i]:.1e}); worst family row {f} (radius {int(r[f])}, "
                   f"{outside[f]:.0f} px outside) hip-f64 {e_h64[f]:.1e} (oracle32-f64 {e_o[f]:.1e}, hip-oracle32 {e_h32[f]:.1e})")
        for m in miss[:4]:
            bad.append(f"{name} row {int(m)} (radius {int(r[m])}, centre {xy[m].tolist()}): hip-f64 {e_h64[m]:.2e} > max(1e-4, 2 x oracle32-f64 {e_o[m]:.2e}); hip {hh[m][:3]}, f64 {a64[m][:3]}")
    print(f"\n[{N}@{W}x{H} {mode} x{smul} frame {frame}] {family.size} family splats, {int(flip_px.sum())} flipped pixels\n   " + "\n   ".join(rep))
    assert not bad, "rows further from the float64 oracle than the bar AND than twice the float32 oracle's own distance:\n   " + "\n   ".join(bad)


def test_single_symmetric_splat_where_every_float32_evaluation_is_off():
    """The other miss the randomised sweep recorded (seed 9001, draw 10962, round 6): ONE Gaussian, 15 px radius, well inside a 128 x 239 image.
    Its footprint is symmetric about its centre, so dL/dmean2D is what is left of terms that cancel between pixels -- sum |gd dx| is ~7 000 x
    the sum -- and the rounding of the float32 TERMS (not of their summation: the same float32 terms added in float64 land in the same
    place) moves the result by 5-8e-4: float32 oracle 0.55225 with one thread and 0.55207 with eight, float64 oracle 0.55254.  No float32
    implementation can meet 1e-4 of the float64 result here; what is asserted is the rule of this file -- no further from the float64
    oracle than the bar or than twice the float32 oracle's own distance -- with the three distances printed."""
    from oracle.oracle import Oracle
    dev = torch.device("cuda:0")
    N, H, W, seed, deg, mode, frame, smul, oshift = 1, 239, 128, 728, 3, "sh_cov", 265, 8.0, -2.0
    d = make_inputs(N, H, W, seed, deg, mode, frame=frame, scale_mul=smul, opacity_shift=oshift)
    o, st = oracle_forward(d)
    assert int(st["radii"][0]) == 15
    o64 = Oracle(np.float64, nthreads=8)
    st64 = o64.forward(**{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()})
    grads = seeded_grads(H, W, 7)
    gb, gb64 = o.backward(st, *grads), o64.backward(st64, *[g.double() for g in grads])
    g, out = hip_forward(d, dev)
    hb = hip_backward(g, out, grads, dev)
    torch.cuda.synchronize()
    bad = []
    for name, h in zip(NAMES, hb):
        if gb.get(name) is None or h.numel() == 0:
            continue
        a32, a64 = np.asarray(gb[name], dtype=np.float64).reshape(N, -1), np.asarray(gb64[name], dtype=np.float64).reshape(N, -1)
        hh = h.cpu().numpy().astype(np.float64).reshape(N, -1)
        scale = float(np.abs(a64).max()) + 1e-30
        e_h64, e_o, e_h32 = np.abs(hh - a64).max() / scale, np.abs(a32 - a64).max() / scale, np.abs(hh - a32).max() / scale
        print(f"   {name}: hip-f64 {e_h64:.1e}, oracle32-f64 {e_o:.1e}, hip-oracle32 {e_h32:.1e}")
        if e_h64 > max(TOL, 2.0 * e_o):
            bad.append(name)
    assert not bad, bad
