"""CPU: the C oracle against the independent differentiable torch restatement (autograd = gradient oracle) and
against closed-form / structural invariants.  These pin the checker itself."""
import math

import numpy as np
import pytest
import torch

from oracle.oracle import Oracle
from oracle.raster_torch import rasterize_torch
from tests.common import make_inputs, seeded_grads, rel_err

CASES = [  # N, H, W, seed, deg, mode, scale_mul, opacity_shift
    (500, 48, 80, 0, 0, "sh_cov", 3.0, 0.0),
    (500, 48, 80, 1, 3, "sh_sr", 3.0, 0.0),
    (2500, 48, 48, 2, 0, "col_sr", 6.0, 4.0),    # opaque splats: alpha clamps at 0.99 and pixels saturate (T < 1e-4)
    (400, 33, 50, 3, 1, "col_cov", 4.0, 1.0),     # ragged image
    (300, 32, 32, 4, 2, "sh_cov", 40.0, 0.0),     # huge splats: frustum clamp of the Jacobian is exercised
]


def _run_both(d, dtype_np, dtype_t):
    dd = {k: (v.to(dtype_t) if torch.is_tensor(v) else v) for k, v in d.items()}
    leaves = [k for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp") if k in dd]
    for k in leaves:
        dd[k] = dd[k].clone().requires_grad_(True)
    m2d = torch.zeros(dd["means3D"].shape[0], 3, dtype=dtype_t, requires_grad=True)
    col, radii, dep, alp, aux = rasterize_torch(means2D=m2d, **dd)
    H, W = d["image_height"], d["image_width"]
    gc, gd, ga = seeded_grads(H, W, 3, dtype_t)
    ((col * gc).sum() + (dep * gd).sum() + (alp * ga).sum()).backward()
    o = Oracle(dtype_np)
    st = o.forward(**{k: (v.detach() if torch.is_tensor(v) else v) for k, v in dd.items()})
    g = o.backward(st, gc, gd, ga)
    return dd, m2d, (col, radii, dep, alp, aux), st, g


@pytest.mark.parametrize("N,H,W,seed,deg,mode,smul,oshift", CASES)
def test_c_oracle_matches_autograd_fp64(N, H, W, seed, deg, mode, smul, oshift):
    d = make_inputs(N, H, W, seed, deg, mode, scale_mul=smul, opacity_shift=oshift)
    dd, m2d, (col, radii, dep, alp, aux), st, g = _run_both(d, np.float64, torch.float64)
    assert st["R"] > 0
    assert np.array_equal(st["radii"], radii.numpy())
    assert np.array_equal(st["keys"], aux["keys"]) and np.array_equal(st["point_list"], aux["point_list"].astype(np.uint32))
    assert np.array_equal(st["n_contrib"], aux["n_contrib"].numpy().astype(np.uint32))
    for a, b in ((st["color"], col), (st["depth"], dep), (st["alpha"], alp)):
        assert rel_err(a, b.detach().numpy()) < 1e-12
    pairs = [("dL_dmeans3D", "means3D"), ("dL_dopacity", "opacities"), ("dL_dsh", "shs"), ("dL_dcolors_precomp", "colors_precomp"),
             ("dL_dscale", "scales"), ("dL_drot", "rotations"), ("dL_dcov3D", "cov3D_precomp")]
    for gname, leaf in pairs:
        if leaf in dd:
            assert rel_err(g[gname].reshape(dd[leaf].shape), dd[leaf].grad.numpy()) < 1e-10, gname
    assert rel_err(g["dL_dmean2D"], m2d.grad.numpy()) < 1e-10


def test_saturating_case_really_saturates():
    d = make_inputs(2500, 48, 48, 2, 0, "col_sr", scale_mul=6.0, opacity_shift=4.0)
    st = Oracle(np.float32).forward(**d)
    assert (st["final_T"] < 1e-3).mean() > 0.5                # most pixels hit the T < 1e-4 stop
    assert (d["opacities"] > 0.99).float().mean() > 0.2       # and many splats clamp alpha at 0.99


@pytest.mark.parametrize("N,H,W,seed,deg,mode,smul,oshift", CASES[:3])
def test_c_oracle_matches_autograd_fp32(N, H, W, seed, deg, mode, smul, oshift):
    d = make_inputs(N, H, W, seed, deg, mode, scale_mul=smul, opacity_shift=oshift)
    dd, m2d, (col, radii, dep, alp, aux), st, g = _run_both(d, np.float32, torch.float32)
    assert np.array_equal(st["radii"], radii.numpy())
    assert rel_err(st["color"], col.detach().numpy()) < 1e-4
    assert rel_err(g["dL_dmeans3D"], dd["means3D"].grad.numpy()) < 1e-3
    assert rel_err(g["dL_dmean2D"], m2d.grad.numpy()) < 1e-3


def test_single_isotropic_gaussian_closed_form():
    """One isotropic Gaussian on the optical axis: alpha(px) = o * exp(-r^2 / (2 (s^2 f^2 / z^2 + 0.3)))."""
    H = W = 64
    z, s, o = 4.0, 0.05, 0.8
    tan = math.tan(math.radians(30))
    f = W / (2 * tan)
    from egogaussian_amd.scene_synth import SynthCamera
    cam = SynthCamera(np.eye(4), H, W, 2 * math.atan(tan), 2 * math.atan(tan))
    st = Oracle(np.float64).forward(
        means3D=np.array([[0.0, 0.0, z]]), opacities=np.array([o]), colors_precomp=np.array([[1.0, 0.5, 0.25]]),
        scales=np.array([[s, s, s]]), rotations=np.array([[1.0, 0, 0, 0]]), viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, campos=cam.camera_center, bg=np.zeros(3), image_height=H, image_width=W,
        tanfovx=tan, tanfovy=tan)
    var = (s * f / z) ** 2 + 0.3
    cx = ((0 + 1) * W - 1) * 0.5
    yy, xx = np.mgrid[0:H, 0:W]
    r2 = (xx - cx) ** 2 + (yy - cx) ** 2
    expect = o * np.exp(-0.5 * r2 / var)
    expect[expect < 1 / 255] = 0
    tiles = st["rects"][0]
    mask = np.zeros((H, W), bool)
    mask[tiles[1] * 16:tiles[3] * 16, tiles[0] * 16:tiles[2] * 16] = True      # only touched tiles are evaluated
    assert np.abs(st["alpha"][0] - expect * mask).max() < 1e-4   # projmatrix is fp32 (scene/cameras.py builds it so)
    assert np.abs(st["color"][1] - 0.5 * expect * mask).max() < 1e-4
    assert np.abs(st["depth"][0] - z * expect * mask).max() < 1e-3
    assert st["radii"][0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))   # lambda = mid + sqrt(max(0.1, mid^2 - det))


def test_structural_invariants():
    d = make_inputs(800, 64, 96, 9, 0, "sh_cov", scale_mul=3.0)
    o = Oracle(np.float32)
    st = o.forward(**d)
    assert int(st["tiles_touched"].sum()) == st["R"] == len(st["keys"])
    assert np.array_equal((st["radii"] > 0), st["tiles_touched"] > 0)
    assert np.all(st["keys"][1:] >= st["keys"][:-1])
    # ranges partition the list by tile id
    t = (st["keys"] >> np.uint64(32)).astype(np.int64)
    for tile, (a, b) in enumerate(st["ranges"]):
        assert np.all(t[a:b] == tile)
    assert int((st["ranges"][:, 1] - st["ranges"][:, 0]).sum()) == st["R"]
    # alpha + final_T = 1 ; colour(bg) - colour(0) = final_T * bg
    assert np.abs(st["alpha"][0] + st["final_T"] - 1).max() < 1e-5
    d0 = dict(d); d0["bg"] = torch.zeros(3)
    st0 = o.forward(**d0)
    assert np.abs((st["color"] - st0["color"]) - st["final_T"][None] * d["bg"].numpy()[:, None, None]).max() < 1e-6
    # permuting the Gaussians leaves the image unchanged (no depth ties in this scene)
    perm = np.random.default_rng(0).permutation(800)
    dp = {k: (v[perm] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == 800 else v) for k, v in d.items()}
    stp = o.forward(**dp)
    assert rel_err(stp["color"], st["color"]) < 1e-5
    assert np.array_equal(stp["radii"], st["radii"][perm])


def test_sort_is_stable_on_depth_ties():
    """Equal depth bits inside one tile keep Gaussian-index order (stable sort on (tile, depth))."""
    d = make_inputs(64, 32, 32, 5, 0, "col_sr", scale_mul=6.0)
    d["means3D"][:, 2] = 5.0                  # identical depth for all; frame 0 camera = identity
    from egogaussian_amd.scene_synth import make_camera
    cam = make_camera(0, 32, 32)
    d.update(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center)
    st = Oracle(np.float32).forward(**d)
    for a, b in st["ranges"]:
        pl = st["point_list"][a:b].astype(np.int64)
        assert np.all(np.diff(pl) > 0)


def test_empty_inputs():
    d = make_inputs(10, 32, 32, 0, 0, "sh_cov")
    d["means3D"][:, 2] = -1.0
    st = Oracle(np.float32).forward(**d)
    assert st["R"] == 0 and np.all(st["radii"] == 0)
    assert np.allclose(st["color"], d["bg"].numpy()[:, None, None])
    g = Oracle(np.float32).backward(st, *seeded_grads(32, 32))
    assert all(np.all(v == 0) for v in g.values() if v is not None)


def test_flip_rule_accepts_only_pixels_with_a_threshold_adjacent_pair():
    """tests/common.py flip_pixels / flip_cause (the parity suite's threshold-flip rule) on the CPU: an output equal to the
    oracle's has no flips; a pixel that is off with NO threshold-adjacent pair in the float64 re-walk of its chain fails the call; the same
    difference on a pixel whose chain holds a pair with alpha within 1e-5 (relative) of 1/255 is accepted as a flip."""
    from tests.common import flip_pixels, flip_cause
    N, H, W = 400, 32, 48
    d = make_inputs(N, H, W, 7, 0, "col_sr", scale_mul=3.0)
    o = Oracle(np.float32)
    st = o.forward(**d)
    assert not flip_pixels(st["color"], st["final_T"], st, st["n_contrib"]).any()
    # a pixel whose chain has no pair anywhere near a threshold (most have none): off by 1e-3 -> refused
    cand = [(y, x) for y in range(H) for x in range(W) if st["n_contrib"][y, x] > 0 and flip_cause(st, y, x) is None]
    assert len(cand) > H * W // 2
    y, x = cand[len(cand) // 2]
    bad = st["color"].copy(); bad[:, y, x] += 1e-3
    with pytest.raises(AssertionError, match="NO threshold-adjacent pair"):
        flip_pixels(bad, st["final_T"], st, st["n_contrib"])
    # the same pixel off by LESS than two float32 evaluations of its chain can differ by (pixel_account's bound, a few 1e-6 here): float
    # noise -- accepted, and NOT marked: it stays under the 1e-4 bar like any other pixel and relaxes no Gaussian's bound
    from tests.common import pixel_account, FLIP_DETECT
    _, noise_c, noise_T = pixel_account(st, y, x)
    assert 1e-8 < noise_c < 2e-5 and noise_T < 2e-5
    hum = st["color"].copy(); hum[:, y, x] += FLIP_DETECT * float(np.abs(st["color"]).max()) + 0.5 * noise_c
    rep = {}
    assert not flip_pixels(hum, st["final_T"], st, st["n_contrib"], report=rep).any() and rep["noise"] == 1 and rep["flips"] == 0
    hum[:, y, x] += noise_c                                          # ... and just beyond that reach: refused
    with pytest.raises(AssertionError, match="NO threshold-adjacent pair"):
        flip_pixels(hum, st["final_T"], st, st["n_contrib"])
    # put the first list entry of that pixel's tile that reaches the pixel at all exactly on the alpha threshold (alpha = (1 + 3e-6) / 255) by
    # editing its opacity
    t = (y // 16) * ((W + 15) // 16) + x // 16
    for g in st["point_list"][int(st["ranges"][t, 0]):int(st["ranges"][t, 1])].astype(int):
        co = st["conic_opacity"][g].astype(np.float64)
        dx, dy = float(st["xy"][g, 0]) - x, float(st["xy"][g, 1]) - y
        G = math.exp(min(0.0, -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy))
        if G > 0.1:
            break
    assert G > 0.1
    g = int(g)
    d2 = dict(d); d2["opacities"] = d["opacities"].clone(); d2["opacities"][g] = (1.0 + 3e-6) / 255.0 / G
    st2 = o.forward(**d2)
    cause = flip_cause(st2, y, x)
    assert cause is not None and cause.startswith("alpha threshold") and f"Gaussian {g})" in cause
    off = st2["color"].copy(); off[:, y, x] += 1e-3
    px = flip_pixels(off, st2["final_T"], st2, st2["n_contrib"])
    assert px[y, x] and px.sum() == 1
