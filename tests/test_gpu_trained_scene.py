"""GPU: the shapes a TRAINED scene adds to the path (/root/reference/trainers/train_static.py:129-133 densifies every 100 iterations;
/root/reference/trainers/fine_all.py:93 and trainers/eval_metric.py:107 render such models): a few screen-filling splats among many
small and faint ones.  bench_data/trained_scene.npz is the model examples/train_synth.py ends with after the reference's 30 000-iteration
schedule (tools/make_trained_scene.py); the synthetic cases build the same shapes in small.

What is specific to them in the HIP path and therefore checked here against the oracle:
  * HOT Gaussians (csrc/egs_common.h): a splat whose alpha >= 1/255 box covers 256 tiles or more spreads its backward sums over
    replica accumulator lines -- codes ranked per 256-Gaussian workgroup, a budget of seven per workgroup, the rest on their own line;
  * the bucketing walk restricted to the tiles of that box (binning.hip) -- lists stay the reference's with culling off."""
import math
import os

import numpy as np
import pytest
import torch

from tests.common import flip_pixels, check_images_isolating_flips, make_inputs, seeded_grads, rel_err, outlier_fraction, tile_culling
from tests.test_gpu_parity import hip_forward, hip_backward, oracle_forward, _dev, TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"]


def _hot_codes(geom, N, radii):
    """HOT code of every visible Gaussian (a culled one has no record), read back from the record's spare box bits
    (csrc/egs_common.h egs_hot_code)."""
    from egogaussian_amd import _C
    rec = _C.geom_views(geom, N)["rec"].cpu().numpy()
    bx, by = rec[:, 10].view(np.uint32), rec[:, 11].view(np.uint32)
    return np.where(radii.cpu().numpy() > 0, ((bx >> 15) & 1) | ((bx >> 30) & 2) | ((by >> 13) & 4), 0)


def _check_grads(hb, gb, st, flip_px, what):
    """Gaussians away from every threshold flip at 1e-4, the ones in a flipped pixel's tile list by a pair's share (tests/common.py)."""
    from tests.common import check_grads_isolating_flips
    return check_grads_isolating_flips(NAMES, hb, gb, st, flip_px, TOL, what=what)[0]


@pytest.mark.parametrize("cull", [False, True], ids=["reference-lists", "tile-culling"])
def test_hot_gaussians_accumulate_through_replica_lines(cull):
    """Twelve screen-filling splats at the head of the array (one workgroup's budget is seven: five stay on their own line), two more in
    another workgroup, 6 000 ordinary ones: every gradient against the oracle, hot codes as documented."""
    from egogaussian_amd import _C
    dev = _dev()
    N, H, W = 6000, 320, 512                                          # 20 x 32 = 640 tiles
    d = make_inputs(N, H, W, 21, 0, "col_sr", scale_mul=1.5)
    big = list(range(12)) + [3000, 3001]
    gen = torch.Generator().manual_seed(9)
    d["means3D"][big, 0] = 0.6 * (torch.rand(len(big), generator=gen) - 0.5)
    d["means3D"][big, 1] = 0.4 * (torch.rand(len(big), generator=gen) - 0.5)
    d["means3D"][big, 2] = 3.0 + 6.0 * torch.rand(len(big), generator=gen)
    d["scales"][big] = 2.0 + 2.0 * torch.rand(len(big), 3, generator=gen)
    d["opacities"][big] = (0.02 + 0.3 * torch.rand(len(big), generator=gen)).view(-1, *d["opacities"].shape[1:])
    o, st = oracle_forward(d)
    with tile_culling(cull):
        g, out = hip_forward(d, dev, debug=True)
        grads = seeded_grads(H, W, 31)
        hb = hip_backward(g, out, grads, dev, debug=True)
    torch.cuda.synchronize()
    assert out[0] == st["R"] and np.array_equal(out[4].cpu().numpy(), st["radii"])
    code = _hot_codes(out[5], N, out[4])
    assert set(np.nonzero(code)[0].tolist()) <= set(big) and (code[:12] != 0).sum() == 7 and code[3000] != 0 and code[3001] != 0
    assert sorted(code[:12][code[:12] != 0].tolist()) == list(range(1, 8)) and sorted(code[[3000, 3001]].tolist()) == [1, 2]
    assert (code[12:3000] == 0).all() and (code[3002:] == 0).all()
    iv = _C.image_views(out[7], W, H)
    flip_px = flip_pixels(out[1].cpu().numpy(), iv["final_T"].cpu().numpy(), st, None if cull else iv["n_contrib"].cpu().numpy().view(np.uint32))
    check_images_isolating_flips((("color", out[1].cpu().numpy(), st["color"]), ("depth", out[2].cpu().numpy(), st["depth"]), ("alpha", out[3].cpu().numpy(), st["alpha"])), st, flip_px, TOL)
    gb = o.backward(st, *grads)
    print("\n   hot replica lines: " + _check_grads(hb, gb, st, flip_px, "hot"))


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_hot_layouts_vs_oracle(seed):
    """Random image sizes (272 .. 1 100 tiles), random numbers of screen-filling splats at random places of the array (clustered, so that
    workgroups run over their budget of seven, or spread), random opacities, either covariance mode, culling on or off: radii, images
    and every gradient against the oracle; hot codes: at most seven per 256-Gaussian workgroup, distinct, only on large boxes."""
    dev = _dev()
    rng = np.random.default_rng(100 + seed)
    H, W = int(rng.integers(260, 420)), int(rng.integers(270, 680))
    N = int(rng.choice([700, 2500, 6000]))
    mode = str(rng.choice(["col_sr", "sh_sr"]))
    cull = bool(rng.integers(0, 2))
    d = make_inputs(N, H, W, 200 + seed, 0, mode, scale_mul=float(rng.choice([1.0, 2.0])))
    n_big = int(rng.integers(1, 24))
    start = int(rng.integers(0, N - 40))
    big = sorted(set((start + rng.integers(0, 40 if rng.integers(0, 2) else N - start, n_big)).tolist()))
    gen = torch.Generator().manual_seed(seed)
    d["means3D"][big, 0] = 0.8 * (torch.rand(len(big), generator=gen) - 0.5)
    d["means3D"][big, 1] = 0.5 * (torch.rand(len(big), generator=gen) - 0.5)
    d["means3D"][big, 2] = 2.5 + 7.0 * torch.rand(len(big), generator=gen)
    d["scales"][big] = 1.0 + 3.0 * torch.rand(len(big), 3, generator=gen)
    d["opacities"][big] = (0.01 + 0.5 * torch.rand(len(big), generator=gen)).view(-1, *d["opacities"].shape[1:])
    o, st = oracle_forward(d)
    with tile_culling(cull):
        g, out = hip_forward(d, dev, debug=True)
        grads = seeded_grads(H, W, 50 + seed)
        hb = hip_backward(g, out, grads, dev, debug=True)
    torch.cuda.synchronize()
    assert out[0] == st["R"] and np.array_equal(out[4].cpu().numpy(), st["radii"])
    code = _hot_codes(out[5], N, out[4])
    for b in range((N + 255) // 256):
        c = code[256 * b:256 * (b + 1)]; c = c[c != 0]
        assert len(c) <= 7 and sorted(c.tolist()) == list(range(1, len(c) + 1)), (b, c)
    assert set(np.nonzero(code)[0].tolist()) <= set(big)
    from egogaussian_amd import _C
    iv = _C.image_views(out[7], W, H)
    flip_px = flip_pixels(out[1].cpu().numpy(), iv["final_T"].cpu().numpy(), st, None if cull else iv["n_contrib"].cpu().numpy().view(np.uint32))
    check_images_isolating_flips((("color", out[1].cpu().numpy(), st["color"]), ("depth", out[2].cpu().numpy(), st["depth"]), ("alpha", out[3].cpu().numpy(), st["alpha"])), st, flip_px, TOL)
    gb = o.backward(st, *grads)
    print(f"\n   [{N}@{W}x{H} {mode} cull={cull}] {len(big)} large splats, {int((code != 0).sum())} hot: " + _check_grads(hb, gb, st, flip_px, "random hot"))


def _trained_inputs():
    from egogaussian_amd.scene_synth import make_camera, SynthGaussians
    z = np.load(os.path.join(ROOT, "bench_data", "trained_scene.npz"))
    scene = {k: z[k] for k in ("xyz", "log_scale", "quat", "opacity_logit", "features")}
    H, W = 540, 960
    pc = SynthGaussians(scene, device="cpu", requires_grad=False, fused=False)
    cam = make_camera(0, H, W)
    return dict(means3D=pc.get_xyz.detach(), opacities=pc.get_opacity.detach(), shs=pc.get_features.detach(),
                cov3D_precomp=pc.get_covariance().detach().contiguous(), viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                campos=cam.camera_center, bg=torch.zeros(3), image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx / 2),
                tanfovy=math.tan(cam.FoVy / 2), sh_degree=0, scale_modifier=1.0), H, W


def test_trained_scene_vs_oracle():
    """The committed densified model (253k Gaussians, 960x540; 4.06 M rectangle instances of which tile culling keeps a fifth; a few
    hundred hot Gaussians): lists bit-exact with culling off, images and every gradient (colour, depth and alpha upstream) against the
    oracle with culling on (the shipped default)."""
    from egogaussian_amd import _C
    dev = _dev()
    d, H, W = _trained_inputs()
    N = d["means3D"].shape[0]
    o, st = oracle_forward(d)
    with tile_culling(False):
        g, out = hip_forward(d, dev)
        bv = _C.binning_views(out[6], N, out[0], W, H, _C.stats["capacity"]); iv = _C.image_views(out[7], W, H)
        assert out[0] == st["R"] and np.array_equal(out[4].cpu().numpy(), st["radii"])
        assert np.array_equal(bv["point_list"].cpu().numpy().view(np.uint32), st["point_list"]), "sorted instance list not bit-exact"
        assert np.array_equal(iv["ranges"].cpu().numpy().view(np.uint32), st["ranges"])
        img_full = [t.clone() for t in out[1:4]]
    with tile_culling(True):
        g, out = hip_forward(d, dev)
        grads = seeded_grads(H, W, 41)
        hb = hip_backward(g, out, grads, dev)
    torch.cuda.synchronize()
    code = _hot_codes(out[5], N, out[4])
    n_hot = int((code != 0).sum())
    assert 50 <= n_hot <= 2000, n_hot
    for a, b in zip(img_full, out[1:4]):
        assert torch.equal(a, b), "tile culling changed an output value on the trained scene"
    iv = _C.image_views(out[7], W, H)
    flip_px = flip_pixels(out[1].cpu().numpy(), iv["final_T"].cpu().numpy(), st)
    check_images_isolating_flips((("color", out[1].cpu().numpy(), st["color"]), ("depth", out[2].cpu().numpy(), st["depth"]), ("alpha", out[3].cpu().numpy(), st["alpha"])), st, flip_px, TOL)
    gb = o.backward(st, *grads)
    n_list = int((iv["ranges"][:, 1] - iv["ranges"][:, 0]).sum())
    print(f"\n   trained scene: R {out[0]}, kept {n_list}, hot Gaussians {n_hot}; " + _check_grads(hb, gb, st, flip_px, "trained"))
