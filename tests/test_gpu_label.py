"""GPU: the colours-only backward (ABI 4, egs_backward grad_mask == EGS_GRAD_COLORS; csrc/render_bwd.hip k_render_backward<0>).

The reference's label call detaches every geometric input (/root/reference/gaussian_renderer/render_helper.py:38-54): only
`colors_precomp` carries a gradient, 30 000 times in stage 1 (/root/reference/trainers/train_static.py:105-109).  The autograd
Function hands ctx.needs_input_grad to the library, which then sums w * dL/dC per Gaussian and nothing else.  Checked here against
the C oracle (1e-4 relative, the north star's bar), against the full backward of the same frame (same sums, different atomic order),
and -- at the benched size, where the oracle takes minutes -- through properties of the exact sum."""
import numpy as np
import pytest
import torch

from tests.common import make_inputs, seeded_grads, rel_err, tile_culling
from tests.test_gpu_parity import hip_forward, hip_backward, oracle_forward, _dev

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _colors_only(g, out, grads, dev):
    from egogaussian_amd import _C
    R, color, depth, alpha, radii, geom, binning, img = out
    e = torch.empty(0, device=dev)
    res = _C.rasterize_gaussians_backward(g["bg"], g["means3D"], radii, g["colors_precomp"], g.get("scales", e), g.get("rotations", e),
                                          g["scale_modifier"], g.get("cov3D_precomp", e), g["viewmatrix"], g["projmatrix"], g["tanfovx"],
                                          g["tanfovy"], grads[0].to(dev), e, e, e, g["sh_degree"], g["campos"], geom, R, binning, img, alpha, False,
                                          grad_mask=_C.GRAD_COLORS)
    assert all(r is None for i, r in enumerate(res) if i != 1), "the colours-only backward produces dL_dcolors and nothing else"
    return res[1]


@pytest.mark.parametrize("N,H,W,seed,mode,smul", [(2000, 96, 128, 3, "col_sr", 2.0), (3000, 70, 100, 1, "col_sr", 4.0),
                                                   (2000, 64, 80, 5, "col_cov", 3.0), (20000, 270, 480, 6, "col_sr", 2.0)])
@pytest.mark.parametrize("cull", [False, True], ids=["reference-lists", "tile-culling"])
def test_colors_only_backward_vs_oracle_and_full_backward(N, H, W, seed, mode, smul, cull):
    dev = _dev()
    d = make_inputs(N, H, W, seed, 0, mode, scale_mul=smul)
    o, st = oracle_forward(d)
    grads = seeded_grads(H, W, seed + 10)
    gb = o.backward(st, grads[0], None, None)                          # colour gradient only: depth / alpha do not reach the colours anyway
    with tile_culling(cull):
        g, out = hip_forward(d, dev)
        dc = _colors_only(g, out, grads, dev)
        full = hip_backward(g, out, grads, dev)[1]
    torch.cuda.synchronize()
    e_or = rel_err(dc.cpu().numpy(), gb["dL_dcolor"])
    e_full = rel_err(dc.cpu().numpy(), full.cpu().numpy())
    print(f"\n[{N}@{W}x{H} {mode}] colours-only dL/dcolors: vs oracle {e_or:.1e}, vs the full backward {e_full:.1e}")
    assert e_full < 2e-6, "same per-pixel terms as the full backward: only the order of the float atomics may differ"
    assert e_or < TOL


def test_label_call_takes_the_colors_only_path_and_matches_the_full_one():
    """get_render_label() through autograd: the library is told that only the colours need a gradient (spy on the C wrapper), `_label.grad`
    equals what the same call yields when every input requires grad (the full backward), and no other leaf receives a gradient."""
    from egogaussian_amd import _C
    from egogaussian_amd.renderer import get_render_label
    from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians
    dev = _dev()
    N, H, W = 30000, 160, 256
    scene = make_scene(N, H, W, 2); scene["log_scale"] += np.log(2.0).astype(np.float32)
    pc = SynthGaussians(scene, device=dev)
    pc._label = torch.rand(N, 1, device=dev).requires_grad_(True)
    cam = make_camera(7, H, W, device=dev)
    bg = torch.zeros(3, device=dev)
    up = seeded_grads(H, W, 3)[0].to(dev)
    seen = []
    orig = _C.rasterize_gaussians_backward
    def spy(*a, **k):
        seen.append(a[-1] if len(a) > 31 else k.get("grad_mask", 0))
        return orig(*a, **k)
    _C.rasterize_gaussians_backward = spy
    try:
        lab = get_render_label(cam, pc, bg)
        (lab * up).sum().backward()
    finally:
        _C.rasterize_gaussians_backward = orig
    assert seen == [_C.GRAD_COLORS]
    assert all(getattr(pc, a).grad is None for a in ("_xyz", "_scaling", "_rotation", "_opacity"))
    g_fast = pc._label.grad.clone(); pc._label.grad = None
    # the same render with a second consumer of the geometry: the mask is wider and the full backward runs
    from egogaussian_amd.rasterizer import GaussianRasterizer
    from egogaussian_amd.renderer import get_raster_settings, gaussians_to_label_rendervar
    var = gaussians_to_label_rendervar(pc)
    var["opacities"] = pc.get_opacity                                   # not detached
    lab2, _, _, _ = GaussianRasterizer(get_raster_settings(cam, pc, bg))(**var)
    (lab2 * up).sum().backward()
    torch.cuda.synchronize()
    assert pc._opacity.grad is not None
    assert rel_err(g_fast.cpu().numpy(), pc._label.grad.cpu().numpy()) < 2e-6
    assert torch.equal(lab, lab2)


@pytest.mark.parametrize("what", ["config_C", "trained_scene"])
def test_colors_only_backward_properties_at_bench_size(what):
    """500k Gaussians @ 960x540 (and the committed trained scene, whose screen-filling splats accumulate through replica lines): the
    oracle is too slow to be the checker of every run, so the exact sum is checked through what it must satisfy.
      * equal to the full backward's dL/dcolors (same terms);
      * linear in the upstream gradient: dC(g1 + g2) = dC(g1) + dC(g2), dC(2 g) = 2 dC(g);
      * the image identity  sum_i c_i . dL/dc_i = sum_px dL/dC . (C - T_final bg)  (C = sum_i w_i c_i + T bg), with the forward's own
        image and transmittance: ties every weight of the backward replay to the forward blend;
      * Gaussians with radii == 0 get exactly zero."""
    from egogaussian_amd import _C
    dev = _dev()
    H, W = 540, 960
    if what == "config_C":
        d = make_inputs(500_000, H, W, 0, 0, "col_sr")
    else:
        import os
        z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data", "trained_scene.npz"))
        N = z["xyz"].shape[0]
        d = make_inputs(N, H, W, 0, 0, "col_sr")
        d["means3D"] = torch.tensor(z["xyz"]); d["scales"] = torch.exp(torch.tensor(z["log_scale"]))
        d["rotations"] = torch.nn.functional.normalize(torch.tensor(z["quat"])); d["opacities"] = torch.sigmoid(torch.tensor(z["opacity_logit"])).reshape(-1, 1)
        d["colors_precomp"] = torch.rand(N, 3, generator=torch.Generator().manual_seed(4))
    g, out = hip_forward(d, dev)
    R, color, depth, alpha, radii, geom, binning, img = out
    g1, g2 = seeded_grads(H, W, 1)[0], seeded_grads(H, W, 2)[0] - 0.5
    dc1 = _colors_only(g, out, (g1,), dev)
    dc2 = _colors_only(g, out, (g2,), dev)
    dc12 = _colors_only(g, out, (g1 + g2,), dev)
    dc1x2 = _colors_only(g, out, (2 * g1,), dev)
    full = hip_backward(g, out, (g1, torch.zeros(1, H, W), torch.zeros(1, H, W)), dev)[1]
    torch.cuda.synchronize()
    scale = float(dc12.abs().max())
    assert float((dc12 - (dc1 + dc2)).abs().max()) <= 2e-5 * scale
    assert float((dc1x2 - 2 * dc1).abs().max()) <= 1e-5 * float(dc1x2.abs().max())
    assert rel_err(dc1.cpu().numpy(), full.cpu().numpy()) < 5e-6
    final_T = _C.image_views(img, W, H)["final_T"]
    lhs = (g["colors_precomp"].double() * dc1.double()).sum().item()
    rhs = (g1.to(dev).double() * (color.double() - final_T.double()[None] * g["bg"].double()[:, None, None])).sum().item()
    print(f"\n[{what}] R={R}  sum c.dL/dc = {lhs:.6f}, sum dL/dC.(C - T bg) = {rhs:.6f}; hot Gaussians {int(((_C.geom_views(geom, d['means3D'].shape[0])['clamped'] >> 3 != 0) & (radii > 0)).sum())}")
    assert abs(lhs - rhs) <= 2e-5 * abs(rhs)
    assert float(dc1[radii <= 0].abs().max()) == 0.0


def test_color_only_forward_changes_no_colour_bit():
    """render(..., color_only=True) (ABI 4: out_depth == out_alpha == NULL; what GraphedTrainStep renders with): the colour image, radii,
    transmittance and contributor counts are those of the full forward bit for bit, depth and alpha are not produced, and the gradients
    of a colour-only loss are the same sums."""
    from egogaussian_amd import _C
    from egogaussian_amd.renderer import render
    from egogaussian_amd.scene_synth import make_scene, make_camera, SynthGaussians, Pipe
    dev = _dev()
    N, H, W = 60000, 270, 480
    scene = make_scene(N, H, W, 4); scene["log_scale"] += np.log(1.5).astype(np.float32)
    cam, bg = make_camera(11, H, W, device=dev), torch.tensor([0.2, 0.1, 0.3], device=dev)
    up = seeded_grads(H, W, 8)[0].to(dev)
    res = {}
    for co in (False, True):
        pc = SynthGaussians(scene, device=dev)
        out = render(cam, pc, Pipe, bg, color_only=co)
        iv = _C.image_views(_C.stats["image_buffer"], W, H)
        (out["render"] * up).sum().backward()
        torch.cuda.synchronize()
        res[co] = (out, iv["final_T"].clone(), iv["n_contrib"].clone(), [p.grad.clone() for p in (pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation)])
    full, only = res[False], res[True]
    assert only[0]["depth"] is None and only[0]["alpha"] is None and full[0]["depth"] is not None
    assert torch.equal(full[0]["render"], only[0]["render"]) and torch.equal(full[0]["radii"], only[0]["radii"])
    assert torch.equal(full[1], only[1]) and torch.equal(full[2], only[2])
    for a, b in zip(full[3], only[3]):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
