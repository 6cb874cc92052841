"""CPU: host-side logic added in round 2 that needs no GPU -- the packed-frame layout of the captured step, the capacity-sized
model's bookkeeping, the synthetic fine_all inputs of bench.py."""
import importlib.util
import math
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_frame_layout_is_aligned_and_round_trips():
    from egogaussian_amd.graph import frame_layout, pack_frame
    from egogaussian_amd.scene_synth import make_camera
    H, W = 37, 50                                        # 3 * 37 * 50 = 5550 floats: not a multiple of 4
    off, size = frame_layout(3 * H * W, H * W, dynamic=True, gated=True)
    assert list(off) == ["gt", "cam", "accum_R", "gate"] and size % 4 == 0
    ends = 0
    for name, (b, e) in off.items():
        assert b % 4 == 0 and b >= ends and e > b, name      # every segment starts on a 16-byte boundary, no overlap
        ends = e
    assert off["cam"][1] - off["cam"][0] == 35 and off["accum_R"][1] - off["accum_R"][0] == 9 and off["gate"][1] - off["gate"][0] == H * W
    off2, size2 = frame_layout(3 * H * W, H * W)
    assert list(off2) == ["gt", "cam"] and size2 < size
    cam = make_camera(17, H, W)
    gt, R, gate = torch.rand(3, H, W), torch.rand(3, 3), torch.rand(H, W)
    f = pack_frame(cam, gt, R, gate)
    assert f.shape == (size,) and f.dtype == torch.float32
    assert torch.equal(f[off["gt"][0]:off["gt"][1]].view(3, H, W), gt) and torch.equal(f[off["accum_R"][0]:off["accum_R"][1]].view(3, 3), R)
    assert torch.equal(f[off["gate"][0]:off["gate"][1]].view(H, W), gate)
    c = f[off["cam"][0]:off["cam"][1]]
    assert torch.equal(c[:16].view(4, 4), cam.world_view_transform) and torch.equal(c[16:32].view(4, 4), cam.full_proj_transform)
    assert torch.equal(c[32:35], cam.camera_center)
    assert pack_frame(cam, gt).shape == (size2,)


def test_capacity_model_bookkeeping_on_cpu():
    from egogaussian_amd.capacity import CapacityGaussians
    from egogaussian_amd.scene_synth import make_scene
    sc = make_scene(100, 32, 32, 0, sh_degree=1)
    pc = CapacityGaussians(sc, 260, device="cpu", sh_degree=1)
    assert (pc.capacity, pc.n_active, int(pc.active_count)) == (260, 100, 100) and pc.active_count.dtype == torch.int32
    for t, k in ((pc._xyz, 3), (pc._scaling, 3), (pc._rotation, 4), (pc._opacity, 1)):
        assert t.shape == (260, k) and t.requires_grad
    assert pc._features_dc.shape == (260, 1, 3) and pc._features_rest.shape == (260, 3, 3)
    assert np.array_equal(pc._xyz.detach().numpy()[:100], sc["xyz"]) and float(pc._xyz.detach()[100:].abs().sum()) == 0
    assert torch.equal(pc._rotation.detach()[100:], torch.tensor([1.0, 0, 0, 0]).expand(160, 4))      # padding rows: identity rotations
    assert pc.live(pc._xyz).shape == (100, 3) and pc.max_radii2D.shape == (260,) and pc.denom.shape == (260, 1)
    pc.set_active(180)
    assert (pc.n_active, int(pc.active_count)) == (180, 180)
    with pytest.raises(ValueError):
        pc.set_active(261)
    with pytest.raises(ValueError):
        CapacityGaussians(sc, 50, device="cpu")
    assert pc.grow(200) == 260                           # never shrinks
    assert pc.grow(400) == 400 and pc._xyz.shape == (400, 3) and pc.n_active == 180 and pc._is_object.shape == (400, 1)
    assert np.array_equal(pc._xyz.detach().numpy()[:100], sc["xyz"])


def test_bench_fine_all_inputs():
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    for k in (0, 37, 150.5, 299):
        R = bench.object_rotation(k, "cpu")
        assert R.shape == (3, 3) and torch.allclose(R @ R.T, torch.eye(3), atol=1e-6) and abs(float(torch.det(R)) - 1.0) < 1e-5
        g = bench.hand_gate(k, 54, 96, "cpu")
        assert g.shape == (54, 96) and set(g.unique().tolist()) <= {0.0, 1.0} and 0.02 < float((g == 0).float().mean()) < 0.2
    assert torch.equal(bench.object_rotation(0, "cpu"), torch.eye(3))
    assert not torch.equal(bench.object_rotation(10, "cpu"), bench.object_rotation(11, "cpu"))
    # the per-unit byte figures of DESIGN.md section 4
    assert bench.algorithmic_bytes("preprocess", 10, 0, 0) == 10 * 104 and bench.algorithmic_bytes("render_backward", 0, 10, 100) == 10 * 92 + 32 * 100
    assert bench.algorithmic_bytes("tile_sort", 0, 7, 0) == 84 and bench.algorithmic_bytes("preprocess_backward", 10, 0, 0) == 2160


def test_ready_to_pin_harness_inputs_are_reproducible():
    """tools/compare_upstream_npz.py: the committed input half (tests/golden/upstream) is what the seeds regenerate -- a holder of the
    reference's CUDA build who regenerates the inputs gets the same bytes, checked by hash before any comparison."""
    import importlib.util
    import os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cmp_up", os.path.join(root, "tools", "compare_upstream_npz.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    want = dict(l.split() for l in open(os.path.join(root, "tests", "golden", "upstream", "input_sha256.txt")).read().splitlines())
    assert set(want) == set(m.CASES)
    for cid in ("A", "E", "F"):
        d, grads = m.make_case(cid)
        assert m.input_hash(d) == want[cid], cid
        f = np.load(os.path.join(root, "tests", "golden", "upstream", f"case_{cid}_inputs.npz"))
        for k in m.INPUT_KEYS:
            if k in d:
                assert np.array_equal(f[k], d[k].numpy()), (cid, k)
        assert np.array_equal(f["gc"], grads[0].numpy())
