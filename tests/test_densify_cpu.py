"""CPU: the torch restatement of the densification bookkeeping (oracle/densify_torch.py) against the fixture captured
from the reference's own GaussianModel (tests/golden/densify.npz, recipe tests/golden/make_golden_densify.py)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
from oracle import densify_torch as D          # noqa: E402


def load():
    return np.load(os.path.join(GOLD, "densify.npz"))


def state_from(g, prefix):
    st = {}
    for k in D.PARAMS:
        st[k] = torch.tensor(g[f"{prefix}{k}"])
        st[k + "_exp_avg"] = torch.tensor(g[f"{prefix}{k}_exp_avg"]); st[k + "_exp_avg_sq"] = torch.tensor(g[f"{prefix}{k}_exp_avg_sq"])
    for k in ("generation", "is_object", "xyz_gradient_accum", "denom", "max_radii2D"):
        st[k] = torch.tensor(g[f"{prefix}{k}"])
    return st


def case_kwargs(a):
    return dict(max_grad=float(a[0]), min_opacity=float(a[1]), extent=float(a[2]), max_screen_size=None if a[3] < 0 else float(a[3]),
                clone=bool(a[4]), split=bool(a[5]), curr_gen=None if a[6] < -1e8 else int(a[6]), prune_prev_gen=bool(a[7]),
                which_object=None if a[9] < -1e8 else int(a[9]))


def assert_state_equal(got, g, prefix, exact_children=False):
    for k in list(D.PARAMS) + [p + m for p in D.PARAMS for m in ("_exp_avg", "_exp_avg_sq")] + ["generation", "is_object",
                                                                                           "xyz_gradient_accum", "denom", "max_radii2D"]:
        want = g[f"{prefix}{k}"]
        have = got[k].detach().cpu().numpy()
        assert have.shape == want.shape, (k, have.shape, want.shape)
        if k in ("xyz", "scaling"):                      # split children are computed (rotation, log): rounding-level freedom
            assert np.allclose(have, want, rtol=2e-6, atol=2e-7), k
        else:
            assert np.array_equal(have, want), k


def test_stats_match_reference():
    g = load()
    st = state_from(g, "in_")
    for grads, radii in zip(g["stats_grads"], g["stats_radii"]):
        D.add_densification_stats(st, torch.tensor(grads), torch.tensor(radii))
    assert np.allclose(st["xyz_gradient_accum"].numpy(), g["stats_xyz_gradient_accum"], rtol=1e-6, atol=0)
    assert np.array_equal(st["denom"].numpy(), g["stats_denom"]) and np.array_equal(st["max_radii2D"].numpy(), g["stats_max_radii2D"])


@pytest.mark.parametrize("k", range(8))
def test_densify_and_prune_matches_reference(k):
    g = load()
    assert int(g["n_cases"]) == 8
    st = D.densify_and_prune(state_from(g, f"case{k}_in_"), percent_dense=float(g["percent_dense"]), z=torch.tensor(g[f"case{k}_z"]),
                             **case_kwargs(g[f"case{k}_args"]))
    assert_state_equal(st, g, f"case{k}_out_")


def test_reset_opacity_matches_reference():
    g = load()
    st = D.reset_opacity(state_from(g, "in_"))
    assert np.allclose(st["opacity"].numpy(), g["reset_opacity"], rtol=1e-6, atol=1e-7)
    assert float(g["reset_exp_avg_abs_sum"]) == 0.0


# ---- PLY hand-off (numpy only) ---------------------------------------------------------------------------------------
class _Model:
    max_sh_degree = 1


def _model_from(g, prefix="in_"):
    m = _Model()
    st = state_from(g, prefix)
    m._xyz, m._features_dc, m._features_rest = st["xyz"], st["f_dc"], st["f_rest"]
    m._opacity, m._scaling, m._rotation, m._label = st["opacity"], st["scaling"], st["rotation"], st["label"]
    m._generation, m._is_object = st["generation"], st["is_object"]
    return m


def test_ply_vertex_table_matches_what_the_reference_hands_to_plyfile():
    from egogaussian_amd import ply
    g = load()
    m = _model_from(g)
    m._opacity = torch.tensor(g["reset_opacity"])          # the capture saved the model right after reset_opacity()
    names, table = ply.vertex_table(m)
    assert list(names) == [str(n) for n in g["ply_names"]] and str(g["ply_element_name"]) == "vertex"
    assert all(str(f) == "<f4" for f in g["ply_formats"])
    assert np.array_equal(table, g["ply_table"])


def test_ply_round_trip_and_upstream_files(tmp_path):
    from egogaussian_amd import ply
    g = load()
    m = _model_from(g)
    path = str(tmp_path / "iteration_7" / "point_cloud.ply")
    ply.save_ply(m, path)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n") + len(b"end_header\n")].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 400\nproperty float x\n") and "property float is_object\n" in head
    assert len(raw) == len(head) + 400 * 4 * len(g["ply_names"])
    back = ply.load_ply(_Model(), path, train_params=True, device="cpu")
    for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_label"):
        assert isinstance(getattr(back, a), torch.nn.Parameter) and torch.equal(getattr(back, a).detach(), getattr(m, a)), a
    assert back._generation.dtype == torch.int32 and torch.equal(back._generation, m._generation) and torch.equal(back._is_object, m._is_object)
    assert back.max_radii2D.shape == (400,) and back.active_sh_degree == 1
    # a file without the three extra columns (upstream 3DGS): the reference's defaults
    names, table = ply.vertex_table(m)
    ply.write_table(str(tmp_path / "og.ply"), names[:-3], table[:, :-3])
    og = ply.load_ply(_Model(), str(tmp_path / "og.ply"), train_params=False, is_object=True, device="cpu")
    assert not isinstance(og._xyz, torch.nn.Parameter) and torch.all(og._label == 0.01) and torch.all(og._generation == 0)
    assert torch.all(og._is_object == 1)
    assert torch.all(ply.load_ply(_Model(), str(tmp_path / "og.ply"), is_object=True, force_bg=True, device="cpu")._is_object == 0)
    with pytest.raises(ValueError):
        bad = _Model(); bad.max_sh_degree = 2
        ply.load_ply(bad, path, device="cpu")
