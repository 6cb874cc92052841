#!/usr/bin/env python
"""Generates tests/golden/densify.npz by IMPORTING the reference's GaussianModel (read-only, /root/reference) in the
build container and running ITS densification / pruning / PLY code on a seeded model (row f-4 of SURVEY.md section 8f).
Only the captured tensors are committed; this script is the recipe.   Run:  python tests/golden/make_golden_densify.py

What is pinned  (/root/reference/scene/gaussian_model.py)
  stats_*      add_densification_stats + the trainer's max_radii2D update          :735-740, trainers/train_static.py:125-127
  case<k>_*    densify_and_prune for several argument sets: parameters, Adam moments, generation / is_object and the
               reset statistics AFTER the call, element for element                :506-709
               The split step draws torch.normal(mean=0, std=scales); the capture replaces that draw by std * z with a
               recorded standard-normal z, so the call is a deterministic function of its inputs.
  reset_*      reset_opacity                                                       :484-490
  ply_*        the vertex table save_ply hands to plyfile: column names, order and values   :340-397
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import CudaToCpu, stub, ROOT  # noqa: E402  (also puts the repo and /root/reference on sys.path)


def npy(t):                                     # a COPY: the reference updates several of these tensors in place afterwards
    return np.array(t.detach().cpu().numpy() if torch.is_tensor(t) else t, copy=True)

CAP = {}


class _PlyElement:
    @staticmethod
    def describe(elements, name):
        CAP["ply_elements"] = elements
        CAP["ply_name"] = name
        return elements


class _PlyData:
    def __init__(self, els):
        self.els = els

    def write(self, path):
        CAP["ply_path"] = path


def build_model(GaussianModel, rng, N, sh_degree=1):
    g = GaussianModel(sh_degree)
    P_ = lambda x: torch.nn.Parameter(torch.tensor(x, dtype=torch.float32))
    K = (sh_degree + 1) ** 2
    g._xyz = P_(rng.normal(size=(N, 3)))
    g._features_dc = P_(rng.normal(size=(N, 1, 3)))
    g._features_rest = P_(rng.normal(size=(N, K - 1, 3)) * 0.1)
    g._scaling = P_(np.log(rng.uniform(0.002, 0.08, size=(N, 3))))
    g._rotation = P_(rng.normal(size=(N, 4)))
    g._opacity = P_(rng.normal(size=(N, 1)) * 2.5)
    g._label = P_(rng.normal(size=(N, 1)))
    g._generation = torch.tensor(rng.integers(0, 3, size=(N, 1)), dtype=torch.int)
    g._is_object = torch.tensor((rng.uniform(size=(N, 1)) < 0.3).astype(np.int32), dtype=torch.int)
    g.max_radii2D = torch.zeros(N)
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                 position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
    g.spatial_lr_scale = 1.0
    g.training_setup(args)
    # two optimizer steps with seeded gradients, so that Adam moments exist and differ per element
    for _ in range(2):
        for p in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g._label):
            p.grad = torch.tensor(rng.normal(size=tuple(p.shape)) * 1e-3, dtype=torch.float32)
        g.optimizer.step()
        g.optimizer.zero_grad(set_to_none=True)
    return g


def snapshot(g, prefix):
    d = {}
    names = {"xyz": g._xyz, "f_dc": g._features_dc, "f_rest": g._features_rest, "opacity": g._opacity, "scaling": g._scaling,
             "rotation": g._rotation, "label": g._label}
    for group in g.optimizer.param_groups:
        p = group["params"][0]
        assert p is names[group["name"]], group["name"]          # the model attribute IS the optimizer's parameter
        st = g.optimizer.state.get(p)
        d[f"{prefix}{group['name']}"] = npy(p)
        d[f"{prefix}{group['name']}_exp_avg"] = npy(st["exp_avg"])
        d[f"{prefix}{group['name']}_exp_avg_sq"] = npy(st["exp_avg_sq"])
        d[f"{prefix}{group['name']}_step"] = np.asarray(float(st["step"]))
    d[f"{prefix}generation"] = npy(g._generation)
    d[f"{prefix}is_object"] = npy(g._is_object)
    d[f"{prefix}xyz_gradient_accum"] = npy(g.xyz_gradient_accum)
    d[f"{prefix}denom"] = npy(g.denom)
    d[f"{prefix}max_radii2D"] = npy(g.max_radii2D)
    return d


def main():
    stub("plyfile", PlyData=_PlyData, PlyElement=_PlyElement)
    stub("pytorch3d")
    stub("pytorch3d.transforms", euler_angles_to_matrix=None)
    stub("simple_knn")
    stub("simple_knn._C", distCUDA2=lambda pts: torch.full((pts.shape[0],), 1e-3))
    out = {}
    with CudaToCpu():
        import scene.gaussian_model as gm
        from scene.gaussian_model import GaussianModel
        gm.mkdir_p = lambda p: None                          # save_ply creates the directory; nothing is written here
        N = 400
        extent = 4.0

        # ---- per-step statistics ---------------------------------------------------------------------------
        rng = np.random.default_rng(77)
        g = build_model(GaussianModel, rng, N)
        out.update(snapshot(g, "in_"))
        vs = torch.zeros(N, 3, requires_grad=True)
        steps = []
        for it in range(3):
            vs.grad = torch.tensor(rng.normal(size=(N, 3)) * 5e-4, dtype=torch.float32)
            radii = torch.tensor(rng.integers(0, 40, size=N) * (rng.uniform(size=N) < 0.7), dtype=torch.int)
            vis = radii > 0
            g.max_radii2D[vis] = torch.max(g.max_radii2D[vis], radii[vis])            # trainers/train_static.py:125
            g.add_densification_stats(vs, vis)
            steps.append((npy(vs.grad), npy(radii)))
        out["stats_grads"] = np.stack([s[0] for s in steps]); out["stats_radii"] = np.stack([s[1] for s in steps])
        out["stats_xyz_gradient_accum"] = npy(g.xyz_gradient_accum); out["stats_denom"] = npy(g.denom)
        out["stats_max_radii2D"] = npy(g.max_radii2D)

        # ---- densify_and_prune, several argument sets, each from the same starting state ------------------------
        cases = [
            dict(max_grad=4e-4, min_opacity=0.05, extent=extent, max_screen_size=20),
            dict(max_grad=4e-4, min_opacity=0.05, extent=extent, max_screen_size=None),
            dict(max_grad=3e-4, min_opacity=0.1, extent=extent, max_screen_size=20, curr_gen=7, prune_prev_gen=False, split_prev_gen=True),
            # (split_prev_gen=False cannot be captured: densify_and_prune passes curr_gen into densify_and_split's `N` slot
            #  (:698 vs :588), so curr_gen is None there and `(get_generation == None).squeeze()` raises.  The same slip means
            #  split children always inherit their parent's generation.)
            dict(max_grad=3e-4, min_opacity=0.1, extent=extent, max_screen_size=20, curr_gen=1, prune_prev_gen=True),
            dict(max_grad=3e-4, min_opacity=0.02, extent=extent, max_screen_size=20, which_object=1),
            dict(max_grad=4e-4, min_opacity=0.3, extent=extent, max_screen_size=10, clone=False, split=False),
            dict(max_grad=4e-4, min_opacity=0.05, extent=extent, max_screen_size=20, clone=True, split=False),
            dict(max_grad=4e-4, min_opacity=0.05, extent=extent, max_screen_size=20, clone=False, split=True),
        ]
        real_normal = torch.normal
        for k, kw in enumerate(cases):
            rng = np.random.default_rng(77)
            g = build_model(GaussianModel, rng, N)
            # statistics with a spread that puts some points on either side of every threshold
            g.xyz_gradient_accum = torch.tensor(rng.gamma(2.0, 2e-4, size=(N, 1)) * rng.integers(0, 4, size=(N, 1)), dtype=torch.float32)
            g.denom = torch.tensor(rng.integers(0, 4, size=(N, 1)).astype(np.float32))       # zeros -> NaN -> 0 path
            g.max_radii2D = torch.tensor(rng.uniform(0, 30, size=N).astype(np.float32))
            pre = snapshot(g, f"case{k}_in_")
            zs = []

            def fake_normal(mean=None, std=None, **kws):
                z = torch.tensor(np.random.default_rng(1000 + k).normal(size=tuple(std.shape)), dtype=torch.float32)
                zs.append(z)
                return mean + std * z
            torch.normal = fake_normal
            try:
                g.densify_and_prune(**kw)
            finally:
                torch.normal = real_normal
            out.update(pre)
            out.update(snapshot(g, f"case{k}_out_"))
            out[f"case{k}_z"] = npy(zs[0]) if zs else np.zeros((0, 3), np.float32)
            out[f"case{k}_args"] = np.array([kw["max_grad"], kw["min_opacity"], kw["extent"],
                                             -1.0 if kw.get("max_screen_size") is None else kw["max_screen_size"],
                                             float(kw.get("clone", True)), float(kw.get("split", True)),
                                             -1e9 if kw.get("curr_gen") is None else kw["curr_gen"],
                                             float(kw.get("prune_prev_gen", True)), float(kw.get("split_prev_gen", True)),
                                             -1e9 if kw.get("which_object") is None else kw["which_object"]], dtype=np.float64)
            print(f"case {k}: {N} -> {g._xyz.shape[0]} points, z {tuple(out[f'case{k}_z'].shape)}")
        out["n_cases"] = np.asarray(len(cases))
        out["percent_dense"] = np.asarray(0.01)

        # ---- reset_opacity -------------------------------------------------------------------------------------
        rng = np.random.default_rng(77)
        g = build_model(GaussianModel, rng, N)
        g.reset_opacity()
        out["reset_opacity"] = npy(g._opacity)
        st = g.optimizer.state[g._opacity]
        out["reset_exp_avg_abs_sum"] = np.asarray(float(st["exp_avg"].abs().sum() + st["exp_avg_sq"].abs().sum()))

        # ---- the vertex table of save_ply --------------------------------------------------------------------
        g.save_ply("/nonexistent/point_cloud.ply")
        el = CAP["ply_elements"]
        out["ply_names"] = np.array(el.dtype.names)
        out["ply_formats"] = np.array([el.dtype[n].str for n in el.dtype.names])
        out["ply_table"] = np.stack([el[n] for n in el.dtype.names], 1)
        out["ply_element_name"] = np.asarray(CAP["ply_name"])
    np.savez_compressed(os.path.join(HERE, "densify.npz"), **out)
    print("densify.npz", os.path.getsize(os.path.join(HERE, "densify.npz")), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
