"""egogaussian_amd.install(): the reference's trainers, unchanged, on this package's kernels (egogaussian_amd/patching.py).

CPU, build container only (the reference's Python is not on the GPU box): after install() the module objects the reference's training
loop calls ARE the replacements -- `trainers.train_static.l1_loss` / `.ssim` (bound by name at import,
/root/reference/trainers/train_static.py:9), `GaussianModel.setup_functions` / `.training_setup`
(/root/reference/scene/gaussian_model.py:27-44,180-198) -- counted by spy counters when the reference's own code paths run.
GPU: the replacement loss functions against the fixture captured from the reference's (tests/golden/losses.npz)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stubs():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("pytorch3d"); mg.stub("pytorch3d.transforms", euler_angles_to_matrix=None)
    mg.stub("wandb", log=lambda *a, **k: None, init=lambda *a, **k: None)
    return mg


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "trainers")), reason="the reference's Python is only present in the build container")
def test_install_repoints_what_the_reference_trainers_call():
    import egogaussian_amd
    from egogaussian_amd import patching
    from egogaussian_amd.optim import FusedAdam
    mg = _stubs()
    touched = [m for m in list(sys.modules) if m.split(".")[0] in ("trainers", "utils", "scene", "gaussian_renderer", "arguments")]
    for m in touched:
        sys.modules.pop(m)
    try:
        import utils.loss_utils as lu
        orig_l1, orig_ssim = lu.l1_loss, lu.ssim
        # a module that bound the names BEFORE install() (a trainer imported too early) is re-pointed too
        early = types.ModuleType("early_trainer"); early.l1_loss, early.ssim = orig_l1, orig_ssim
        sys.modules["early_trainer"] = early
        assert torch.autograd.is_multithreading_enabled()
        rep = egogaussian_amd.install()
        assert egogaussian_amd.install() is rep                                  # idempotent
        assert not torch.autograd.is_multithreading_enabled() and "autograd" in rep   # backward on the trainer's thread while installed
        assert "scene.gaussian_model.GaussianModel.get_features" in rep["model"]      # the class's concatenation remembers its halves (provenance.py)
        assert ("early_trainer", "l1_loss") in rep["rebound"] and ("early_trainer", "ssim") in rep["rebound"]
        import trainers.train_static as ts                                        # the reference's trainer module, as it is
        assert ts.l1_loss is lu.l1_loss is early.l1_loss and ts.l1_loss is not orig_l1 and ts.l1_loss.__wrapped__ is orig_l1
        assert ts.ssim is lu.ssim is early.ssim and ts.ssim.__wrapped__ is orig_ssim
        for mod in ("trainers.fine_all", "trainers.coarse_obj_pose", "trainers.fine_obj"):
            try:
                m = __import__(mod, fromlist=["x"])
            except Exception:                                                     # (a module needing something the container lacks: skip it)
                continue
            for name in ("l1_loss", "ssim"):
                if hasattr(m, name):
                    assert getattr(m, name) is getattr(lu, name), (mod, name)
        # CPU tensors: the replacements step aside and the reference's own functions answer (spy counters)
        a, b = torch.rand(3, 24, 32), torch.rand(3, 24, 32)
        n0 = dict(patching.calls)
        assert torch.equal(ts.l1_loss(a, b), orig_l1(a, b)) and torch.equal(ts.ssim(a, b), orig_ssim(a, b))
        assert patching.calls["l1_loss_fallback"] == n0["l1_loss_fallback"] + 1 and patching.calls["ssim_fallback"] == n0["ssim_fallback"] + 1
        # the model: constructing the reference's GaussianModel runs the wrapped setup_functions -> the covariance producers are installed
        with mg.CudaToCpu():
            from scene.gaussian_model import GaussianModel
            n_setup = patching.calls["setup_functions"]
            g = GaussianModel(0)
            assert patching.calls["setup_functions"] == n_setup + 1
            assert g.covariance_activation.__name__ == "covariance_activation" and hasattr(g.covariance_activation_w_rot, "calls")
            assert g.covariance_activation_w_rot is g.build_covariance_from_scaling_rotation_w_rot
            # ... and its training_setup() (the reference's own, with its argument object) ends with a FusedAdam over the same groups
            P_ = lambda x: torch.nn.Parameter(torch.tensor(x, dtype=torch.float32))
            n = 50
            g._xyz, g._features_dc, g._features_rest = P_(np.zeros((n, 3))), P_(np.zeros((n, 1, 3))), P_(np.zeros((n, 0, 3)))
            g._scaling, g._rotation, g._opacity, g._label = P_(np.zeros((n, 3))), P_(np.ones((n, 4))), P_(np.zeros((n, 1))), P_(np.zeros((n, 1)))
            from arguments import OptimizationParams
            import argparse
            targs = OptimizationParams(argparse.ArgumentParser())
            n_train = patching.calls["training_setup"]
            g.training_setup(targs)
            assert patching.calls["training_setup"] == n_train + 1
            assert isinstance(g.optimizer, FusedAdam)
            names = [gr["name"] for gr in g.optimizer.param_groups]
            assert names[:6] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"], names
            assert g.optimizer.param_groups[0]["params"][0] is g._xyz
            # the reference's OWN getters now return results that remember their raw parameters (provenance.py): what its render() hands the
            # rasterizer can be traced back to the leaves (the rasterizer's raw-parameter path takes them on a HIP device)
            from egogaussian_amd import provenance as prov
            o_op, o_cov, o_f = prov.origin(g.get_opacity, "opacity"), prov.origin(g.get_covariance(1.0), "covariance"), prov.origin(g.get_features, "features")
            assert o_op is not None and o_op.raws[0] is g._opacity
            assert o_cov is not None and o_cov.raws[0] is g._scaling and o_cov.raws[1] is g._rotation and o_cov.extra == (1.0, None)
            assert o_f is not None and o_f.raws[0] is g._features_dc and o_f.raws[1] is g._features_rest
            assert torch.equal(g.get_opacity, torch.sigmoid(g._opacity)) and g.get_scaling is g.get_scaling
            # the reference's covariance entry point runs the installed producer
            c0 = g.covariance_activation_w_rot.calls
            g._is_object = torch.zeros(n, 1)
            g.get_rotated_covariance(torch.eye(3), 1, False, 1.0)
            assert g.covariance_activation_w_rot.calls == c0 + 1
        egogaussian_amd.uninstall()
        assert torch.autograd.is_multithreading_enabled()
        assert isinstance(GaussianModel.__dict__["get_features"], property) and GaussianModel.__dict__["get_features"].fget.__name__ == "get_features" \
            and "_fget" not in GaussianModel.__dict__["get_features"].fget.__code__.co_varnames
        assert lu.l1_loss is orig_l1 and lu.ssim is orig_ssim and ts.l1_loss is orig_l1 and early.ssim is orig_ssim
        assert not hasattr(GaussianModel.setup_functions, "__wrapped__")
    finally:
        egogaussian_amd.uninstall()
        sys.modules.pop("early_trainer", None)
        for k in ("plyfile", "pytorch3d", "pytorch3d.transforms", "wandb"):
            sys.modules.pop(k, None)
        for m in [m for m in list(sys.modules) if m.split(".")[0] in ("trainers", "utils", "scene", "gaussian_renderer", "arguments")]:
            sys.modules.pop(m)
        if REF in sys.path:
            sys.path.remove(REF)


def test_replacement_losses_step_aside_on_cpu_and_agree_with_the_mirrors():
    from egogaussian_amd import patching, losses
    l1, ss = patching.make_loss_functions()
    a, b = torch.rand(3, 40, 56), torch.rand(3, 40, 56)
    assert torch.allclose(l1(a, b), losses.l1_loss(a, b)) and torch.allclose(ss(a, b), losses.ssim(a, b))
    assert torch.allclose(ss(a, b, 11, True), losses.ssim(a, b))


@pytest.mark.gpu
def test_replacement_losses_match_the_reference_fixture_on_the_gpu():
    """losses.npz holds l1_loss / ssim of the reference's own functions on seeded image pairs (tests/golden/make_golden.py): the HIP
    replacements reproduce the values and, through autograd, the gradient of the reference's loss expression
    (1 - lambda) l1 + lambda (1 - ssim), /root/reference/trainers/train_static.py:92-95."""
    from egogaussian_amd import patching, losses
    g = np.load(os.path.join(HERE, "golden", "losses.npz"))
    l1, ss = patching.make_loss_functions()
    dev = "cuda:0"
    a = torch.tensor(g["a"], device=dev).requires_grad_(True); b = torch.tensor(g["b"], device=dev)
    n0 = dict(patching.calls)
    v_l1, v_ss = l1(a, b), ss(a, b)
    assert v_l1.grad_fn is not None and v_l1.grad_fn is v_ss.grad_fn          # ONE autograd node (one launch each way) behind the two calls
    assert patching.calls["l1_loss"] == n0["l1_loss"] + 1 and patching.calls["ssim"] == n0["ssim"] + 1      # the HIP path ran
    assert abs(float(v_l1) - float(g["l1"])) <= 1e-6 * abs(float(g["l1"])) + 1e-8
    assert abs(float(v_ss) - float(g["ssim"])) <= 2e-6
    loss = 0.8 * v_l1 + 0.2 * (1.0 - v_ss)
    loss.backward()
    a2 = torch.tensor(g["a"], device=dev).requires_grad_(True)
    losses.training_loss(a2, b).backward()
    assert float((a.grad - a2.grad).abs().max()) <= 2e-5 * float(a2.grad.abs().max())
    # the hand-mask hook of the reference's loop composes with it (train_static.py:91)
    a3 = torch.tensor(g["a"], device=dev).requires_grad_(True)
    img = a3 * 1.0
    mask = (torch.rand(1, *a3.shape[1:], device=dev) > 0.5).float()
    img.register_hook(lambda grad: grad * (1 - mask))
    (0.8 * l1(img, b) + 0.2 * (1.0 - ss(img, b))).backward()
    assert float((a3.grad - a2.grad * (1 - mask)).abs().max()) <= 2e-5 * float(a2.grad.abs().max())
