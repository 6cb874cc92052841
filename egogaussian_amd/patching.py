"""install(): ONE call, made before the reference's trainers are imported, after which they run unchanged on this package's kernels.

Swapping the two import names (`diff_gaussian_rasterization`, `simple_knn`) already runs the HIP rasterizer under the reference's
loop, but the loop then spends 0.28 ms in it and 3.9 ms in the PyTorch ops around it (bench.py `reference_shaped_step`, round 4:
240 it/s): the image loss (six grouped 11x11 convolutions and a dozen element-wise launches each way,
/root/reference/utils/loss_utils.py:57-107), the covariance producers (/root/reference/scene/gaussian_model.py:29-63) and
torch.optim.Adam over six groups (:180-198).  The trainers bind the loss functions BY NAME at import
(`from utils.loss_utils import l1_loss, ssim`, /root/reference/trainers/train_static.py:9, fine_all.py, coarse_obj_pose.py,
fine_obj.py) and build the model through `GaussianModel(...)` / `training_setup()`, so replacing the attributes of those two modules
before the trainers are imported is enough -- no reference file is edited, none is copied, nothing of the reference ships:

    import egogaussian_amd
    egogaussian_amd.install()                 # needs the reference's root on sys.path; idempotent
    from trainers.train_static import ...     # the reference, as it is

What is replaced (by attribute; each replacement falls back to the original for anything it does not cover -- CPU tensors,
batched images, other window sizes):
  utils.loss_utils.l1_loss     -> mean |x - gt|          } fused.l1_and_ssim(x, gt): called one after the other on the same pair (as the trainers
  utils.loss_utils.ssim        -> the 11x11 SSIM mean    } do), ONE HIP launch each way serves both; alone, one launch each
  scene.gaussian_model.GaussianModel.setup_functions -> the original, then adapter.attach(self, optimizer=False): the covariance
                                  producers the reference's render() calls (`covariance_activation`, `covariance_activation_w_rot`)
  scene.gaussian_model.GaussianModel.training_setup  -> the original, then adapter.attach(self): its torch.optim.Adam becomes a
                                  FusedAdam over the same groups (the reference's densification keeps editing optimizer.state)
Modules that had ALREADY bound the loss names when install() runs (a trainer imported too early) are re-pointed as well.
`uninstall()` restores everything.  `patching.calls` counts how often each replacement ran (tests spy on it).
"""
import importlib
import sys

import torch

_STATE = {}
calls = {"l1_loss": 0, "ssim": 0, "setup_functions": 0, "training_setup": 0, "l1_loss_fallback": 0, "ssim_fallback": 0}


def _hip_image_pair(a, b):
    return (torch.is_tensor(a) and torch.is_tensor(b) and a.is_cuda and b.is_cuda and a.dim() == 3 and a.shape == b.shape
            and a.dtype == torch.float32 and b.dtype == torch.float32)


def _rebind(old, new, skip):
    """Every already-imported module attribute that IS `old` (a `from utils.loss_utils import l1_loss` made before install())."""
    hits = []
    for name, mod in list(sys.modules.items()):
        if mod is None or mod in skip:
            continue
        d = getattr(mod, "__dict__", None)
        if not isinstance(d, dict):
            continue
        for k, v in list(d.items()):
            if v is old:
                d[k] = new
                hits.append((name, k))
    return hits


def make_loss_functions(orig_l1=None, orig_ssim=None):
    """(l1_loss, ssim) with the reference's signatures (/root/reference/utils/loss_utils.py:57-58,79-88) on the fused HIP kernel; what
    they do not cover goes to `orig_*` (default: this package's PyTorch mirrors, losses.py)."""
    from . import fused, losses
    orig_l1 = orig_l1 or losses.l1_loss
    orig_ssim = orig_ssim or (lambda a, b, window_size=11, size_average=True: losses.ssim(a, b))

    # The trainers call l1_loss(x, gt) and ssim(x, gt) on the SAME pair one after the other (train_static.py:92-95): the first call runs
    # the fused kernel once for both values, the second takes its half -- and one backward launch serves both (fused.l1_and_ssim).
    pending = {}

    def _pair(a, b, want):
        key = (id(a), a._version, b.data_ptr(), b._version)
        hit = pending.pop("pair", None)
        if hit is not None and hit[0] == key and hit[1] is a and hit[2] != want:
            return hit[3]
        l1v, ssv = fused.l1_and_ssim(a, b.detach())
        pending["pair"] = (key, a, want, ssv if want == "l1" else l1v)     # the other half, for the call that follows
        return l1v if want == "l1" else ssv

    def l1_loss(network_output, gt):
        if _hip_image_pair(network_output, gt) and not gt.requires_grad:
            calls["l1_loss"] += 1
            return _pair(network_output, gt, "l1")
        calls["l1_loss_fallback"] += 1
        return orig_l1(network_output, gt)

    def ssim(img1, img2, window_size=11, size_average=True):
        if window_size == 11 and size_average and _hip_image_pair(img1, img2) and not img2.requires_grad:
            calls["ssim"] += 1
            return _pair(img1, img2, "ssim")
        calls["ssim_fallback"] += 1
        return orig_ssim(img1, img2, window_size, size_average)
    l1_loss.__wrapped__, ssim.__wrapped__ = orig_l1, orig_ssim
    return l1_loss, ssim


def install(loss=True, model=True, optimizer=True, capturable=False, autograd_on_calling_thread=True):
    """loss: replace utils.loss_utils.{l1_loss, ssim}; model: wrap GaussianModel.setup_functions (covariance producers, activations that
    remember their raw parameters: provenance.py) and the class's get_features; optimizer: wrap GaussianModel.training_setup (FusedAdam;
    capturable as adapter.attach); autograd_on_calling_thread: torch.autograd.set_multithreading_enabled(False) -- loss.backward() then
    runs on the trainer's own thread instead of being handed to autograd's device thread and waited for, which is worth 0.1-0.15 ms per
    iteration of a loop whose every kernel is a few microseconds of host work (the reference trains on ONE device from one thread; a
    process that also drives other devices from other threads should pass False).  Returns a dict of what was replaced."""
    if _STATE.get("installed"):
        return _STATE["report"]
    from .adapter import attach
    report = {"loss": [], "model": [], "rebound": []}
    saved = {}
    if autograd_on_calling_thread:
        saved["autograd_mt"] = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)
        report["autograd"] = "backward on the calling thread (torch.autograd.set_multithreading_enabled(False))"
    if loss:
        lu = importlib.import_module("utils.loss_utils")             # the reference's module: its root must be on sys.path
        orig_l1, orig_ssim = lu.l1_loss, lu.ssim
        l1_loss, ssim = make_loss_functions(orig_l1, orig_ssim)
        lu.l1_loss, lu.ssim = l1_loss, ssim
        saved["loss"] = (lu, orig_l1, orig_ssim, l1_loss, ssim)
        report["loss"] = ["utils.loss_utils.l1_loss", "utils.loss_utils.ssim"]
        report["rebound"] += _rebind(orig_l1, l1_loss, (lu,)) + _rebind(orig_ssim, ssim, (lu,))
    if model or optimizer:
        gm = importlib.import_module("scene.gaussian_model")
        cls = gm.GaussianModel
        orig_setup, orig_train = cls.setup_functions, cls.training_setup
        if model:
            def setup_functions(self):
                orig_setup(self)
                calls["setup_functions"] += 1
                attach(self, optimizer=False)
            setup_functions.__wrapped__ = orig_setup
            cls.setup_functions = setup_functions
            report["model"].append("scene.gaussian_model.GaussianModel.setup_functions")
        if optimizer:
            def training_setup(self, training_args):
                orig_train(self, training_args)
                calls["training_setup"] += 1
                attach(self, optimizer=True, capturable=capturable)
            training_setup.__wrapped__ = orig_train
            cls.training_setup = training_setup
            report["model"].append("scene.gaussian_model.GaussianModel.training_setup")
        # get_features (gaussian_model.py:157-160) is a property of the class: its concatenation remembers its two halves for models that
        # adapter.attach() marked (provenance.py: the rasterizer then takes the halves, no copy either way)
        orig_feat = cls.__dict__.get("get_features")
        if model and isinstance(orig_feat, property):
            from .provenance import tag_features

            def get_features(self, _fget=orig_feat.fget):
                f = _fget(self)
                if getattr(self, "_egs_tag_features", False):
                    tag_features(f, self._features_dc, self._features_rest)
                return f
            cls.get_features = property(get_features)
            report["model"].append("scene.gaussian_model.GaussianModel.get_features")
        saved["model"] = (cls, orig_setup, orig_train, orig_feat)
    _STATE.update(installed=True, saved=saved, report=report)
    return report


def uninstall():
    if not _STATE.get("installed"):
        return
    saved = _STATE["saved"]
    if "autograd_mt" in saved:
        torch.autograd.set_multithreading_enabled(saved["autograd_mt"])
    if "loss" in saved:
        lu, orig_l1, orig_ssim, l1_new, ssim_new = saved["loss"]
        lu.l1_loss, lu.ssim = orig_l1, orig_ssim
        _rebind(l1_new, orig_l1, (lu,)); _rebind(ssim_new, orig_ssim, (lu,))
    if "model" in saved:
        cls, orig_setup, orig_train, orig_feat = saved["model"]
        cls.setup_functions, cls.training_setup = orig_setup, orig_train
        if isinstance(orig_feat, property):
            cls.get_features = orig_feat
    _STATE.clear()
