"""Whole-training-step hipGraph capture (PyTorch's torch.cuda.graph on ROCm = hipGraph).

Once the kernels are fast the iteration of /root/reference/trainers/train_static.py:67-138 is bound by the host: ~60
kernel launches, three Python autograd Functions and the optimizer cost ~0.75 ms of CPU per step against ~0.55 ms of GPU
work at 500k Gaussians.  The step has static shapes (the data-dependent instance count is handled by the capacity-bounded
`egs_forward_enqueue`, the optimizer by `FusedAdam(capturable=True)`), so it is captured ONCE -- covariance, render
forward, loss, backward, Adam -- and replayed with one launch per iteration; per-iteration inputs (camera matrices and
ground-truth image) are copied into static device tensors first.

Validity: a replayed frame whose instance count exceeded the captured capacity is clipped, hence wrong.  The graph
tracks the running maximum of R on the device (`max_instances()`); callers check it at their next synchronisation point
(the reference synchronises every iteration anyway through `loss.item()`, trainers/train_static.py:112) and re-capture
with a larger capacity if needed (`recapture()`).

What is baked into the captured launches and therefore needs `recapture()` when it changes: the tensors themselves (densification
and pruning replace them), the number of Gaussians, the image size, `pc.active_sh_degree` (the reference raises it every 1000
iterations, scene/gaussian_model.py:176-178), the background tensor's address, `lambda_dssim`.  What does not: camera and
ground-truth contents (copied in per call) and the learning rates (device scalars; `__call__` pushes host-side edits of
`param_groups[i]["lr"]` -- the reference's per-iteration `update_learning_rate` -- before each replay).
"""
import torch

from . import _C
from .fused import l1_ssim_loss
from .renderer import render
from .scene_synth import Pipe


def pack_camera(cam):
    """The three camera tensors render() reads, as one float32[35] block: world_view_transform, full_proj_transform, camera_center."""
    return torch.cat([cam.world_view_transform.reshape(-1), cam.full_proj_transform.reshape(-1), cam.camera_center.reshape(-1)]).float()


def pack_frame(cam, gt):
    """One resident tensor per training frame: the ground-truth image followed by the camera block.  GraphedTrainStep(frame) then
    refreshes both static inputs of the captured step with ONE device copy."""
    return torch.cat([gt.reshape(-1).float(), pack_camera(cam).to(gt.device)])


class _StaticCamera:
    """Camera whose tensors are fixed device buffers; `load(cam)` copies another camera of the same intrinsics in."""

    def __init__(self, cam, storage=None):
        """storage: an existing float32[35] device view to live in (the tail of a packed frame), else its own block."""
        self.image_height, self.image_width, self.FoVx, self.FoVy = cam.image_height, cam.image_width, cam.FoVx, cam.FoVy
        packed = pack_camera(cam)
        if storage is None:
            self.packed = packed
        else:
            self.packed = storage
            self.packed.copy_(packed)
        self.world_view_transform = self.packed[0:16].view(4, 4)
        self.full_proj_transform = self.packed[16:32].view(4, 4)
        self.camera_center = self.packed[32:35]

    def load(self, cam):
        same = (cam.image_height, cam.image_width, cam.FoVx, cam.FoVy) == (self.image_height, self.image_width, self.FoVx, self.FoVy)
        assert same, "a captured step is specific to one image size and field of view"
        packed = getattr(cam, "packed", None)
        if packed is not None and packed.shape == self.packed.shape:   # cameras that keep the three in one block: one copy
            self.packed.copy_(packed, non_blocking=True)
            return
        self.world_view_transform.copy_(cam.world_view_transform, non_blocking=True)
        self.full_proj_transform.copy_(cam.full_proj_transform, non_blocking=True)
        self.camera_center.copy_(cam.camera_center, non_blocking=True)


class GraphedTrainStep:
    def __init__(self, pc, optimizer, bg, lambda_dssim=0.2, pipe=Pipe, render_kwargs=None, densify_stats=False):
        """densify_stats: the captured step also keeps the per-iteration densification statistics (trainers/train_static.py:125-127:
        max_radii2D, xyz_gradient_accum, denom) -- updated by the rasterizer's backward itself, no launch of their own."""
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FusedAdam(capturable=True)")
        self.pc, self.opt, self.bg, self.lam, self.pipe = pc, optimizer, bg, lambda_dssim, pipe
        self.densify_stats = densify_stats
        self.render_kwargs = render_kwargs or {}
        self.graph = None
        self._max_r = None
        self.loss_sum = None

    def _body(self):
        out = render(self.cam, self.pc, self.pipe, self.bg, fused_densify_stats=self.densify_stats, **self.render_kwargs)
        # the loss value and the running sum are produced by the loss BACKWARD kernel (nothing reads them before): two launches less
        loss = l1_ssim_loss(out["render"], self.gt, self.lam, running_sum=self.loss_sum, defer_value=True)
        loss.backward(gradient=self._one)                            # a resident 1.0: no fill kernel per iteration
        self.opt.step()
        return loss.detach(), out

    def capture(self, cam, gt, warmup=3, capacity_margin=1.25):
        """Runs `warmup` eager iterations on (cam, gt) -- they are real training steps -- then records (without executing) one
        more into the graph."""
        dev = gt.device
        if isinstance(cam, _StaticCamera):                           # recapture: keep the static buffers
            if gt is not self.gt:
                self.gt.copy_(gt)
        else:
            self._frame = torch.empty(gt.numel() + 35, device=dev, dtype=torch.float32)      # image, then camera: one copy target
            self.gt = self._frame[:gt.numel()].view(gt.shape)
            self.gt.copy_(gt)
            self.cam = _StaticCamera(cam, storage=self._frame[gt.numel():])
        self._one = torch.ones((), device=dev)
        if getattr(self, "loss_sum", None) is None:
            self.loss_sum = torch.zeros((), device=dev)                  # sum of the losses of every iteration run through this object
        P = self.pc.get_xyz.shape[0]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                          # eager: sets the capacity hint, allocator pools, lazy state
                self.opt.zero_grad(set_to_none=True)
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.capacity = max(int(_C.stats["num_rendered"] * capacity_margin), _C.stats["capacity"])
        _C.set_capacity_hint(self.capacity, dev)
        self.P = P
        self._max_r = torch.zeros(1, dtype=torch.int64, device=dev)      # raised by every replayed forward (library side)
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        _C.set_running_max(dev, self._max_r)
        # capture_begin/capture_end directly: the torch.cuda.graph context manager also runs gc.collect() and
        # torch.cuda.empty_cache() (3 ms at this size), which a trainer that re-captures after every densification pays each time
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(side):
                # thread_local: calls other threads make meanwhile (e.g. the RCCL watchdog polling its events) must not abort the capture
                self.graph.capture_begin(capture_error_mode="thread_local")
                try:
                    self.loss, out = self._body()
                    self.image = out["render"].detach()
                    self.radii = out["radii"]
                    self.viewspace_grad = out["viewspace_points"].grad
                    self._total = _C.stats["total_view"]
                finally:
                    self.graph.capture_end()
        finally:
            _C.set_running_max(dev, None)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        return self

    def recapture(self, cam=None, gt=None, warmup=1, capacity_margin=1.25):
        """Capture again with the model as it is now -- after densification / pruning replaced the parameters, or after ok()
        reported a frame that outgrew the capacity.  Like capture(), the `warmup` eager iterations are real training steps."""
        cam = self.cam if cam is None else cam
        gt = self.gt if gt is None else gt
        self.graph = None                                            # drop the old graph and its private memory pool first
        return self.capture(cam, gt, warmup=warmup, capacity_margin=capacity_margin)

    def __call__(self, cam, gt=None):
        """One training iteration: copy inputs in, replay.  Returns the (device, static) loss tensor.
        Either (camera, ground-truth image) or one packed frame from pack_frame() (a single copy)."""
        if gt is None:
            self._frame.copy_(cam, non_blocking=True)
        else:
            self.cam.load(cam)
            self.gt.copy_(gt, non_blocking=True)
        self.opt.sync_lr()                                           # a fill per group whose learning rate was edited since the last call
        self.graph.replay()
        return self.loss

    def last_instance_count(self):
        """Instances the most recent replay bucketed; call after synchronising.  More than `capacity` means that frame was
        clipped."""
        return int(self._total.item())

    def max_instances(self):
        """Largest R over every replay so far (reads a device scalar: synchronises)."""
        return int(self._max_r.item())

    def ok(self):
        """True if no replayed frame exceeded the captured capacity (i.e. every one of them was rendered completely)."""
        return self.max_instances() <= self.capacity
