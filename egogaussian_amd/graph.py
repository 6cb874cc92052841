"""Whole-training-step hipGraph capture (PyTorch's torch.cuda.graph on ROCm = hipGraph).

Once the kernels are fast the iteration of /root/reference/trainers/train_static.py:67-138 is bound by the host: ~60
kernel launches, three Python autograd Functions and the optimizer cost ~0.75 ms of CPU per step against ~0.55 ms of GPU
work at 500k Gaussians.  The step has static shapes (the data-dependent instance count is handled by the capacity-bounded
`egs_forward_enqueue`, the optimizer by `FusedAdam(capturable=True)`), so it is captured ONCE -- covariance, render
forward, loss, backward, Adam -- and replayed with one launch per iteration; per-iteration inputs (camera matrices,
ground-truth image and, for the `fine_all` call shape, the object's accumulated rotation and the hand-mask gate) are
copied into static device tensors first, as ONE copy when the caller keeps them packed (`pack_frame`).

Overflow safety.  A replayed frame whose instance count exceeds the captured capacity is clipped, hence wrong.  The
forward chain writes an overflow word on the device (`_C.StepGuard`); the backward's fused densification statistics and
the Adam launch read it and do NOTHING for such a frame -- parameters, moments, step counts and statistics stay bit for
bit what they were, so every optimizer step follows a complete render, as in the reference (train_static.py:110-138).
The frame is merely lost.  `ok()` (one host read of the running maximum) tells whether that ever happened; with
`check_every=K` the object looks itself every K calls and re-captures with a larger capacity.

What is baked into the captured launches and therefore needs `recapture()` when it changes: the tensors themselves
(densification and pruning of a plain model replace them; a capacity.CapacityGaussians model keeps them, and its live row
count is a device word the kernels read), the image size, `pc.active_sh_degree` (the reference raises it every 1000
iterations, scene/gaussian_model.py:176-178), the background tensor's address, `lambda_dssim`.  What does not: camera,
ground truth, accum_R and gate contents (copied in per call) and the learning rates (device scalars; `__call__` pushes
host-side edits of `param_groups[i]["lr"]` -- the reference's per-iteration `update_learning_rate` -- before each replay).
"""
import torch

from . import _C
from .fused import l1_ssim_loss
from .renderer import render
from .scene_synth import Pipe


def pack_camera(cam):
    """The three camera tensors render() reads, as one float32[35] block: world_view_transform, full_proj_transform, camera_center."""
    return torch.cat([cam.world_view_transform.reshape(-1), cam.full_proj_transform.reshape(-1), cam.camera_center.reshape(-1)]).float()


def frame_layout(n_img, n_pix, dynamic=False, gated=False):
    """Float offsets of the segments of a packed frame (each starts on a 16-byte boundary): -> ({name: (begin, end)}, size)."""
    up4 = lambda x: (x + 3) & ~3
    off = {"gt": (0, n_img)}
    end = up4(n_img)
    off["cam"] = (end, end + 35); end = up4(end + 35)
    if dynamic:
        off["accum_R"] = (end, end + 9); end = up4(end + 9)
    if gated:
        off["gate"] = (end, end + n_pix); end = up4(end + n_pix)
    return off, end


def pack_frame(cam, gt, accum_R=None, gate=None):
    """One resident tensor per training frame: the ground-truth image, the camera block and -- for a step captured with
    dynamic=True / gated=True -- the object's accumulated rotation (3x3) and the per-pixel gradient gate (1 - hand mask, [H,W]).
    GraphedTrainStep(frame) then refreshes every static input of the captured step with ONE device copy."""
    off, size = frame_layout(gt.numel(), gt.shape[-2] * gt.shape[-1], accum_R is not None, gate is not None)
    f = torch.zeros(size, device=gt.device, dtype=torch.float32)
    f[off["gt"][0]:off["gt"][1]] = gt.reshape(-1)
    f[off["cam"][0]:off["cam"][1]] = pack_camera(cam).to(gt.device)
    if accum_R is not None:
        f[off["accum_R"][0]:off["accum_R"][1]] = accum_R.reshape(-1).to(gt.device)
    if gate is not None:
        f[off["gate"][0]:off["gate"][1]] = gate.reshape(-1).to(gt.device)
    return f


class _StaticCamera:
    """Camera whose tensors are fixed device buffers; `load(cam)` copies another camera of the same intrinsics in."""

    def __init__(self, cam, storage=None):
        """storage: an existing float32[35] device view to live in (part of a packed frame), else its own block."""
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
    def __init__(self, pc, optimizer, bg, lambda_dssim=0.2, pipe=Pipe, render_kwargs=None, densify_stats=False, dynamic=False,
                 which_object=1, gated=False, check_every=0, steps_per_replay=1, fuse_optimizer=True, double_buffer=False, loss_grad_in_blend=True):
        """densify_stats: the captured step also keeps the per-iteration densification statistics (trainers/train_static.py:125-127:
                       max_radii2D, xyz_gradient_accum, denom) -- updated by the rasterizer's backward itself, no launch of their own.
        dynamic:       the `fine_all` call shape (/root/reference/trainers/fine_all.py:88-93): render(..., rot_cov=True,
                       accum_R=<static 3x3, refreshed per call>, which_object=which_object, during_training=False).
        gated:         the image gradient is multiplied by a per-pixel gate refreshed per call -- the reference's
                       `render_image.register_hook(lambda grad: grad * (1 - hand_mask))` (train_static.py:91, fine_all.py:94).
        check_every:   K > 0: every K calls read the overflow maximum (one host synchronisation) and re-capture with a larger
                       instance capacity if a frame was clipped (its update was skipped, see the module docstring).
        steps_per_replay: S > 1 captures S complete iterations back to back, each on its own static frame; one launch then runs
                       S training steps on S frames (`__call__` takes the S packed frames as one [S, frame] tensor or a list).  A
                       graph launch leaves the GPU idle for ~9 us before its first node; this divides that by S.
        double_buffer: capture the step TWICE, on two sets of static frame buffers, and alternate between the two graphs: the copy of
                       call k + 1's frames then runs on a side stream while call k's replay is still working (it only has to wait for
                       replay k - 1, the last user of its buffer), instead of between two replays (packed frames only; 31 MB per five
                       960x540 frames: 2.7 us per step).  Results are those of the single-buffered step.  MEASURED at config C: 1.6 % slower than
                       the single graph (3 175 vs 3 225 it/s) -- the event waits between the streams cost more than the hidden copy --,
                       so it is off by default and bench.py does not use it.
        loss_grad_in_blend: the captured step has NO loss-backward launch: the rasterizer's backward blend computes the image loss's gradient
                       for its tile itself, from the maps the loss forward leaves, bit-identical to the launch it replaces
                       (fused.l1_ssim_loss(raster_lossgrad=True), include/egs_raster.h egs_backward_lossgrad): nine launches per step -> eight.
        fuse_optimizer: the parameters render() hands to the rasterizer as stored take their Adam step inside its backward
                       (renderer.render, optimizer=): no gradient arrays, no optimizer launch for them; the step's loss must then
                       depend on the model through that one render only -- which is the step this class captures.  Results are
                       bit-identical either way."""
        self.fuse_optimizer = bool(fuse_optimizer)
        import os
        self.loss_grad_in_blend = bool(loss_grad_in_blend) and not os.environ.get("EGS_NO_LOSS_GRAD_IN_BLEND")      # (A/B switch for bench.py)
        self.double_buffer = bool(double_buffer)
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FusedAdam(capturable=True)")
        self.pc, self.opt, self.bg, self.lam, self.pipe = pc, optimizer, bg, lambda_dssim, pipe
        self.densify_stats = densify_stats
        self.dynamic, self.which_object, self.gated = bool(dynamic), which_object, bool(gated)
        self.render_kwargs = dict(render_kwargs or {})
        self.check_every = int(check_every)
        self.steps_per_replay = max(1, int(steps_per_replay))
        self.graph = None
        self._sets = None
        self.guard = None
        self.loss_sum = None
        self.recaptures = 0               # re-captures this object did on its own (overflow)
        self.skipped_frames_seen = 0      # overflow events noticed by check()
        self._calls = 0

    def _body(self, k=0):
        """One training iteration on static frame k."""
        f = self._slots[k]
        kw = dict(self.render_kwargs)
        if self.dynamic:
            kw.update(rot_cov=True, accum_R=f["accum_R"], which_object=self.which_object, during_training=False)
        out = render(f["cam"], self.pc, self.pipe, self.bg, fused_densify_stats=self.densify_stats, guard=self.guard,
                     optimizer=self.opt if self.fuse_optimizer else None, color_only=True, **kw)      # (the loss reads the colour image only)
        # the loss value and the running sum are produced by the loss BACKWARD kernel (nothing reads them before): two launches less
        loss = l1_ssim_loss(out["render"], f["gt"], self.lam, grad_gate=f["gate"] if self.gated else None, running_sum=self.loss_sum,
                            defer_value=True, raster_prologue=True, raster_lossgrad=self.loss_grad_in_blend)
        loss.backward(gradient=self._one)                            # a resident 1.0: no fill kernel per iteration
        self.opt.step()                                              # whatever the backward did not step itself (fuse_optimizer)
        return loss.detach(), out

    def _frame_layout(self, gt):
        return frame_layout(gt.numel(), gt.shape[-2] * gt.shape[-1], self.dynamic, self.gated)

    def capture(self, cam, gt, warmup=3, capacity_margin=1.25, accum_R=None, gate=None, capacity_cams=None, capacity=None):
        """Runs `warmup` eager iterations on (cam, gt) -- they are real training steps -- then records (without executing) one
        more into the graph.  capacity_cams: further cameras whose instance counts size the captured capacity (a forward-only
        render each); without them the capacity is `capacity_margin` x the count of `cam` alone, and R varies across views.
        capacity: the instance capacity to capture with, as is (overrides the margin rule; tests use it to provoke an overflow)."""
        dev = gt.device
        if isinstance(cam, _StaticCamera):                           # recapture: keep the static buffers
            if gt is not self.gt:
                self.gt.copy_(gt)
        else:
            off, size = self._frame_layout(gt)
            # per captured iteration: image, camera[, accum_R][, gate] -- ONE copy target for all of them
            self._frames = torch.empty((self.steps_per_replay, size), device=dev, dtype=torch.float32)
            self._slots = []
            for k in range(self.steps_per_replay):
                fr = self._frames[k]
                slot = {"gt": fr[off["gt"][0]:off["gt"][1]].view(gt.shape), "accum_R": None, "gate": None}
                slot["gt"].copy_(gt)
                slot["cam"] = _StaticCamera(cam, storage=fr[off["cam"][0]:off["cam"][1]])
                if self.dynamic:
                    slot["accum_R"] = fr[off["accum_R"][0]:off["accum_R"][1]].view(3, 3)
                    slot["accum_R"].copy_(torch.eye(3, device=dev) if accum_R is None else accum_R)
                if self.gated:
                    slot["gate"] = fr[off["gate"][0]:off["gate"][1]].view(gt.shape[-2], gt.shape[-1])
                    slot["gate"].copy_(torch.ones(gt.shape[-2:], device=dev) if gate is None else gate)
                self._slots.append(slot)
            self._frame = self._frames[0]
            first = self._slots[0]                                   # (the single-iteration names)
            self.gt, self.cam, self.accum_R, self.gate = first["gt"], first["cam"], first["accum_R"], first["gate"]
        self._one = torch.ones((), device=dev)
        if getattr(self, "loss_sum", None) is None:
            self.loss_sum = torch.zeros((), device=dev)                  # sum of the losses of every iteration run through this object
        self.guard = _C.StepGuard(dev)
        self.opt.guard = self.guard                                  # the Adam launch of an overflowed frame does nothing
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            r_seen = 0
            if capacity_cams:
                kw = dict(self.render_kwargs)
                if self.dynamic:
                    kw.update(rot_cov=True, accum_R=self.accum_R, which_object=self.which_object, during_training=False)
                with torch.no_grad():
                    for c in capacity_cams:
                        render(c, self.pc, self.pipe, self.bg, **kw)
                        r_seen = max(r_seen, _C.stats["num_rendered"])
            for _ in range(max(1, warmup)):                          # eager: sets the capacity hint, allocator pools, lazy state
                self.opt.zero_grad(set_to_none=True)
                self._body()
                r_seen = max(r_seen, _C.stats["num_rendered"])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.capacity = max(int(r_seen * capacity_margin), _C.stats["capacity"], getattr(self, "_min_capacity", 0))
        if capacity is not None:
            self.capacity = max(int(capacity), 1)
        _C.set_capacity_hint(self.capacity, dev)
        self.P = self.pc.get_xyz.shape[0]
        self._model_version = getattr(self.pc, "model_version", 0)
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        # capture_begin/capture_end directly: the torch.cuda.graph context manager also runs gc.collect() and
        # torch.cuda.empty_cache() (3 ms at this size), which a trainer that re-captures after every densification pays each time
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # thread_local: calls other threads make meanwhile (e.g. the RCCL watchdog polling its events) must not abort the capture
            self.graph.capture_begin(capture_error_mode="thread_local")
            try:
                self.losses = []
                for k in range(self.steps_per_replay):
                    if k:
                        self.opt.zero_grad(set_to_none=True)         # (the next backward writes fresh gradients instead of accumulating)
                    self.loss, out = self._body(k)
                    self.losses.append(self.loss)
                self.image = out["render"].detach()
                self.radii = out["radii"]
                self.visibility_filter = out["visibility_filter"]      # follows every replay (a view of the rasterizer's saved state)
                self.viewspace_grad = out["viewspace_points"].grad
            finally:
                self.graph.capture_end()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._sets = None
        if self.double_buffer:
            # the same iterations recorded once more on a second set of static frames; what the two captures share -- parameters,
            # optimizer state, guard words, loss_sum -- is shared by address
            first = dict(graph=self.graph, frames=self._frames, slots=self._slots, loss=self.loss, losses=self.losses, image=self.image,
                         radii=self.radii, visibility_filter=self.visibility_filter, viewspace_grad=self.viewspace_grad)
            frames2 = self._frames.clone()
            off, _ = self._frame_layout(self.gt)
            slots2 = []
            for k in range(self.steps_per_replay):
                fr, src = frames2[k], self._slots[k]
                sl = {"gt": fr[off["gt"][0]:off["gt"][1]].view(src["gt"].shape), "accum_R": None, "gate": None}
                sl["cam"] = _StaticCamera(src["cam"], storage=fr[off["cam"][0]:off["cam"][1]])
                if self.dynamic:
                    sl["accum_R"] = fr[off["accum_R"][0]:off["accum_R"][1]].view(3, 3)
                if self.gated:
                    sl["gate"] = fr[off["gate"][0]:off["gate"][1]].view(src["gt"].shape[-2], src["gt"].shape[-1])
                slots2.append(sl)
            self._slots = slots2
            self.opt.zero_grad(set_to_none=True)
            g2 = torch.cuda.CUDAGraph()
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                g2.capture_begin(capture_error_mode="thread_local")
                try:
                    losses2 = []
                    for k in range(self.steps_per_replay):
                        if k:
                            self.opt.zero_grad(set_to_none=True)
                        loss2, out2 = self._body(k)
                        losses2.append(loss2)
                finally:
                    g2.capture_end()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            second = dict(graph=g2, frames=frames2, slots=slots2, loss=loss2, losses=losses2, image=out2["render"].detach(), radii=out2["radii"],
                          visibility_filter=out2["visibility_filter"], viewspace_grad=out2["viewspace_points"].grad)
            self._slots = first["slots"]
            self._sets = [first, second]
            self._copy_stream = torch.cuda.Stream(device=dev)
            self._done = [torch.cuda.Event(), torch.cuda.Event()]       # replay of set b finished reading its frames
            self._copied = [torch.cuda.Event(), torch.cuda.Event()]
            self._used = [False, False]
        self.guard.running_max.zero_(); self.guard.overflow.zero_()
        return self

    def recapture(self, cam=None, gt=None, warmup=1, capacity_margin=1.25, capacity_cams=None):
        """Capture again with the model as it is now -- after densification / pruning replaced the parameters, or after ok()
        reported a frame that outgrew the capacity.  Like capture(), the `warmup` eager iterations are real training steps."""
        cam = self.cam if cam is None else cam
        gt = self.gt if gt is None else gt
        self.graph = None                                            # drop the old graph and its private memory pool first
        self._sets = None
        return self.capture(cam, gt, warmup=warmup, capacity_margin=capacity_margin, capacity_cams=capacity_cams)

    def __call__(self, cam, gt=None, accum_R=None, gate=None, ready=None):
        """One training iteration (steps_per_replay of them): copy inputs in, replay.  Returns the (device, static) loss tensor of
        the last iteration (`self.losses` has all).  Either (camera, ground-truth image[, accum_R][, gate]) or packed frames from
        pack_frame(): one for a single-iteration step, a [S, frame] tensor (one copy) or a list of S for steps_per_replay = S.
        ready (double_buffer only): a torch.cuda.Event recorded after the packed frames were COMPLETELY written; the side-stream copy
        waits for it.  Without it the copy waits for everything enqueued on the current stream so far -- always correct, but frames
        produced on the current stream right before the call then serialise behind the replay that is still running."""
        if gt is None and self._sets is not None:
            return self._call_double_buffered(cam, ready)
        if gt is None:
            if isinstance(cam, (list, tuple)):
                for k, fr in enumerate(cam):
                    self._frames[k].copy_(fr, non_blocking=True)
            else:
                self._frames.copy_(cam.view(self._frames.shape), non_blocking=True)
        else:
            if self.steps_per_replay != 1:
                raise ValueError("steps_per_replay > 1 takes packed frames")
            self.cam.load(cam)
            self.gt.copy_(gt, non_blocking=True)
            if self.dynamic and accum_R is not None:
                self.accum_R.copy_(accum_R, non_blocking=True)
            if self.gated and gate is not None:
                self.gate.copy_(gate, non_blocking=True)
        if getattr(self.pc, "model_version", 0) != self._model_version:
            raise RuntimeError("GraphedTrainStep: the model reallocated its arrays (CapacityGaussians.grow) after this step was captured; "
                               "the captured launches point at freed memory -- call recapture() first")
        self.opt.sync_lr()                                           # a fill per group whose learning rate was edited since the last call
        self.graph.replay()
        self._calls += 1
        if self.check_every > 0 and self._calls % self.check_every == 0:
            self.check()
        return self.loss

    def _call_double_buffered(self, frames, ready=None):
        """Packed frames, two captured graphs: the copy into set b's static frames runs on a side stream as soon as set b's previous
        replay has finished, i.e. under the replay of the other set that is still running.  The copy reads the caller's tensors on
        that side stream, so it must be ordered after whatever WROTE them: the caller's `ready` event, else the current stream as it
        stands now; and the tensors are marked as used by the side stream so that the allocator does not recycle them under the copy."""
        b = self._calls & 1
        st = self._sets[b]
        main = torch.cuda.current_stream(st["frames"].device)
        if self._used[b]:
            self._copy_stream.wait_event(self._done[b])
        if ready is not None:
            self._copy_stream.wait_event(ready)
        else:
            self._copy_stream.wait_stream(main)                      # whatever produced the frames (and, the first time, the capture)
        with torch.cuda.stream(self._copy_stream):
            if isinstance(frames, (list, tuple)):
                for k, fr in enumerate(frames):
                    st["frames"][k].copy_(fr, non_blocking=True)
                    fr.record_stream(self._copy_stream)
            else:
                st["frames"].copy_(frames.view(st["frames"].shape), non_blocking=True)
                frames.record_stream(self._copy_stream)
            self._copied[b].record(self._copy_stream)
        if getattr(self.pc, "model_version", 0) != self._model_version:
            raise RuntimeError("GraphedTrainStep: the model reallocated its arrays (CapacityGaussians.grow) after this step was captured; "
                               "the captured launches point at freed memory -- call recapture() first")
        self.opt.sync_lr()
        main.wait_event(self._copied[b])
        st["graph"].replay()
        self._done[b].record(main)
        self._used[b] = True
        self.loss, self.losses, self.image, self.radii = st["loss"], st["losses"], st["image"], st["radii"]
        self.visibility_filter, self.viewspace_grad = st["visibility_filter"], st["viewspace_grad"]
        self._calls += 1
        if self.check_every > 0 and self._calls % self.check_every == 0:
            self.check()
        return self.loss

    def check(self, capacity_margin=1.25):
        """Reads the running maximum of the instance count (synchronises).  If a replayed frame was clipped -- its parameter
        update was skipped on the device -- re-captures with room for it and returns False; True otherwise."""
        if self.ok():
            return True
        self.skipped_frames_seen += 1
        self._min_capacity = int(self.max_instances() * capacity_margin) + 65536
        self.recaptures += 1
        self.recapture(warmup=1, capacity_margin=capacity_margin)
        return False

    def last_instance_count(self):
        """Instances the most recent replay bucketed; call after synchronising.  More than `capacity` means that frame was
        clipped (and its update skipped)."""
        return int(self.guard.overflow[1].item()) & 0xffffffff

    def last_frame_overflowed(self):
        return bool(int(self.guard.overflow[0].item()))

    def max_instances(self):
        """Largest R over every replay since the capture (reads a device scalar: synchronises)."""
        return int(self.guard.running_max.item())

    def ok(self):
        """True if no replayed frame exceeded the captured capacity (i.e. every one of them was rendered completely)."""
        return self.max_instances() <= self.capacity
