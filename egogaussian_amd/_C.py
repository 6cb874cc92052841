"""`_C` namespace of the drop-in `diff_gaussian_rasterization` package, backed by libegs_raster.so.

Mirrors the pybind functions of the upstream CUDA extension the reference imports at
/root/reference/gaussian_renderer/__init__.py:14 (names, argument order and return tuples as listed in
SURVEY.md section 8b): rasterize_gaussians, rasterize_gaussians_backward, mark_visible.  "Absent" optional
inputs are zero-length tensors, exactly as upstream passes them.

Tensors must live on a HIP device (torch device type "cuda" on ROCm).  There is no CPU path.
"""
import ctypes as C
import os

import torch

from . import lib as _lib
from . import _hip


# Diagnostics of the most recent forward of this process (instance count, capacity, retries, the device-side total, the image
# buffer for the measurement views).  Read by bench.py, tools and tests; NOT part of any data path -- nothing the package
# computes depends on it.
stats = {"num_rendered": 0, "capacity": 0, "retries": 0}
_capacity_hint = {}                # device index -> instance capacity the next forward is enqueued against
_pinned_counts = {}                # device index -> page-locked host buffer for the per-workgroup instance counts


class StepGuard:
    """Device words a captured training step shares between its forward, backward and optimizer launches (include/egs_raster.h,
    "overflow word"):  `overflow` uint32[2] -- written by every forward: [0] = 1 when the frame needed more instances than the captured
    capacity (its image is clipped), else 0, [1] = the frame's instance count; the backward's fused statistics and FusedAdam(capturable) read it and do nothing
    for such a frame.  `running_max` int64[1] -- raised to the instance count of every forward, so one host read tells whether
    ANY replay overflowed and by how much."""

    def __init__(self, device, deferred=False):
        self.overflow = torch.zeros(2, dtype=torch.int32, device=device)      # [0] flag, [1] instance count of the latest frame
        self.running_max = torch.zeros(1, dtype=torch.int64, device=device)
        # deferred=True (EAGER loops): forwards that carry this guard are enqueued against the capacity hint WITHOUT the host wait for
        # the instance count (egs_forward_enqueue, as under graph capture); the count and the overflow word of a frame are read from
        # page-locked memory at a FOLLOWING forward (the next one whose call finds the copy landed; never a wait).  A frame that did not fit is then already voided on the device (the backward's
        # statistics and the Adam step of a FusedAdam(capturable=True) honour the overflow word), `overflows` counts it, and the next
        # frame runs with room.  Its IMAGE and loss value are clipped: use for training loops, not for evaluation renders.
        self.deferred = bool(deferred)
        self.overflows = 0                    # frames of this guard that were clipped (found at the following forward / check())
        self.frames = 0
        self.last_R = 0                       # rectangle instances of the last CHECKED frame
        self._pending = []                    # frames not checked yet, oldest first: (page-locked counts, P, capacity, device index, event behind its copies)
        self._pinned = [None] * _DEFER_RING

    def check(self):
        """Settle every frame that has not been checked yet (waits for them).  -> True when no frame of this guard has been clipped."""
        _settle(self, wait=True)
        return self.overflows == 0


_DEFER_RING = 8                       # deferred frames in flight before a forward waits for the oldest (a host that runs far ahead)


def _settle(guard, wait=False):
    """Read the counts the earlier deferred forwards of `guard` copied out, oldest first, as far as they have landed (a sentinel
    word tells); raise the capacity hint when one did not fit.  Never waits for the GPU unless `wait` or the ring is full: a host
    that runs ahead of the GPU (no synchronisation in its loop) checks a frame a few forwards later instead."""
    while guard._pending:
        pinned, P, cap, key, ev = guard._pending[0]
        nb = (P + 255) // 256
        tail = pinned[nb:nb + 2]
        if int(tail[0]) == -1:                # the copy that ends the frame's chain has not landed yet (the sentinel is still there)
            if not (wait or len(guard._pending) >= _DEFER_RING - 1):
                return
            ev.synchronize()                  # THIS frame's copies only: later frames and other streams keep running
        guard._pending.pop(0)
        R = int(_lib.load().egs_sum_counts(int(P), C.c_void_p(pinned.data_ptr())))
        clipped, kept = int(tail[0]), int(tail[1]) & 0xffffffff
        guard.last_R = R
        if clipped:
            guard.overflows += 1
            stats["retries"] += 1
        if clipped or R > cap:
            _capacity_hint[key] = max(_capacity_hint.get(key, 0), int(max(R, kept) * 1.25) + 65536)


def binning_passes(P, W, H):
    """Upper bound on the 9-bit radix passes of the per-tile (depth, index) sort for P Gaussians (a tile runs
    ceil(bits(depth range) / 9) depth passes, and the index passes only if two of its entries share a depth)."""
    lay = _lib.BinningLayout()
    _lib.check(_lib.load().egs_get_binning_layout(int(P), 0, int(W), int(H), C.byref(lay)))
    return int(lay.index_passes) + 4


# ---- per-call flags (include/egs_raster.h EGS_CALL_*, ABI 6) ----------------------------------------------------------------------
# The library keeps no switches: every forward says in its own flags word whether it culls, fuses the count pass, sorts inside the blend.
# A caller picks them per call -- rasterize_gaussians(..., debug=<bits>), or GaussianRasterizationSettings(debug=<bits>): the reference's
# `debug` field is that word; True / 1 = CALL_SYNC, its old meaning -- and two trainers in one process with different settings do not
# meet.  What follows is the DEFAULT this Python layer ORs into calls that do not say: the environment's A/B switches, and the
# set_* helpers the tests use as context managers (tests/common.py) -- Python state of this module, not of the library.
CALL_SYNC, CALL_KEEP_ALL_INSTANCES, CALL_SEPARATE_COUNT, CALL_SEPARATE_SORT, CALL_BALLOT_RANK = (
    _lib.CALL_SYNC, _lib.CALL_KEEP_ALL_INSTANCES, _lib.CALL_SEPARATE_COUNT, _lib.CALL_SEPARATE_SORT, _lib.CALL_BALLOT_RANK)
_env_on = lambda name: os.environ.get(name, "") not in ("", "0")
default_call_flags = (CALL_SEPARATE_COUNT if _env_on("EGS_NO_FUSED_COUNT") else 0) | (CALL_SEPARATE_SORT if _env_on("EGS_NO_SORT_IN_BLEND") else 0)


def call_flags(debug=0):
    """The flags word of a forward: the caller's `debug` (bool: CALL_SYNC; int: EGS_CALL_* bits) OR this module's defaults."""
    return (int(debug) if not isinstance(debug, bool) else (CALL_SYNC if debug else 0)) | default_call_flags


def _set_default(bit, want_bit):
    global default_call_flags
    old = bool(default_call_flags & bit)
    default_call_flags = (default_call_flags | bit) if want_bit else (default_call_flags & ~bit)
    return old


def set_tile_culling(on):
    """Default of calls that do not say: tile culling of never-contributing instances on (the library's default) or off
    (CALL_KEEP_ALL_INSTANCES: the reference's full rectangles, internal lists bit for bit the reference algorithm's).  -> the previous setting."""
    return not _set_default(CALL_KEEP_ALL_INSTANCES, not on)


def set_sort_in_blend(on):
    """Default of calls that do not say: the per-tile sort inside the forward blend's launch (on) or as launches of its own
    (CALL_SEPARATE_SORT).  -> the previous setting."""
    return not _set_default(CALL_SEPARATE_SORT, not on)


def set_fused_count(on):
    """Default of calls that do not say: the count pass of the tile bucketing inside the preprocess launch whenever a forward has a
    placement buffer (on) or as a launch of its own (CALL_SEPARATE_COUNT).  -> the previous setting."""
    return not _set_default(CALL_SEPARATE_COUNT, not on)


def force_ballot_rank(on):
    """Default of calls that do not say: the per-tile sort's ballot-based ranking fallback (test hook).  -> the previous setting."""
    return _set_default(CALL_BALLOT_RANK, bool(on))


def forward_fuses_count(P, W, H, debug=0):
    """Whether a forward of this size with a placement buffer and these flags folds the count pass into the preprocess launch."""
    return bool(_lib.load().egs_forward_fuses_count(int(P), int(W), int(H), call_flags(debug)))


def last_instance_count(device=None, P=None):
    """R (rectangle instances) of the most recent EAGER forward on `device`, summed from the page-locked per-workgroup counts
    that call copied out.  Captured forwards skip the copy; they are checked through set_running_max / stats["total_view"]."""
    key = torch.cuda.current_device() if device is None else torch.device(device).index
    pinned = _pinned_counts.get(key)
    if pinned is None:
        return 0
    P = stats.get("P", 0) if P is None else P
    return int(_lib.load().egs_sum_counts(int(P), C.c_void_p(pinned.data_ptr())))


_placement = {}


_NO_PLACEMENT = bool(os.environ.get("EGS_NO_PLACEMENT"))      # switch for A/B measurements (read once)
_sizes = {}                      # (P, W, H, capacity) -> (geometry bytes, image bytes, binning bytes, offset of the device-side instance count)


def _buffer_sizes(L, P, W, H, cap):
    """Sizes of the three opaque buffers for this problem size, asked of the library once per (P, W, H, capacity)."""
    key = (P, W, H, cap)
    ent = _sizes.get(key)
    if ent is None:
        if len(_sizes) > 256:
            _sizes.clear()
        lay = _lib.BinningLayout()
        L.egs_get_binning_layout(P, cap, W, H, C.byref(lay))
        ent = _sizes[key] = (int(L.egs_geom_bytes(P)), int(L.egs_image_bytes(W, H)), int(L.egs_binning_bytes(P, cap, W, H)), int(lay.total))
    return ent


def _carve(dev, shapes):
    """One float32 allocation, one contiguous view per shape (None -> None), every view starting on a 256-byte boundary: an
    allocator call costs ~1.5 us of host time, the backward used to make eight."""
    offs, tot = [], 0
    for sh in shapes:
        offs.append(tot)
        if sh is not None:
            n = 1
            for d in sh:
                n *= d
            tot += (n + 63) & ~63
    buf = torch.empty((max(tot, 1),), device=dev, dtype=torch.float32)
    out = []
    for sh, o in zip(shapes, offs):
        if sh is None:
            out.append(None); continue
        n = 1
        for d in sh:
            n *= d
        out.append(buf[o:o + n].view(sh))
    return out


def placement_buffer(dev, W, H):
    """The persistent placement buffer (include/egs_raster.h) of forwards at W x H on the current stream of `dev`: what the previous
    such forward's blend spent per quadrant, by which the next one places its tiles.  Zero-filled when created."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, int(W), int(H), _hip._raw_stream(idx))
    t = _placement.get(key)
    if t is None:
        # (never dropped: a captured hipGraph keeps the address; 40 KB per resolution and stream at 960x540)
        L = _lib.load()
        t = _placement[key] = torch.zeros((L.egs_placement_bytes(int(W), int(H)),), device=dev, dtype=torch.uint8)
        # the owner's one-time initialisation (include/egs_raster.h: the sums region must be zero when a forward starts; the library keeps no
        # record of buffers).  torch.zeros already cleared it; under hipGraph capture only that fill exists (and is replayed: harmless)
        if not torch.cuda.is_current_stream_capturing():
            _lib.check(L.egs_placement_init(t.data_ptr(), int(W), int(H), _hip.stream_of(dev)))
    return t


def set_capacity_hint(instances, device=None):
    key = torch.cuda.current_device() if device is None else torch.device(device).index
    _capacity_hint[key] = int(instances)


def _ptr(t):
    """Device address as a plain int (ctypes converts it for a c_void_p parameter; building the c_void_p object here cost ~7 us per
    eager step over fifty arguments), None for an absent / empty tensor."""
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _object_rotation_struct(object_rotation, dev, P):
    """(egs_object_rotation struct or None, tensors to keep alive) from (M [3,3] or [9], selected uint8[P] / bool[P] or None, row-0 gradient
    multiplier: float or float32[1] device tensor) -- fused.object_selection's result plugs in as (M,) + selection."""
    if object_rotation is None:
        return None, ()
    M, sel, mult = object_rotation
    M = _f32c(M.detach(), "object rotation").reshape(9)
    keep = [M]
    st = _lib.ObjectRotation()
    st.M9 = M.data_ptr()
    if sel is not None:
        sel = sel.to(torch.uint8).contiguous()
        if sel.numel() != P or sel.device != dev:
            raise RuntimeError("object_rotation: `selected` must hold one byte per Gaussian on the rasterizer's device")
        st.selected = sel.data_ptr(); keep.append(sel)
    if torch.is_tensor(mult):
        mult = mult.detach().float().reshape(1).contiguous()
        st.row0_grad_mult, st.row0_grad_mult_dev = 1.0, mult.data_ptr(); keep.append(mult)
    else:
        st.row0_grad_mult = float(mult)
    return st, tuple(keep)


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the MI355X rasterizer runs on HIP devices only (no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_stream = _hip.stream_of           # (device) -> c_void_p of its current stream


def _opt(t):
    """zero-length tensor -> None"""
    return None if t is None or t.numel() == 0 else t


ACT_LOG_SCALES, ACT_RAW_QUATS, ACT_LOGIT_OPACITY = 1, 2, 4      # include/egs_raster.h: EGS_ACT_*
ACT_RAW_PARAMETERS = ACT_LOG_SCALES | ACT_RAW_QUATS | ACT_LOGIT_OPACITY
# include/egs_raster.h: EGS_GRAD_* (which inputs' gradients the caller reads; 0 = all)
GRAD_MEANS3D, GRAD_MEANS2D, GRAD_SH, GRAD_COLORS, GRAD_OPACITY, GRAD_SCALES, GRAD_ROTATIONS, GRAD_COV3D = 1, 2, 4, 8, 16, 32, 64, 128
COLORS_ONLY_BACKWARD = True        # measurement switch (bench.py label_phase_shape): False withholds the mask, the full backward runs


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, activation_flags=0, sh_rest=None, active_count=None, guard=None, object_rotation=None, color_only=False):
    """-> (num_rendered, color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer)
    color_only (extension, ABI 4): depth and alpha are not produced (None) -- the blend skips their sums and planes.
    active_count (extension): int32[1] device tensor, the number of live rows of a capacity-sized model (include/egs_raster.h);
    guard (extension): a StepGuard whose words a captured forward writes.
    object_rotation (extension): (M, selected, row-0 gradient multiplier) -- the `fine_all` call shape's rotated covariance built
    inside the rasterizer from scales + rotations (include/egs_raster.h egs_object_rotation); pass the same to the backward.
    sh_rest (extension): `sh` is then the DC block [P,1,3] and `sh_rest` the other coefficients [P,M-1,3] -- the two parameters the
    reference's GaussianModel stores, without the torch.cat of get_features (include/egs_raster.h: split spherical harmonics)."""
    L = _lib.load()
    flags = call_flags(debug)
    means3D = _f32c(means3D, "means3D")
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    background, opacity = _f32c(background, "background"), _f32c(opacity, "opacity")
    viewmatrix, projmatrix, campos = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
    colors, scales, rotations = _opt(_f32c(colors, "colors")), _opt(_f32c(scales, "scales")), _opt(_f32c(rotations, "rotations"))
    cov3D_precomp, sh = _opt(_f32c(cov3D_precomp, "cov3D_precomp")), _opt(_f32c(sh, "sh"))
    sh_rest = _opt(_f32c(sh_rest, "sh_rest"))
    if active_count is not None and not (active_count.is_cuda and active_count.dtype == torch.int32 and active_count.numel() == 1):
        raise RuntimeError("active_count: an int32[1] tensor on the rasterizer's device")
    M = 0 if sh is None else sh.shape[1]
    if sh_rest is not None:
        if sh is None or sh.shape[1] != 1 or sh_rest.shape[0] != P:
            raise RuntimeError("sh_rest goes with sh = the DC block [P, 1, 3]")
        M = 1 + sh_rest.shape[1]
    with _hip.device_ctx(dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        cap = _capacity_hint.get(key, 0)
        gb, ib, bb, total_off = _buffer_sizes(L, P, W, H, cap)
        planes = torch.empty((3 if color_only else 5, H, W), device=dev, dtype=torch.float32)       # colour, depth, alpha: one allocation, three contiguous views
        out_color, out_depth, out_alpha = planes[0:3], (None if color_only else planes[3:4]), (None if color_only else planes[4:5])
        radii = torch.empty((P,), device=dev, dtype=torch.int32)
        state = torch.empty((gb + ib + bb,), device=dev, dtype=torch.uint8)    # the three opaque buffers (their sizes are multiples of 256 bytes)
        geom, img, binning = state[:gb], state[gb:gb + ib], state[gb + ib:]
        R = C.c_int64(0)
        # The instance count R is data dependent.  Instead of stopping the GPU while the host reads it, the whole chain
        # is enqueued against a capacity guess (1.25 x the largest R seen on this device); only a too-small guess costs
        # a second binning + blend launch.
        nb = (P + 255) // 256
        pinned = _pinned_counts.get(key)
        if pinned is None or pinned.numel() < nb:
            pinned = _pinned_counts[key] = torch.empty((max(nb, 4096),), dtype=torch.int32, pin_memory=True)
        capturing = torch.cuda.is_current_stream_capturing()
        place = None if _NO_PLACEMENT else placement_buffer(dev, W, H)
        rot_st, _rot_keep = _object_rotation_struct(object_rotation, dev, P)
        rot_arg = C.byref(rot_st) if rot_st is not None else None
        deferred = guard is not None and getattr(guard, "deferred", False) and not capturing and P != 0
        if deferred:
            _settle(guard)                                       # the previous frame of this guard: did it fit?
            new_cap = _capacity_hint.get(key, 0)
            if new_cap != cap:                                   # it did not (or nothing is known yet): lay the buffers out again
                cap = new_cap
                gb, ib, bb, total_off = _buffer_sizes(L, P, W, H, cap)
                state = torch.empty((gb + ib + bb,), device=dev, dtype=torch.uint8)
                geom, img, binning = state[:gb], state[gb:gb + ib], state[gb + ib:]
            deferred = cap > 0                                   # no capacity known yet: this call establishes it the waiting way
        if deferred:
            i = guard.frames % _DEFER_RING
            pin = guard._pinned[i]
            if pin is None or pin.numel() < nb + 2:
                pin = guard._pinned[i] = torch.empty((max(nb + 2, 4096),), dtype=torch.int32, pin_memory=True)
            pin[nb] = -1                                         # sentinel: overwritten (0 / 1) by the copy that ends the chain
            _lib.check(L.egs_forward_enqueue(
                P, int(degree), M, _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors), _ptr(opacity), _ptr(scales), float(scale_modifier),
                _ptr(rotations), _ptr(cov3D_precomp), int(activation_flags), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), _ptr(background), W, H,
                float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), _ptr(radii), _ptr(geom), cap, _ptr(binning), _ptr(img),
                _ptr(out_color), _ptr(out_depth), _ptr(out_alpha), C.c_void_p(pin.data_ptr()), _ptr(guard.running_max),
                _ptr(active_count), _ptr(guard.overflow), _ptr(place), rot_arg, _stream(dev), flags))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))             # behind the frame's last copy (egs_forward_enqueue queued it on this stream)
            guard._pending.append((pin, P, cap, key, ev))
            guard.frames += 1
            R = C.c_int64(cap)                      # layout size, as under capture; the frame's own count is read at the next call
            rc = 0
        elif capturing:
            # hipGraph capture of a whole training step (egogaussian_amd/graph.py): nothing may wait on the host, so the
            # chain is enqueued against the capacity established by earlier eager calls and R is checked after replays.
            if cap <= 0 or P == 0:
                raise RuntimeError("rasterize_gaussians under graph capture needs a capacity from an earlier eager call")
            _lib.check(L.egs_forward_enqueue(
                P, int(degree), M, _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors), _ptr(opacity), _ptr(scales), float(scale_modifier),
                _ptr(rotations), _ptr(cov3D_precomp), int(activation_flags), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), _ptr(background), W, H,
                float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), _ptr(radii), _ptr(geom), cap, _ptr(binning), _ptr(img),
                _ptr(out_color), _ptr(out_depth), _ptr(out_alpha), None, _ptr(None if guard is None else guard.running_max),
                _ptr(active_count), _ptr(None if guard is None else guard.overflow), _ptr(place), rot_arg, _stream(dev), flags))
            R = C.c_int64(cap)                      # layout size; the true count is stats["total_view"] after a sync
            rc = 0
        else:
            rc = L.egs_forward(P, int(degree), M, _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors), _ptr(opacity), _ptr(scales),
                               float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), int(activation_flags), _ptr(viewmatrix), _ptr(projmatrix),
                               _ptr(campos), _ptr(background), W, H, float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                               _ptr(radii), _ptr(geom), cap, _ptr(binning), _ptr(img), _ptr(out_color), _ptr(out_depth),
                               _ptr(out_alpha), C.c_void_p(pinned.data_ptr()), C.byref(R), _ptr(active_count), _ptr(place), rot_arg, _stream(dev),
                               flags)
        if rc == _lib.RETRY_LARGER:
            cap = int(R.value * 1.25) + 65536
            total_off = _buffer_sizes(L, P, W, H, cap)[3]
            binning = torch.empty((_buffer_sizes(L, P, W, H, cap)[2],), device=dev, dtype=torch.uint8)
            rc = L.egs_forward_render(P, cap, _ptr(background), W, H, _ptr(geom), _ptr(binning), _ptr(img), _ptr(out_color),
                                      _ptr(out_depth), _ptr(out_alpha), _stream(dev), flags)
            stats["retries"] += 1
        _lib.check(rc)
        if P and not capturing and not deferred:
            _capacity_hint[key] = max(cap, 0) if R.value else _capacity_hint.get(key, 0)
        stats["capacity"] = cap
    stats["num_rendered"] = int(guard.last_R) if deferred else int(R.value)
    stats["P"] = P
    stats["image_buffer"] = img
    if cap > 0:                                     # device-side instance count of this forward (int64[1] view, for graph replays)
        stats["total_view"] = binning[total_off:total_off + 8].view(torch.int64)
    return int(R.value), out_color, out_depth, out_alpha, radii, geom, binning, img


def visible_view(geom, P):
    """radii > 0 as a torch.bool VIEW of the bytes the preprocess kernel wrote into the geometry buffer (no compare kernel).  It
    aliases `geom`, which the backward reads: treat it as read-only."""
    if P == 0:
        return torch.zeros(0, dtype=torch.bool, device=geom.device)
    glay = _lib.GeomLayout()
    _lib.check(_lib.load().egs_get_geom_layout(int(P), C.byref(glay)))
    return geom[glay.visible:glay.visible + P].view(torch.bool)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alpha,
                                 debug, activation_flags=0, sh_rest=None, densify_stats=None, guard=None, sink=None,
                                 prologue_scratch=None, object_rotation=None, grad_mask=0, loss_grad=None):
    """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3],
           dL_dscales[P,3], dL_drotations[P,4]); with sh_rest, dL_dsh is [P,1,3] and a ninth element dL_dsh_rest[P,M-1,3] follows.
    densify_stats (extension): (xyz_gradient_accum[P,1], denom[P,1], max_radii2D[P] or None), float32, updated in place by the kernel
    that produces dL_dmeans2D (include/egs_raster.h) -- the caller then skips its add_densification_stats for this iteration.
    guard (extension): the StepGuard of the forward; the statistics are left untouched when its overflow word is set.
    sink (extension): an optim.AdamSink -- the leaves it owns take their Adam step inside this backward (include/egs_raster.h,
    egs_backward_adam); their gradients are not produced: those positions of the result are None.
    prologue_scratch (extension): the scratch buffer a preceding egs_l1_ssim_backward_ex prepared for THIS backward (tile order,
    cleared accumulator, optimizer bookkeeping: fused.l1_ssim_loss(raster_prologue=True)); the backward then starts at its blend kernel.
    loss_grad (extension, ABI 5): a lib.LossGrad -- dL_dout_color is then NOT read (it only gives the image size): the blend computes the
    image loss's gradient itself from what the loss forward left (include/egs_raster.h egs_backward_lossgrad; fused.l1_ssim_loss(raster_lossgrad=True)).
    grad_mask (extension, ABI 4): GRAD_* bits of the inputs whose gradient the caller reads (autograd's needs_input_grad), 0 = all.
    GRAD_COLORS alone with `colors` given -- the reference's label call, /root/reference/gaussian_renderer/render_helper.py:38-54 --
    takes the colours-only backward: only dL_dcolors is produced, every other position of the result is None."""
    L = _lib.load()
    means3D = _f32c(means3D, "means3D")
    dev = means3D.device
    P = means3D.shape[0]
    H, W = dL_dout_color.shape[-2], dL_dout_color.shape[-1]
    background = _f32c(background, "background")
    viewmatrix, projmatrix, campos = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
    colors, scales, rotations = _opt(_f32c(colors, "colors")), _opt(_f32c(scales, "scales")), _opt(_f32c(rotations, "rotations"))
    cov3D_precomp, sh = _opt(_f32c(cov3D_precomp, "cov3D_precomp")), _opt(_f32c(sh, "sh"))
    g_color = _f32c(dL_dout_color, "dL_dout_color")
    g_depth = _opt(_f32c(dL_dout_depth, "dL_dout_depth"))
    g_alpha = _opt(_f32c(dL_dout_alpha, "dL_dout_alpha"))
    sh_rest = _opt(_f32c(sh_rest, "sh_rest"))
    M = 0 if sh is None else sh.shape[1] + (0 if sh_rest is None else sh_rest.shape[1])
    if COLORS_ONLY_BACKWARD and P != 0 and grad_mask == GRAD_COLORS and colors is not None and sink is None and densify_stats is None and object_rotation is None \
            and loss_grad is None:          # (a loss gradient computed in the blend needs the full path: dL_dout_color is uninitialised then)
        with _hip.device_ctx(dev):
            dcolors = torch.empty((P, 3), device=dev, dtype=torch.float32)
            scratch = prologue_scratch if prologue_scratch is not None else torch.empty((L.egs_backward_scratch_bytes(P),), device=dev, dtype=torch.uint8)
            _lib.check(L.egs_backward_adam(
                P, int(degree), M, int(R), _ptr(background), _ptr(means3D), None, None, _ptr(colors), _ptr(scales),
                float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), int(activation_flags), _ptr(viewmatrix), _ptr(projmatrix),
                _ptr(campos), W, H, float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer),
                _ptr(imageBuffer), _ptr(g_color), None, None, None, _ptr(dcolors), None, None, None, None, None, None, None,
                None, None, None, None, None, 1 if prologue_scratch is not None else 0, None, GRAD_COLORS, _ptr(scratch), _stream(dev),
                (call_flags(debug) & CALL_SYNC)))
        return None, dcolors, None, None, None, None, None, None
    with _hip.device_ctx(dev):
        e = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
        own_cov = cov3D_precomp is None
        owned = set() if sink is None or P == 0 else sink.check(means3D=means3D, scales=scales, rotations=rotations, sh=sh, sh_rest=sh_rest,
                                                                own_cov=own_cov, colors=colors)
        keep = sink is not None and getattr(sink, "keep_grads", False)      # tests: the owned leaves' gradients are written as well
        fused = lambda leaf: leaf in owned and not keep
        no_colors = bool(owned and not keep and sh is not None and sh_rest is None and M == 1)      # dcolors would be scratch nobody reads
        need_m3d_arg = fused(_lib.SINK_MEANS3D) and sh_rest is not None
        # every gradient array of this backward out of ONE allocation (contiguous views on 256-byte boundaries)
        (dmeans2D, dcolors, dopacity, dmeans3D, dmeans3D_tmp, dcov3D, dsh, dsh_rest, dscales, drots) = _carve(dev, [
            (P, 3), None if no_colors else (P, 3), None if fused(_lib.SINK_OPACITY) else (P, 1), None if fused(_lib.SINK_MEANS3D) else (P, 3),
            (P, 3) if need_m3d_arg else None, None if own_cov else (P, 6),
            None if (fused(_lib.SINK_SH) or sh is None) else tuple(sh.shape),
            None if (fused(_lib.SINK_SH_REST) or sh_rest is None) else tuple(sh_rest.shape),
            None if (fused(_lib.SINK_SCALES) or not own_cov) else (P, 3), None if (fused(_lib.SINK_ROTATIONS) or not own_cov) else (P, 4)])
        # (split harmonics: the positions' gradient is finished by the spherical-harmonics launch and travels there through this array)
        dmeans3D_arg = dmeans3D_tmp if need_m3d_arg else dmeans3D
        if own_cov:
            dcov3D = e(0, 6)                                 # not produced when the library built the covariance itself
        if sh is None:
            dsh = e(0, 0, 3)
        if not own_cov:                                      # absent inputs get empty gradients (the autograd Function maps them to None)
            dscales = None if fused(_lib.SINK_SCALES) else e(0, 3)
            drots = None if fused(_lib.SINK_ROTATIONS) else e(0, 4)
        rot_st, _rot_keep = _object_rotation_struct(object_rotation, dev, P)
        if P != 0 and loss_grad is not None:
            if g_depth is not None or g_alpha is not None:
                raise RuntimeError("loss_grad: the loss must depend on the colour output only")
            scratch = prologue_scratch if prologue_scratch is not None else torch.empty((L.egs_backward_scratch_bytes(P),), device=dev, dtype=torch.uint8)
            _lib.check(L.egs_backward_lossgrad(
                P, int(degree), M, int(R), _ptr(background), _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors), _ptr(scales),
                float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), int(activation_flags), _ptr(viewmatrix), _ptr(projmatrix),
                _ptr(campos), W, H, float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer),
                _ptr(imageBuffer), C.byref(loss_grad), _ptr(dmeans2D), _ptr(dcolors),
                _ptr(dopacity), _ptr(dmeans3D_arg), None if own_cov else _ptr(dcov3D), _ptr(dsh), _ptr(dsh_rest), _ptr(dscales) if own_cov else None,
                _ptr(drots) if own_cov else None, *_stat_ptrs(densify_stats, P, dev), _ptr(None if guard is None else guard.overflow),
                C.byref(sink.struct) if owned else None, 1 if prologue_scratch is not None else 0,
                C.byref(rot_st) if rot_st is not None else None, 0, _ptr(scratch), _stream(dev), (call_flags(debug) & CALL_SYNC)))
            if owned:
                sink.mark_stepped()
        elif P != 0 and (owned or prologue_scratch is not None or rot_st is not None):
            scratch = prologue_scratch if prologue_scratch is not None else torch.empty((L.egs_backward_scratch_bytes(P),), device=dev, dtype=torch.uint8)
            _lib.check(L.egs_backward_adam(
                P, int(degree), M, int(R), _ptr(background), _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors), _ptr(scales),
                float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), int(activation_flags), _ptr(viewmatrix), _ptr(projmatrix),
                _ptr(campos), W, H, float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer),
                _ptr(imageBuffer), _ptr(g_color), _ptr(g_depth), _ptr(g_alpha), _ptr(dmeans2D), _ptr(dcolors),
                _ptr(dopacity), _ptr(dmeans3D_arg), None if own_cov else _ptr(dcov3D), _ptr(dsh), _ptr(dsh_rest), _ptr(dscales) if own_cov else None,
                _ptr(drots) if own_cov else None, *_stat_ptrs(densify_stats, P, dev), _ptr(None if guard is None else guard.overflow),
                C.byref(sink.struct) if owned else None, 1 if prologue_scratch is not None else 0,
                C.byref(rot_st) if rot_st is not None else None, 0, _ptr(scratch), _stream(dev), (call_flags(debug) & CALL_SYNC)))
            if owned:
                sink.mark_stepped()
        elif P != 0:
            scratch = torch.empty((L.egs_backward_scratch_bytes(P),), device=dev, dtype=torch.uint8)
            _lib.check(L.egs_backward(
                P, int(degree), M, int(R), _ptr(background), _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors), _ptr(scales),
                float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), int(activation_flags), _ptr(viewmatrix), _ptr(projmatrix),
                _ptr(campos), W, H, float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer),
                _ptr(imageBuffer), _ptr(g_color), _ptr(g_depth), _ptr(g_alpha), _ptr(dmeans2D), _ptr(dcolors),
                _ptr(dopacity), _ptr(dmeans3D), None if own_cov else _ptr(dcov3D), _ptr(dsh), _ptr(dsh_rest), _ptr(dscales) if own_cov else None,
                _ptr(drots) if own_cov else None, *_stat_ptrs(densify_stats, P, dev), _ptr(None if guard is None else guard.overflow),
                0, _ptr(scratch), _stream(dev), (call_flags(debug) & CALL_SYNC)))
    if sh_rest is not None:
        return dmeans2D, dcolors, dopacity, dmeans3D, dcov3D, dsh, dscales, drots, dsh_rest
    return dmeans2D, dcolors, dopacity, dmeans3D, dcov3D, dsh, dscales, drots


def _stat_ptrs(stats_tensors, P, dev):
    if stats_tensors is None:
        return None, None, None
    acc, den, mr = stats_tensors
    for t in (acc, den) + ((mr,) if mr is not None else ()):
        if not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == P):
            raise RuntimeError("densify_stats: contiguous float32 tensors of P elements on the rasterizer's device")
    return _ptr(acc), _ptr(den), _ptr(mr)


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool[P]: Gaussians in front of the near plane (view.z > 0.2)."""
    L = _lib.load()
    means3D = _f32c(means3D, "means3D")
    P = means3D.shape[0]
    present = torch.zeros((P,), device=means3D.device, dtype=torch.uint8)
    if P:
        with _hip.device_ctx(means3D.device):
            _lib.check(L.egs_mark_visible(P, _ptr(means3D), _ptr(_f32c(viewmatrix, "viewmatrix")),
                                          _ptr(_f32c(projmatrix, "projmatrix")), _ptr(present), _stream(means3D.device)))
    return present.bool()


# spellings of the C++ symbols behind the pybind names (BASELINE.json refers to `_C.RasterizeGaussians`)
RasterizeGaussians = RasterizeGaussiansCUDA = rasterize_gaussians
RasterizeGaussiansBackward = RasterizeGaussiansBackwardCUDA = rasterize_gaussians_backward
markVisible = mark_visible


# ---- debugging / test helpers: typed views into the opaque byte buffers ------------------------------
def geom_views(geom, P):
    lay = _lib.GeomLayout()
    _lib.check(_lib.load().egs_get_geom_layout(P, C.byref(lay)))
    v = lambda off, n, dt: geom[off:off + n].view(dt)
    return dict(rec=v(lay.rec, P * 48, torch.float32).view(P, 12), rect=v(lay.rect, P * 8, torch.int32).view(P, 2),
                offsets=v(lay.offsets, P * 4, torch.int32), clamped=geom[lay.clamped:lay.clamped + P],
                visible=geom[lay.visible:lay.visible + P].view(torch.bool))


def binning_views(binning, P, R, W, H, capacity=None):
    """Typed views; `capacity` = the size the buffer was laid out for (stats["capacity"] right after a forward)."""
    lay = _lib.BinningLayout()
    _lib.check(_lib.load().egs_get_binning_layout(P, R if capacity is None else capacity, W, H, C.byref(lay)))
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    return dict(point_list=binning[lay.point_list:lay.point_list + R * 4].view(torch.int32),
                pairs=binning[lay.pairs:lay.pairs + R * 8].view(torch.int64),
                table=binning[lay.table:lay.table + nt * lay.table_stride * 4].view(torch.int32).view(nt, lay.table_stride)[:, :lay.bin_blocks] if R else None,
                key_bits=lay.key_bits, index_passes=lay.index_passes, bin_blocks=lay.bin_blocks)


def image_views(img, W, H):
    lay = _lib.ImageLayout()
    _lib.check(_lib.load().egs_get_image_layout(W, H, C.byref(lay)))
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    return dict(ranges=img[lay.ranges:lay.ranges + nt * 8].view(torch.int32).view(nt, 2),
                final_T=img[lay.final_T:lay.final_T + H * W * 4].view(torch.float32).view(H, W),
                n_contrib=img[lay.n_contrib:lay.n_contrib + H * W * 4].view(torch.int32).view(H, W),
                quad_work=img[lay.quad_work:lay.quad_work + nt * 16].view(torch.int32).view(nt, 4),
                quad_pairs=img[lay.quad_pairs:lay.quad_pairs + nt * 16].view(torch.int32).view(nt, 4),
                quad_visits=img[lay.quad_pairs + nt * 16:lay.quad_pairs + nt * 32].view(torch.int32).view(nt, 4))
