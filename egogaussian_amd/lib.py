"""ctypes binding of libegs_raster.so (C ABI: include/egs_raster.h).

There is no fallback: if the shared library is missing or cannot be loaded this module raises, and so
does every op built on it.  Build it with `python __graft_entry__.py` or `make -C egogaussian_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EGS_RASTER_LIB", os.path.join(_HERE, "libegs_raster.so"))      # override for A/B builds
ABI_VERSION = 6
RETRY_LARGER = -100
# per-call flags (include/egs_raster.h EGS_CALL_*, ABI 6): bits of the `debug` / `flags` word of the forwards
CALL_SYNC, CALL_KEEP_ALL_INSTANCES, CALL_SEPARATE_COUNT, CALL_SEPARATE_SORT, CALL_BALLOT_RANK = 1, 2, 4, 8, 16

vp, f32, i32, i64 = C.c_void_p, C.c_float, C.c_int, C.c_int64


class GeomLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("rec", "rect", "offsets", "clamped", "visible", "scan_scratch", "total")]


class BinningLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("point_list", "pairs", "scratch", "table", "spine", "total")] + \
               [(n, C.c_int) for n in ("bin_blocks", "key_bits", "index_passes", "table_stride")]


class ImageLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("ranges", "final_T", "n_contrib", "quad_work", "tile_order", "quad_pairs")]


class AdamLeaf(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("param", "exp_avg", "exp_avg_sq", "lr", "step")]


class AdamSink(C.Structure):                     # egs_adam_sink
    _fields_ = [("leaf", AdamLeaf * 6), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("coef", C.c_void_p),
                ("active_rows", C.c_void_p)]


class ObjectRotation(C.Structure):               # egs_object_rotation
    _fields_ = [("M9", C.c_void_p), ("selected", C.c_void_p), ("row0_grad_mult", C.c_float), ("row0_grad_mult_dev", C.c_void_p)]


class BackwardPrologue(C.Structure):             # egs_backward_prologue
    _fields_ = [("P", C.c_int), ("width", C.c_int), ("height", C.c_int), ("image_buffer", C.c_void_p), ("scratch", C.c_void_p),
                ("sink", C.POINTER(AdamSink)), ("skip_flag", C.c_void_p), ("geom_buffer", C.c_void_p)]


class LossGrad(C.Structure):                     # egs_loss_grad
    _fields_ = [("image", C.c_void_p), ("gt", C.c_void_p), ("dm_dmu1", C.c_void_p), ("dm_dexx", C.c_void_p), ("dm_dexy", C.c_void_p),
                ("gate", C.c_void_p), ("upstream_grad", C.c_void_p), ("lambda_dssim", C.c_float), ("deferred_partial_sums", C.c_void_p),
                ("deferred_loss", C.c_void_p), ("loss_running_sum", C.c_void_p)]


SINK_MEANS3D, SINK_OPACITY, SINK_SCALES, SINK_ROTATIONS, SINK_SH, SINK_SH_REST = range(6)      # EGS_SINK_*

# name -> (restype, argtypes); every symbol include/egs_raster.h declares
SIGNATURES = {
    "egs_abi_version": (C.c_int, []),
    "egs_source_hash": (C.c_char_p, []),
    "egs_error_string": (C.c_char_p, [C.c_int]),
    "egs_device_info": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "egs_geom_bytes": (C.c_size_t, [i32]),
    "egs_binning_bytes": (C.c_size_t, [i32, i64, i32, i32]),
    "egs_image_bytes": (C.c_size_t, [i32, i32]),
    "egs_backward_scratch_bytes": (C.c_size_t, [i32]),
    "egs_order_words": (C.c_int, [i32, i32]),
    "egs_get_geom_layout": (C.c_int, [i32, C.POINTER(GeomLayout)]),
    "egs_get_binning_layout": (C.c_int, [i32, i64, i32, i32, C.POINTER(BinningLayout)]),
    "egs_get_image_layout": (C.c_int, [i32, i32, C.POINTER(ImageLayout)]),
    "egs_forward_geometry": (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp, i32, i32, f32, f32,
                                       i32, vp, vp, C.POINTER(i64), vp, C.POINTER(ObjectRotation), vp, i32]),
    "egs_forward": (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp, vp, i32, i32, f32, f32, i32, vp, vp, i64,
                               vp, vp, vp, vp, vp, vp, C.POINTER(i64), vp, vp, C.POINTER(ObjectRotation), vp, i32]),
    "egs_forward_enqueue": (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp, vp, i32, i32, f32, f32, i32, vp, vp,
                                       i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(ObjectRotation), vp, i32]),
    "egs_placement_bytes": (C.c_size_t, [i32, i32]),
    "egs_placement_init": (C.c_int, [vp, i32, i32, vp]),
    "egs_sum_counts": (C.c_int64, [i32, vp]),
    "egs_forward_render": (C.c_int, [i32, i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32]),
    "egs_backward": (C.c_int, [i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp, i32, i32, f32, f32,
                               vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32]),
    "egs_backward_adam": (C.c_int, [i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp, i32, i32, f32, f32,
                                    vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(AdamSink), i32, C.POINTER(ObjectRotation), i32, vp, vp, i32]),
    "egs_mark_visible": (C.c_int, [i32, vp, vp, vp, vp, vp]),
    "egs_cov3d_forward": (C.c_int, [i32, vp, i32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "egs_cov3d_dm_scratch_floats": (C.c_size_t, [i32]),
    "egs_cov3d_backward": (C.c_int, [i32, vp, i32, f32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "egs_l1_ssim_partial_count": (C.c_size_t, [i32, i32, i32]),
    "egs_l1_ssim_forward": (C.c_int, [i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]),
    "egs_l1_ssim_backward": (C.c_int, [i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "egs_l1_ssim_pair_forward": (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),

    "egs_l1_ssim_pair_backward": (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(BackwardPrologue), vp]),
    "egs_l1_ssim_backward_ex": (C.c_int, [i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(BackwardPrologue), vp]),
    "egs_l1_ssim_forward_ex": (C.c_int, [i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, C.POINTER(BackwardPrologue), vp]),
    "egs_backward_lossgrad": (C.c_int, [i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp, i32, i32, f32, f32,
                                        vp, vp, vp, vp, C.POINTER(LossGrad), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                        C.POINTER(AdamSink), i32, C.POINTER(ObjectRotation), i32, vp, vp, i32]),
    "egs_adam_step": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, vp]),
    "egs_adam_workgroups": (C.c_int64, [i64]),
    "egs_adam_step_capturable": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, vp, vp, vp, vp]),
    "egs_densify_stats": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, vp]),
    "egs_densify_plan_scratch_bytes": (C.c_size_t, [i32]),
    "egs_densify_plan": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp,
                                    vp, vp, vp]),
    "egs_prune_plan": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, vp]),
    "egs_gather_rows_f32": (C.c_int, [i64, i32, vp, vp, i32, vp, vp, vp]),
    "egs_gather_i32": (C.c_int, [i64, vp, vp, i32, i32, vp, vp, vp]),
    "egs_split_children": (C.c_int, [i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]),
    "egs_knn3_mean_dist2": (C.c_int, [i32, vp, vp, vp]),
    "egs_knn3_grid_scratch_bytes": (C.c_size_t, [i32]),
    "egs_knn3_grid": (C.c_int, [i32, vp, vp, vp, vp]),
    "egs_forward_fuses_count": (C.c_int, [i32, i32, i32, i32]),
    "egs_profile_begin": (C.c_int, [i32]),
    "egs_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "egs_profile_stage_name": (C.c_char_p, [i32]),
}
N_STAGES = 8


def profile_begin(max_records=65536):
    check(load().egs_profile_begin(max_records))


def profile_end():
    """-> {stage: (total_ms, launches)}"""
    ms = (C.c_double * N_STAGES)()
    n = (C.c_int * N_STAGES)()
    L = load()
    check(L.egs_profile_end(ms, n))
    return {L.egs_profile_stage_name(k).decode(): (float(ms[k]), int(n[k])) for k in range(N_STAGES)}

_lib = None


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the library's sources -- csrc/*.hip, csrc/*.h, include/egs_raster.h, the Makefile -- in
    name order.  Counter files under profiles/ carry the hash they were collected at; bench.py reports their numbers only
    while it still equals this one (a profile of other kernels describes other kernels).  The library embeds the same hash at build
    time: built_source_hash()."""
    from .srchash import source_hash
    return source_hash()


def built_source_hash():
    """The source hash `make` embedded into the loaded library (egs_source_hash()): equal to kernel_source_hash() unless the build is stale."""
    return load().egs_source_hash().decode()


def library_path():
    """Path of the shared library this process loads (EGS_RASTER_LIB overrides the in-tree build)."""
    return LIB_PATH


def load():
    """Load (once) and return the ctypes handle.  Raises RuntimeError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X rasterizer has no CPU or PyTorch fallback. "
            "Build it first (python -c 'import __graft_entry__ as g; g.build()' or make -C egogaussian_amd/csrc).")
    # PyTorch-ROCm bundles its own libamdhip64.so while this library links the system one (libamdhip64.so.7): two copies of the HIP
    # runtime live in the process.  That works as long as TORCH's copy is the one that opened the device first; the other way round --
    # this library loaded (its kernels registered) before torch's first HIP call -- every launch of ours then fails with
    # hipErrorNoDevice (found by running build() and smoke() in one process).  So: bring torch's runtime up first when a device exists.
    # ... and, since this creates the HIP context, first export what a later multi-process RCCL group needs from the environment while
    # it can still be set (dist.REQUIRED_ENV; dmabuf-only hosts fail with hipIpcGetMemHandle otherwise) -- only if the caller left it unset.
    for k, v in (("HSA_ENABLE_IPC_MODE_LEGACY", "0"),):
        os.environ.setdefault(k, v)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:                  # (a C-ABI user without torch: nothing to order)
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.egs_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libegs_raster.so ABI {lib.egs_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise RuntimeError(f"libegs_raster: {load().egs_error_string(code).decode()} (code {code})")
