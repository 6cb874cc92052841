"""CPU: the C-ABI library loads, exports every symbol include/egs_raster.h declares, sizes/layouts are sane and
argument errors are reported before any device work.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "egs_raster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(egs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from egogaussian_amd import lib
    L = lib.load()
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/egs_raster.h but not exported"
    assert set(lib.SIGNATURES) == set(names), "python binding table and header disagree"
    assert L.egs_abi_version() == lib.ABI_VERSION


def test_sizes_and_layouts():
    from egogaussian_amd import lib
    L = lib.load()
    assert L.egs_geom_bytes(0) >= 0 and L.egs_geom_bytes(1000) < L.egs_geom_bytes(2000)
    assert L.egs_geom_bytes(500000) >= 500000 * (48 + 8 + 4 + 1)
    assert L.egs_binning_bytes(500000, 10**6, 960, 540) >= 10**6 * 20
    assert L.egs_image_bytes(960, 540) >= 960 * 540 * 8 + 60 * 34 * 8
    assert L.egs_backward_scratch_bytes(1000) >= 48000
    g = lib.GeomLayout(); assert L.egs_get_geom_layout(1000, C.byref(g)) == 0
    offs = [g.rec, g.rect, g.offsets, g.clamped, g.scan_scratch, g.total]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs) and offs[-1] + 8 <= L.egs_geom_bytes(1000)
    b = lib.BinningLayout(); assert L.egs_get_binning_layout(500000, 5000, 960, 540, C.byref(b)) == 0
    assert b.key_bits == 32 + 11 and b.index_passes == 3 and b.bin_blocks == 489     # 60 x 34 = 2040 tiles -> 11 bits; 19 index bits
    assert b.point_list == 0 < b.pairs < b.scratch < b.table < b.spine       # the backward only needs point_list (offset 0)
    assert L.egs_get_binning_layout(1000000, 5000, 1920, 1080, C.byref(b)) == 0 and b.key_bits == 45 and b.index_passes == 3
    assert L.egs_get_binning_layout(200, 5000, 64, 64, C.byref(b)) == 0 and b.key_bits == 37 and b.index_passes == 1 and b.bin_blocks == 1     # workgroups of whole 256-Gaussian blocks, sized by P (binning.hip egs_bin_gpb)
    i = lib.ImageLayout(); assert L.egs_get_image_layout(100, 70, C.byref(i)) == 0 and i.ranges < i.final_T < i.n_contrib
    assert i.n_contrib < i.quad_work < i.tile_order < i.quad_pairs and i.quad_pairs + 7 * 5 * 8 * 4 <= L.egs_image_bytes(100, 70)
    assert L.egs_abi_version() == 6 and L.egs_knn3_grid_scratch_bytes(0) == 0 and L.egs_knn3_grid_scratch_bytes(100000) > 100000 * 20
    assert L.egs_knn3_grid(5, None, None, None, None) == -1 and L.egs_knn3_grid(0, None, None, None, None) == 0
    # ABI 5: the placement buffer = 4 cost words per tile + the tile-order words + (256-byte aligned) eight sums words per tile;
    # which forwards fold the count pass of the bucketing into the preprocess launch (one round of <= 16 groups per workgroup)
    nt = 60 * 34
    assert L.egs_placement_bytes(960, 540) == ((nt * 4 + L.egs_order_words(960, 540)) * 4 + 255) // 256 * 256 + 8 * nt * 4
    assert L.egs_placement_bytes(0, 540) == 0 and L.egs_placement_init(None, 960, 540, None) == -1
    fuses = lambda P, W, H, flags=0: L.egs_forward_fuses_count(P, W, H, flags)
    assert fuses(500000, 960, 540) == 1 and fuses(100000, 960, 540) == 1 and fuses(253202, 960, 540) == 1 and fuses(1, 64, 64) == 1
    assert fuses(1000000, 1920, 1080) == 0        # four rounds of eight groups per workgroup: the separate count launch
    assert fuses(0, 960, 540) == 0
    # ABI 6: a per-call flag, not a process-wide switch -- the answer for one call's flags says nothing about the next call's
    assert fuses(500000, 960, 540, lib.CALL_SEPARATE_COUNT) == 0 and fuses(500000, 960, 540) == 1
    assert fuses(500000, 960, 540, lib.CALL_KEEP_ALL_INSTANCES | lib.CALL_SYNC | lib.CALL_SEPARATE_SORT | lib.CALL_BALLOT_RANK) == 1
    # ... and the library exports no setter of process-wide behaviour any more
    for gone in ("egs_debug_set_tile_culling", "egs_debug_set_fused_count", "egs_debug_set_sort_in_blend", "egs_debug_force_ballot_rank", "egs_debug_set_lossgrad"):
        assert not hasattr(L, gone), gone


def test_argument_errors_precede_device_work():
    from egogaussian_amd import lib
    L = lib.load()
    R = C.c_int64(-7)
    none = None
    # null required pointers
    rc = L.egs_forward_geometry(10, 0, 1, none, none, none, none, none, none, 1.0, none, none, 0, none, none, none, 64, 64, 1.0, 1.0, 0,
                                none, none, C.byref(R), none, none, none, 0)
    assert rc == -1 and R.value == 0
    assert L.egs_forward_geometry(-1, 0, 1, none, none, none, none, none, none, 1.0, none, none, 0, none, none, none, 64, 64, 1.0, 1.0, 0,
                                  none, none, C.byref(R), none, none, none, 0) == -1
    assert L.egs_forward_geometry(10, 0, 1, none, none, none, none, none, none, 1.0, none, none, 0, none, none, none, 70000, 64, 1.0, 1.0,
                                  0, none, none, C.byref(R), none, none, none, 0) == -3
    assert L.egs_forward_geometry(10, 0, 1, none, none, none, none, none, none, 1.0, none, none, 0, none, none, none, 8192, 8192, 1.0, 1.0,
                                  0, none, none, C.byref(R), none, none, none, 0) == -3      # 262144 tiles > 36864 (one LDS counter per tile)
    # P == 0 is a valid no-op
    assert L.egs_forward_geometry(0, 0, 0, none, none, none, none, none, none, 1.0, none, none, 0, none, none, none, 64, 64, 1.0, 1.0, 0,
                                  none, none, C.byref(R), none, none, none, 0) == 0 and R.value == 0
    # mode errors: both shs and colours given (fake non-null pointers are never dereferenced before the check)
    p = C.c_void_p(4096)
    assert L.egs_forward_geometry(10, 0, 1, p, p, none, p, p, none, 1.0, none, p, 0, p, p, p, 64, 64, 1.0, 1.0, 0, p, p, C.byref(R),
                                  none, none, none, 0) == -2
    assert L.egs_forward_geometry(10, 0, 1, p, p, none, none, p, p, 1.0, none, none, 0, p, p, p, 64, 64, 1.0, 1.0, 0, p, p, C.byref(R),
                                  none, none, none, 0) == -2      # scales without rotations
    assert L.egs_forward_geometry(10, 4, 25, p, p, none, none, p, none, 1.0, none, p, 0, p, p, p, 64, 64, 1.0, 1.0, 0, p, p, C.byref(R),
                                  none, none, none, 0) == -3      # SH degree 4 unsupported
    assert L.egs_forward_geometry(10, 0, 1, p, p, none, none, p, none, 1.0, none, p, 1, p, p, p, 64, 64, 1.0, 1.0, 0, p, p, C.byref(R),
                                  none, none, none, 0) == -2      # log-scale activation asked for, but the covariance is given
    assert L.egs_forward_geometry(10, 0, 1, p, p, none, none, p, p, 1.0, p, none, 8, p, p, p, 64, 64, 1.0, 1.0, 0, p, p, C.byref(R),
                                  none, none, none, 0) == -2      # unknown activation flag
    assert L.egs_forward_geometry(10, 0, 1, p, p, p, none, p, none, 1.0, none, p, 0, p, p, p, 64, 64, 1.0, 1.0, 0, p, p, C.byref(R),
                                  none, none, none, 0) == -2      # split spherical harmonics need at least two coefficients
    assert b"exactly one" in L.egs_error_string(-2) and b"capacity" in L.egs_error_string(lib.RETRY_LARGER)
    # one-call forward: argument errors before any device work
    assert L.egs_forward(10, 0, 1, none, none, none, none, none, none, 1.0, none, none, 0, none, none, none, none, 64, 64, 1.0, 1.0, 0,
                         none, none, 100, none, none, none, none, none, none, C.byref(R), none, none, none, none, 0) == -1
    # opaque buffers must be 256-byte aligned (include/egs_raster.h, Conventions): refused before any device work
    q = C.c_void_p(4096 + 16)
    assert L.egs_forward_geometry(10, 0, 1, p, p, none, none, p, none, 1.0, none, p, 0, p, p, p, 64, 64, 1.0, 1.0, 0, p, q, C.byref(R),
                                  none, none, none, 0) == -1
    assert L.egs_forward_render(10, 100, p, 64, 64, p, q, p, p, p, p, none, 0) == -1
    assert L.egs_forward_render(10, 100, p, 64, 64, p, p, q, p, p, p, none, 0) == -1
    assert L.egs_placement_bytes(960, 540) >= (2040 * 4 + 2040) * 4 and L.egs_placement_bytes(0, 5) == 0
    assert L.egs_mark_visible(5, none, none, none, none, none) == -1


def test_python_surface_validation_and_no_cpu_fallback():
    import diff_gaussian_rasterization as dgr
    from egogaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    assert dgr.GaussianRasterizer is GaussianRasterizer and hasattr(dgr._C, "rasterize_gaussians") \
        and hasattr(dgr._C, "rasterize_gaussians_backward") and hasattr(dgr._C, "mark_visible")
    rs = GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                       scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                       campos=torch.zeros(3), prefiltered=False, debug=False)
    r = GaussianRasterizer(raster_settings=rs)
    assert GaussianRasterizer(rs).raster_settings is rs                       # positional ctor (render_helper.py:61)
    x, o = torch.zeros(4, 3), torch.ones(4, 1)
    with pytest.raises(Exception, match="exactly one"):
        r(means3D=x, means2D=x, opacities=o, shs=None, colors_precomp=None, cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(Exception, match="exactly one"):
        r(means3D=x, means2D=x, opacities=o, shs=torch.zeros(4, 1, 3), colors_precomp=torch.zeros(4, 3), cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(Exception, match="exactly one"):
        r(means3D=x, means2D=x, opacities=o, shs=torch.zeros(4, 1, 3), scales=torch.ones(4, 3))
    with pytest.raises(Exception, match="exactly one"):
        r(means3D=x, means2D=x, opacities=o, shs=torch.zeros(4, 1, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4),
          cov3D_precomp=torch.zeros(4, 6))
    # CPU tensors: loud failure, never a silent fallback
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=x, means2D=x, opacities=o, shs=torch.zeros(4, 1, 3), cov3D_precomp=torch.zeros(4, 6))


def test_library_is_built_from_the_present_sources():
    """Guards against benchmarking a stale build: the hash of the sources `make` embedded into the library (egs_source_hash) is the
    hash of the sources in the tree -- wherever the tree is (the sources travel with the library), whatever the file dates say."""
    from egogaussian_amd import lib
    assert lib.built_source_hash() == lib.kernel_source_hash(), \
        "libegs_raster.so is stale: run EGS_CLEAN=1 python -c 'import __graft_entry__ as g; g.build()'"


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "egogaussian_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in text
    assert not re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, "diff_gaussian_rasterization", "__init__.py")).read(), flags=re.M)
