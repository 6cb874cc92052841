"""ctypes front-end of the C oracle (oracle/raster_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  Nothing under egogaussian_amd/ may import this module.  PARITY UNPINNED (see the
header of raster_oracle.c): the reference holds neither the rasterizer's source nor tests for it.

The stage functions mirror the pipeline of SURVEY.md section 8a (a-4 .. a-11) so the HIP path can be
compared stage by stage: preprocess -> scan -> duplicate -> sort -> ranges -> render, and
render-backward -> preprocess-backward.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile both oracle variants with the committed Makefile (gcc only)."""
    targets = [os.path.join(_HERE, n) for n in ("liboracle_f32.so", "liboracle_f64.so")]
    src = os.path.join(_HERE, "raster_oracle.c")
    stale = force or any((not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


def _lib(dtype):
    dtype = np.dtype(dtype)
    name = {np.dtype(np.float32): "liboracle_f32.so", np.dtype(np.float64): "liboracle_f64.so"}[dtype]
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        assert lib.egso_real_bytes() == dtype.itemsize
        lib.egso_scan.restype = C.c_int64
        _LIBS[name] = lib
    return _LIBS[name]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


class Oracle:
    """One rasterizer call, stage by stage.  All arrays are numpy; `dtype` selects fp32 / fp64."""

    def __init__(self, dtype=np.float32, nthreads=1):
        self.dt = np.dtype(dtype)
        self.lib = _lib(self.dt)
        self.real = C.c_float if self.dt == np.float32 else C.c_double
        self.nthreads = int(nthreads)

    # ---- forward -------------------------------------------------------------------------
    def forward(self, *, means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height, image_width,
                tanfovx, tanfovy, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                scale_modifier=1.0, sh_degree=0, stop_after=None):
        dt = self.dt
        st = {}
        m = _arr(means3D, dt).reshape(-1, 3)
        P = m.shape[0]
        H, W = int(image_height), int(image_width)
        st.update(P=P, H=H, W=W, tanfovx=float(tanfovx), tanfovy=float(tanfovy), scale_modifier=float(scale_modifier),
                  D=int(sh_degree))
        st["means3D"] = m
        st["opacities"] = _arr(opacities, dt).reshape(-1)
        st["shs"] = None if shs is None else _arr(shs, dt).reshape(P, -1, 3)
        st["M"] = 0 if st["shs"] is None else st["shs"].shape[1]
        st["colors_precomp"] = None if colors_precomp is None else _arr(colors_precomp, dt).reshape(P, 3)
        st["scales"] = None if scales is None else _arr(scales, dt).reshape(P, 3)
        st["rotations"] = None if rotations is None else _arr(rotations, dt).reshape(P, 4)
        st["cov3D_precomp"] = None if cov3D_precomp is None else _arr(cov3D_precomp, dt).reshape(P, 6)
        assert (st["shs"] is None) != (st["colors_precomp"] is None)
        assert (st["cov3D_precomp"] is None) != (st["scales"] is None or st["rotations"] is None)
        st["viewmatrix"] = _arr(viewmatrix, dt).reshape(16)
        st["projmatrix"] = _arr(projmatrix, dt).reshape(16)
        st["campos"] = _arr(campos, dt).reshape(3)
        st["bg"] = _arr(bg, dt).reshape(3)
        gx, gy = (W + 15) // 16, (H + 15) // 16
        st["grid"] = (gx, gy)

        st["radii"] = np.zeros(P, np.int32)
        st["xy"] = np.zeros((P, 2), dt)
        st["depths"] = np.zeros(P, dt)
        st["cov3D"] = np.zeros((P, 6), dt)
        st["rgb"] = np.zeros((P, 3), dt)
        st["conic_opacity"] = np.zeros((P, 4), dt)
        st["tiles_touched"] = np.zeros(P, np.uint32)
        st["clamped"] = np.zeros((P, 3), np.uint8)
        st["rects"] = np.zeros((P, 4), np.int32)
        rc = self.lib.egso_preprocess(
            P, st["D"], st["M"], _p(m), _p(st["scales"]), self.real(scale_modifier), _p(st["rotations"]),
            _p(st["opacities"]), _p(st["shs"]), _p(st["colors_precomp"]), _p(st["cov3D_precomp"]), _p(st["viewmatrix"]),
            _p(st["projmatrix"]), _p(st["campos"]), W, H, self.real(tanfovx), self.real(tanfovy), _p(st["radii"]),
            _p(st["xy"]), _p(st["depths"]), _p(st["cov3D"]), _p(st["rgb"]), _p(st["conic_opacity"]),
            _p(st["tiles_touched"]), _p(st["clamped"]), _p(st["rects"]))
        assert rc == 0
        st["offsets"] = np.zeros(P, np.uint32)
        R = int(self.lib.egso_scan(P, _p(st["tiles_touched"]), _p(st["offsets"]))) if P else 0
        st["R"] = R
        if stop_after == "preprocess":
            return st
        st["keys_unsorted"] = np.zeros(R, np.uint64)
        st["vals_unsorted"] = np.zeros(R, np.uint32)
        self.lib.egso_duplicate(P, _p(st["depths"]), _p(st["offsets"]), _p(st["radii"]), _p(st["rects"]), W,
                                _p(st["keys_unsorted"]), _p(st["vals_unsorted"]))
        st["key_bits"] = int(self.lib.egso_key_bits(gx * gy))
        st["keys"] = np.zeros(R, np.uint64)
        st["point_list"] = np.zeros(R, np.uint32)
        rc = self.lib.egso_sort_pairs(C.c_int64(R), _p(st["keys_unsorted"]), _p(st["vals_unsorted"]), _p(st["keys"]),
                                      _p(st["point_list"]), st["key_bits"])
        assert rc == 0
        st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
        self.lib.egso_tile_ranges(C.c_int64(R), _p(st["keys"]), gx * gy, _p(st["ranges"]))
        if stop_after == "binning":
            return st
        st["color"] = np.zeros((3, H, W), dt)
        st["depth"] = np.zeros((1, H, W), dt)
        st["alpha"] = np.zeros((1, H, W), dt)
        st["final_T"] = np.zeros((H, W), dt)
        st["n_contrib"] = np.zeros((H, W), np.uint32)
        self.lib.egso_render_forward(W, H, _p(st["ranges"]), _p(st["point_list"]), _p(st["xy"]), _p(st["rgb"]),
                                     _p(st["depths"]), _p(st["conic_opacity"]), _p(st["bg"]), _p(st["color"]),
                                     _p(st["depth"]), _p(st["alpha"]), _p(st["final_T"]), _p(st["n_contrib"]),
                                     self.nthreads)
        return st

    # ---- backward ------------------------------------------------------------------------
    def backward(self, st, dL_dcolor, dL_ddepth=None, dL_dalpha=None):
        dt = self.dt
        P, H, W = st["P"], st["H"], st["W"]
        gcol = _arr(dL_dcolor, dt).reshape(3, H, W)
        gdep = np.zeros((H, W), dt) if dL_ddepth is None else _arr(dL_ddepth, dt).reshape(H, W)
        galp = np.zeros((H, W), dt) if dL_dalpha is None else _arr(dL_dalpha, dt).reshape(H, W)
        g = {}
        g["dL_dmean2D"] = np.zeros((P, 3), dt)
        g["dL_dconic"] = np.zeros((P, 4), dt)
        g["dL_dopacity"] = np.zeros((P, 1), dt)
        g["dL_dcolor"] = np.zeros((P, 3), dt)
        g["dL_ddepth"] = np.zeros(P, dt)
        self.lib.egso_render_backward(W, H, _p(st["ranges"]), _p(st["point_list"]), _p(st["xy"]), _p(st["rgb"]),
                                      _p(st["depths"]), _p(st["conic_opacity"]), _p(st["bg"]), _p(st["final_T"]),
                                      _p(st["n_contrib"]), _p(gcol), _p(gdep), _p(galp), _p(g["dL_dmean2D"]),
                                      _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolor"]), _p(g["dL_ddepth"]),
                                      self.nthreads)
        M = st["M"]
        g["dL_dmeans3D"] = np.zeros((P, 3), dt)
        g["dL_dcov3D"] = np.zeros((P, 6), dt)
        g["dL_dsh"] = np.zeros((P, M, 3), dt) if st["shs"] is not None else None
        own_cov = st["cov3D_precomp"] is None
        g["dL_dscale"] = np.zeros((P, 3), dt) if own_cov else None
        g["dL_drot"] = np.zeros((P, 4), dt) if own_cov else None
        self.lib.egso_preprocess_backward(
            P, st["D"], M, _p(st["means3D"]), _p(st["radii"]), _p(st["shs"]), _p(st["clamped"]), _p(st["scales"]),
            _p(st["rotations"]), self.real(st["scale_modifier"]), _p(st["cov3D"]), _p(st["viewmatrix"]),
            _p(st["projmatrix"]), _p(st["campos"]), W, H, self.real(st["tanfovx"]), self.real(st["tanfovy"]),
            _p(g["dL_dmean2D"]), _p(g["dL_dconic"]), _p(g["dL_dcolor"]), _p(g["dL_ddepth"]), _p(g["dL_dmeans3D"]),
            _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscale"]), _p(g["dL_drot"]))
        if st["colors_precomp"] is not None:
            g["dL_dcolors_precomp"] = g["dL_dcolor"]
        return g

    def mark_visible(self, means3D, viewmatrix):
        m = _arr(means3D, self.dt).reshape(-1, 3)
        out = np.zeros(m.shape[0], np.uint8)
        self.lib.egso_mark_visible(m.shape[0], _p(m), _p(_arr(viewmatrix, self.dt).reshape(16)), _p(out))
        return out.astype(bool)
