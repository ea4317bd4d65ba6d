"""ctypes front-end of the CPU raster oracle (oracle/raster_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.  PARITY UNPINNED
against the reference binary (CUDA-only, no tests/goldens upstream); see the
header of raster_oracle.c for how it is pinned instead.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BLOCK = 16


def build(force=False):
    so = os.path.join(_HERE, "_build", "libraster_oracle.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.gvdo_scan.restype = ctypes.c_int64
        _LIB.gvdo_higher_msb.restype = ctypes.c_uint32
    return _LIB


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tan_fovx, tan_fovy,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            sh_degree=0, scale_modifier=1.0):
    """Full forward; returns a dict holding outputs and every intermediate state array
    (names follow rasterizer_impl.h:21-72)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    opacities = _f32(opacities).reshape(-1)
    viewmatrix = _f32(viewmatrix).reshape(-1)
    projmatrix = _f32(projmatrix).reshape(-1)
    campos = _f32(campos).reshape(-1)
    bg = _f32(bg).reshape(-1)
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (shs, colors_precomp, scales, rotations, cov3D_precomp))
    M = 0 if shs is None else shs.shape[1]
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    st = dict(P=P, W=W, H=H, M=M, D=sh_degree)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), np.float32)
    st["depths"] = np.zeros(P, np.float32)
    st["cov3D"] = np.zeros((P, 6), np.float32)
    st["rgb"] = np.zeros((P, 3), np.float32)
    st["conic_opacity"] = np.zeros((P, 4), np.float32)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    st["rects"] = np.zeros((P, 4), np.int32)
    L.gvdo_preprocess(P, int(sh_degree), M, _p(means3D), _p(scales), ctypes.c_float(scale_modifier),
                      _p(rotations), _p(opacities), _p(shs), _p(cov3D_precomp), _p(colors_precomp),
                      _p(viewmatrix), _p(projmatrix), _p(campos), W, H,
                      ctypes.c_float(tan_fovx), ctypes.c_float(tan_fovy),
                      _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["cov3D"]),
                      _p(st["rgb"]), _p(st["conic_opacity"]), _p(st["tiles_touched"]),
                      _p(st["clamped"]), _p(st["rects"]))
    st["point_offsets"] = np.zeros(P, np.uint32)
    R = int(L.gvdo_scan(P, _p(st["tiles_touched"]), _p(st["point_offsets"])))
    st["R"] = R
    st["keys_unsorted"] = np.zeros(max(R, 1), np.uint64)
    st["vals_unsorted"] = np.zeros(max(R, 1), np.uint32)
    st["keys"] = np.zeros(max(R, 1), np.uint64)
    st["point_list"] = np.zeros(max(R, 1), np.uint32)
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    L.gvdo_bin(P, W, H, _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["point_offsets"]),
               ctypes.c_uint64(R), _p(st["keys_unsorted"]), _p(st["vals_unsorted"]), _p(st["keys"]),
               _p(st["point_list"]), _p(st["ranges"]))
    for k in ("keys_unsorted", "vals_unsorted", "keys", "point_list"):
        st[k] = st[k][:R]
    feats = colors_precomp if colors_precomp is not None else st["rgb"]
    st["features"] = feats
    st["color"] = np.zeros((3, H, W), np.float32)
    st["depth"] = np.zeros((1, H, W), np.float32)
    st["alpha"] = np.zeros((1, H, W), np.float32)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    L.gvdo_render(W, H, _p(st["ranges"]), _p(np.ascontiguousarray(st["point_list"])), _p(st["means2D"]),
                  _p(feats), _p(st["depths"]), _p(st["conic_opacity"]), _p(bg),
                  _p(st["color"]), _p(st["depth"]), _p(st["alpha"]), _p(st["n_contrib"]))
    st["_in"] = dict(means3D=means3D, opacities=opacities, viewmatrix=viewmatrix, projmatrix=projmatrix,
                     campos=campos, bg=bg, shs=shs, colors_precomp=colors_precomp, scales=scales,
                     rotations=rotations, cov3D_precomp=cov3D_precomp, tan_fovx=tan_fovx, tan_fovy=tan_fovy,
                     scale_modifier=scale_modifier)
    return st


def backward(st, dL_dcolor, dL_ddepth, dL_dalpha):
    """Backward for a state produced by forward(); returns the reference's 8-tuple
    (rasterize_points.cu:207) as a dict, plus the intermediate per-Gaussian grads."""
    L = lib()
    P, W, H, M, D = st["P"], st["W"], st["H"], st["M"], st["D"]
    i = st["_in"]
    dL_dcolor = _f32(dL_dcolor).reshape(3, H, W)
    dL_ddepth = _f32(dL_ddepth).reshape(H, W)
    dL_dalpha = _f32(dL_dalpha).reshape(H, W)
    g = dict()
    g["dL_dmeans2D"] = np.zeros((P, 3), np.float32)
    g["dL_dconic"] = np.zeros((P, 4), np.float32)
    g["dL_dopacity"] = np.zeros((P, 1), np.float32)
    g["dL_dcolors"] = np.zeros((P, 3), np.float32)
    g["dL_ddepths"] = np.zeros((P, 1), np.float32)
    L.gvdo_render_backward(P, W, H, _p(st["ranges"]), _p(np.ascontiguousarray(st["point_list"])), _p(i["bg"]),
                           _p(st["means2D"]), _p(st["conic_opacity"]), _p(st["features"]), _p(st["depths"]),
                           _p(st["alpha"]), _p(st["n_contrib"]), _p(dL_dcolor), _p(dL_ddepth), _p(dL_dalpha),
                           _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]),
                           _p(g["dL_dcolors"]), _p(g["dL_ddepths"]))
    derived_from_sums(st, g)
    return g


def derived_from_sums(st, g):
    """Second half of the backward (backward.cu:144-412: cov2D, projection, SH, cov3D) from the per-Gaussian sums
    g[dL_dmeans2D, dL_dconic, dL_dcolors, dL_ddepths]; fills the derived gradients into g.  Separate so that tests can
    run the chain on another implementation's sums."""
    L = lib()
    P, W, H, M, D = st["P"], st["W"], st["H"], st["M"], st["D"]
    i = st["_in"]
    for k in ("dL_dmeans2D", "dL_dconic", "dL_dcolors", "dL_ddepths"):
        g[k] = np.ascontiguousarray(g[k], np.float32)
    g["dL_dmeans3D"] = np.zeros((P, 3), np.float32)
    g["dL_dcov3D"] = np.zeros((P, 6), np.float32)
    g["dL_dsh"] = np.zeros((P, M, 3), np.float32)
    g["dL_dscales"] = np.zeros((P, 3), np.float32)
    g["dL_drotations"] = np.zeros((P, 4), np.float32)
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(i["tan_fovy"]))
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(i["tan_fovx"]))
    cov3D = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else st["cov3D"]
    L.gvdo_preprocess_backward(P, int(D), M, _p(i["means3D"]), _p(st["radii"]), _p(i["shs"]), _p(st["clamped"]),
                               _p(i["scales"]), _p(i["rotations"]), ctypes.c_float(i["scale_modifier"]),
                               _p(cov3D), _p(i["viewmatrix"]), _p(i["projmatrix"]),
                               ctypes.c_float(focal_x), ctypes.c_float(focal_y),
                               ctypes.c_float(i["tan_fovx"]), ctypes.c_float(i["tan_fovy"]), _p(i["campos"]),
                               _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]),
                               _p(g["dL_dcolors"]), _p(g["dL_ddepths"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]),
                               _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    L = lib()
    means3D = _f32(means3D)
    out = np.zeros(means3D.shape[0], np.uint8)
    L.gvdo_mark_visible(means3D.shape[0], _p(means3D), _p(_f32(viewmatrix).reshape(-1)),
                        _p(_f32(projmatrix).reshape(-1)), _p(out))
    return out.astype(bool)
