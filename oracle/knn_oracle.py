"""ctypes wrapper of oracle/knn_oracle.c.  TEST INFRASTRUCTURE ONLY (see the C file's header); PARITY UNPINNED."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(os.path.join(_HERE, "_build", "libknn_oracle.so"))
    return _LIB


def dist2(points):
    """points [P,3] float32 -> (mean squared distance to the 3 nearest neighbours [P] f32, their ids [P,3] i32)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    means = np.zeros(P, np.float32)
    idx = np.zeros((P, 3), np.int32)
    lib().gvdo_knn_mean_dist(pts.ctypes.data_as(ctypes.c_void_p), P, means.ctypes.data_as(ctypes.c_void_p),
                             idx.ctypes.data_as(ctypes.c_void_p))
    return means, idx


def morton_order(points):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    codes = np.zeros(P, np.uint32)
    order = np.zeros(P, np.uint32)
    lib().gvdo_knn_morton_order(pts.ctypes.data_as(ctypes.c_void_p), P, codes.ctypes.data_as(ctypes.c_void_p),
                                order.ctypes.data_as(ctypes.c_void_p))
    return codes, order
