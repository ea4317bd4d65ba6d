"""`simple_knn._C` -- same name and call as the reference's pybind module (submodules/simple-knn/ext.cpp:15-17):

    distCUDA2(points[P,3] float32 on the GPU) -> (meanDists[P] float32, nearestIndices[P,3] int32)      (spatial.cu:15-26)

backed by lib/libgvd_knn.so (csrc/knn.hip) through the C-ABI of include/gvd_knn.h.  No CPU path: a CPU tensor raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GVD_KNN_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libgvd_knn.so")  # env: A/B and sanitizer builds
_LIB = None


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOOP = _Noop()


def _on(dev):
    """Device guard that is free when `dev` is already the current device (torch.cuda.device() costs ~25 us per use)."""
    idx = dev.index
    return _NOOP if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(dev)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: build the HIP extensions first "
                               f"(python -c 'import __graft_entry__ as g; g.build()')")
        L = ctypes.CDLL(_LIB_PATH)
        L.gvd_knn_workspace_bytes.restype = ctypes.c_size_t
        L.gvd_knn_workspace_bytes.argtypes = [ctypes.c_int]
        L.gvd_knn_last_error.restype = ctypes.c_char_p
        L.gvd_knn_mean_dist.restype = ctypes.c_int
        L.gvd_knn_mean_dist.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.c_void_p]
        _LIB = L
    return _LIB


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("simple_knn.distCUDA2: points must live on a ROCm device (this build has no CPU path)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError(f"simple_knn.distCUDA2: expected [P, 3] points, got {tuple(points.shape)}")
    pts = points.detach().contiguous().float()
    P = pts.shape[0]
    means = torch.zeros(P, dtype=torch.float32, device=pts.device)          # torch::full({P}, 0.0), spatial.cu:22
    nearest = torch.empty((P, 3), dtype=torch.int32, device=pts.device)
    if P == 0:
        return means, nearest
    L = lib()
    nbytes = L.gvd_knn_workspace_bytes(P)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=pts.device)
    base = (ws.data_ptr() + 255) & ~255
    with _on(pts.device):
        rc = L.gvd_knn_mean_dist(pts.data_ptr(), P, means.data_ptr(), nearest.data_ptr(), base, nbytes,
                                 _stream())
    if rc != 0:
        raise RuntimeError(f"gvd_knn error {rc}: {L.gvd_knn_last_error().decode()}")
    return means, nearest
