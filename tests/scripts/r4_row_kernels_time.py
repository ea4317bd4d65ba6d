"""Time of the row kernels the guided step keeps under autograd (gvd_layer_norm_bwd_add, gvd_geglu, gvd_geglu_bwd) at its shapes.  (dev tool)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd import ops

DEV = "cuda:0"
_P = ctypes.c_void_p
for M, C in ((112000, 320), (28000, 640), (7000, 1280), (1750, 1280), (230400, 320)):
    x, dy, add = (torch.randn(M, C, device=DEV).half() for _ in range(3))
    gamma = torch.randn(C, device=DEV).half()
    dx = torch.empty_like(x)

    def run():
        ops._check(ops.lib().gvd_layer_norm_bwd_add(_P(x.data_ptr()), _P(dy.data_ptr()), _P(gamma.data_ptr()), _P(add.data_ptr()), _P(dx.data_ptr()),
                                                    ctypes.c_longlong(M), C, ctypes.c_float(1e-5), 0, _P(ops._stream())))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print(f"M {M:6d} C {C:4d}: {us:7.1f} us  {4 * M * C * 2 / us / 1e6:5.2f} TB/s", flush=True)


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, C in ((112000, 1280), (28000, 2560), (7000, 5120)):
    h = (torch.randn(M, 2 * C, device=DEV) * 1.5).half()
    gy = torch.randn(M, C, device=DEV).half()
    gh = torch.empty_like(h)
    t_f = timeit(lambda: ops._hip_geglu(h))
    t_b = timeit(lambda: ops._check(ops.lib().gvd_geglu_bwd(_P(h.data_ptr()), _P(gy.data_ptr()), _P(gh.data_ptr()), ctypes.c_longlong(M), C, 0, _P(ops._stream()))))
    print(f"geglu M {M:6d} C {C:4d}: fwd {t_f:7.1f} us {3 * M * C * 2 / t_f / 1e6:5.2f} TB/s   bwd {t_b:7.1f} us {5 * M * C * 2 / t_b / 1e6:5.2f} TB/s", flush=True)
