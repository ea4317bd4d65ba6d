"""GroupNorm forward-statistics / backward kernels at the VAE and U-Net shapes: time and effective bandwidth.
python tests/scripts/bench_gn.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from bench_conv import timeit
from lvdm_amd import ops

dev = "cuda:0"
for (N, S, C) in [(1, 589824, 128), (5, 589824, 128), (5, 147456, 256), (5, 36864, 512), (25, 9216, 320), (25, 2304, 640), (25, 576, 1280)]:
    x = torch.randn(N, S, C, device=dev).half()
    gy = torch.randn(N, S, C, device=dev).half()
    w = torch.ones(C, device=dev).half()
    b = torch.zeros(C, device=dev).half()
    y, xc, g32, stats, _ = ops._hip_group_norm(x, 32, w, b, 1e-6, True, True, keep=True)
    t_f = timeit(lambda: ops._hip_group_norm(x, 32, w, b, 1e-6, True, True, keep=True))
    t_b = timeit(lambda: ops._hip_group_norm_bwd(x, gy, g32, stats, 32, 1e-6, True, True))
    by = N * S * C * 2
    print(f"N={N} S={S} C={C}: fwd (stats+apply, 3 passes) {t_f * 1e3:7.1f} us {3 * by / t_f / 1e9:6.2f} TB/s | "
          f"bwd (stats+apply, 5 passes) {t_b * 1e3:7.1f} us {5 * by / t_b / 1e9:6.2f} TB/s", flush=True)
