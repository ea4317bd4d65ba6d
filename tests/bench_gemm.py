"""Micro-benchmark of the MFMA GEMM on the U-Net's Linear shapes (one MI355X): python tests/bench_gemm.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
from lvdm_amd import gemm  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print(f"{'shape (M, N, K)':32s} {'ours ms':>8s} {'TF/s':>7s} {'hipBLASLt ms':>12s} {'TF/s':>7s}   fused forms")
for (M, N, K) in [(230400, 320, 320), (230400, 960, 320), (230400, 2560, 320), (230400, 320, 1280), (57600, 640, 640), (57600, 1920, 640),
                  (57600, 5120, 640), (57600, 640, 2560), (14400, 1280, 1280), (14400, 3840, 1280), (14400, 10240, 1280), (14400, 1280, 5120),
                  (3600, 1280, 1280), (3600, 10240, 1280), (25 * 256, 640, 1024), (9216, 9216, 512), (9216, 512, 9216)]:
    x = torch.randn(M, K, device=dev, generator=g).half()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).half()
    b = torch.randn(N, device=dev, generator=g)
    bh = b.half()
    t1 = timeit(lambda: gemm.gemm_nt(x, w, bias=b))
    t2 = timeit(lambda: F.linear(x, w, bh))
    fl = 2.0 * M * N * K / 1e9
    extra = ""
    if K in (320, 640, 1280) and N == 8 * K:
        ln = torch.nn.LayerNorm(K).to(dev).half().requires_grad_(False)
        lin = torch.nn.Linear(K, N).to(dev).half().requires_grad_(False)
        with torch.no_grad():
            t3 = timeit(lambda: gemm.linear(x, lin.weight, lin.bias, ln=ln, geglu=True))
        extra = f"LN + GEMM + GEGLU in one launch (+ row stats): {t3:.3f} ms"
    print(f"{str((M, N, K)):32s} {t1:8.3f} {fl / t1:7.0f} {t2:12.3f} {fl / t2:7.0f}   {extra}")
