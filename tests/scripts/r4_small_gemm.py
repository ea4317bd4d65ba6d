"""Small / skinny GEMMs of the U-Net (context projections, per-frame embeddings, the deepest level): time per launch.  (dev tool;
A/B of library builds through GVD_DIFFUSION_LIB)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd import gemm

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (M, N, K) in [(77, 2560, 1024), (256, 2560, 1024), (77, 640, 1024), (256, 1280, 1024), (25, 1280, 1280), (25, 320, 1280), (1, 1280, 1280), (3600, 1280, 1280), (3600, 1280, 5120),
                  (3600, 10240, 1280), (3600, 3840, 1280), (7000, 1280, 1280), (7000, 1280, 10240), (7000, 10240, 1280), (1750, 1280, 1280), (1750, 1280, 5120)]:
    x = torch.randn(M, K, device=dev, generator=g).half()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).half()
    b = torch.randn(N, device=dev, generator=g)
    y = gemm.gemm_nt(x, w, bias=b)
    ref = x.float() @ w.float().t() + b
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: gemm.gemm_nt(x, w, bias=b))
    print(f"{str((M, N, K)):24s} {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF/s  err {err:.1e}", flush=True)
