"""Micro-benchmark of the MFMA attention kernel at the ViewCrafter shapes.  GPU only.
usage: python tests/bench_attention.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch

from lvdm_amd import ops

dev = "cuda:0"
shapes = [("L0 self 25x5 N=9216", 25, 5, 9216, 9216, False), ("L1 self 25x10 N=2304", 25, 10, 2304, 2304, False),
          ("L2 self 25x20 N=576", 25, 20, 576, 576, False), ("L0 cross text Nk=77", 25, 5, 9216, 77, False),
          ("L0 cross img Nk=256", 25, 5, 9216, 256, False), ("L0 temporal T=25 P=9216", 9216, 5, 25, 25, True)]
g = torch.Generator(device=dev).manual_seed(0)
for name, B, H, Nq, Nk, fm in shapes:
    if fm:
        q, k, v = (torch.randn(Nq, B, H * 64, device=dev, generator=g).half() for _ in range(3))
    else:
        q = torch.randn(B, Nq, H * 64, device=dev, generator=g).half()
        k, v = (torch.randn(B, Nk, H * 64, device=dev, generator=g).half() for _ in range(2))
    for _ in range(3):
        ops.attention(q, k, v, H, frame_major=fm)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        ops.attention(q, k, v, H, frame_major=fm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 4.0 * B * H * Nq * Nk * 64
    # backward (guided sampler): dQ, dK, dV; algorithmic flops = 5 products (S, dP, dV, dK, dQ) = 2.5x forward
    out, lse = ops._hip_attention_fwd(q, k, v, H, fm, want_lse=True)
    go = torch.randn_like(out)
    for _ in range(2):
        ops._hip_attention_bwd(q, k, v, out, go, lse, H, fm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ops._hip_attention_bwd(q, k, v, out, go, lse, H, fm)
    torch.cuda.synchronize()
    db = (time.perf_counter() - t0) / n
    print(f"{name:28s} fwd {dt * 1e3:8.3f} ms {fl / dt / 1e12:7.1f} TFLOP/s   bwd {db * 1e3:8.3f} ms {2.5 * fl / db / 1e12:7.1f} TFLOP/s")
