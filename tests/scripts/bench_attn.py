"""Flash-attention forward / backward at the U-Net shapes: time and algorithmic TFLOP/s.  python tests/scripts/bench_attn.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from bench_conv import timeit
from lvdm_amd import ops
g = torch.Generator(device="cuda:0").manual_seed(0)
for (B, N, Nk, H, bwd) in [(25, 9216, 9216, 5, True), (25, 2304, 2304, 10, True), (25, 576, 576, 20, True), (25, 9216, 77, 5, False),
                           (25, 9216, 256, 5, False)]:
    q = torch.randn(B, N, H * 64, device="cuda:0", generator=g).half()
    k, v = (torch.randn(B, Nk, H * 64, device="cuda:0", generator=g).half() for _ in range(2))
    o, lse = ops._hip_attention_fwd(q, k, v, H, False, want_lse=True)
    t = timeit(lambda: ops._hip_attention_fwd(q, k, v, H, False, want_lse=True), n=5, warm=2)
    fl = 4.0 * B * H * N * Nk * 64
    line = f"B={B} N={N} Nk={Nk} H={H}: fwd {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TF"
    if bwd:
        go = torch.randn_like(o)
        tb = timeit(lambda: ops._hip_attention_bwd(q, k, v, o, go, lse, H, False), n=3, warm=1)
        line += f" | bwd {tb * 1e3:8.1f} us {2.5 * fl / tb / 1e9:7.1f} TF (algorithmic)"
    print(line, flush=True)
