"""VAE mid-block attention (single head, d = 512) at the 320x448 guided step's shape (25 frames x 2240 tokens) and at 576x1024's (5 frames
x 9216 tokens per decoder group): forward + backward time against the score-buffer budget that sets the query-chunk size.  (dev tool)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd import wide_attention as W

dev = "cuda:0"


def timeit(fn, n=10):
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


for B, N in ((25, 2240), (5, 9216), (25, 560)):
    g = torch.Generator(device=dev).manual_seed(N)
    q, k, v = (torch.randn(B, N, 512, device=dev, generator=g).half().requires_grad_(True) for _ in range(3))
    go = torch.randn(B, N, 512, device=dev, generator=g).half()
    row = f"B {B:2d} N {N:5d}:"
    for mb in (48, 96, 160, 256, 512, 1024):
        W.SCORE_BYTES = mb << 20

        def step():
            o = W.attention(q, k, v)
            torch.autograd.grad(o, (q, k, v), go)
        row += f"  {mb:4d} MB (chunk {W._chunk_rows(B, N):5d}) {timeit(step):8.1f} us"
    print(row, flush=True)
