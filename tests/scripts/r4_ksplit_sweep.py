"""Split-K factor sweep on the few-workgroup convolutions of both bench resolutions (dev tool): time of fused_conv (GroupNorm + SiLU
prologue, residual, statistics -- what the U-Net calls) for forced slice counts; the row `auto` is what conv._ksplit picks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
import torch.nn as nn
from lvdm_amd import conv as C

dev = "cuda:0"


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
KS = (1, 2, 3, 4, 5, 8)
print(f"{'shape':44s} " + " ".join(f"k={k:<5d}" for k in KS) + "  auto")
cases = [("t", (2, 25, 140), 1280, 1280), ("t", (2, 25, 35), 1280, 1280), ("t", (2, 25, 560), 640, 640), ("t", (25, 144), 1280, 1280), ("t", (25, 576), 1280, 1280),
         ("t", (25, 2304), 640, 640), ("s", (50, 5, 7), 1280, 1280), ("s", (50, 5, 7), 2560, 1280), ("s", (50, 10, 14), 1280, 1280), ("s", (50, 10, 14), 2560, 1280),
         ("s", (25, 9, 16), 1280, 1280), ("s", (25, 9, 16), 2560, 1280), ("s", (25, 18, 32), 1280, 1280), ("s", (5, 40, 56), 512, 512)]
for kind, shape, Cin, Cout in cases:
    mode = C.TEMPORAL if kind == "t" else C.SPATIAL
    x = torch.randn(*shape, Cin, device=dev, generator=g).half()
    res = torch.randn(*shape, Cout, device=dev, generator=g).half()
    m = (nn.Conv3d(Cin, Cout, (3, 1, 1), padding=(1, 0, 0)) if kind == "t" else nn.Conv2d(Cin, Cout, 3, padding=1)).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cin).to(dev).half().requires_grad_(False)
    n_stat = (shape[0] if len(shape) == 3 else 1) if kind == "t" else shape[0]
    row = f"{kind} {str(shape):18s} {Cin:5d} -> {Cout:5d}           "[:44]
    with torch.no_grad():
        ns = C.norm_state(gn, x=x, n_stat=n_stat)
        f = lambda: C.fused_conv(x, m, mode=mode, gn=gn, norm=ns, silu=True, residual=res, stats_groups=32)
        for k in KS:
            C.FORCE_KSPLIT = k
            row += f" {timeit(f):7.1f}"
        C.FORCE_KSPLIT = None
        row += f"  {timeit(f):7.1f}"
    print(row, flush=True)
