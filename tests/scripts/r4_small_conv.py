"""Small-level convolutions of the 320x448 guided step (dev tool, round 4): the temporal (3,1,1) form at the four U-Net levels for
one sample and for the batch-2 CFG pair in ONE launch, and the 3x3 form on the 10x14 / 5x7 latents -- the launches that under-fill the
chip (40-560 workgroups).   python tests/scripts/r4_small_conv.py        (GVD_DIFFUSION_LIB selects an A/B build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
import torch.nn as nn
from lvdm_amd import conv as C

dev = "cuda:0"


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


g = torch.Generator(device=dev).manual_seed(0)
print("lib:", os.environ.get("GVD_DIFFUSION_LIB", "default"))
for (T, P, Cc) in [(25, 2240, 320), (25, 560, 640), (25, 140, 1280), (25, 35, 1280)]:
    m = nn.Conv3d(Cc, Cc, (3, 1, 1), padding=(1, 0, 0)).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cc).to(dev).half().requires_grad_(False)
    fl = 2.0 * T * P * Cc * Cc * 3
    row = f"temporal T={T} P={P:5d} C={Cc:5d}:"
    for S in (1, 2):
        x = torch.randn(S, T, P, Cc, device=dev, generator=g).half()
        xs = x if S > 1 else x[0]
        with torch.no_grad():
            ns = C.norm_state(gn, x=xs, n_stat=S)
            t0 = timeit(lambda: C.fused_conv(xs, m, mode=C.TEMPORAL))
            t1 = timeit(lambda: C.fused_conv(xs, m, mode=C.TEMPORAL, gn=gn, norm=ns, silu=True, residual=xs, stats_groups=32))
        row += f"  samples {S}: plain {t0*1e3:6.1f} us {S*fl/t0/1e9:6.0f} TF, prologue+residual+stats {t1*1e3:6.1f} us {S*fl/t1/1e9:6.0f} TF |"
    print(row, flush=True)
for (N, H, W, Cin, Cout) in [(50, 10, 14, 1280, 1280), (50, 10, 14, 2560, 1280), (50, 5, 7, 1280, 1280), (50, 5, 7, 2560, 1280), (50, 20, 28, 640, 640), (50, 20, 28, 1280, 640)]:
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).half()
    m = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cin).to(dev).half().requires_grad_(False)
    fl = 2.0 * N * H * W * Cin * Cout * 9
    with torch.no_grad():
        ns = C.norm_state(gn, x=x, n_stat=N)
        t0 = timeit(lambda: C.fused_conv(x, m))
        t1 = timeit(lambda: C.fused_conv(x, m, gn=gn, norm=ns, silu=True, stats_groups=32))
    print(f"conv3x3 N={N} {H}x{W} {Cin}->{Cout}: plain {t0*1e3:6.1f} us {fl/t0/1e9:6.0f} TF | prologue+stats {t1*1e3:6.1f} us {fl/t1/1e9:6.0f} TF   tile {C.config(0, N, H, W, Cin, Cout)}", flush=True)
