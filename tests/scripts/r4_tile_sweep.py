"""Which tile configuration should each small / short-K convolution take?  (dev tool, round 4)  Run once per GVD_CONV_FORCE_CFG value
(0: 160x256, 1: 320x128, 2: 128x256, 4: 128x128; unset: the shipped rule) -- the library reads it once per process.
python tests/scripts/r4_tile_sweep.py"""
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
out = [f"cfg={os.environ.get('GVD_CONV_FORCE_CFG', 'rule')}"]
for (S, T, P, Cc) in [(2, 25, 2240, 320), (2, 25, 560, 640), (2, 25, 140, 1280), (2, 25, 35, 1280), (1, 25, 9216, 320), (1, 25, 2304, 640), (1, 25, 576, 1280), (1, 25, 144, 1280)]:
    if Cc % 160 and os.environ.get("GVD_CONV_FORCE_CFG") in ("0", "1"):
        continue
    m = nn.Conv3d(Cc, Cc, (3, 1, 1), padding=(1, 0, 0)).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cc).to(dev).half().requires_grad_(False)
    x = torch.randn(S, T, P, Cc, device=dev, generator=g).half()
    xs = x if S > 1 else x[0]
    with torch.no_grad():
        ns = C.norm_state(gn, x=xs, n_stat=S)
        t1 = timeit(lambda: C.fused_conv(xs, m, mode=C.TEMPORAL, gn=gn, norm=ns, silu=True, residual=xs, stats_groups=32))
    out.append(f"t{S}x{P}x{Cc}:{t1*1e3:.0f}")
for (N, H, W, Cin, Cout) in [(50, 10, 14, 1280, 1280), (50, 10, 14, 2560, 1280), (50, 5, 7, 1280, 1280), (50, 5, 7, 2560, 1280), (50, 20, 28, 640, 640), (50, 40, 56, 320, 320),
                             (25, 9, 16, 1280, 1280), (25, 18, 32, 1280, 1280), (25, 36, 64, 640, 640), (25, 72, 128, 320, 320),
                             (25, 40, 56, 512, 512), (25, 80, 112, 512, 512), (25, 160, 224, 256, 256), (25, 320, 448, 128, 128)]:
    if Cout % 160 and os.environ.get("GVD_CONV_FORCE_CFG") in ("0", "1"):
        continue
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).half()
    m = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cin).to(dev).half().requires_grad_(False)
    with torch.no_grad():
        ns = C.norm_state(gn, x=x, n_stat=N)
        t1 = timeit(lambda: C.fused_conv(x, m, gn=gn, norm=ns, silu=True, stats_groups=32), n=8, warm=3)
    out.append(f"s{N}x{H}x{W}x{Cin}>{Cout}:{t1*1e3:.0f}")
    del x
print("  ".join(out), flush=True)
