"""Where does the VAE's full-resolution convolution stage lose its time?  (dev tool, round 4)  Times one convolution of each VAE
decoder stage at the 320x448 guided-step geometry (25 frames in one pass) in the forms the guided step launches: plain, with the
GroupNorm+SiLU prologue, with prologue + statistics epilogue (+ residual), and the input-gradient launch with the GroupNorm-backward
statistics epilogue (+ the apply pass).   python tests/scripts/r4_conv_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
import torch.nn as nn
from lvdm_amd import conv as C

dev = "cuda:0"
SHAPES = [(25, 320, 448, 128, 128), (25, 320, 448, 256, 128), (25, 160, 224, 256, 256), (25, 160, 224, 512, 256), (25, 80, 112, 512, 512),
          (25, 40, 56, 512, 512)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]


def timeit(fn, n=6, warm=2):
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
for (N, H, W, Cin, Cout) in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).half()
    m = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cin, eps=1e-6).to(dev).half().requires_grad_(False)
    res = torch.randn(N, H, W, Cout, device=dev, generator=g).half()
    fl = 2.0 * N * H * W * Cin * Cout * 9
    tf = lambda t: fl / t / 1e9
    with torch.no_grad():
        ns = C.norm_state(gn, x=x, n_stat=N)
        t0 = timeit(lambda: C.fused_conv(x, m))
        t1 = timeit(lambda: C.fused_conv(x, m, gn=gn, norm=ns, silu=True))
        t2 = timeit(lambda: C.fused_conv(x, m, gn=gn, norm=ns, silu=True, stats_groups=32))
        t3 = timeit(lambda: C.fused_conv(x, m, gn=gn, norm=ns, silu=True, stats_groups=32, residual=res))
        t4 = timeit(lambda: C.fused_conv(x, m, stats_groups=32))
    # input-gradient launch: d/dx of conv(silu(GN(x))) -- dgrad convolution with the norm-backward statistics epilogue + apply pass
    xg = x.clone().requires_grad_(True)
    y, _ = C.fused_conv(xg, m, gn=gn, norm=ns, silu=True)
    gy = torch.randn_like(y)
    tb = timeit(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True))
    xp = x.clone().requires_grad_(True)
    yp, _ = C.fused_conv(xp, m)
    tbp = timeit(lambda: torch.autograd.grad(yp, xp, gy, retain_graph=True))
    print(f"{N}x{H}x{W} {Cin}->{Cout}: plain {t0*1e3:7.0f} us {tf(t0):6.0f} TF | +prologue {t1*1e3:7.0f} {tf(t1):6.0f} | +prologue+stats {t2*1e3:7.0f} {tf(t2):6.0f} | "
          f"+residual {t3*1e3:7.0f} {tf(t3):6.0f} | stats only {t4*1e3:7.0f} {tf(t4):6.0f} | dgrad plain {tbp*1e3:7.0f} {tf(tbp):6.0f} | "
          f"dgrad + norm-bwd stats + apply {tb*1e3:7.0f} ({tf(tb):6.0f} on the conv flops)", flush=True)
    del x, res, xg, y, gy, xp, yp
    torch.cuda.empty_cache()
