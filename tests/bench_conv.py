"""Micro-benchmark of the MFMA convolution kernel on the ViewCrafter U-Net / VAE shapes, next to torch's Conv2d (MIOpen) on the
same channels-last fp16 tensors.  Prints one line per shape: TFLOP/s of both, and the time of the GroupNorm+SiLU pass the fused
prologue replaces.   python tests/bench_conv.py [--quick] [--temporal]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.nn as nn

UNET = [  # (N, H, W, Cin, Cout)
    (25, 72, 128, 320, 320), (25, 72, 128, 960, 320), (25, 72, 128, 640, 640),
    (25, 36, 64, 640, 640), (25, 36, 64, 1920, 640), (25, 36, 64, 1280, 1280),
    (25, 18, 32, 1280, 1280), (25, 18, 32, 2560, 1280),
    (25, 9, 16, 1280, 1280), (25, 9, 16, 2560, 1280),
    (25, 72, 128, 320, 4), (25, 72, 128, 8, 320),
]
VAE = [(1, 72, 128, 512, 512), (1, 144, 256, 512, 512), (1, 288, 512, 256, 256), (1, 576, 1024, 128, 128), (1, 576, 1024, 256, 128)]
TEMPORAL = [(25, 9216, 320), (25, 2304, 640), (25, 576, 1280), (25, 144, 1280)]


def timeit(fn, n=10, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-miopen", action="store_true")
    ap.add_argument("--temporal-only", action="store_true")
    args = ap.parse_args()
    from lvdm_amd import conv as C, ops
    dev = "cuda:0"
    torch.backends.cudnn.benchmark = True
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [] if args.temporal_only else UNET[:4] if args.quick else UNET + VAE
    for (N, H, W, Cin, Cout) in shapes:
        x = torch.randn(N, H, W, Cin, device=dev, generator=g).half()
        m = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half()
        for p in m.parameters():
            p.requires_grad_(False)
        flops = 2.0 * N * H * W * Cin * Cout * 9
        with torch.no_grad():
            t_ours = timeit(lambda: C.fused_conv(x, m))
            line = f"conv3x3 N={N} {H}x{W} {Cin}->{Cout}: ours {t_ours * 1e3:8.1f} us {flops / t_ours / 1e9:7.1f} TF"
            if Cin % 32 == 0 and Cin >= 32:
                gn = nn.GroupNorm(32, Cin).to(dev).half()
                for p in gn.parameters():
                    p.requires_grad_(False)
                t_f = timeit(lambda: C.fused_conv(x, m, gn=gn, silu=True, stats_groups=32 if Cout % 32 == 0 else 0))
                ns = C.norm_state(gn, x=x, n_stat=N)
                t_f2 = timeit(lambda: C.fused_conv(x, m, gn=gn, norm=ns, silu=True))
                t_gn = timeit(lambda: ops.group_norm(x.reshape(N, H * W, Cin), 32, gn.weight, gn.bias, gn.eps, silu=True, channels_last=True))
                line += f" | +stats pass+GN/SiLU prologue+stats out {t_f * 1e3:8.1f} us, prologue only {t_f2 * 1e3:8.1f} us | separate GN+SiLU pass {t_gn * 1e3:7.1f} us"
            if not args.no_miopen:
                xc = x.permute(0, 3, 1, 2)  # channels_last view
                mc = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half()
                mc.weight.data = mc.weight.data.contiguous(memory_format=torch.channels_last)
                t_m = timeit(lambda: mc(xc))
                line += f" | MIOpen {t_m * 1e3:8.1f} us {flops / t_m / 1e9:7.1f} TF"
        print(line, flush=True)
    for (T, Pp, Cc) in ([] if args.quick else TEMPORAL):
        x = torch.randn(T, Pp, Cc, device=dev, generator=g).half()
        m = nn.Conv3d(Cc, Cc, (3, 1, 1), padding=(1, 0, 0)).to(dev).half()
        gn = nn.GroupNorm(32, Cc).to(dev).half()
        for p in list(m.parameters()) + list(gn.parameters()):
            p.requires_grad_(False)
        flops = 2.0 * T * Pp * Cc * Cc * 3
        with torch.no_grad():
            t0 = timeit(lambda: C.fused_conv(x, m, mode=C.TEMPORAL))
            ns = C.norm_state(gn, x=x, n_stat=1)
            t1 = timeit(lambda: C.fused_conv(x, m, mode=C.TEMPORAL, gn=gn, norm=ns, silu=True, residual=x, stats_groups=32))
            tp = timeit(lambda: C.fused_conv(x, m, mode=C.TEMPORAL, gn=gn, norm=ns, silu=True))
            tr = timeit(lambda: C.fused_conv(x, m, mode=C.TEMPORAL, residual=x))
            ts = timeit(lambda: C.fused_conv(x, m, mode=C.TEMPORAL, stats_groups=32))
            from lvdm_amd.unet import TemporalConvBlock
            taps = m.weight[:, :, :, 0, 0].permute(2, 0, 1).contiguous()
            t2 = timeit(lambda: TemporalConvBlock._temporal_gemm(x[None], taps, m.bias))
        print(f"temporal T={T} P={Pp} C={Cc}: ours {t0 * 1e3:8.1f} us {flops / t0 / 1e9:7.1f} TF | fused prologue/residual/stats {t1 * 1e3:8.1f} us (prologue only {tp * 1e3:.1f}, residual only {tr * 1e3:.1f}, stats only {ts * 1e3:.1f}) | 3 hipBLASLt GEMMs {t2 * 1e3:8.1f} us {flops / t2 / 1e9:7.1f} TF",
              flush=True)


if __name__ == "__main__":
    main()
