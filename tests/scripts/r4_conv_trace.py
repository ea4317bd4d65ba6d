"""Per-workgroup timeline of the convolution kernel (dev tool, round 4): needs the -DGVD_CONV_TRACE build
(tests/scripts/build_conv_trace.sh -> lib/libgvd_diffusion_ctrace.so, selected with GVD_DIFFUSION_LIB).  For each shape: mean cycles
(s_memtime, 100 MHz ticks) of the three phases of a workgroup -- staging of chunk 0, K loop, epilogue -- and the span of the launch."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import numpy as np
import torch
import torch.nn as nn
from lvdm_amd import conv as C, ops

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def trace(fn, nblocks):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (2048 * 6))()
    assert ops.lib().gvd_conv_trace_read(buf, 2048 * 6) == 0
    a = np.array(buf[:], dtype=np.uint64).reshape(2048, 6)[:min(nblocks, 2048)].astype(np.int64)
    t = a[:, :4] - a[:, :1].min()
    ns = 10.0          # s_memtime ticks at 100 MHz
    return {"fill_us": float((t[:, 1] - t[:, 0]).mean() * ns / 1e3), "loop_us": float((t[:, 2] - t[:, 1]).mean() * ns / 1e3),
            "epilogue_us": float((t[:, 3] - t[:, 2]).mean() * ns / 1e3), "wg_us": float((t[:, 3] - t[:, 0]).mean() * ns / 1e3),
            "span_us": float(t[:, 3].max() * ns / 1e3), "blocks": int(len(a)), "cus": int(len(set((int(r[4]) & 0xF00) | (int(r[4]) >> 13 & 7) << 12 | int(r[5]) << 16 for r in a)))}


for (N, H, W, Cin, Cout) in [(25, 320, 448, 128, 128), (25, 160, 224, 256, 256), (25, 80, 112, 512, 512), (50, 10, 14, 1280, 1280), (50, 5, 7, 1280, 1280)]:
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).half()
    m = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cin, eps=1e-6).to(dev).half().requires_grad_(False)
    res = torch.randn(N, H, W, Cout, device=dev, generator=g).half()
    bn, pix, tw = C.config(0, N, H, W, Cin, Cout)
    nb = N * (-(-H // (pix // tw))) * (-(-W // tw))
    with torch.no_grad():
        ns = C.norm_state(gn, x=x, n_stat=N)
        for tag, fn in (("plain", lambda: C.fused_conv(x, m)), ("prologue+stats+residual", lambda: C.fused_conv(x, m, gn=gn, norm=ns, silu=True, stats_groups=32, residual=res))):
            r = trace(fn, nb)
            print(f"conv {N}x{H}x{W} {Cin}->{Cout} tile {bn}x{pix} [{tag}]: " + "  ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in r.items()), flush=True)
for (S, T, P, Cc) in [(2, 25, 2240, 320), (2, 25, 560, 640), (2, 25, 140, 1280), (2, 25, 35, 1280)]:
    m = nn.Conv3d(Cc, Cc, (3, 1, 1), padding=(1, 0, 0)).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cc).to(dev).half().requires_grad_(False)
    x = torch.randn(S, T, P, Cc, device=dev, generator=g).half()
    bn, pix, _ = C.config(1, T, S, P, Cc, Cc)
    pb = min(32, max(1, pix // T))
    nb = S * (-(-P // pb))
    with torch.no_grad():
        ns = C.norm_state(gn, x=x, n_stat=S)
        for tag, fn in (("plain", lambda: C.fused_conv(x, m, mode=C.TEMPORAL)), ("prologue+stats+residual", lambda: C.fused_conv(x, m, mode=C.TEMPORAL, gn=gn, norm=ns, silu=True, residual=x, stats_groups=32))):
            r = trace(fn, nb)
            print(f"temporal S={S} T={T} P={P} C={Cc} tile {bn}x{pix} [{tag}]: " + "  ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in r.items()), flush=True)
