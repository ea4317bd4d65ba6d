"""A few launches of the dominant MFMA kernels of the DDIM path at the BASELINE configs[2] shapes, for rocprofv3 --pmc runs:
L0 self-attention (25 frames x 5 heads, N = 9216), the 3x3 convolution (L0 320->320 with the fused GroupNorm+SiLU prologue and
statistics epilogue; L0 640->640; L1 1280->1280), the temporal (3,1,1) convolution at L0, and the VAE 128-channel convolution at
576x1024."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
import torch.nn as nn
from lvdm_amd import conv as C, ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(25, 9216, 320, device=dev, generator=g).half() for _ in range(3))
for _ in range(3):
    o, lse = ops._hip_attention_fwd(q, k, v, 5, False, want_lse=True)
go = torch.randn_like(o)
for _ in range(2):   # the guided step's backward pair (k_attn_bwd_dkv, k_attn_bwd_dq)
    ops._hip_attention_bwd(q, k, v, o, go, lse, 5, False)
del q, k, v, o, go, lse
# temporal self-attention at level 0: the 25 frames of each of the 9216 pixels, read in place from the token-major tensor (bandwidth-bound)
qt, kt, vt = (torch.randn(25, 9216, 320, device=dev, generator=g).half() for _ in range(3))
for _ in range(3):
    ops._hip_attention_fwd(qt, kt, vt, 5, True)
del qt, kt, vt


def conv_case(N, H, W, Cin, Cout, prologue):
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).half()
    m = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half().requires_grad_(False)
    gn = nn.GroupNorm(32, Cin).to(dev).half().requires_grad_(False)
    ns = C.norm_state(gn, x=x, n_stat=N) if prologue else None
    with torch.no_grad():
        for _ in range(3):
            C.fused_conv(x, m, gn=gn if prologue else None, norm=ns, silu=prologue, stats_groups=32 if prologue else 0)


conv_case(25, 72, 128, 320, 320, True)
conv_case(25, 72, 128, 640, 640, False)
conv_case(25, 36, 64, 1280, 1280, True)
conv_case(1, 576, 1024, 128, 128, True)
xt = torch.randn(25, 9216, 320, device=dev, generator=g).half()
m3 = nn.Conv3d(320, 320, (3, 1, 1), padding=(1, 0, 0)).to(dev).half().requires_grad_(False)
with torch.no_grad():
    for _ in range(3):
        C.fused_conv(xt, m3, mode=C.TEMPORAL)
del xt
# the MFMA GEMM (csrc/gemm_mfma.hip) on three of the U-Net's Linear shapes: the level-0 feed-forward in (LayerNorm fold + GEGLU),
# the level-0 attention out-projection with the residual add, the level-2 feed-forward in
from lvdm_amd import gemm as Gm
for (M, N, K, kind) in [(230400, 2560, 320, "geglu"), (230400, 320, 320, "residual"), (14400, 10240, 1280, "geglu"), (57600, 1920, 640, "ln")]:
    xg = torch.randn(M, K, device=dev, generator=g).half()
    lin = nn.Linear(K, N).to(dev).half().requires_grad_(False)
    ln = nn.LayerNorm(K).to(dev).half().requires_grad_(False)
    with torch.no_grad():
        for _ in range(3):
            if kind == "geglu":
                Gm.linear(xg, lin.weight, lin.bias, ln=ln, geglu=True)
            elif kind == "ln":
                Gm.linear(xg, lin.weight, lin.bias, ln=ln)
            else:
                Gm.linear(xg, lin.weight, lin.bias, residual=xg)
torch.cuda.synchronize()
