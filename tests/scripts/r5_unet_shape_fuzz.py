"""Whole-network shape fuzz (round 5 dev tool): a three-level miniature of the ViewCrafter U-Net (every module kind of the shipped one:
ResBlocks with temporal convolutions, spatial + temporal transformers, image cross-attention, Down / Upsample) evaluated on random
(batch, frames, latent height, latent width) -- frames 1-25, maps 8 x 8 to 48 x 64, also sizes whose deepest level is 1 x 1 or odd x odd --
on the fp16 HIP path against the package's fp32 torch form of the same weights, forward and input gradient.  Bars: 3e-2 / 6e-2 of the
largest entry (the fp16-vs-fp32 bar of tests/test_diffusion_gpu.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import copy, warnings
import numpy as np, torch
from fill_by_name import fill_by_name
from lvdm_amd.unet import UNetModel
dev = "cuda:0"
CFG = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[1, 2, 4], num_res_blocks=1, channel_mult=[1, 2, 4], dropout=0.0,
           num_head_channels=64, transformer_depth=1, context_dim=64, use_linear=True, use_checkpoint=False, temporal_conv=True,
           temporal_attention=True, temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False, temporal_length=16,
           addition_attention=True, image_cross_attention=True, default_fs=10, fs_condition=True)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
base = fill_by_name(UNetModel(**CFG), std=0.05).eval().to(dev).requires_grad_(False)
u32 = base
u16 = copy.deepcopy(base).half().to_token_major()
g = torch.Generator(device=dev).manual_seed(1)
worst = [0.0, 0.0]
warnings.simplefilter("ignore")
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    b = int(rng.choice([1, 1, 2])); T = int(rng.choice([1, 2, 3, 5, 16, 25])); H = 4 * int(rng.integers(2, 13)); W = 4 * int(rng.integers(2, 17))
    ctx_len = int(rng.choice([77 + 16 * T, 93, 333]))
    x = torch.randn(b, 8, T, H, W, device=dev, generator=g)
    ctx = torch.randn(b, ctx_len, 64, device=dev, generator=g)
    t = torch.randint(0, 1000, (b,), device=dev, generator=g); fs = torch.full((b,), 10, device=dev)
    x16 = x.half().requires_grad_(True); x32 = x.clone().requires_grad_(True)
    y16 = u16(x16, t, context=ctx.half(), fs=fs)
    y32 = u32(x32, t, context=ctx, fs=fs)
    e = float((y16.float() - y32).abs().max() / y32.abs().max())
    gy = torch.randn(y32.shape, device=dev, generator=g)
    y16.backward(gy.half()); y32.backward(gy)
    eg = float((x16.grad.float() - x32.grad).abs().max() / x32.grad.abs().max())
    worst = [max(worst[0], e), max(worst[1], eg)]
    print(f"b {b} T {T} {H}x{W} ctx {ctx_len}: forward {e:.2e}, input gradient {eg:.2e}", flush=True)
    assert e < 3e-2 and eg < 6e-2 and torch.isfinite(y16).all(), (b, T, H, W, ctx_len, e, eg)
print(f"unet shape fuzz ok; worst forward {worst[0]:.2e}, worst input gradient {worst[1]:.2e}")
