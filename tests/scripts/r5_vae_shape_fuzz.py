"""VAE decoder shape fuzz (round 5 dev tool): a three-level miniature of the KL-VAE decoder with the shipped one's structure (ch 128 x [1, 2, 4]:
the mid block's single attention head is 512 wide like the shipped decoder's -- the chunked wide-head path --, nearest x2 Upsample convolutions,
128-channel full-resolution ResnetBlocks, conv_out to 3 channels) on random (frames, latent height, latent width), token-major fp16 HIP path
against the fp32 torch form of the same weights, forward and input gradient, frames decoded in one call and in groups (`perframe`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import copy, warnings
import numpy as np, torch
from fill_by_name import fill_by_name
from lvdm_amd.vae import AutoencoderKLDecoder
dev = "cuda:0"
CFG = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4], num_res_blocks=1, attn_resolutions=[], dropout=0.0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
base = fill_by_name(AutoencoderKLDecoder(CFG), std=0.03).eval().to(dev).requires_grad_(False)
v32 = base
v16 = copy.deepcopy(base).half().to_token_major()
g = torch.Generator(device=dev).manual_seed(2)
warnings.simplefilter("ignore")
worst = [0.0, 0.0]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    n = int(rng.choice([1, 2, 3, 5, 7])); H = int(rng.integers(3, 25)); W = int(rng.integers(3, 33))
    z = torch.randn(n, 4, H, W, device=dev, generator=g)
    z16 = z.half().requires_grad_(True); z32 = z.clone().requires_grad_(True)
    k = int(rng.choice([0, 1, 2]))
    y16 = v16.perframe(lambda zz: v16.decode(zz), z16, frames_per_call=k or None)
    y32 = v32.decode(z32)
    assert y16.shape == y32.shape == (n, 3, 4 * H, 4 * W), (y16.shape, y32.shape)
    e = float((y16.float() - y32).abs().max() / y32.abs().max())
    gy = torch.randn(y32.shape, device=dev, generator=g)
    y16.backward(gy.half()); y32.backward(gy)
    eg = float((z16.grad.float() - z32.grad).abs().max() / z32.grad.abs().max())
    worst = [max(worst[0], e), max(worst[1], eg)]
    print(f"frames {n} latent {H}x{W} group {k or 'all'}: forward {e:.2e}, input gradient {eg:.2e}", flush=True)
    assert e < 3e-2 and eg < 6e-2 and torch.isfinite(y16).all(), (n, H, W, k, e, eg)
print(f"vae shape fuzz ok; worst forward {worst[0]:.2e}, worst input gradient {worst[1]:.2e}")
