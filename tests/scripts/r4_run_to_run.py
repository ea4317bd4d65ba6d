"""Run-to-run spread of the full-size fp16 U-Net forward (25 x 72 x 128 latents): the only non-deterministic ingredient is the order of
the fp64 atomics behind the GroupNorm statistics.  Prints max |ya - yb| / max |ya| over a few repeats.  (dev tool)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd.model import VIEWCRAFTER_UNET
from lvdm_amd.unet import UNetModel

DEV = "cuda:0"
torch.manual_seed(0)
with torch.device(DEV):
    unet = UNetModel(**VIEWCRAFTER_UNET)
g = torch.Generator(device=DEV).manual_seed(0)
with torch.no_grad():
    for p_ in unet.parameters():
        if float(p_.abs().max()) == 0.0:
            p_.copy_(torch.randn(p_.shape, device=DEV, generator=g) * 0.02)
unet.eval().requires_grad_(False).half().to_token_major()
ctx = torch.randn(1, 333, 1024, device=DEV, generator=g).half()
t, fs = torch.tensor([500], device=DEV), torch.tensor([10], device=DEV)
x = torch.randn(1, 8, 25, 72, 128, device=DEV, generator=g).half()
with torch.no_grad():
    ys = [unet(x, t, context=ctx, fs=fs).float() for _ in range(5)]
ref = ys[0]
for i, y in enumerate(ys[1:], 1):
    d = (y - ref).abs()
    print(f"run {i}: max |diff| / max |y| = {float(d.max() / ref.abs().max()):.2e}   differing elements {float((d > 0).float().mean()):.3f}   rms diff / rms y = {float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}", flush=True)
