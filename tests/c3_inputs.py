"""Inputs of the C3-resolution U-Net anchor (tests/golden/make_golden_c3_unet.py writes the fixture from them in the build container,
tests/test_diffusion_goldens_gpu.py feeds the same tensors to the HIP path): seeded CPU generators, nothing stored."""
import torch

T, HL, WL, STD = 2, 72, 128, 0.02   # frames, latent height / width (576 x 1024 video), weight std


def inputs():
    g = torch.Generator().manual_seed(576)
    mk = lambda *s: torch.randn(*s, generator=g)
    return dict(x=mk(1, 8, T, HL, WL), ctx=mk(1, 333, 1024), t=torch.tensor([500]), fs=torch.tensor([10]))
