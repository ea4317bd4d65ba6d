"""Inputs of the full-width guided-step anchor (tests/golden/make_golden_fullwidth_guided.py writes the fixture from them in the build
container, tests/test_fullwidth_guided_gpu.py feeds the same tensors to the HIP path): seeded CPU generators, nothing stored."""
import torch

T, HL, WL, INDEX, STD = 3, 40, 56, 30, 0.02   # frames, latent height / width (320 x 448 video), DDIM index (of 50), weight std


def inputs():
    g = torch.Generator().manual_seed(2025)
    mk = lambda *s: torch.randn(*s, generator=g)
    d = dict(x=mk(1, 4, T, HL, WL), concat=mk(1, 4, T, HL, WL) * 0.2, ctx_c=mk(1, 333, 1024), ctx_uc=mk(1, 333, 1024),
             noise0=mk(1, 4, T, HL, WL), noise1=mk(1, 4, T, HL, WL))
    d["guide_imgs"] = torch.rand(T, 3, 8 * HL, 8 * WL, generator=g)
    d["guide_masks"] = (torch.rand(T, 1, 8 * HL, 8 * WL, generator=g) > 0.3).float()
    return d
