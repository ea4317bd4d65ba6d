"""Deterministic, architecture-order-independent weights: every parameter/buffer is filled from a
generator seeded by the CRC32 of its state-dict key.  The golden script applies it to the REFERENCE
modules, the tests apply it to the rebuilt modules: equal key names => equal weights, nothing stored.
(Also solves SURVEY's 'random-init U-Net is degenerate': zero-init convs / proj_out become non-zero.)"""
import zlib

import torch


def fill_by_name(module, std=0.05):
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
            r = torch.randn(v.shape, generator=g) * std
            if k.endswith("norm.weight") or ".norm" in k and k.endswith("weight") or k.split(".")[-2:] == ["0", "weight"] and v.dim() == 1:
                r = r + 1.0  # keep normalisation gains around 1
            v.copy_(r)
    return module
