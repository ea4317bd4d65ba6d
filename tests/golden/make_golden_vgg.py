"""Generates tests/golden/vgg_loss_ref.npz by IMPORTING THE REFERENCE'S utils/vgg_loss.py (build container only).

`torchvision` is not installed here, so the two things the reference takes from it are provided by a stand-in module for the
duration of this script: `torchvision.models.vgg19(...).features` (the published VGG-19 configuration 'E': 16 3x3 convolutions
with ReLU, 5 max-poolings, in torchvision's layer order) and `torchvision.transforms.functional.normalize`.  What the golden
pins is therefore the REFERENCE'S OWN logic -- block cuts features[:4],[4:9],[9:18],[18:27],[27:36], ImageNet normalisation,
224x224 bilinear resize, nearest mask, sum of per-block MSE -- and its gradient; the VGG weights are derived from parameter
names (tests/fill_by_name.py), not stored.  Only arrays are written.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fill_by_name import fill_by_name


def _vgg19(pretrained=False, **kw):
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    torch.manual_seed(0)
    return m


def main():
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.models.vgg19 = _vgg19
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
    tv.transforms.functional.normalize = lambda t, mean, std: (t - torch.tensor(mean, dtype=t.dtype)[None, :, None, None]) / torch.tensor(std, dtype=t.dtype)[None, :, None, None]
    for name, mod in (("torchvision", tv), ("torchvision.models", tv.models), ("torchvision.transforms", tv.transforms),
                      ("torchvision.transforms.functional", tv.transforms.functional)):
        sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location("ref_vgg_loss", "/root/reference/utils/vgg_loss.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    vl = ref.VggLoss("cpu")
    fill_by_name(vl, std=0.035)
    g = torch.Generator().manual_seed(11)
    out = {"keys": np.array(sorted(vl.state_dict().keys()))}
    for tag, shape, with_mask in (("a", (1, 3, 40, 56), True), ("b", (2, 3, 64, 48), False)):
        x = torch.rand(shape, generator=g).requires_grad_(True)
        y = torch.rand(shape, generator=g)
        m = (torch.rand(shape[0], 1, *shape[2:], generator=g) > 0.3).float() if with_mask else None
        loss = vl(x, y, mask=m)
        (gx,) = torch.autograd.grad(loss, x)
        out[f"{tag}_x"], out[f"{tag}_y"], out[f"{tag}_loss"], out[f"{tag}_gx"] = x.detach().numpy(), y.numpy(), loss.detach().numpy(), gx.numpy()
        if m is not None:
            out[f"{tag}_mask"] = m.numpy()
    np.savez_compressed(os.path.join(HERE, "vgg_loss_ref.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "vgg_loss_ref.npz")), "bytes", {k: v.shape for k, v in out.items() if "loss" in k}, float(out["a_loss"]))


if __name__ == "__main__":
    main()
