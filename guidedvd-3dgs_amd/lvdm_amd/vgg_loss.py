"""VGG-19 perceptual distance of the guidance loss (the reference calls it `lpips_guidance`).

Restates `VggLoss` of utils/vgg_loss.py:4-52 as used by LossGuidance (utils/viewcrafter_wrapper.py:79-80,157-159): ImageNet
normalisation, bilinear resize of both images to 224x224 (masks: nearest, applied after the resize), five VGG-19 feature blocks
cut after the activations in front of each pooling layer (torchvision `features[:4], [4:9], [9:18], [18:27], [27:36]`), loss =
sum over the blocks of mse(block(x), block(y)).

The 16 convolutions are 3x3 / pad 1: on a ROCm device they run the package's MFMA convolution kernel (forward and input
gradient -- the guided sampler differentiates the loss w.r.t. the decoded frame), fp16 activations, fp32 loss.  Parameter names
follow torchvision's `vgg19().features` indices inside the reference's five `blocks` (`blocks.1.5.weight`, ...), so a state dict
of the reference module loads strictly.  Weights: torchvision's pretrained VGG-19 when torchvision is importable (what the
reference downloads), or a state dict given by `weights=` / $GVD_VGG19_WEIGHTS; `pretrained=False` leaves the random init (tests).
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv as mconv, ops

_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
_CUTS = [(0, 4), (4, 9), (9, 18), (18, 27), (27, 36)]
VGG_MEAN, VGG_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _vgg19_features():
    layers, cin = [], 3
    for v in _CFG_E:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=False)]
            cin = v
    return layers


class _ScaleGrad(torch.autograd.Function):
    """Identity whose backward multiplies the gradient by a constant.  The per-block MSE divides by millions of elements, so the
    gradients entering the fp16 feature chain are ~1e-7 -- below fp16's normal range.  The fp16 part of the backward therefore
    runs on gradients scaled by GRAD_SCALE (applied where the fp32 loss hands over to the fp16 features, undone in fp32 where
    the chain reaches the input image): plain loss scaling, local to this module."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


GRAD_SCALE = 2.0 ** 14


class VggLoss(nn.Module):
    def __init__(self, device=None, resize=True, pretrained=True, weights=None):
        super().__init__()
        feats = _vgg19_features()
        self.blocks = nn.ModuleList([nn.Sequential(OrderedDict((str(i), feats[i]) for i in range(a, b))) for a, b in _CUTS])
        self.loss_blocks = [0, 1, 2, 3, 4]
        self.resize = resize
        weights = weights or os.environ.get("GVD_VGG19_WEIGHTS")
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            sd = sd.get("state_dict", sd)
            if any(k.startswith("features.") for k in sd):       # a torchvision vgg19 state dict
                idx2blk = {i: b for b, (a, e) in enumerate(_CUTS) for i in range(a, e)}
                sd = {f"blocks.{idx2blk[int(k.split('.')[1])]}.{k.split('.', 1)[1]}": v for k, v in sd.items()
                      if k.startswith("features.") and int(k.split(".")[1]) < 36}
            self.load_state_dict(sd, strict=True)
        elif pretrained:
            try:
                import torchvision
            except ImportError as e:
                raise RuntimeError("VggLoss needs the pretrained VGG-19 weights: install torchvision (what the reference uses), or pass "
                                   "weights=<state dict / path> or set GVD_VGG19_WEIGHTS") from e
            tv = torchvision.models.vgg19(pretrained=True).features
            for blk in self.blocks:
                for name, m in blk.named_children():
                    if isinstance(m, nn.Conv2d):
                        m.load_state_dict(tv[int(name)].state_dict())
        for p in self.parameters():
            p.requires_grad = False
        self.eval()
        if device is not None:
            self.to(device)

    def _features(self, x):
        """x [n, 3, H, W] fp32 -> list of the five block outputs."""
        outs = []
        fused = x.is_cuda and not ops._REFERENCE_MATH
        if fused:
            if x.requires_grad:
                x = _ScaleGrad.apply(x, 1.0 / GRAD_SCALE)       # (fp32) undo the gradient scaling of the fp16 chain
            t = x.permute(0, 2, 3, 1).half().contiguous()      # token-major fp16 for the MFMA convolutions
        for bi, blk in enumerate(self.blocks):
            for m in blk:
                if isinstance(m, nn.Conv2d):
                    if fused:
                        t = mconv.fused_conv(t, m)[0]
                    else:
                        x = m(x)
                elif isinstance(m, nn.ReLU):
                    t, x = (F.relu(t), x) if fused else (None, F.relu(x))
                else:  # 2x2 max pooling
                    if fused:
                        t = F.max_pool2d(t.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()
                    else:
                        x = F.max_pool2d(x, 2, 2)
            outs.append(_ScaleGrad.apply(t.float(), GRAD_SCALE) if (fused and t.requires_grad) else (t.float() if fused else x))
            if bi == self.loss_blocks[-1]:
                break
        return outs

    def forward(self, input, target, mask=None):
        if mask is not None:
            mask = mask.to(torch.float32)
        if input.shape[1] != 3:
            input, target = input.repeat(1, 3, 1, 1), target.repeat(1, 3, 1, 1)
        mean = torch.tensor(VGG_MEAN, device=input.device, dtype=input.dtype)[None, :, None, None]
        std = torch.tensor(VGG_STD, device=input.device, dtype=input.dtype)[None, :, None, None]
        input, target = (input - mean) / std, (target - mean) / std
        if self.resize:
            input = F.interpolate(input, mode="bilinear", size=(224, 224), align_corners=False)
            target = F.interpolate(target, mode="bilinear", size=(224, 224), align_corners=False)
            if mask is not None:
                mask = F.interpolate(mask, mode="nearest", size=(224, 224))
                input, target = input * mask, target * mask
        if input.is_cuda and not ops._REFERENCE_MATH and any(p.dtype != torch.float16 for p in self.parameters()):
            self.half()                                         # fp16 weights for the MFMA convolutions (once)
        fx, fy = self._features(input), self._features(target)
        loss = 0.0
        for i in self.loss_blocks:
            loss = loss + F.mse_loss(fx[i], fy[i])
        return loss
