"""Frozen OpenCLIP ViT-H/14 towers that produce the once-per-video conditioning:

    FrozenOpenCLIPEmbedder          text  [b] str      -> [b, 77, 1024]   (penultimate layer, ln_final)   condition.py:174-232
    FrozenOpenCLIPImageEmbedderV2   image [b,3,H,W]    -> [b, 257, 1280]  (all tokens of the last block)  condition.py:295-372

The reference builds both through the un-vendored pip dependency `open_clip` (`open_clip.create_model_and_transforms
("ViT-H-14", pretrained="laion2b_s32b_b79k")`, requirements.txt; not importable in this image) and `kornia`.  Here the
towers are plain modules with open_clip's parameter tree (`model.visual.transformer.resblocks.N.attn.in_proj_weight`, ...)
so the `cond_stage_model.*` / `embedder.*` keys of the ViewCrafter checkpoint load strict without either package; only the
BPE tokenizer (a vocabulary file that ships inside open_clip) is still taken from open_clip when a prompt must be tokenised.
**Parity unpinned**: with open_clip absent no reference output can be generated here; the architecture follows open_clip's
published ViT-H-14 configuration (vision: width 1280, 32 layers, 16 heads, patch 14, 224 px; text: width 1024, 24 layers,
16 heads, 77 tokens, vocabulary 49408; GELU MLP x4; pre-LN residual blocks) and the call sequence the reference spells out.
These run once per video, outside the DDIM loop: attention is torch's scaled_dot_product_attention (d = 80 / causal).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attention(nn.Module):
    """nn.MultiheadAttention's parameter names (in_proj_weight, in_proj_bias, out_proj), self-attention only."""

    def __init__(self, width, heads):
        super().__init__()
        self.heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, x, causal=False):   # x [b, n, c]
        b, n, c = x.shape
        q, k, v = F.linear(x, self.in_proj_weight, self.in_proj_bias).reshape(b, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        return self.out_proj(o.transpose(1, 2).reshape(b, n, c))


class _Block(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _Attention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, 4 * width)), ("gelu", nn.GELU()),
                                              ("c_proj", nn.Linear(4 * width, width))]))

    def forward(self, x, causal=False):
        x = x + self.attn(self.ln_1(x), causal)
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_Block(width, heads) for _ in range(layers)])

    def forward(self, x, causal=False, skip_last=0):
        for blk in self.resblocks[:len(self.resblocks) - skip_last]:
            x = blk(x, causal)
        return x


class _Visual(nn.Module):
    def __init__(self, width=1280, layers=32, heads=16, patch=14, image=224, out_dim=1024):
        super().__init__()
        self.grid = image // patch
        self.conv1 = nn.Conv2d(3, width, patch, stride=patch, bias=False)
        self.class_embedding = nn.Parameter(width ** -0.5 * torch.randn(width))
        self.positional_embedding = nn.Parameter(width ** -0.5 * torch.randn(self.grid ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(width ** -0.5 * torch.randn(width, out_dim))


class _CLIP(nn.Module):
    """The attribute tree of open_clip's CLIP('ViT-H-14') as far as a state dict sees it."""

    def __init__(self, text=True, visual=True, embed_dim=1024, text_width=1024, text_layers=24, text_heads=16, vocab=49408, ctx=77,
                 vision_cfg=None):
        super().__init__()
        if visual:
            self.visual = _Visual(**(vision_cfg or {}), out_dim=embed_dim)
        if text:
            self.transformer = _Transformer(text_width, text_layers, text_heads)
        self.token_embedding = nn.Embedding(vocab, text_width)
        self.positional_embedding = nn.Parameter(torch.empty(ctx, text_width).normal_(std=0.01))
        self.ln_final = nn.LayerNorm(text_width)
        self.text_projection = nn.Parameter(torch.empty(text_width, embed_dim).normal_(std=text_width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592)
        self.register_buffer("attn_mask", torch.full((ctx, ctx), float("-inf")).triu_(1), persistent=False)


def _tokenize(texts):
    try:
        import open_clip
    except ImportError as e:
        raise RuntimeError("tokenising a prompt needs open_clip's BPE vocabulary (pip package `open_clip_torch`, the reference's "
                           "own dependency); pass pre-tokenised int64 [b, 77] tensors instead") from e
    return open_clip.tokenize(texts)


class FrozenOpenCLIPEmbedder(nn.Module):
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 model_cfg=None):
        super().__init__()
        assert layer in self.LAYERS
        if arch != "ViT-H-14" and model_cfg is None:
            raise NotImplementedError("only the ViT-H-14 configuration of the ViewCrafter yaml is built")
        self.model = _CLIP(text=True, visual=False, **(model_cfg or {}))
        self.device, self.max_length, self.layer = device, max_length, layer
        self.layer_idx = 0 if layer == "last" else 1
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, text):
        tokens = text if torch.is_tensor(text) else _tokenize(text)
        return self.encode_with_transformer(tokens.to(self.model.positional_embedding.device))

    def encode_with_transformer(self, text):
        x = self.model.token_embedding(text) + self.model.positional_embedding
        x = self.model.transformer(x, causal=True, skip_last=self.layer_idx)
        return self.model.ln_final(x)

    def encode(self, text):
        return self(text)


def _gaussian_blur(x, ks, sigma):
    """Separable Gaussian blur with reflect borders (kornia.filters.gaussian_blur2d semantics)."""
    def k1(n, s):
        t = torch.arange(n, dtype=torch.float32, device=x.device) - n // 2
        if n % 2 == 0:
            t = t + 0.5
        g = torch.exp(-t * t / (2.0 * s * s))
        return (g / g.sum()).to(x.dtype)
    c = x.shape[1]
    ky, kx = k1(ks[0], sigma[0]), k1(ks[1], sigma[1])
    x = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
    x = F.conv2d(x, kx.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
    return F.conv2d(x, ky.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)


def _resize_antialias(x, size):
    """kornia.geometry.resize(x, size, 'bicubic', align_corners=True, antialias=True) as kornia documents it: when
    down-scaling, a Gaussian pre-filter with sigma = max((factor - 1) / 2, 0.001), kernel size max(int(4 sigma), 3) made
    odd, then F.interpolate(bicubic, align_corners=True)."""
    h, w = x.shape[-2:]
    fy, fx = h / size[0], w / size[1]
    if max(fy, fx) > 1:
        sig = (max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * sig[0], 3)), int(max(2.0 * 2 * sig[1], 3))]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        x = _gaussian_blur(x, ks, sig)
    return F.interpolate(x, size=size, mode="bicubic", align_corners=True)


class FrozenOpenCLIPImageEmbedderV2(nn.Module):
    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled", antialias=True,
                 model_cfg=None):
        super().__init__()
        if arch != "ViT-H-14" and model_cfg is None:
            raise NotImplementedError("only the ViT-H-14 configuration of the ViewCrafter yaml is built")
        if layer == "penultimate":
            raise NotImplementedError()
        self.model = _CLIP(text=False, visual=True, **(model_cfg or {}))
        self.device, self.layer, self.antialias = device, layer, antialias
        if freeze:
            self.freeze()
        self.register_buffer("mean", torch.Tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer("std", torch.Tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)

    def freeze(self):
        self.model = self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    def preprocess(self, x):
        size = (self.model.visual.grid * self.model.visual.conv1.kernel_size[0],) * 2
        x = _resize_antialias(x, size) if self.antialias else F.interpolate(x, size=size, mode="bicubic", align_corners=True)
        x = (x + 1.) / 2.
        return (x - self.mean.to(x)[None, :, None, None]) / self.std.to(x)[None, :, None, None]

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    def encode_with_vision_transformer(self, x):
        v = self.model.visual
        x = self.preprocess(x).to(v.conv1.weight.dtype)
        x = v.conv1(x).flatten(2).transpose(1, 2)                                   # [b, grid^2, width]
        cls = v.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + v.positional_embedding.to(x.dtype)
        x = v.ln_pre(x)
        return v.transformer(x)
