"""Golden for the CLIP towers (round-2 verdict item 9): the REFERENCE'S OWN call sequences --
`FrozenOpenCLIPEmbedder.encode_with_transformer` / `text_transformer_forward` and
`FrozenOpenCLIPImageEmbedderV2.encode_with_vision_transformer` (third_party/ViewCrafter/lvdm/modules/encoders/condition.py:215-232,
345-372), imported from /root/reference in the build container -- driving a model whose blocks are built here from
`torch.nn.MultiheadAttention` exactly as open_clip's published `ResidualAttentionBlock` is (ln_1, attn = nn.MultiheadAttention,
ln_2, mlp = c_fc / gelu / c_proj; x + attn(ln_1 x, attn_mask); x + mlp(ln_2 x); sequence-first layout), with open_clip's parameter
tree.  `open_clip` and `kornia` themselves are not installed (no network): they are stubbed at import time, the embedder objects are
created without their `__init__` (which downloads the pretrained model), and the kornia resize in front of the vision tower is taken
out on both sides (`preprocess` = identity here; lvdm_amd's restatement of it stays unpinned, and says so).  What this pins: the
layer that is read (penultimate for text), the causal mask, the class-token / position / ln_pre sequence, "all tokens of the last
block, no ln_post, no projection", and the block arithmetic against torch's own multi-head attention.  Tiny widths; weights by
parameter NAME (fill_by_name), so only inputs and outputs are stored."""
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VC = "/root/reference/third_party/ViewCrafter"
sys.path.insert(0, VC)
for stub in ("cv2", "kornia", "open_clip"):
    sys.modules.setdefault(stub, types.ModuleType(stub))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fill_by_name import fill_by_name  # noqa: E402

import lvdm.modules.encoders.condition as cond  # noqa: E402  (the reference file)

TEXT = dict(width=64, layers=3, heads=4, vocab=96, ctx=77)
VIS = dict(width=96, layers=3, heads=4, patch=8, image=32)


class Block(nn.Module):
    """open_clip.transformer.ResidualAttentionBlock (no layer scale, GELU MLP x4), sequence-first."""

    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, 4 * width)), ("gelu", nn.GELU()), ("c_proj", nn.Linear(4 * width, width))]))

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    grad_checkpointing = False

    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([Block(width, heads) for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for r in self.resblocks:
            x = r(x, attn_mask=attn_mask)
        return x


class Visual(nn.Module):
    input_patchnorm = False

    def __init__(self, width, layers, heads, patch, image):
        super().__init__()
        g = image // patch
        self.grid_size, self.patch_size = (g, g), (patch, patch)
        self.conv1 = nn.Conv2d(3, width, patch, stride=patch, bias=False)
        self.class_embedding = nn.Parameter(torch.zeros(width))
        self.positional_embedding = nn.Parameter(torch.zeros(g * g + 1, width))
        self.patch_dropout = nn.Identity()
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.zeros(width, 32))


class TextModel(nn.Module):
    def __init__(self, width, layers, heads, vocab, ctx):
        super().__init__()
        self.transformer = Transformer(width, layers, heads)
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.zeros(ctx, width))
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.zeros(width, 32))
        self.logit_scale = nn.Parameter(torch.ones([]))
        self.register_buffer("attn_mask", torch.full((ctx, ctx), float("-inf")).triu_(1), persistent=False)


class VisionModel(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.visual = Visual(**kw)


def bare(cls):
    obj = cls.__new__(cls)
    nn.Module.__init__(obj)
    return obj


def main():
    out = {}
    g = torch.Generator().manual_seed(11)
    # ---- text tower through the reference's encode_with_transformer, both `layer` settings ----
    tm = fill_by_name(TextModel(**TEXT), std=0.08).eval()
    tokens = torch.randint(0, TEXT["vocab"], (2, TEXT["ctx"]), generator=g)
    out["text_tokens"] = tokens.numpy()
    out["text_keys"] = np.array(sorted(tm.state_dict().keys()))
    for layer, idx in (("last", 0), ("penultimate", 1)):
        emb = bare(cond.FrozenOpenCLIPEmbedder)
        emb.model, emb.layer, emb.layer_idx, emb.device, emb.max_length = tm, layer, idx, "cpu", 77
        with torch.no_grad():
            out[f"text_{layer}"] = emb.encode_with_transformer(tokens).numpy()
    # ---- vision tower through the reference's encode_with_vision_transformer (kornia preprocess taken out) ----
    vm = fill_by_name(VisionModel(**VIS), std=0.08).eval()
    img = torch.randn(2, 3, VIS["image"], VIS["image"], generator=g)
    out["vis_image"], out["vis_keys"] = img.numpy(), np.array(sorted(vm.state_dict().keys()))
    ve = bare(cond.FrozenOpenCLIPImageEmbedderV2)
    ve.model, ve.device, ve.layer, ve.antialias = vm, "cpu", "pooled", True
    ve.preprocess = lambda x: x
    with torch.no_grad():
        out["vis_tokens"] = ve.encode_with_vision_transformer(img).numpy()
    np.savez_compressed(os.path.join(HERE, "clip_ref.npz"), **out)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "clip_ref.npz")), "bytes")


if __name__ == "__main__":
    main()
