"""Image-conditioning projector that feeds the U-Net's cross attention (once per video):
`Resampler` — a 4-layer perceiver that turns the CLIP ViT-H/14 token grid [b, 257, 1280] into 16 x 16 frame-wise query
tokens [b, 256, 1024] — and the single-token `ImageProjModel` variant.

    third_party/ViewCrafter/lvdm/modules/encoders/resampler.py:9-24     ImageProjModel
    .../resampler.py:27-35     FeedForward (LayerNorm, Linear, GELU, Linear; no biases)
    .../resampler.py:49-93     PerceiverAttention: queries = normed latents, keys/values = [normed image tokens ; latents]
    .../resampler.py:96-144    Resampler

Parameter names equal the reference's, so `image_proj_model.*` of a ViewCrafter checkpoint loads with strict=True.  The
reference scales q and k by d^-1/4 each before the product (fp16 range); the flash kernel keeps scores in fp32, so the
single d^-1/2 factor inside `ops.attention` is the same function.  Attention (12 heads x 64, 256 queries over 513 keys)
and the LayerNorms run on the HIP kernels of libgvd_diffusion.so for 16-bit activations; the Linear layers are hipBLASLt.
"""
import torch
import torch.nn as nn

from . import ops


class _LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class ImageProjModel(nn.Module):
    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = _LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        tokens = self.proj(image_embeds.to(self.proj.weight.dtype))
        return self.norm(tokens.reshape(-1, self.clip_extra_context_tokens, self.cross_attention_dim))


def FeedForward(dim, mult=4):
    inner = int(dim * mult)
    return nn.Sequential(_LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.dim_head, self.heads = dim_head, heads
        inner = dim_head * heads
        self.norm1 = _LayerNorm(dim)
        self.norm2 = _LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, latents):
        """x: image tokens [b, n1, D]; latents: query tokens [b, n2, D] -> [b, n2, D]."""
        x = self.norm1(x)
        latents = self.norm2(latents)
        q = self.to_q(latents)
        k, v = self.to_kv(torch.cat((x, latents), dim=-2)).chunk(2, dim=-1)
        return self.to_out(ops.attention(q, k, v, self.heads))


class Resampler(nn.Module):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, video_length=None):
        super().__init__()
        self.num_queries = num_queries            # per frame
        self.video_length = video_length
        total = num_queries * video_length if video_length is not None else num_queries
        self.latents = nn.Parameter(torch.randn(1, total, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = _LayerNorm(output_dim)
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])

    def forward(self, x):
        latents = self.latents.expand(x.size(0), -1, -1)
        x = self.proj_in(x)
        for attn, ff in self.layers:
            latents = attn(x, latents) + latents
            latents = ff(latents) + latents
        return self.norm_out(self.proj_out(latents))      # b, (frames * queries), output_dim
