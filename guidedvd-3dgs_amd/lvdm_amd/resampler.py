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
runs on the flash kernel; every Linear is the MFMA GEMM of csrc/gemm_mfma.hip with the LayerNorm in front of it folded into its
epilogue and the residual add behind it fused (lvdm_amd/gemm.py) for 16-bit activations.
"""
import torch
import torch.nn as nn

from . import gemm, ops


class _LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters, HIP row kernel on 16-bit activations."""

    def forward(self, t):
        return ops.layer_norm(t, self.weight, self.bias, self.eps)


def _linear(n_in, n_out, bias=True):
    return nn.Linear(n_in, n_out, bias=bias)


class ImageProjModel(nn.Module):
    """One CLIP embedding -> `clip_extra_context_tokens` context tokens (keys: proj, norm)."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim, self.clip_extra_context_tokens = cross_attention_dim, clip_extra_context_tokens
        self.add_module("proj", _linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim))
        self.add_module("norm", _LayerNorm(cross_attention_dim))

    @ops.checkpoint_dtype_tower
    def forward(self, image_embeds):
        wide = gemm.linear(image_embeds.to(self.proj.weight.dtype), self.proj.weight, self.proj.bias)
        return self.norm(wide.view(-1, self.clip_extra_context_tokens, self.cross_attention_dim))


def FeedForward(dim, mult=4):
    """Sequential indices 0 (norm), 1 and 3 (bias-free Linears) carry the parameters; 2 is the GELU."""
    hidden = int(dim * mult)
    stack = [_LayerNorm(dim), _linear(dim, hidden, False), nn.GELU(), _linear(hidden, dim, False)]
    return nn.Sequential(*stack)


class PerceiverAttention(nn.Module):
    """Latent queries attend over [image tokens ; latents]; keys norm1, norm2, to_q, to_kv (k | v stacked), to_out."""

    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        width = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        for name, module in (("norm1", _LayerNorm(dim)), ("norm2", _LayerNorm(dim)), ("to_q", _linear(dim, width, False)),
                             ("to_kv", _linear(dim, 2 * width, False)), ("to_out", _linear(width, dim, False))):
            self.add_module(name, module)

    def forward(self, x, latents, residual=None):
        """x: image tokens [b, n1, D]; latents: query tokens [b, n2, D] -> [b, n2, D] (+ residual, in to_out's epilogue).  norm1 /
        norm2 are folded into the k | v and q projections (no normalised tensor is rounded to 16 bit in between)."""
        w = self.heads * self.dim_head
        kv = torch.cat([gemm.linear(x, self.to_kv.weight, ln=self.norm1), gemm.linear(latents, self.to_kv.weight, ln=self.norm2)], dim=1)
        q = gemm.linear(latents, self.to_q.weight, ln=self.norm2)
        return gemm.linear(ops.attention(q, kv[..., :w], kv[..., w:], self.heads), self.to_out.weight, residual=residual)


class Resampler(nn.Module):
    """`depth` x (PerceiverAttention, FeedForward) over `num_queries` learned tokens per frame (keys: latents, proj_in, proj_out,
    norm_out, layers.<i>.<0|1>.*)."""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, video_length=None):
        super().__init__()
        self.num_queries, self.video_length = num_queries, video_length
        n_tokens = num_queries * (video_length if video_length is not None else 1)
        self.latents = nn.Parameter(torch.randn(1, n_tokens, dim) * dim ** -0.5)
        self.proj_in, self.proj_out = _linear(embedding_dim, dim), _linear(dim, output_dim)
        self.norm_out = _LayerNorm(output_dim)
        self.layers = nn.ModuleList()
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads), FeedForward(dim, ff_mult)]))

    @ops.checkpoint_dtype_tower
    def forward(self, x):
        tokens = gemm.linear(x, self.proj_in.weight, self.proj_in.bias)
        state = self.latents.expand(tokens.shape[0], -1, -1).to(tokens.dtype)
        for attend, feed_forward in self.layers:
            state = attend(tokens, state, residual=state)
            hidden = feed_forward[2](gemm.linear(state, feed_forward[1].weight, ln=feed_forward[0]))
            state = gemm.linear(hidden, feed_forward[3].weight, residual=state)
        return self.norm_out(gemm.linear(state, self.proj_out.weight, self.proj_out.bias))      # [b, frames * queries, output_dim]
