"""3D (2D + temporal) denoising U-Net of ViewCrafter, rebuilt.

Operator boundary kept: `UNetModel(**yaml_params).forward(x[b,8,T,h,w], timesteps[b], context[b,L,1024], fs[b])
-> [b,4,T,h,w]` (lvdm/modules/networks/openaimodel3d.py:548-603), and the parameter tree carries the
reference's state-dict key names (`input_blocks.4.0.in_layers.2.weight`, `...temopral_conv.conv3.3.weight`,
`...transformer_blocks.0.attn2.to_k_ip.weight`, ...) so the ViewCrafter checkpoint loads with strict=True.

Layout (the MI355X-first part).  Every feature map lives in ONE memory layout for the whole network:
token-major  [frame, y, x, channel]  (= torch channels_last for the 4-D [(b t), C, H, W] view):

  * 2-D convolutions consume / produce it directly (NHWC implicit-GEMM kernels, no NCHW<->NHWC transposes);
  * a spatial transformer's tokens [(b t), h*w, C] are the same bytes -- no 'b c h w -> b (h w) c' copies;
  * the temporal (3,1,1) convolution is three [T*h*w, C] x [C, C] GEMMs on frame-shifted slices of the same
    bytes (see TemporalConvBlock);
  * temporal self-attention (sequence = the T frames of one pixel) reads its q/k/v in place with the strided
    MFMA attention kernel (`frame_major=True`) instead of materialising '(b h w) t c' tensors;
  * GroupNorm runs the channels-last kernel on whatever token range the statistics span
    (one frame for 2-D norms, all T frames for the temporal ones).
The reference performs 4-6 full-tensor rearrange copies per transformer / temporal block.

Also: the 77 text + 256 image context tokens are projected ONCE per layer for all T frames when the context
is frame-invariant (the reference projects 25 identical copies), and activation checkpointing is
off by default (288 GB of HBM; `use_checkpoint=True` restores it).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint as _ckpt

from . import conv as mconv, gemm, ops, parallel
from .schedule import timestep_embedding


def _fused(t, module=None):
    """The hand-written MFMA convolution path (csrc/conv_mfma.hip) applies: fp16 activations on a ROCm device, inference
    (dropout inactive).  Everything else -- fp32 parity runs, the CPU reference form -- takes the module-by-module form below
    (torch convolutions)."""
    ok = t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) and not (module is not None and module.training)
    if t.is_cuda and not ok:   # never silent: a device tensor is about to meet torch's convolutions instead of the MFMA kernel
        ops._torch_form("convolution", f"dtype {t.dtype}{' in training mode' if (module is not None and module.training) else ''}")
    return ok


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


def _cl(x):
    """4-D [(b t), C, H, W] tensor in token-major (channels_last) memory."""
    return x.contiguous(memory_format=torch.channels_last)


def _tok(x):
    """[(b t), C, H, W] channels_last  ->  [(b t), H, W, C] contiguous VIEW (a copy only if x was not channels_last)."""
    return x.permute(0, 2, 3, 1).contiguous()


def _img(tok):
    """[(b t), H, W, C] contiguous -> [(b t), C, H, W] channels_last view."""
    return tok.permute(0, 3, 1, 2)


def _cell(t):
    """A gradient hand-over cell (ops.GradCell) for a tensor that feeds a norm AND a later `+ t`, when a gradient will flow."""
    return ops.GradCell() if (torch.is_grad_enabled() and t.requires_grad) else None


def _gn_tokens(gn, tok, n_stat, silu=False, shard=None, tokens_total=None, grad_add=None):
    """GroupNorm over token-major data; statistics span tok.numel() / (n_stat * C) tokens per group -- completed
    across the frame-shard group when `shard` is given (tokens_total = tokens per sample over all ranks)."""
    C = tok.shape[-1]
    y = ops.group_norm(tok.reshape(n_stat, -1, C), gn.num_groups, gn.weight, gn.bias, gn.eps, silu=silu, channels_last=True,
                       group=None if shard is None else shard.group, S_total=tokens_total, grad_add=grad_add)
    return y.reshape(tok.shape)


class GroupNorm32(nn.GroupNorm):
    """fp32-statistics GroupNorm of a 4-D feature map (lvdm/basics.py:76-86); `silu=True` fuses the activation."""

    def forward(self, x, silu=False):
        tok = _tok(x)
        return _img(_gn_tokens(self, tok, tok.shape[0], silu))


def _run(fn, use_checkpoint, *args):
    return _ckpt(fn, *args, use_reentrant=False) if (use_checkpoint and torch.is_grad_enabled()) else fn(*args)


# ------------------------------------------------------------------------------------------------
# attention side (lvdm/modules/attention.py)
# ------------------------------------------------------------------------------------------------
class CrossAttention(nn.Module):
    """attention.py:42-144.  Self-attention when context is None; with `image_cross_attention` the
    context is [text 77 | image tokens] and the two softmaxes are separate and summed."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0., image_cross_attention=False,
                 image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False, text_context_len=77):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale = image_cross_attention_scale
        self.image_cross_attention_scale_learnable = image_cross_attention_scale_learnable
        self.text_context_len = text_context_len
        if image_cross_attention:
            self.to_k_ip = nn.Linear(context_dim, inner, bias=False)
            self.to_v_ip = nn.Linear(context_dim, inner, bias=False)
            if image_cross_attention_scale_learnable:
                self.register_parameter("alpha", nn.Parameter(torch.tensor(0.)))

    def _kv(self, context, frames):
        """K / V (and image-prompt K / V) of the context, each pair as ONE GEMM ([k | v] column blocks, read in place by the
        attention kernel).  `frames` > 1 means `context` holds ONE copy of a context shared by `frames` consecutive batch rows:
        projected once and never expanded -- with a single context row the attention kernel broadcasts it (batch stride 0); with
        one row per sample (the batch-2 CFG pair) forward() folds the frames of a sample into the query dimension instead."""
        C = self.to_k.weight.shape[0]
        ctx_t = context[:, :self.text_context_len]
        kv = gemm.linear_cat(ctx_t, [self.to_k.weight, self.to_v.weight])
        kv_ip = None
        if self.image_cross_attention:
            kv_ip = gemm.linear_cat(context[:, self.text_context_len:], [self.to_k_ip.weight, self.to_v_ip.weight])
        split = lambda t: (None, None) if t is None else (t[..., :C], t[..., C:])
        return split(kv) + split(kv_ip)

    def forward(self, x, context=None, shared_frames=1, frame_major=False, norm=None, residual=None):
        """norm: the LayerNorm in front of this attention (attention.py:283-285) -- folded into the q / k / v GEMMs instead of
        being applied; residual: the block's `+ x`, added in to_out's epilogue.  Under autograd the gradient of that `+ x` is
        handed to the q / k / v projection's backward through a GradCell and summed inside its LayerNorm-backward kernel."""
        C = self.to_q.weight.shape[0]
        cell = _cell(x) if residual is x else None
        if context is None:
            qkv = gemm.linear_cat(x, [self.to_q.weight, self.to_k.weight, self.to_v.weight], ln=norm, grad_add=cell)   # one launch, LayerNorm folded
            # q | k | v are read in place as column blocks (and, under autograd, their gradients written in place)
            if frame_major:  # x [b, T, pixels, C]: one T-long sequence per pixel, read in place
                out = ops.self_attention_packed(qkv, self.heads, frame_major=True)   # [b, T, pixels, 3 C]: samples looped inside the op
            else:
                out = ops.self_attention_packed(qkv, self.heads)
        else:
            q = gemm.linear(x, self.to_q.weight, ln=norm, grad_add=cell)
            k, v, k_ip, v_ip = self._kv(context, shared_frames)
            q_shape = q.shape
            if shared_frames > 1 and k.shape[0] > 1 and os.environ.get("GVD_XATTN_EXPAND", "0") == "1":   # (A/B: K / V copied per frame, as before)
                rep = lambda t_: None if t_ is None else t_.repeat_interleave(shared_frames, dim=0)
                k, v, k_ip, v_ip = rep(k), rep(v), rep(k_ip), rep(v_ip)
            elif shared_frames > 1 and k.shape[0] > 1:
                # one context per SAMPLE, shared by its `shared_frames` consecutive rows of q: attention is independent per query
                # row, so the frames of a sample are just more queries of ONE batch entry -- a view of q and of the output instead
                # of K / V copied `shared_frames` times (repeat_interleave + copy: 1.4 ms per guided step at 320x448)
                if q.shape[0] != k.shape[0] * shared_frames:
                    raise RuntimeError(f"CrossAttention: {q.shape[0]} query rows for {k.shape[0]} contexts x {shared_frames} frames")
                q = q.reshape(k.shape[0], shared_frames * q.shape[1], q.shape[2])
            out = ops.attention(q, k, v, self.heads)
            if k_ip is not None:
                s = self.image_cross_attention_scale
                if self.image_cross_attention_scale_learnable:
                    out = out + s * ops.attention(q, k_ip, v_ip, self.heads) * (torch.tanh(self.alpha) + 1)
                else:   # out + s * out_ip in the second attention's epilogue
                    out = ops.attention(q, k_ip, v_ip, self.heads, accum=out, accum_scale=float(s))
            out = out.reshape(q_shape)
        lo, drop = self.to_out[0], self.to_out[1]
        if drop.training and drop.p > 0:
            # the reference drops the projection, not the residual stream: dropout(linear(out)) [+ x] (attention.py:144, :241-244);
            # with or without a residual (advisor finding, round 4: the residual-less call skipped the dropout)
            y = drop(gemm.linear(out, lo.weight, lo.bias))
            return y if residual is None else y + residual
        return gemm.linear(out, lo.weight, lo.bias, residual=residual, res_grad_to=cell)   # eval / p = 0: dropout is the identity, `+ x` in the epilogue


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state-dict keys) routed through the HIP row kernel on 16-bit activations."""

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x, norm=None, grad_add=None):
        return gemm.linear(x, self.proj.weight, self.proj.bias, ln=norm, geglu=True, grad_add=grad_add)   # LayerNorm fold + gate in ONE launch


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim))

    def forward(self, x, norm=None, residual=None):
        """norm: the LayerNorm in front (folded into the projection); residual: the block's `+ x` (second GEMM's epilogue)."""
        drop = self.net[1]
        if not (drop.training and drop.p > 0):
            # guided sampler (autograd, frozen weights): projection + gate + output projection and their backward without the gate row kernels
            y = gemm.feed_forward(x, self.net[0].proj.weight, self.net[0].proj.bias, self.net[2].weight, self.net[2].bias, ln=norm, residual=residual)
            if y is not None:
                return y
        cell = _cell(x) if residual is x else None
        h = self.net[1](self.net[0](x, norm=norm, grad_add=cell))
        return gemm.linear(h, self.net[2].weight, self.net[2].bias, residual=residual, res_grad_to=cell)


class BasicTransformerBlock(nn.Module):
    """attention.py:212-246: x += attn1(LN x); x += attn2(LN x, context); x += FF(LN x)."""

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, use_checkpoint=False, **attn2_kw):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head, dropout)
        self.ff = FeedForward(dim, dropout=dropout)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head, dropout, **attn2_kw)
        self.norm1, self.norm2, self.norm3 = LayerNorm(dim), LayerNorm(dim), LayerNorm(dim)
        self.use_checkpoint = use_checkpoint

    def _fwd(self, x, context, shared_frames, frame_major):
        # every LayerNorm is folded into the GEMM behind it and every `+ x` into the GEMM in front of it (gemm.py): a block is
        # 7 GEMM + 3 (4) attention + 3 row-statistics launches, no normalised / gated / summed intermediate tensor
        x = self.attn1(x, frame_major=frame_major, norm=self.norm1, residual=x)
        x = self.attn2(x, context, shared_frames, frame_major=frame_major, norm=self.norm2, residual=x)
        return self.ff(x, norm=self.norm3, residual=x)

    def forward(self, x, context=None, shared_frames=1, frame_major=False):
        return _run(lambda a, c: self._fwd(a, c, shared_frames, frame_major), self.use_checkpoint, x, context)


class SpatialTransformer(nn.Module):
    """attention.py:249-310 (use_linear=True: GN -> tokens -> Linear in -> blocks -> Linear out -> + x)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, use_checkpoint=False,
                 use_linear=True, image_cross_attention=False, image_cross_attention_scale_learnable=False):
        super().__init__()
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.use_linear = use_linear
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout, context_dim, use_checkpoint,
                                  image_cross_attention=image_cross_attention,
                                  image_cross_attention_scale_learnable=image_cross_attention_scale_learnable)
            for _ in range(depth)])
        self.proj_out = zero_module(nn.Linear(inner, in_channels) if use_linear else nn.Conv2d(inner, in_channels, 1))

    def forward(self, x, context=None, shared_frames=1):
        n, c, h, w = x.shape
        tok = _tok(x)
        cell = _cell(tok)     # tok feeds the norm and the closing `+ x_in`: one fused gradient sum (ops.GradCell)
        t = _gn_tokens(self.norm, tok, n, grad_add=cell).reshape(n, h * w, c)
        t = gemm.linear(t, self.proj_in.weight, self.proj_in.bias)     # (a 1x1 Conv2d is a Linear with weight [out, in, 1, 1])
        for blk in self.transformer_blocks:
            t = blk(t, context, shared_frames)
        t = gemm.linear(t, self.proj_out.weight, self.proj_out.bias, residual=tok.reshape(n, h * w, c), res_grad_to=cell)   # `+ x_in` in the epilogue
        return _img(t.reshape(n, h, w, c))


class TemporalTransformer(nn.Module):
    """attention.py:313-412, only_self_att=True, no relative position, no causal mask (ViewCrafter yaml).
    One sequence of T frames per pixel; both attn1 and attn2 are self-attention.  Works on the token-major
    [b, T, h*w, C] view: per-token ops (GN affine, LayerNorm, Linear, FF) are order-agnostic, and the attention
    kernel reads the frames of a pixel with stride h*w*C."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., use_checkpoint=False, use_linear=False):
        super().__init__()
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.use_linear = use_linear
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv1d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout, None, use_checkpoint) for _ in range(depth)])
        self.proj_out = zero_module(nn.Linear(inner, in_channels) if use_linear else nn.Conv1d(inner, in_channels, 1))

    def forward(self, x, batch_size):  # x [(b T), C, H, W]
        bt, c, h, w = x.shape
        b, T = batch_size, bt // batch_size
        tok = _tok(x)
        shard = parallel.active()
        cell = _cell(tok) if shard is None else None
        if shard is None:
            t = _gn_tokens(self.norm, tok, b, grad_add=cell).reshape(b, T, h * w, c)  # statistics over (C/32, T, h, w) per sample
        else:  # frames are sharded: finish the statistics across the group, then re-shard frames -> pixels
            t = _gn_tokens(self.norm, tok, b, shard=shard, tokens_total=shard.T * h * w)
            t = parallel.frames_to_pixels(t.reshape(T, h * w, c), shard)[None]  # [1, all T, this rank's pixels, c]
        t = gemm.linear(t, self.proj_in.weight, self.proj_in.bias)     # (Conv1d(k = 1) == per-token Linear with weight [out, in, 1])
        for blk in self.transformer_blocks:
            t = blk(t, frame_major=True)
        if shard is not None:
            t = gemm.linear(t, self.proj_out.weight, self.proj_out.bias)
            t = parallel.pixels_to_frames(t[0], shard, h * w)
            return _img(t.reshape(bt, h, w, c) + tok)
        t = gemm.linear(t, self.proj_out.weight, self.proj_out.bias, residual=tok.reshape(t.shape[:-1] + (c,)), res_grad_to=cell)
        return _img(t.reshape(bt, h, w, c))


# ------------------------------------------------------------------------------------------------
# convolutional side (openaimodel3d.py)
# ------------------------------------------------------------------------------------------------
class TemporalConvBlock(nn.Module):
    """openaimodel3d.py:239-279: 4 x [GN32 -> SiLU -> (Dropout) -> Conv3d k=(3,1,1) pad (1,0,0)] + identity.

    The (3,1,1) convolution mixes channels over three neighbouring FRAMES of one pixel; on MI355X it is run
    as what it is -- three [T*h*w, C] x [C, C] GEMMs on the token-major [b, T, h*w, C] view, the t-1 / t+1 taps
    accumulated in the GEMM epilogue (addmm_, beta = 1) on frame-shifted slices.  No im2col, no 3-D
    convolution library call (MIOpen's grouped-conv kernel ran these at ~47 TFLOP/s: 40 % of a U-Net forward)."""

    def __init__(self, channels, dropout=0.0):
        super().__init__()
        conv = lambda: nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0))
        self.conv1 = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), conv())
        self.conv2 = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Dropout(dropout), conv())
        self.conv3 = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Dropout(dropout), conv())
        self.conv4 = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Dropout(dropout), conv())
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)
        self._taps = {}

    def _tap_weights(self, conv):
        """[3, Cout, Cin] contiguous per-frame-tap matrices, cached until the parameter changes."""
        w = conv.weight
        if w.requires_grad:
            return w[:, :, :, 0, 0].permute(2, 0, 1).contiguous()
        hit = self._taps.get(id(conv))
        if hit is None or hit[0] != w._version or hit[1].dtype != w.dtype or hit[1].device != w.device:
            self._taps[id(conv)] = hit = (w._version, w[:, :, :, 0, 0].permute(2, 0, 1).contiguous())
        return hit[1]

    @staticmethod
    def _temporal_gemm(h, taps, bias):
        """h [b, t, n, c] -> [b, t, n, co]:  out[t] = W1 h[t] + W0 h[t-1] + W2 h[t+1] + bias (zero padding in t)."""
        b, t, n, c = h.shape
        co = taps.shape[1]
        out = F.linear(h, taps[1], bias)
        if t > 1:
            od = out.dtype  # under autocast F.linear returns fp16 while the in-place addmm_ is not autocast-wrapped
            hs, w0, w2 = h.to(od), taps[0].to(od).t(), taps[2].to(od).t()
            for bi in range(b):
                out[bi, 1:].reshape(-1, co).addmm_(hs[bi, :-1].reshape(-1, c), w0)
                out[bi, :-1].reshape(-1, co).addmm_(hs[bi, 1:].reshape(-1, c), w2)
        return out

    def _forward_tokens_fused(self, tok, b, stats=None):
        """Four launches of the temporal MFMA kernel (all samples of the batch in each): GroupNorm+SiLU in the operand load, the identity add in the
        last epilogue, and each convolution leaves the statistics its successor's norm needs (openaimodel3d.py:270-278).
        Frame-sharded (parallel.py): the block works on [all T, this rank's pixels] between one all-to-all each way, and
        the per-video norm statistics are completed across the shard group by a 2 G-double all-reduce per norm."""
        bt, hh, ww, c = tok.shape
        shard = parallel.active()
        if shard is not None:
            x_all = parallel.frames_to_pixels(tok.reshape(bt, hh * ww, c), shard)      # [T, local pixels, c]
            group, total = shard.group, shard.T * hh * ww
            samples = [x_all]
        else:
            # all samples of the batch in ONE launch per convolution ([b, T, pixels, c]: the samples are extra pixel tiles; the 5-D
            # norms and the statistics stay per sample) -- the batch-2 CFG pair at 320x448 launches 140-1100 workgroups per
            # convolution instead of twice 70-560, and no per-sample split / concatenation nodes exist under autograd
            T = bt // b
            group = total = None
            samples = [tok.reshape(b, T, hh * ww, c) if b > 1 else tok.reshape(T, hh * ww, c)]
        outs = []
        for x0 in samples:
            h, part = x0, (stats if (shard is None or b == 1) else None)
            cell = _cell(x0)          # x0 feeds conv1's norm and the closing identity add
            seqs = (self.conv1, self.conv2, self.conv3, self.conv4)
            n_stat = b if (shard is None and b > 1) else 1
            for k, seq in enumerate(seqs):
                gn, conv = seq[0], seq[-1]
                if part is None:
                    ns = mconv.norm_state(gn, x=h.detach(), n_stat=n_stat, group=group, S_total=total)
                else:  # per-frame sums of a 2-D producer merge into the per-video statistics of this 5-D norm
                    ns = mconv.norm_state(gn, partial=part, merge=part.N // n_stat, group=group, S_total=total)
                last = k == len(seqs) - 1
                h, part = mconv.fused_conv(h, conv, mode=mconv.TEMPORAL, gn=gn, norm=ns, silu=True,
                                           residual=x0 if last else None, stats_groups=0 if last else gn.num_groups,
                                           grad_add=cell if k == 0 else None, res_grad_to=cell if last else None)
            outs.append(h)
        if shard is not None:
            return parallel.pixels_to_frames(outs[0], shard, hh * ww).reshape(bt, hh, ww, c)
        return outs[0].reshape(bt, hh, ww, c)

    def forward_tokens(self, tok, b, stats=None):  # tok [(b t), H, W, C] contiguous
        if _fused(tok, self):
            return self._forward_tokens_fused(tok, b, stats)
        bt, hh, ww, c = tok.shape
        shard = parallel.active()
        if shard is None:
            h, total = tok.reshape(b, bt // b, hh * ww, c), None
        else:  # the frame taps need t-1 / t+1 of a pixel: work on [all T, this rank's pixels] for the whole block
            h, total = parallel.frames_to_pixels(tok.reshape(bt, hh * ww, c), shard)[None], shard.T * hh * ww
        for seq in (self.conv1, self.conv2, self.conv3, self.conv4):
            gn, conv = seq[0], seq[-1]
            h = _gn_tokens(gn, h, b, silu=True, shard=shard, tokens_total=total)
            h = self._temporal_gemm(h, self._tap_weights(conv), conv.bias)
        if shard is not None:
            h = parallel.pixels_to_frames(h[0], shard, hh * ww)
        return tok + h.reshape(bt, hh, ww, c)

    def forward(self, x):  # reference signature: [b, c, t, h, w]
        b, c, t, hh, ww = x.shape
        tok = x.permute(0, 2, 3, 4, 1).reshape(b * t, hh, ww, c)
        return self.forward_tokens(tok, b).reshape(b, t, hh, ww, c).permute(0, 4, 1, 2, 3)


class ResBlock(nn.Module):
    """openaimodel3d.py:109-236 (no scale-shift norm, no up/down variants: the yaml uses neither)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_checkpoint=False, use_temporal_conv=False):
        super().__init__()
        out_channels = out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(out_channels, out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)
        self.use_temporal_conv = use_temporal_conv
        if use_temporal_conv:
            self.temopral_conv = TemporalConvBlock(out_channels, dropout=0.1)  # (sic) reference attribute name

    def _fwd(self, x, emb, batch_size):
        h = self.in_layers[2](self.in_layers[0](x, silu=True))
        h = h + gemm.linear(F.silu(emb), self.emb_layers[1].weight, self.emb_layers[1].bias).to(h.dtype)[:, :, None, None]
        h = self.out_layers[3](self.out_layers[2](self.out_layers[0](h, silu=True)))
        h = self.skip_connection(x) + h
        if self.use_temporal_conv and batch_size:
            h = _img(self.temopral_conv.forward_tokens(_tok(h), batch_size))
        return h

    def _fwd_fused(self, x, emb, batch_size, norm1=None):
        """Two launches of the MFMA convolution (+ the 1x1 skip GEMM when channels change): both GroupNorm+SiLU pairs live in
        the operand loads, `+ emb_out`, `+ skip` and the 16-bit rounding in the epilogues; the first convolution leaves the
        statistics of its output for the second norm, the second for the temporal block's first norm."""
        tok = _tok(x)
        gn1, conv1 = self.in_layers[0], self.in_layers[2]
        gn2, conv2 = self.out_layers[0], self.out_layers[3]
        emb_out = gemm.linear(F.silu(emb), self.emb_layers[1].weight, self.emb_layers[1].bias).to(tok.dtype)
        # tok feeds conv1's norm and the skip path: the skip side hands its gradient to conv1's backward (ops.GradCell), which sums
        # it inside the GroupNorm-backward apply kernel -- no accumulation kernel where the two branches meet
        cell = _cell(tok)
        # norm1: the state of gn1 when the caller already has it (the decoder's fused skip concatenation, UNetModel.forward)
        h, part = mconv.fused_conv(tok, conv1, gn=gn1, norm=norm1, silu=True, add_nc=emb_out, stats_groups=gn2.num_groups, grad_add=cell)
        ns2 = mconv.norm_state(gn2, partial=part)
        identity = isinstance(self.skip_connection, nn.Identity)
        if identity:
            skip = tok
        else:  # 1x1 convolution == per-token GEMM on the same bytes
            sc = self.skip_connection
            skip = gemm.linear(tok, sc.weight, sc.bias, grad_to=cell)
        temporal = self.use_temporal_conv and batch_size
        h, part = mconv.fused_conv(h, conv2, gn=gn2, norm=ns2, silu=True, residual=skip, res_grad_to=cell if identity else None,
                                   stats_groups=self.temopral_conv.conv1[0].num_groups if temporal else 0)
        if temporal:
            h = self.temopral_conv.forward_tokens(h, batch_size, stats=part)
        return _img(h)

    def forward(self, x, emb, batch_size=None, norm1=None):
        if _fused(x, self):
            return _run(lambda a, e: self._fwd_fused(a, e, batch_size, norm1), self.use_checkpoint, x, emb)
        return _run(lambda a, e: self._fwd(a, e, batch_size), self.use_checkpoint, x, emb)


class Downsample(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.op = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=1)

    def forward(self, x):
        if _fused(x, self):   # stride-2 form of the MFMA convolution (forward) / zero-stuffed form (input gradient)
            return _img(mconv.fused_conv(_tok(x), self.op, mode=mconv.STRIDE2)[0])
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x):
        if _fused(x, self):  # the x2 nearest upsampling happens in the kernel's patch addressing: no 4x-sized intermediate
            return _img(mconv.fused_conv(_tok(x), self.conv, upsample=True)[0])
        return self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))


class TimestepEmbedSequential(nn.Sequential):
    """openaimodel3d.py:30-48: routes (emb | context | batch size) to the children that take them."""

    def forward(self, x, emb, context=None, batch_size=None, shared_frames=1, norm1=None):
        for i, layer in enumerate(self):
            if isinstance(layer, ResBlock):
                x = layer(x, emb, batch_size=batch_size, norm1=norm1 if i == 0 else None)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context, shared_frames)
            elif isinstance(layer, TemporalTransformer):
                x = layer(x, batch_size)
            elif isinstance(layer, nn.Conv2d) and layer.kernel_size == (3, 3) and layer.stride == (1, 1) and _fused(x, layer):
                x = _img(mconv.fused_conv(_tok(x), layer)[0])
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    """openaimodel3d.py:281-603.  Accepts the reference's constructor keywords (yaml `unet_config.params`);
    options the shipped ViewCrafter config does not use raise instead of being silently ignored."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None, use_scale_shift_norm=False,
                 resblock_updown=False, num_heads=-1, num_head_channels=-1, transformer_depth=1, use_linear=False,
                 use_checkpoint=False, temporal_conv=False, tempspatial_aware=False, temporal_attention=True,
                 use_relative_position=True, use_causal_attention=False, temporal_length=None, use_fp16=False,
                 addition_attention=False, temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False):
        super().__init__()
        if (dims != 2 or use_scale_shift_norm or resblock_updown or tempspatial_aware or use_relative_position
                or use_causal_attention or not temporal_selfatt_only or not conv_resample or num_head_channels == -1):
            raise NotImplementedError("UNetModel (MI355X build) covers the ViewCrafter configuration "
                                      "(configs/inference_pvd_1024.yaml:33-64) only")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.temporal_attention, self.addition_attention = temporal_attention, addition_attention
        self.default_fs, self.fs_condition = default_fs, fs_condition
        self.dtype = torch.float16 if use_fp16 else torch.float32
        ted = model_channels * 4
        mlp = lambda: nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.time_embed = mlp()
        if fs_condition:
            self.fps_embedding = mlp()
            nn.init.zeros_(self.fps_embedding[-1].weight)
            nn.init.zeros_(self.fps_embedding[-1].bias)

        def res(cin, cout):
            return ResBlock(cin, ted, dropout, cout, use_checkpoint, temporal_conv)

        def attn_layers(ch):
            heads = ch // num_head_channels
            out = [SpatialTransformer(ch, heads, num_head_channels, transformer_depth, 0., context_dim, use_checkpoint,
                                      use_linear, image_cross_attention, image_cross_attention_scale_learnable)]
            if temporal_attention:
                out.append(TemporalTransformer(ch, heads, num_head_channels, transformer_depth, 0., use_checkpoint, use_linear))
            return out

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        if addition_attention:
            self.init_attn = TimestepEmbedSequential(
                TemporalTransformer(model_channels, 8, num_head_channels, transformer_depth, 0., use_checkpoint, use_linear=False))
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers += attn_layers(ch)
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, ch)))
                chans.append(ch)
                ds *= 2
        mid = [res(ch, ch)] + attn_layers(ch) + [res(ch, ch)]
        self.middle_block = TimestepEmbedSequential(*mid)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + chans.pop(), mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers += attn_layers(ch)
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(),
                                 zero_module(nn.Conv2d(model_channels, out_channels, 3, padding=1)))

    def to_token_major(self):
        """Store every Conv2d weight channels_last so the NHWC convolution kernels get their native filter layout
        (values unchanged; state_dict round-trips)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def forward(self, x, timesteps, context=None, features_adapter=None, fs=None, **kwargs):
        b, _, t, hh, ww = x.shape
        wdtype = self.input_blocks[0][0].weight.dtype  # fp32 weights (+autocast) or a model converted with .half()
        xin_dtype = x.dtype
        x = x.to(wdtype)
        context = context.to(wdtype)
        mlp = lambda seq, e: gemm.linear(F.silu(gemm.linear(e, seq[0].weight, seq[0].bias)), seq[2].weight, seq[2].bias)
        emb = mlp(self.time_embed, timestep_embedding(timesteps, self.model_channels).type(x.dtype))
        # Context routing (openaimodel3d.py:555-562): per-frame image tokens only when L == 77 + 16 T,
        # otherwise the SAME context for every frame -> keep ONE copy per batch row and let the
        # cross-attention project it once (shared_frames = T) instead of T identical copies.
        shared = 1
        shard = parallel.active()
        if shard is not None:  # x holds this rank's frames only (parallel.py); the context must be the shared kind
            if b != 1 or context.shape[1] == 77 + shard.T * 16:
                raise NotImplementedError("frame-sharded U-Net: batch 1 and a frame-independent context only")
            shared = t
        elif context.shape[1] == 77 + t * 16:
            ctx_text = context[:, :77].repeat_interleave(t, dim=0)
            ctx_img = context[:, 77:].reshape(b * t, 16, context.shape[-1])
            context = torch.cat([ctx_text, ctx_img], dim=1)
        else:
            shared = t
        emb = emb.repeat_interleave(t, dim=0)
        if self.fs_condition:
            if fs is None:
                fs = torch.full((b,), self.default_fs, dtype=torch.long, device=x.device)
            fs_emb = mlp(self.fps_embedding, timestep_embedding(fs, self.model_channels).type(x.dtype))
            emb = emb + fs_emb.repeat_interleave(t, dim=0)
        h = _cl(x.transpose(1, 2).reshape(b * t, -1, hh, ww))  # -> token-major for the whole network
        hs = []
        for i, module in enumerate(self.input_blocks):
            h = module(h, emb, context, b, shared)
            if i == 0 and self.addition_attention:
                h = self.init_attn(h, emb, context, b, shared)
            hs.append(h)
        h = self.middle_block(h, emb, context, b, shared)
        for module in self.output_blocks:
            skip = hs.pop()
            first = module[0]
            if (shard is None and isinstance(first, ResBlock) and not first.training and h.is_cuda and h.dtype in (torch.float16, torch.bfloat16)
                    and h.is_contiguous(memory_format=torch.channels_last) and skip.is_contiguous(memory_format=torch.channels_last)
                    and mconv.cat_with_stats_ok(_tok(h), _tok(skip), first.in_layers[0])):
                # the skip concatenation and the statistics of the ResBlock's first norm in one pass (inference; conv.cat_with_stats)
                cat, ns = mconv.cat_with_stats(_tok(h), _tok(skip), first.in_layers[0])
                h = module(_img(cat), emb, context, b, shared, norm1=ns)
            else:
                h = module(torch.cat([h, skip], dim=1), emb, context, b, shared)
        if _fused(h, self):
            y = _img(mconv.fused_conv(_tok(h), self.out[2], gn=self.out[0], silu=True)[0]).to(xin_dtype)
        else:
            y = self.out[2](self.out[0](h, silu=True)).to(xin_dtype)
        return y.reshape(b, t, -1, hh, ww).transpose(1, 2)
