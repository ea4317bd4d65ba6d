"""KL-VAE decoder used inside the guided DDIM step.

Rebuild of `AutoencoderKL.decode` = post_quant_conv (1x1) -> Decoder (lvdm/models/autoencoder.py:104-107,
lvdm/modules/networks/ae_modules.py:466-578, AttnBlock :26-78, ResnetBlock :151-210) with the reference's
parameter names (`decoder.mid.attn_1.q.weight`, `decoder.up.3.block.0.nin_shortcut.weight`, ...) so the
`first_stage_model.*` part of the ViewCrafter checkpoint loads strict.  eps of every GroupNorm is 1e-6.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv as mconv, gemm, ops


def _fused(t, module=None):
    """fp16 activations on a ROCm device in inference mode take the hand-written MFMA convolutions (csrc/conv_mfma.hip) on
    token-major data; fp32 parity runs and the CPU reference form take the torch convolutions."""
    ok = t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) and not (module is not None and module.training)
    if t.is_cuda and not ok:   # never silent: a device tensor is about to meet torch's convolutions instead of the MFMA kernel
        ops._torch_form("convolution", f"dtype {t.dtype}{' in training mode' if (module is not None and module.training) else ''}")
    return ok


def _tok(x):
    return x.permute(0, 2, 3, 1).contiguous()   # a view when x is channels_last


def _img(tok):
    return tok.permute(0, 3, 1, 2)


def _norm(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


def _gn(gn, x, silu):
    """GroupNorm(+swish) of a 4-D feature map in whichever memory format it is in: a channels_last tensor is
    normalised through its [N, H, W, C] view by the token-major kernels (no NCHW round trip)."""
    if x.dim() == 4 and x.shape[1] % 8 == 0 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last):
        y = ops.group_norm(x.permute(0, 2, 3, 1), 32, gn.weight, gn.bias, gn.eps, silu=silu, channels_last=True)
        return y.permute(0, 3, 1, 2)
    return ops.group_norm(x, 32, gn.weight, gn.bias, gn.eps, silu=silu)


def _gn_swish(gn, x):
    return _gn(gn, x, True)  # x*sigmoid(x) == SiLU


class ResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.norm1 = _norm(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = _norm(out_channels)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1)

    def forward_tokens(self, tok, stats=None, want_stats=False):
        """tok [N, H, W, C] -> (out tokens, statistics of out for the next norm | None).  Two MFMA convolution launches:
        norm + swish in the operand load, the shortcut add in the second epilogue (ae_modules.py:186-210)."""
        ns1 = mconv.norm_state(self.norm1, partial=stats) if stats is not None else None
        # tok feeds norm1 and the shortcut: the shortcut side's gradient is summed inside conv1's GroupNorm-backward kernel (ops.GradCell)
        cell = ops.GradCell() if (torch.is_grad_enabled() and tok.requires_grad) else None
        h, part = mconv.fused_conv(tok, self.conv1, gn=self.norm1, norm=ns1, silu=True, stats_groups=32, grad_add=cell)
        ns2 = mconv.norm_state(self.norm2, partial=part)
        nin = hasattr(self, "nin_shortcut")
        if nin:
            skip = gemm.linear(tok, self.nin_shortcut.weight, self.nin_shortcut.bias, grad_to=cell)
        else:
            skip = tok
        return mconv.fused_conv(h, self.conv2, gn=self.norm2, norm=ns2, silu=True, residual=skip, res_grad_to=None if nin else cell,
                                stats_groups=32 if want_stats else 0)

    def forward(self, x):
        if _fused(x, self):
            return _img(self.forward_tokens(_tok(x))[0])
        h = self.conv1(_gn_swish(self.norm1, x))
        h = self.conv2(_gn_swish(self.norm2, h))
        return (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x) + h


class AttnBlock(nn.Module):
    """Single-head attention over h*w tokens with c channels (ae_modules.py:26-78); q/k/v/proj are 1x1 convs."""

    def __init__(self, c):
        super().__init__()
        self.norm = _norm(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward_tokens(self, tok):
        """tok [N, H, W, C]: norm -> q/k/v (1x1 convolutions == per-token GEMMs) -> single-head attention -> proj_out -> + x."""
        n, h, w, c = tok.shape
        t = tok.reshape(n, h * w, c)
        cell = ops.GradCell() if (torch.is_grad_enabled() and t.requires_grad) else None
        hn = ops.group_norm(t, 32, self.norm.weight, self.norm.bias, self.norm.eps, silu=False, channels_last=True, grad_add=cell)
        # q | k | v as ONE MFMA GEMM (column blocks read in place), the wide single-head attention (ops.attention -> flash kernel
        # at d = 64, chunked GEMMs at d = 512: wide_attention.py), proj_out with the `+ x` in its epilogue
        qkv = gemm.linear_cat(hn, [self.q.weight, self.k.weight, self.v.weight], [self.q.bias, self.k.bias, self.v.bias])
        o = ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], heads=1)
        return gemm.linear(o, self.proj_out.weight, self.proj_out.bias, residual=t, res_grad_to=cell).reshape(n, h, w, c)

    def forward(self, x):
        if _fused(x, self):
            return _img(self.forward_tokens(_tok(x)))
        b, c, h, w = x.shape
        hn = _gn(self.norm, x, False)
        tok = lambda t: t.flatten(2).transpose(1, 2)  # [b, hw, c]
        o = ops.attention(tok(self.q(hn)), tok(self.k(hn)), tok(self.v(hn)), heads=1)
        return x + self.proj_out(o.transpose(1, 2).reshape(b, c, h, w))


class Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        if _fused(x, self):
            return _img(mconv.fused_conv(_tok(x), self.conv, upsample=True)[0])
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 in_channels=3, resolution=256, z_channels=4, **ignored):
        super().__init__()
        if len(attn_resolutions) != 0:
            raise NotImplementedError("per-level attention is not used by the ViewCrafter VAE (yaml: attn_resolutions: [])")
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block_out = ch * ch_mult[i_level]
            up = nn.Module()
            up.block = nn.ModuleList()
            up.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                up.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if i_level != 0:
                up.upsample = Upsample(block_in)
            self.up.insert(0, up)
        self.norm_out = _norm(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)

    def _forward_fused(self, z):
        """The whole decoder on token-major data with the MFMA convolutions; every convolution epilogue hands the next
        GroupNorm its statistics, so no separate statistics / normalisation pass touches the (up to 576x1024x128) maps."""
        t, st = mconv.fused_conv(_tok(z), self.conv_in, stats_groups=32)
        t, _ = self.mid.block_1.forward_tokens(t, st)
        t = self.mid.attn_1.forward_tokens(t)
        t, st = self.mid.block_2.forward_tokens(t, None, want_stats=True)
        for i_level in reversed(range(self.num_resolutions)):
            blocks = self.up[i_level].block
            for k, blk in enumerate(blocks):
                last = k == len(blocks) - 1
                t, st = blk.forward_tokens(t, st, want_stats=not (last and i_level != 0))
            if i_level != 0:
                t, st = mconv.fused_conv(t, self.up[i_level].upsample.conv, upsample=True, stats_groups=32)
        ns = mconv.norm_state(self.norm_out, partial=st)
        return _img(mconv.fused_conv(t, self.conv_out, gn=self.norm_out, norm=ns, silu=True)[0])

    def forward(self, z):
        if _fused(z, self):
            return self._forward_fused(z)
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            for blk in self.up[i_level].block:
                h = blk(h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        return self.conv_out(_gn_swish(self.norm_out, h))


class Downsample(nn.Module):
    """ae_modules.py:90-109: stride-2 3x3 convolution on an input padded by one pixel on the right / bottom only."""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        if _fused(x, self) and not (torch.is_grad_enabled() and x.requires_grad):
            return _img(mconv.fused_conv(_tok(x), self.conv, mode=mconv.STRIDE2_PAD_HI)[0])
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Encoder(nn.Module):
    """ae_modules.py:360-462 (runs once per video, before the DDIM loop): conv-in, per level
    `num_res_blocks` ResnetBlocks + Downsample, mid (Res, single-head attention, Res), GN -> swish -> conv-out to
    2*z_channels moments.  Same parameter names as the reference (`encoder.down.1.block.0.norm1.weight`, ...)."""

    def __init__(self, *, ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0, in_channels=3,
                 resolution=256, z_channels=4, double_z=True, **ignored):
        super().__init__()
        if len(attn_resolutions) != 0:
            raise NotImplementedError("per-level attention is not used by the ViewCrafter VAE (yaml: attn_resolutions: [])")
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            down = nn.Module()
            down.block = nn.ModuleList()
            down.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                down.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = _norm(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)

    def _forward_fused(self, x):
        """Token-major encoder on the MFMA convolutions (conv-in pads its 3 input channels to 8; stride-2 Downsample mode)."""
        t, st = mconv.fused_conv(_tok(x), self.conv_in, stats_groups=32)
        for i_level in range(self.num_resolutions):
            blocks = self.down[i_level].block
            for k, blk in enumerate(blocks):
                t, st = blk.forward_tokens(t, st, want_stats=not (k == len(blocks) - 1 and i_level != self.num_resolutions - 1))
            if i_level != self.num_resolutions - 1:
                t, st = mconv.fused_conv(t, self.down[i_level].downsample.conv, mode=mconv.STRIDE2_PAD_HI, stats_groups=32)
        t, _ = self.mid.block_1.forward_tokens(t, st)
        t = self.mid.attn_1.forward_tokens(t)
        t, st = self.mid.block_2.forward_tokens(t, None, want_stats=True)
        ns = mconv.norm_state(self.norm_out, partial=st)
        return _img(mconv.fused_conv(t, self.conv_out, gn=self.norm_out, norm=ns, silu=True)[0])

    def forward(self, x):
        if _fused(x, self) and not (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_fused(x)
        h = self.conv_in(x)
        for i_level in range(self.num_resolutions):
            for blk in self.down[i_level].block:
                h = blk(h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(_gn_swish(self.norm_out, h))


class DiagonalGaussianDistribution:
    """lvdm/distributions.py:24-41 (the parts the sampling path uses)."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape)   # drawn on the CPU generator, exactly like the reference (:36-37)
        return self.mean + self.std * noise.to(device=self.parameters.device)

    def mode(self):
        return self.mean


class AutoencoderKLDecoder(nn.Module):
    """`first_stage_model` as far as the DDIM loop needs it: decode() (autoencoder.py:104-107).  `with_encoder=True`
    adds the encoder + quant_conv so that `encode()` (:97-102) works too and the whole
    `first_stage_model.*` part of the checkpoint loads strict."""

    def __init__(self, ddconfig, embed_dim=4, with_encoder=False):
        super().__init__()
        if with_encoder:
            self.encoder = Encoder(**ddconfig)
            self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.decoder = Decoder(**ddconfig)
        self._ch, self._ch_mult = int(ddconfig["ch"]), [int(m) for m in ddconfig["ch_mult"]]
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)

        self._token_major = False

    def to_token_major(self):
        """Keep every feature map channels_last (NHWC convolution kernels, token-major GroupNorm): same values, same
        state_dict; Conv2d weights are stored channels_last."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        self._token_major = True
        return self

    def encode(self, x, **kwargs):
        if not hasattr(self, "encoder"):
            raise RuntimeError("AutoencoderKLDecoder was built without the encoder (with_encoder=True)")
        return DiagonalGaussianDistribution(self._conv1x1(self.quant_conv, self.encoder(x)))

    def _conv1x1(self, conv, z):
        """A 1x1 convolution (`post_quant_conv` 4 -> 4, `quant_conv` 8 -> 8; autoencoder.py:97-107) as a per-token projection on
        the MFMA GEMM (channels padded to its granule of 8) -- on the 16-bit device path no library convolution is called."""
        if z.is_cuda and z.dtype in (torch.float16, torch.bfloat16) and z.dtype == conv.weight.dtype and not conv.weight.requires_grad:
            y = gemm.linear(z.permute(0, 2, 3, 1), conv.weight, conv.bias).permute(0, 3, 1, 2)     # token rows [N, H, W, C]
            return y.contiguous(memory_format=torch.channels_last) if self._token_major else y.contiguous()
        return conv(z)

    def decode(self, z, **kwargs):
        if self._token_major:
            z = z.contiguous(memory_format=torch.channels_last)
        return self.decoder(self._conv1x1(self.post_quant_conv, z))

    #: frames per decoder / encoder call under `perframe_ae` (None: as many as fit the bounds below)
    PERFRAME_GROUP_MAX = 32

    def _frames_that_fit(self, x, latent):
        """The convolution kernels address with 32-bit element offsets: the largest feature map of one call (full
        resolution x ch * ch_mult[1] channels) has to stay below 2^31 elements."""
        full = x.shape[-2] * x.shape[-1] * ((2 ** (len(self._ch_mult) - 1)) ** 2 if latent else 1)
        return max(1, min(self.PERFRAME_GROUP_MAX, (2 ** 31 - 1) // max(1, full * self._ch * max(self._ch_mult[:2]))))

    def perframe(self, fn, x, frames_per_call=None, latent=True):
        """`perframe_ae` (ddpm3d.py:630-667) bounds VAE memory by running one frame per call -- sized for 40/80 GB parts.
        Every normalisation in the VAE is per sample, so frames grouped into one call produce the same per-frame values;
        with 288 GB the 25-frame video goes through in a few calls (full waves of workgroups per launch instead of
        25 x 1.1), and `frames_per_call` brings the bound back for callers that need it."""
        k = self._frames_that_fit(x, latent)
        if frames_per_call:
            k = max(1, min(k, int(frames_per_call)))
        if x.shape[0] <= k:
            return fn(x)
        return torch.cat([fn(x[i:i + k]) for i in range(0, x.shape[0], k)], dim=0)
