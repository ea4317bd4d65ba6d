"""Hand-written MFMA convolutions of the diffusion path (csrc/conv_mfma.hip through the C-ABI `gvd_conv_mfma`).

    fused_conv(x, conv, ...)     out = conv(silu?(GroupNorm(x))) + bias + add_nc + residual   (+ statistics for the next norm)

`x` / `out` are token-major [N, H, W, C] 16-bit tensors; the temporal (3,1,1) form takes [T, pixels, C] or, for a batch,
[samples, T, pixels, C] (the samples are extra pixel tiles of ONE launch; norms and statistics stay per sample).  The weights stay
the module's `nn.Conv2d` / `nn.Conv3d` parameters (reference state-dict keys); the packed MFMA image is cached on the
parameter.  Forward and input gradient (the guided sampler differentiates w.r.t. x_t with frozen weights) run the same
kernel -- the input gradient is the convolution with the transposed, tap-flipped weights.

No CPU path: on CPU tensors `fused_conv` raises unless `ops.use_reference_math(True)` is active (tests / cpu_baseline only),
in which case it evaluates the same expression with torch ops (the reference's own formulation).
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import ops

SPATIAL, TEMPORAL, STRIDE2, STRIDE2_PAD_HI = 0, 1, 2, 3   # C-ABI `mode` values (include/gvd_diffusion.h)
NEAREST, ZERO_STUFF = 1, 2                               # C-ABI `upsample` values
UP2, UP2_BWD = 4, 5                                      # C-ABI modes: nearest x2 upsampling + 3x3 as four phase convolutions of the input map; its input gradient
UP2_PHASES = os.environ.get("GVD_CONV_UP2_PHASES", "1") == "1"   # 0: the on-the-fly upsampled patch with 9 taps (A/B runs, tests)
STATS_REPLICAS = 8


def config(mode, N, H, W, Cin, Cout):
    """(BN, tile pixels, tile width) the kernel uses for this problem (BN is the weight-packing granule)."""
    bn, pix, tw = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    ops._check(ops.lib().gvd_conv_config(mode, N, H, W, Cin, Cout, ctypes.byref(bn), ctypes.byref(pix), ctypes.byref(tw)))
    return bn.value, pix.value, tw.value


def pack_weight(w3, BN):
    """[Cout, Cin, taps] -> the LDS image layout of include/gvd_diffusion.h (gvd_conv_mfma): [co tiles][chunks][taps][BN][4][8],
    slot s of row r holds input channels (s ^ ((r >> 2) & 3)) * 8 .. +7 of the chunk."""
    Cout, Cin, taps = w3.shape
    nct, nch = -(-Cout // BN), -(-Cin // 32)
    wp = w3.new_zeros(nct * BN, nch * 32, taps)
    wp[:Cout, :Cin] = w3
    wp = wp.reshape(nct, BN, nch, 4, 8, taps).permute(0, 2, 5, 1, 3, 4)          # [nct, nch, taps, BN, 4, 8]
    r = torch.arange(BN, device=w3.device)
    idx = torch.arange(4, device=w3.device)[None, :] ^ ((r >> 2) & 3)[:, None]     # [BN, 4]
    idx = idx[None, None, None, :, :, None].expand(nct, nch, taps, BN, 4, 8)
    return wp.gather(4, idx).contiguous()


def _taps(weight):
    """Conv2d [Co, Ci, 3, 3] / Conv3d [Co, Ci, 3, 1, 1] -> [Co, Ci, taps]."""
    return weight.reshape(weight.shape[0], weight.shape[1], -1)


def invalidate_packed(module_or_weight):
    """Drop the cached MFMA images.  The cache is keyed on (tensor._version, data_ptr): in-place autograd-visible updates
    (`p.copy_`, `p.mul_`, optimizer steps, load_state_dict) bump the version and re-pack by themselves; writes THROUGH `.data`
    (`p.data.copy_(ema)`, EMA swaps) do not -- call this after them."""
    ws = module_or_weight.parameters() if isinstance(module_or_weight, torch.nn.Module) else [module_or_weight]
    for w in ws:
        for attr in ("_gvd_packed", "_gvd_f32", "_gvd_gemm"):
            if hasattr(w, attr):
                try:
                    delattr(w, attr)
                except AttributeError:
                    pass


def packed(weight, BN, backward=False, cin_pad=0, dtype=None):
    """Packed image of a conv weight in `dtype` (the activations' 16-bit type; fp32 master weights under autocast are cast
    here, once), cached on the parameter until it is modified (see invalidate_packed for `.data` writes).  backward: the
    input-gradient operator (Cout <-> Cin transposed, taps flipped).  cin_pad: zero input channels appended (Cin not a
    multiple of 8)."""
    dtype = dtype or weight.dtype
    key = (BN, backward, cin_pad, dtype, weight.device)
    cache = getattr(weight, "_gvd_packed", None)
    tag = (weight._version, weight.data_ptr())
    if cache is None or cache[0] != tag:
        cache = (tag, {})
        try:
            weight._gvd_packed = cache
        except AttributeError:
            pass
    hit = cache[1].get(key)
    if hit is None:
        if backward == "up2":
            # nearest x2 upsampling + 3x3 as four 2x2 convolutions of the input map (kernel mode 4): output pixel 2 i + a reads
            # x[i + a + k - 1], k = 0, 1, per dimension; the taps of W that land on the same input pixel are summed (fp32, one rounding):
            # kernel index sets per (phase a, tap k):  a = 0: {0}, {1, 2};  a = 1: {0, 1}, {2}
            sets = (((0,), (1, 2)), ((0, 1), (2,)))
            w4 = weight.detach().float()
            phases = []
            for ay in (0, 1):
                for ax in (0, 1):
                    taps = [sum(w4[:, :, jy, jx] for jy in sets[ay][ky] for jx in sets[ax][kx]) for ky in (0, 1) for kx in (0, 1)]
                    w3 = torch.stack(taps, dim=2).to(dtype)                       # [Cout, Cin, 4]
                    if cin_pad:
                        w3 = F.pad(w3, (0, 0, 0, cin_pad))
                    phases.append(pack_weight(w3.contiguous(), BN))
            hit = cache[1][key] = torch.stack(phases).contiguous()               # [4 phases][co tiles][chunks][4 taps][BN][4][8]
            return hit
        if backward == "up2_bwd":
            # input gradient of the phase form (kernel mode 5): gx[i] = sum_u K_u g[2 i + u], u = -1 .. 2 per dimension, K_u the sum of
            # the transposed taps W_j^T over the index sets {2}, {1, 2}, {0, 1}, {0} (u = -1, 0, 1, 2); over the phase images
            # g_b[j] = g[2 j + b] tap k of phase b is u = 2 k - b.  The four phases are consecutive runs of input-channel chunks.
            usets = {-1: (2,), 0: (1, 2), 1: (0, 1), 2: (0,)}
            w4 = weight.detach().float()
            cg = w4.shape[0] + cin_pad
            cg32 = -(-cg // 32) * 32
            runs = []
            for by in (0, 1):
                for bx in (0, 1):
                    taps = [sum(w4[:, :, jy, jx] for jy in usets[2 * ky - by] for jx in usets[2 * kx - bx]).t() for ky in (0, 1) for kx in (0, 1)]
                    w3 = torch.stack(taps, dim=2).to(dtype)                       # [Cin_fwd, Cout_fwd, 4]
                    runs.append(F.pad(w3, (0, 0, 0, cg32 - w3.shape[1])))
            hit = cache[1][key] = pack_weight(torch.cat(runs, dim=1).contiguous(), BN)   # [co tiles][4 x chunks per phase][4 taps][BN][4][8]
            return hit
        w3 = _taps(weight.detach()).to(dtype)
        if backward:
            w3 = w3.flip(2).transpose(0, 1)
        if cin_pad:
            w3 = F.pad(w3, (0, 0, 0, cin_pad))
        hit = cache[1][key] = pack_weight(w3.contiguous(), BN)
    return hit


class NormState:
    """GroupNorm statistics + per-(sample, channel) affine in the layout of gvd_group_norm (fp64 sums [N][G][2], then fp32
    (a, b) [N][C]); what the fused convolution's prologue and the GroupNorm backward kernel read."""
    __slots__ = ("buf", "N", "C", "G", "S", "eps", "gamma32", "group")

    def __init__(self, buf, N, C, G, S, eps, gamma32, group=None):
        # S: elements per (sample, channel) the statistics span -- over ALL ranks of `group` when the norm is sharded
        self.buf, self.N, self.C, self.G, self.S, self.eps, self.gamma32, self.group = buf, N, C, G, S, eps, gamma32, group

    @property
    def coef_ptr(self):
        return self.buf.data_ptr() + 16 * self.N * self.G


class PartialStats:
    """Sum / sum of squares per (sample, group) as accumulated by a convolution epilogue: fp64 [R, N, G, 2]."""
    __slots__ = ("sums", "R", "N", "G", "S")

    def __init__(self, sums, R, N, G, S):
        self.sums, self.R, self.N, self.G, self.S = sums, R, N, G, S


def norm_state(gn, x=None, partial=None, n_stat=None, merge=1, group=None, S_total=None):
    """Norm state of GroupNorm module `gn` for input x [n_stat, ..., C] (token-major), from a statistics pass over x or
    from the partial sums `partial` a producing convolution left (merge consecutive samples: per-frame -> per-video).
    group / S_total: the statistics span the slices held by the ranks of `group` (frame-sharded temporal norms): the local
    sums are all-reduced (2 G doubles per sample) before the affine is formed; S_total = global elements per channel."""
    P, LL = ctypes.c_void_p, ctypes.c_longlong
    G = gn.num_groups
    g32, b32 = ops._f32_param(gn.weight), ops._f32_param(gn.bias)
    C = g32.numel()
    if partial is not None:
        N, S, dev = partial.N // merge, partial.S * merge, partial.sums.device
    else:
        x = x.contiguous()
        N, dev = n_stat, x.device
        S = x.numel() // (N * C)
    buf = torch.empty(2 * N * G + N * C, dtype=torch.float64, device=dev)
    L, eps, st = ops.lib(), ctypes.c_float(gn.eps), P(ops._stream())
    with ops._on(dev):
        if partial is not None:   # merge replicas / samples (and, unsharded, form the affine in the same call)
            ops._check(L.gvd_group_norm_coef(P(buf.data_ptr()), P(partial.sums.data_ptr()), partial.R, merge, P(g32.data_ptr()),
                                             P(b32.data_ptr()), N, C, LL(S), G, eps, st))
        else:
            ops._check(L.gvd_group_norm_stats(P(x.data_ptr()), P(buf.data_ptr()), N, C, LL(S), G, 1,
                                              1 if x.dtype == torch.bfloat16 else 0, st))
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(buf[:2 * N * G], group=group)
            S = int(S_total) if S_total is not None else ops._global_count(S, group, dev)
        if partial is None or group is not None:
            ops._check(L.gvd_group_norm_coef(P(buf.data_ptr()), None, 1, 1, P(g32.data_ptr()), P(b32.data_ptr()), N, C, LL(S), G, eps, st))
    return NormState(buf, N, C, G, S, gn.eps, g32, group)


def cat_with_stats(a, b, gn):
    """torch.cat([a, b], -1) of two token-major tensors [N, H, W, Ca] / [N, H, W, Cb] plus the NormState of GroupNorm `gn` over the result
    (per-sample statistics), in ONE pass over the sources (gvd_cat2_group_norm_stats): the U-Net decoder's skip concatenation feeds a
    ResBlock whose first norm would otherwise re-read the 0.3-0.6 GB tensor the copy kernel just wrote.  Inference path only (no autograd)."""
    P, LL = ctypes.c_void_p, ctypes.c_longlong
    N, H, W, Ca = a.shape
    Cb = b.shape[-1]
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((N, H, W, Ca + Cb), dtype=a.dtype, device=a.device)
    G = gn.num_groups
    g32, b32 = ops._f32_param(gn.weight), ops._f32_param(gn.bias)
    C, S = Ca + Cb, H * W
    buf = torch.empty(2 * N * G + N * C, dtype=torch.float64, device=a.device)
    L, st = ops.lib(), P(ops._stream())
    with ops._on(a.device):
        ops._check(L.gvd_cat2_group_norm_stats(P(a.data_ptr()), Ca, P(b.data_ptr()), Cb, P(out.data_ptr()), P(buf.data_ptr()), N, LL(S), G,
                                               1 if a.dtype == torch.bfloat16 else 0, st))
        ops._check(L.gvd_group_norm_coef(P(buf.data_ptr()), None, 1, 1, P(g32.data_ptr()), P(b32.data_ptr()), N, C, LL(S), G,
                                         ctypes.c_float(gn.eps), st))
    return out, NormState(buf, N, C, G, S, gn.eps, g32, None)


def cat_with_stats_ok(a, b, gn):
    """The fused concatenation applies: 16-bit token-major device tensors of one dtype, channel counts in octets, no gradient wanted."""
    return (a.is_cuda and a.dtype in (torch.float16, torch.bfloat16) and b.dtype == a.dtype and a.shape[:-1] == b.shape[:-1]
            and a.shape[-1] % 8 == 0 and b.shape[-1] % 8 == 0 and (a.shape[-1] + b.shape[-1]) % gn.num_groups == 0
            and not (torch.is_grad_enabled() and (a.requires_grad or b.requires_grad)) and not os.environ.get("GVD_NO_FUSED_CAT"))


class _ZeroArena:
    """Zeroed fp64 scratch for the statistics accumulators of the convolution epilogues.  Every statistics-producing launch needs its
    own zeroed [replicas, samples, groups, 2] block (~110 per U-Net forward, 0.5-200 KB each); a `torch.zeros` per launch is a ~4 us
    fill kernel in front of every convolution.  Blocks are carved from an 8 MB tensor zeroed by ONE fill and never handed out twice
    (the tensor dies with its last block), so the zeroing costs one launch per ~40 convolutions."""
    WORDS = 1 << 20
    _cur = {}

    @classmethod
    def take(cls, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 15) & ~15                                  # 128-byte granules
        # (inside a hipGraph capture the fill must belong to the graph that uses the block: no sharing across captures)
        if n_al > cls.WORDS // 4 or (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            return torch.zeros(shape, dtype=torch.float64, device=device)
        key = (device.type, device.index, torch.cuda.current_stream(device).stream_id if device.type == "cuda" else 0)
        buf, used = cls._cur.get(key, (None, cls.WORDS))
        if used + n_al > cls.WORDS:
            buf, used = torch.zeros(cls.WORDS, dtype=torch.float64, device=device), 0
        cls._cur[key] = (buf, used + n_al)
        return buf[used:used + n].view(shape)


def _launch(x, wpk, Cout, mode, N, H, W, Cin, *, coef_ptr=None, coef_per_n=1, silu=False, bias=None, add_nc=None,
            residual=None, stats_groups=0, upsample=0, H_in=0, W_in=0, norm_bwd=None):
    """One launch of the convolution kernel.  norm_bwd = (norm_x, NormState, silu): this is an input-gradient launch whose
    output is the gradient w.r.t. silu?(GroupNorm(norm_x)); the returned sums are then the GroupNorm-BACKWARD statistics
    (gvd_conv_mfma_norm_bwd) instead of the forward ones."""
    P = ctypes.c_void_p
    # (temporal: N = frames, H = samples, W = pixels per frame -- the samples of a batch are extra pixel tiles of ONE launch)
    out = torch.empty(((N, W, Cout) if H == 1 else (H, N, W, Cout)) if mode == TEMPORAL else ((N, H // 2, W // 2, Cout) if mode == UP2_BWD else (N, H, W, Cout)),
                      dtype=x.dtype, device=x.device)
    sums = None
    if stats_groups:
        n_stat = H if mode == TEMPORAL else N
        sums = _ZeroArena.take((STATS_REPLICAS, n_stat, stats_groups, 2), x.device)
    if norm_bwd is not None:
        nx, ns, nsilu = norm_bwd
        with ops._on(x.device):
            rc = ops.lib().gvd_conv_mfma_norm_bwd(P(x.data_ptr()), P(wpk.data_ptr()), P(out.data_ptr()), P(sums.data_ptr()),
                                                  STATS_REPLICAS, stats_groups, mode, N, H, W, Cin, Cout, P(nx.data_ptr()),
                                                  P(ns.coef_ptr), 1, P(ns.gamma32.data_ptr()),
                                                  int(bool(nsilu)), 1 if x.dtype == torch.bfloat16 else 0, P(ops._stream()))
        ops._check(rc)
        return out, PartialStats(sums, STATS_REPLICAS, sums.shape[1], stats_groups, 0)
    with ops._on(x.device):
        rc = ops.lib().gvd_conv_mfma(P(x.data_ptr()), P(wpk.data_ptr()), P(coef_ptr), coef_per_n,
                                     P(bias.data_ptr() if bias is not None else None),
                                     P(add_nc.data_ptr() if add_nc is not None else None),
                                     P(residual.data_ptr() if residual is not None else None), P(out.data_ptr()),
                                     P(sums.data_ptr() if sums is not None else None), STATS_REPLICAS, stats_groups, mode,
                                     N, H, W, H_in, W_in, Cin, Cout, int(upsample), int(bool(silu)),
                                     1 if x.dtype == torch.bfloat16 else 0, P(ops._stream()))
    ops._check(rc)
    if sums is not None:
        S = N * W if mode == TEMPORAL else H * W
        return out, PartialStats(sums, STATS_REPLICAS, sums.shape[1], stats_groups, S)
    return out, None


SHEETS = os.environ.get("GVD_CONV_SHEETS", "1") == "1"   # 0: small maps run one tile (or two) per frame, as before round 4 (A/B runs, tests)
SHEET_MAX_PIXELS = 256


def _sheet_plan(N, H, W):
    """Maps per sheet row (Q >= 1) if the N maps of H x W pixels should run as ONE frame sheet (csrc/conv_mfma.hip: k_sheet_in), else
    0.  A map smaller than the 16 x 8 tile wastes the rest of it (5 x 7: 73 %, 10 x 14 and 9 x 16: 44 %); on the sheet the tiles span
    maps.  Taken when the padded tile area shrinks by >= 17 %."""
    if not SHEETS or N < 2 or H * W > SHEET_MAX_PIXELS:
        return 0

    def padded(h, w):   # tile slots of the better of the two 128-pixel tile shapes (16 x 8, 32 x 4)
        return min(-(-h // (128 // tw)) * (128 // tw) * (-(-w // tw) * tw) for tw in (16, 32))

    now = N * padded(H, W)
    best_q, best = 0, now
    for q in (1, 2, 4):
        if q > N:
            break
        a = padded(-(-N // q) * (H + 1) - 1, q * (W + 1) - 1)
        if a < best:
            best_q, best = q, a
    return best_q if best * 1.2 <= now else 0


SPLITK = os.environ.get("GVD_CONV_SPLITK", "1") == "1"   # 0: no split-K launches (A/B runs, tests)
FORCE_KSPLIT = None


def _groups(mode, N, H, W, Cin, Cout):
    """Workgroups gvd_conv_mfma launches for this problem (csrc/conv_mfma.hip: choose / conv_launch)."""
    BN, pix, tw = config(mode, N, H, W, Cin, Cout)
    cols = -(-Cout // BN)
    if mode == TEMPORAL:
        return -(-W // max(1, min(pix // N, 32))) * H * cols
    return N * -(-H // (pix // tw)) * -(-W // tw) * cols


def _ksplit(mode, N, H, W, Cin, Cout):
    """Split-K factor: launches of fewer workgroups than the chip has slots (2 x 256) whose reduction is long -- the deepest U-Net
    level: 140-320 workgroups walking K = 9 x 1280 ... 9 x 2560, the temporal form at 35 / 140 / 144 pixels -- are cut along the input
    channels into ~900 workgroups of >= 8 chunks of 32 channels each.  The fp32 sum of the slices reads slices x output bytes: capped
    at 40 MB (~10 us), which keeps the 7000-row temporal launch at two slices (tests/scripts/r4_ksplit_sweep.py: 131 us at 2, 155 at
    5, 139 unsplit) and leaves launches of 340+ workgroups alone (a 400-workgroup VAE group measured 91 -> 109 us split)."""
    nchunks = -(-Cin // 32)
    if not SPLITK or nchunks < 16 or mode not in (SPATIAL, TEMPORAL):
        return 1
    if FORCE_KSPLIT is not None:          # experiments (tests/scripts/r4_ksplit_sweep.py)
        return max(1, min(int(FORCE_KSPLIT), nchunks // 2))
    g = _groups(mode, N, H, W, Cin, Cout)
    if g > 340:
        return 1
    k = min(8, nchunks // 8, int(900 / g + 0.5), max(1, (40 << 20) // (2 * N * H * W * Cout)))
    return k if k >= 2 else 1


def _launch_split(x, wpk, Cout, mode, N, H, W, Cin, ksplit, *, coef_ptr=None, silu=False):
    """The convolution as `ksplit` input-channel slices (gvd_conv_mfma_splitk): 16-bit partial sums [slices, ...out...], slices."""
    P = ctypes.c_void_p
    shape = ((N, W, Cout) if H == 1 else (H, N, W, Cout)) if mode == TEMPORAL else (N, H, W, Cout)
    part = torch.empty((ksplit,) + shape, dtype=x.dtype, device=x.device)
    n_sl = ctypes.c_int(0)
    with ops._on(x.device):
        ops._check(ops.lib().gvd_conv_mfma_splitk(P(x.data_ptr()), P(wpk.data_ptr()), P(coef_ptr), 1, P(part.data_ptr()), int(ksplit),
                                                  ctypes.byref(n_sl), mode, N, H, W, Cin, Cout, int(bool(silu)),
                                                  1 if x.dtype == torch.bfloat16 else 0, P(ops._stream())))
    return part, n_sl.value


def _sum_slices(part, slices, bias, residual, n_stat=1, stats_groups=0):
    """out = sum of the split-K slices + bias + residual (fp32, one rounding); with stats_groups also the PartialStats of out per
    (sample of n_stat, group), accumulated in the same pass."""
    P, LL = ctypes.c_void_p, ctypes.c_longlong
    out = torch.empty(part.shape[1:], dtype=part.dtype, device=part.device)
    C = out.shape[-1]
    sums = _ZeroArena.take((1, n_stat, stats_groups, 2), part.device) if stats_groups else None
    with ops._on(part.device):
        ops._check(ops.lib().gvd_conv_sum_slices(P(part.data_ptr()), int(slices), P(out.data_ptr()), P(None if bias is None else bias.data_ptr()),
                                                 P(None if residual is None else residual.data_ptr()), P(None if sums is None else sums.data_ptr()),
                                                 int(stats_groups), int(n_stat), LL(out.numel() // C), C,
                                                 1 if part.dtype == torch.bfloat16 else 0, P(ops._stream())))
    return out, (PartialStats(sums, 1, n_stat, stats_groups, out.numel() // (n_stat * C)) if stats_groups else None)


def _split_conv(x, wpk, Cout, mode, N, H, W, Cin, ksplit, *, coef_ptr=None, silu=False, bias=None, residual=None, n_stat=1, stats_groups=0):
    """conv(act(x)) + bias + residual as split-K slices and their sum: (out, PartialStats | None)."""
    part, slices = _launch_split(x, wpk, Cout, mode, N, H, W, Cin, ksplit, coef_ptr=coef_ptr, silu=silu)
    return _sum_slices(part, slices, bias, residual, n_stat, stats_groups)


def _sheet_conv(x, weight, backward, Cout, N, H, W, Cin, Q, *, pad=0, coef_ptr=None, silu=False, bias=None, add_nc=None, residual=None, stats_groups=0):
    """conv(act(x)) + bias + add_nc + residual of N small maps through one frame sheet: 3 launches (sheet in with the norm's affine,
    the plain convolution on a single image, sheet out with the per-frame adds)."""
    P = ctypes.c_void_p
    R = -(-N // Q)
    Hv, Wv = R * (H + 1) - 1, Q * (W + 1) - 1
    bf = 1 if x.dtype == torch.bfloat16 else 0
    sheet = torch.empty((1, Hv, Wv, Cin), dtype=x.dtype, device=x.device)
    with ops._on(x.device):
        ops._check(ops.lib().gvd_conv_sheet_in(P(x.data_ptr()), P(sheet.data_ptr()), P(coef_ptr), int(bool(silu)), N, H, W, Cin, Q, bf,
                                               P(ops._stream())))
    BN, _, _ = config(SPATIAL, 1, Hv, Wv, Cin, Cout)
    wpk = packed(weight, BN, backward, pad, x.dtype)
    k = _ksplit(SPATIAL, 1, Hv, Wv, Cin, Cout)
    if k > 1:   # one sheet is few workgroups with a long reduction: cut it along the input channels
        out_v, slices = _launch_split(sheet, wpk, Cout, SPATIAL, 1, Hv, Wv, Cin, k)
    else:
        (out_v, _), slices = _launch(sheet, wpk, Cout, SPATIAL, 1, Hv, Wv, Cin), 1
    out = torch.empty((N, H, W, Cout), dtype=x.dtype, device=x.device)
    sums = _ZeroArena.take((1, N, stats_groups, 2), x.device) if stats_groups else None
    with ops._on(x.device):
        ops._check(ops.lib().gvd_conv_sheet_out(P(out_v.data_ptr()), slices, P(out.data_ptr()), P(None if bias is None else bias.data_ptr()),
                                                P(None if add_nc is None else add_nc.data_ptr()),
                                                P(None if residual is None else residual.data_ptr()),
                                                P(None if sums is None else sums.data_ptr()), int(stats_groups), N, H, W, Cout, Q, bf,
                                                P(ops._stream())))
    return out, (PartialStats(sums, 1, N, stats_groups, H * W) if stats_groups else None)


FUSE_NORM_BACKWARD_STATS = os.environ.get("GVD_FUSE_NORM_BWD", "1") == "1"   # 0: separate statistics pass (k_gn_bwd_stats_*), for A/B runs and tests


def _dgrad_with_norm_backward(g, wT, x, ns, silu, mode, N, H, W, Cg, Cn, add=None):
    """gx = d/dx of conv(act(GroupNorm(x))) given g = d/d(conv output): the input-gradient convolution with the GroupNorm-backward
    statistics in its epilogue (gvd_conv_mfma_norm_bwd), the replica merge, (sharded norms: a 2 G-double all-reduce) and the apply pass.
    add: the gradient x receives along a residual branch (ops.GradCell), summed in by the apply kernel."""
    P, LL = ctypes.c_void_p, ctypes.c_longlong
    x = x.contiguous()
    dev, bf = x.device, 1 if x.dtype == torch.bfloat16 else 0
    gx = torch.empty_like(x)
    n_stat = H if mode == TEMPORAL else N
    if n_stat != ns.N:
        raise RuntimeError(f"fused_conv backward: norm state spans {ns.N} samples, the convolution {n_stat}")
    d_act, partial = _launch(g, wT, Cn, mode, N, H, W, Cg, stats_groups=ns.G, norm_bwd=(x, ns, silu))
    part = partial.sums
    scratch = torch.empty(2 * ns.N * ns.G + ns.N * ns.C, dtype=torch.float64, device=dev)
    S = x.numel() // (ns.N * ns.C)
    L, st = ops.lib(), P(ops._stream())
    with ops._on(dev):
        ops._check(L.gvd_group_norm_merge(P(scratch.data_ptr()), P(part.data_ptr()), STATS_REPLICAS, 1, ns.N, ns.G, st))
        if ns.group is not None:
            import torch.distributed as dist
            dist.all_reduce(scratch[:2 * ns.N * ns.G], group=ns.group)
        ops._check(L.gvd_group_norm_bwd_apply_add(P(x.data_ptr()), P(d_act.data_ptr()), P(None if add is None else add.data_ptr()),
                                                  P(gx.data_ptr()), P(ns.buf.data_ptr()), P(scratch.data_ptr()), ns.N, ns.C, LL(S),
                                                  LL(ns.S if ns.group is not None else S), ns.G, ctypes.c_float(ns.eps),
                                                  int(bool(silu)), 1, bf, st))
    return gx


def _geometry(x, mode, upsample):
    """(N, H, W, Cin, H_in, W_in): output geometry of the convolution of x, and the input's for the stride-2 modes."""
    if mode == TEMPORAL:      # [T, pixels, C], or [samples, T, pixels, C]: frames, samples, pixels per frame
        T, Pp, Cin = x.shape[-3:]
        return T, (x.shape[0] if x.dim() == 4 else 1), Pp, Cin, 0, 0
    N, Hin, Win, Cin = x.shape
    if mode == SPATIAL:
        return N, (2 * Hin if upsample else Hin), (2 * Win if upsample else Win), Cin, 0, 0
    pad_lo = 1 if mode == STRIDE2 else 0
    return N, (Hin + pad_lo - 2) // 2 + 1, (Win + pad_lo - 2) // 2 + 1, Cin, Hin, Win


def _run_forward(x, weight, bias, mode, upsample, ns, silu, add_nc, residual, stats_groups):
    N, H, W, Cin, H_in, W_in = _geometry(x, mode, upsample)
    Cout = weight.shape[0]
    pad = (-Cin) % 8
    if pad:
        if ns is not None:
            raise RuntimeError("fused_conv: a GroupNorm prologue needs Cin % 8 == 0")
        x = F.pad(x, (0, pad))
    up2 = bool(upsample) and mode == SPATIAL and UP2_PHASES and W // 2 >= 24 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
    if up2:   # nearest x2 + 3x3 as four 2x2 phase convolutions of the input map: 16 tap evaluations per input pixel instead of 36
        BN, _, _ = config(UP2, N, H, W, Cin + pad, Cout)
        b32 = None if bias is None else ops._f32_param(bias)
        if ns is not None and (ns.C != Cin or ns.N != N):
            raise RuntimeError(f"fused_conv: norm state is for {ns.N} x {ns.C} channels, input has {N} x {Cin}")
        return _launch(x.contiguous(), packed(weight, BN, "up2", pad, x.dtype), Cout, UP2, N, H, W, Cin + pad,
                       coef_ptr=None if ns is None else ns.coef_ptr, coef_per_n=1, silu=silu, bias=b32,
                       add_nc=None if add_nc is None else add_nc.contiguous(),
                       residual=None if residual is None else residual.contiguous(), stats_groups=stats_groups)
    BN, _, _ = config(mode, N, H, W, Cin + pad, Cout)
    wpk = packed(weight, BN, False, pad, x.dtype)
    b32 = None if bias is None else ops._f32_param(bias)
    n_norm = H if mode == TEMPORAL else N
    if ns is not None and (ns.C != Cin or ns.N != n_norm):
        raise RuntimeError(f"fused_conv: norm state is for {ns.N} x {ns.C} channels, input has {n_norm} x {Cin}")
    Q = _sheet_plan(N, H, W) if (mode == SPATIAL and not upsample and Cout % 8 == 0) else 0
    if Q:   # maps smaller than a tile: one frame sheet (the per-frame prologue / epilogue terms move into the sheet kernels)
        return _sheet_conv(x.contiguous(), weight, False, Cout, N, H, W, Cin + pad, Q, pad=pad, coef_ptr=None if ns is None else ns.coef_ptr,
                           silu=silu, bias=b32, add_nc=None if add_nc is None else add_nc.contiguous(),
                           residual=None if residual is None else residual.contiguous(), stats_groups=stats_groups)
    k = _ksplit(mode, N, H, W, Cin + pad, Cout) if (not upsample and add_nc is None and Cout % 8 == 0) else 1
    if k > 1:   # few workgroups, long reduction: split-K slices, then the sum with the epilogue terms (and a statistics pass)
        return _split_conv(x.contiguous(), wpk, Cout, mode, N, H, W, Cin + pad, k, coef_ptr=None if ns is None else ns.coef_ptr, silu=silu, bias=b32,
                           residual=None if residual is None else residual.contiguous(), n_stat=n_norm, stats_groups=stats_groups)
    return _launch(x.contiguous(), wpk, Cout, mode, N, H, W, Cin + pad,
                   coef_ptr=None if ns is None else ns.coef_ptr, coef_per_n=1,
                   silu=silu, bias=b32, add_nc=None if add_nc is None else add_nc.contiguous(),
                   residual=None if residual is None else residual.contiguous(), stats_groups=stats_groups,
                   upsample=NEAREST if upsample else 0, H_in=H_in, W_in=W_in)


class _FusedConvFn(torch.autograd.Function):
    """out = conv(act(x)) + bias + add_nc + residual; gradients w.r.t. x and residual only (weights frozen)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, mode, upsample, ns, silu, add_nc, stats_groups, cells=(None, None)):
        out, part = _run_forward(x, weight, bias, mode, upsample, ns, silu, add_nc, residual, stats_groups)
        ctx.set_materialize_grads(False)    # (the statistics output never carries a gradient: without this autograd fills a zero
        ctx.save_for_backward(x, weight)    #  tensor of its shape for every backward call -- ~190 fill launches per guided step)
        # ops.GradCell hand-overs: grad_add is taken in backward and summed into d/dx by the GroupNorm-backward apply kernel;
        # the residual's gradient goes into res_to (if its taker armed it) instead of to autograd's accumulation
        grad_add, res_to = cells
        if grad_add is not None:
            grad_add.arm()
        res_to = res_to if (res_to is not None and res_to.armed and residual is not None) else None
        ctx.cells = (grad_add, res_to)
        ctx.cfg = (mode, upsample, ns, silu, residual is not None)
        ctx.mark_non_differentiable(*([] if part is None else [part.sums]))
        ctx.part = part
        return (out,) if part is None else (out, part.sums)

    @staticmethod
    def backward(ctx, gout, *unused):
        x, weight = ctx.saved_tensors
        mode, upsample, ns, silu, has_res = ctx.cfg
        if gout is None:                    # nothing flows into `out` (only possible if the caller differentiates the statistics)
            return (None,) * 11
        gout = gout.contiguous()
        gx = None
        grad_add, res_to = ctx.cells
        extra = None if grad_add is None else grad_add.take(like=x)
        g_res = gout if has_res else None
        if g_res is not None and res_to is not None and res_to.put(g_res):
            g_res = None
        tail = (None,) * 9
        if ctx.needs_input_grad[0]:
            N, H, W, Cin, H_in, W_in = _geometry(x, mode, upsample)
            Cout = weight.shape[0]
            pad = (-Cout) % 8
            g = F.pad(gout, (0, pad)) if pad else gout
            if mode == STRIDE2_PAD_HI:
                raise NotImplementedError("fused_conv: the input gradient of the VAE-encoder Downsample is not on the guided path")
            if mode == STRIDE2:
                # transposed convolution = stride-1 convolution (transposed, tap-flipped weights) of the zero-stuffed gradient
                BN, _, _ = config(SPATIAL, N, 2 * H, 2 * W, Cout + pad, Cin)
                d_act, _ = _launch(g, packed(weight, BN, True, pad, gout.dtype), Cin, SPATIAL, N, 2 * H, 2 * W, Cout + pad,
                                   upsample=ZERO_STUFF)
                if (2 * H, 2 * W) != (H_in, W_in):
                    d_act = d_act[:, :H_in, :W_in].contiguous()
            elif (upsample and mode == SPATIAL and UP2_PHASES and W // 2 >= 24 and Cin % 8 == 0 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
                # nearest x2 + 3x3: the input gradient as ONE 2x2-per-phase convolution over the four phase images of the gradient
                # (16 tap evaluations per input pixel instead of 36, no 2x2 sum pass behind it)
                BN, _, _ = config(UP2_BWD, N, H, W, Cout + pad, Cin)
                d_act, _ = _launch(g, packed(weight, BN, "up2_bwd", pad, gout.dtype), Cin, UP2_BWD, N, H, W, Cout + pad)
                upsample = False                # (d_act is already at the input resolution)
            else:
                Q = _sheet_plan(N, H, W) if (mode == SPATIAL and not upsample and Cin % 8 == 0) else 0
                BN, _, _ = config(mode, N, H, W, Cout + pad, Cin)
                wT = None if Q else packed(weight, BN, True, pad, gout.dtype)
                ks = 1 if (Q or upsample or Cin % 8) else _ksplit(mode, N, H, W, Cout + pad, Cin)
                if Q:       # small maps: the input-gradient convolution on one frame sheet; the norm's backward runs its own two passes
                    d_act, _ = _sheet_conv(g, weight, True, Cin, N, H, W, Cout + pad, Q, pad=pad)
                elif ks > 1:   # few workgroups, long reduction: split-K slices + their sum (the norm's backward: two passes)
                    d_act, _ = _split_conv(g, wT, Cin, mode, N, H, W, Cout + pad, ks)
                elif ns is not None and not upsample and FUSE_NORM_BACKWARD_STATS:
                    # the GroupNorm-backward statistics come out of the input-gradient convolution's epilogue
                    gx = _dgrad_with_norm_backward(g, wT, x, ns, silu, mode, N, H, W, Cout + pad, Cin, add=extra)
                    return (gx, g_res) + tail
                else:
                    d_act, _ = _launch(g, wT, Cin, mode, N, H, W, Cout + pad)
            if upsample:   # nearest x2 backward: each input pixel fed a 2x2 block
                d_act = d_act.reshape(N, H // 2, 2, W // 2, 2, Cin).sum(dim=(2, 4))
            if ns is None:
                gx = d_act
            else:
                xs = x.reshape(ns.N, -1, ns.C)
                gx = ops._hip_group_norm_bwd(xs, d_act.reshape(xs.shape), ns.gamma32, ns.buf, ns.G, ns.eps, silu, True,
                                             ns.group, ns.S if ns.group is not None else None,
                                             add=None if extra is None else extra.reshape(xs.shape)).reshape(x.shape)
                extra = None
            if extra is not None:
                gx = gx + extra
        elif extra is not None:
            gx = extra
        return (gx, g_res) + tail


def _reference(x, weight, bias, mode, upsample, gn, silu, add_nc, residual, n_stat):
    """The same expression with stock torch ops (reference formulation): tests and the CPU baseline only."""
    if gn is not None:
        xs = x.reshape(n_stat, -1, x.shape[-1])
        x = ops.group_norm_math(xs, gn.num_groups, gn.weight, gn.bias, gn.eps, silu=silu, channels_last=True).reshape(x.shape)
    if mode != TEMPORAL:
        xi = x.permute(0, 3, 1, 2)
        if upsample:
            xi = F.interpolate(xi, scale_factor=2, mode="nearest")
        if mode == STRIDE2_PAD_HI:
            xi = F.pad(xi, (0, 1, 0, 1))
        y = F.conv2d(xi, weight.to(x.dtype), None if bias is None else bias.to(x.dtype), stride=1 if mode == SPATIAL else 2,
                     padding=0 if mode == STRIDE2_PAD_HI else 1).permute(0, 2, 3, 1)
        if add_nc is not None:
            y = y + add_nc.to(y.dtype)[:, None, None, :]
    elif x.dim() == 4:                                                             # [samples, T, P, C]
        xi = x.permute(0, 3, 1, 2)[..., None]                                      # [S, C, T, P, 1]
        y = F.conv3d(xi, weight.to(x.dtype), None if bias is None else bias.to(x.dtype), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1)
    else:
        xi = x.permute(2, 0, 1)[None, :, :, :, None]                              # [1, C, T, P, 1]
        y = F.conv3d(xi, weight.to(x.dtype), None if bias is None else bias.to(x.dtype), padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0)
    if residual is not None:
        y = y + residual
    return y


def fused_conv(x, conv, *, mode=SPATIAL, upsample=False, gn=None, norm=None, n_stat=None, silu=False, add_nc=None,
               residual=None, stats_groups=0, group=None, S_total=None, grad_add=None, res_grad_to=None):
    """x token-major ([N, H, W, Cin] or, temporal, [T, pixels, Cin]); `conv` the nn.Conv2d(3x3, pad 1) / nn.Conv3d((3,1,1))
    module; mode STRIDE2 / STRIDE2_PAD_HI: the stride-2 Downsample convolutions of the U-Net / the VAE encoder.
    gn: GroupNorm module applied (with `silu`) in the kernel's prologue; norm: a NormState for it if the caller
    already has one (from a producer's statistics), else a statistics pass over x runs first; n_stat: samples the norm
    statistics span separately (frames for 2-D norms, 1 for the temporal ones).
    grad_add / res_grad_to: ops.GradCell hand-overs under autograd (x feeds this convolution's norm AND a later `+ x`: the residual
    side delivers its gradient into the cell, this side adds it inside the GroupNorm-backward apply kernel).
    Returns (out, PartialStats | None): the statistics of `out` for a following GroupNorm with `stats_groups` groups."""
    on_dev = ops._require_device(x, "fused_conv")
    if not on_dev or x.dtype not in (torch.float16, torch.bfloat16):
        if on_dev and not ops._REFERENCE_MATH and x.dtype != torch.float32:
            raise RuntimeError(f"fused_conv: unsupported dtype {x.dtype}")
        if on_dev:
            ops._torch_form("fused_conv", f"dtype {x.dtype}")
        n_stat = n_stat if n_stat is not None else ((x.shape[0] if x.dim() == 4 else 1) if mode == TEMPORAL else x.shape[0])
        return _reference(x, conv.weight, conv.bias, mode, upsample, gn, silu, add_nc, residual, n_stat), None
    if torch.is_grad_enabled() and (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)):
        raise RuntimeError("fused_conv: only the input gradient is implemented (freeze the weights)")
    if torch.is_grad_enabled() and add_nc is not None and add_nc.requires_grad:
        raise RuntimeError("fused_conv: no gradient is produced for add_nc (the ResBlock's timestep / fs embedding term); detach it "
                           "or differentiate w.r.t. x only, as the guided sampler does")
    ns = None
    if gn is not None:
        n_stat = n_stat if n_stat is not None else ((x.shape[0] if x.dim() == 4 else 1) if mode == TEMPORAL else x.shape[0])
        ns = norm if norm is not None else norm_state(gn, x=x.detach(), n_stat=n_stat, group=group, S_total=S_total)
    need_grad = torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad))
    if need_grad:
        res = _FusedConvFn.apply(x, residual, conv.weight, conv.bias, mode, upsample, ns, silu, add_nc, stats_groups,
                                 (grad_add if x.requires_grad else None, res_grad_to))
        out = res[0]
        part = None
        if stats_groups:
            N = x.shape[0]
            S = out.shape[-3] * out.shape[-2] if mode == TEMPORAL else out.shape[1] * out.shape[2]
            part = PartialStats(res[1], res[1].shape[0], res[1].shape[1], stats_groups, S)
        return out, part
    return _run_forward(x, conv.weight, conv.bias, mode, upsample, ns, silu, add_nc, residual, stats_groups)
