"""Hand-written MFMA GEMM for every Linear / 1x1 convolution on token rows (csrc/gemm_mfma.hip through `gvd_gemm_nt`).

    linear(x, weight, bias, ln=..., residual=..., geglu=...)     y = [geglu]( LayerNorm?(x) W^T + b ) [+ residual]
    linear_cat(x, [w_q, w_k, w_v], ln=...)                       several Linears of one input as ONE launch (q | k | v columns)
    gemm_nt(x, w, ...)                                           the raw  Y = alpha X W^T  with strides / batch (attention chunks)

The LayerNorm never materialises: LN(x) W^T = rstd (x W'^T - mean s) + c with W' = W gamma, s = rowsum(W'), c = W beta + b -- the
kernel runs on the raw tokens, `row_stats` supplies (mean, rstd) per row, (s, c) are per-column vectors of the epilogue.  The
folded / permuted / concatenated weight images are cached on the parameter objects (invalidated by in-place updates; see
conv.invalidate_packed for `.data` writes).

No CPU path: CPU tensors raise unless `ops.use_reference_math(True)` (tests / cpu_baseline), which evaluates the same expression
with the reference's torch ops (F.layer_norm, F.linear, gelu).  fp32 device tensors (parity runs) take that form too, with the
usual one-time RuntimeWarning.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import ops

_P, _LL = ctypes.c_void_p, ctypes.c_longlong
_SIG = False


def _lib():
    global _SIG
    L = ops.lib()
    if not _SIG:
        L.gvd_gemm_nt.argtypes = [_P, _LL, _LL, _P, _LL, _LL, _P, _LL, _LL, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_float, _P, _P, _P, _P, _LL, _LL, ctypes.c_int, ctypes.c_int, _P]
        L.gvd_gemm_nt.restype = ctypes.c_int
        if hasattr(L, "gvd_gemm_nt_gate"):
            L.gvd_gemm_nt_gate.argtypes = [_P, _LL, _LL, _P, _LL, _LL, _P, _LL, _LL, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_float, _P, _P, _P, ctypes.c_int, _P, _LL, _LL, ctypes.c_int, _P]
            L.gvd_gemm_nt_gate.restype = ctypes.c_int
        L.gvd_row_stats.argtypes = [_P, _LL, _P, _LL, ctypes.c_int, ctypes.c_float, ctypes.c_int, _P]
        L.gvd_row_stats.restype = ctypes.c_int
        _SIG = True
    return L


def _mat(t, what):
    """(rows, cols, row stride, batch, batch stride) of a 2-D [R, C] or 3-D [B, R, C] tensor whose last dim is contiguous."""
    if t.dim() == 2:
        R, C = t.shape
        b, bs = 1, 0
        ld = t.stride(0)
    elif t.dim() == 3:
        b, R, C = t.shape
        bs, ld = t.stride(0), t.stride(1)
    else:
        raise ValueError(f"gemm_nt: {what} must be 2-D or 3-D, got {tuple(t.shape)}")
    if t.stride(-1) != 1 and C > 1:
        raise ValueError(f"gemm_nt: {what} must be contiguous along its last dimension")
    if R == 1:
        ld = max(ld, C)
    return R, C, ld, b, bs


def gemm_nt(x, w, *, alpha=1.0, bias=None, row_stats=None, col_sum=None, residual=None, geglu=False, out=None):
    """Y = epilogue(alpha X W^T).  x [M, K] or [B, M, K], w [N, K] or [B, N, K] (a 2-D operand is shared by the batch), 16-bit,
    last dim contiguous (row / batch strides free: views are read in place).  Returns [.., M, N] (N / 2 with geglu)."""
    M, K, ldx, bx, sx = _mat(x, "x")
    N, Kw, ldw, bw, sw = _mat(w, "w")
    if K != Kw or x.dtype != w.dtype or x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"gemm_nt: x {tuple(x.shape)} {x.dtype} / w {tuple(w.shape)} {w.dtype}")
    batch = max(bx, bw)
    if (bx not in (1, batch)) or (bw not in (1, batch)):
        raise ValueError("gemm_nt: batch sizes differ")
    No = N // 2 if geglu else N
    if out is None:
        out = torch.empty((batch, M, No) if (x.dim() == 3 or w.dim() == 3) else (M, No), dtype=x.dtype, device=x.device)
    Mo, Noo, ldy, _, sy = _mat(out, "out")
    if (Mo, Noo) != (M, No):
        raise ValueError(f"gemm_nt: out {tuple(out.shape)} for M = {M}, N = {No}")
    ldr = sr = 0
    if residual is not None:
        Mr, Nr, ldr, br, sr = _mat(residual, "residual")
        if (Mr, Nr) != (M, No) or residual.dtype != x.dtype:
            raise ValueError(f"gemm_nt: residual {tuple(residual.shape)} {residual.dtype}")
    p = lambda t: _P(None if t is None else t.data_ptr())
    with ops._on(x.device):
        rc = _lib().gvd_gemm_nt(p(x), ldx, sx if bx > 1 else 0, p(w), ldw, sw if bw > 1 else 0, p(out), ldy, sy if batch > 1 else 0,
                                M, N, K, batch, float(alpha), p(bias), p(row_stats), p(col_sum), p(residual), ldr, sr,
                                int(bool(geglu)), 1 if x.dtype == torch.bfloat16 else 0, _P(ops._stream()))
    ops._check(rc)
    return out


def row_stats(x2, eps):
    """fp32 (mean, rstd) pairs [M, 2] of the rows of x2 [M, C] (16-bit; row stride free)."""
    M, C = x2.shape
    out = torch.empty((M, 2), dtype=torch.float32, device=x2.device)
    with ops._on(x2.device):
        ops._check(_lib().gvd_row_stats(_P(x2.data_ptr()), x2.stride(0) if M > 1 else C, _P(out.data_ptr()), M, C, float(eps),
                                        1 if x2.dtype == torch.bfloat16 else 0, _P(ops._stream())))
    return out


# ---- weight images ------------------------------------------------------------------------------------------------------------
def _tag(t):
    return None if t is None else (t._version, t.data_ptr(), t.dtype, t.device)


def _geglu_perm(n2, device):
    """Row order of the GEGLU projection for the kernel's register epilogue (include/gvd_diffusion.h): per block of 32 tile rows,
    rows 0-15 are the VALUES and rows 16-31 the GATES of 16 consecutive outputs, each half in natural order -- a lane of the
    16 x 16 x 32 MFMA accumulator layout then owns value and gate of 4 consecutive outputs.  (value rows 0 .. n-1, gate rows
    n .. 2n-1 of the projection.)  A library of rounds 3-5 (A/B runs through GVD_DIFFUSION_LIB) has no gvd_gemm_geglu_layout and
    wants the 32 x 32 x 16 order: within each half, row c = 8 rg + 4 hi + e holds output 8 hi + 4 rg + e."""
    n = n2 // 2
    c = torch.arange(16, device=device)
    L = ops.lib()
    if hasattr(L, "gvd_gemm_geglu_layout") and L.gvd_gemm_geglu_layout() == 1:
        u = c
    else:
        u = 8 * ((c >> 2) & 1) + 4 * (c >> 3) + (c & 3)                   # output index within the block for half-row c
    blocks = torch.arange(n // 16, device=device)[:, None] * 16 + u[None, :]   # [n / 16, 16] value rows
    return torch.cat([blocks, blocks + n], dim=1).reshape(-1)


def _prepared(weights, biases, ln, geglu, dtype):
    """(W image [N, K] in `dtype`, bias / c vector fp32 [N] or None, col_sum fp32 [N] or None) for a list of Linear weights that
    share their input: rows concatenated, LayerNorm `ln` folded in, GEGLU row order applied.  Cached on the first weight, one
    slot per VARIANT (which LayerNorm is folded in, GEGLU order or natural order, dtype): the Resampler's `to_kv` is used with two
    different norms back to back (resampler.py:69) and a GEGLU projection is used in both row orders by the guided step (forward:
    GEGLU order; backward: the gate recompute in natural order) -- a single slot rebuilt the image on every one of those calls."""
    w0 = weights[0]
    variant = (None if ln is None else (id(ln.weight), id(ln.bias)), bool(geglu), dtype, len(weights))
    key = (tuple(_tag(w) for w in weights), tuple(_tag(b) for b in biases), None if ln is None else (_tag(ln.weight), _tag(ln.bias)))
    cache = getattr(w0, "_gvd_gemm", None)
    if isinstance(cache, dict):
        hit = cache.get(variant)
        if hit is not None and hit[0] == key:
            return hit[1]
    with torch.no_grad():
        W = torch.cat([w.detach().reshape(w.shape[0], -1).float() for w in weights], dim=0)        # (1x1 conv weights flatten to [N, K])
        have_bias = any(b is not None for b in biases)
        c = None
        if have_bias or ln is not None:
            c = torch.cat([(torch.zeros(w.shape[0], device=W.device) if b is None else b.detach().float()) for w, b in zip(weights, biases)])
        s = None
        if ln is not None:
            gamma = torch.ones(W.shape[1], device=W.device) if ln.weight is None else ln.weight.detach().float()
            if ln.bias is not None:
                c = c + W @ ln.bias.detach().float()
            Wd = (W * gamma[None, :]).to(dtype)
            s = Wd.float().sum(dim=1)              # of the ROUNDED image: acc - mean s == sum_k (x_k - mean) W'_k exactly
        else:
            Wd = W.to(dtype)
        if geglu:
            perm = _geglu_perm(Wd.shape[0], Wd.device)
            Wd = Wd[perm]
            c = None if c is None else c[perm]
            s = None if s is None else s[perm]
        pad = (-Wd.shape[1]) % 8
        if pad:
            Wd = F.pad(Wd, (0, pad))
        val = (Wd.contiguous(), None if c is None else c.contiguous(), None if s is None else s.contiguous())
    try:
        if not isinstance(cache, dict):
            cache = w0._gvd_gemm = {}
        if len(cache) >= 8:      # (a module rebuilt with fresh norm objects over and over: keep the table small)
            cache.clear()
        cache[variant] = (key, val)
    except AttributeError:
        pass
    return val


def _transposed(weights, dtype):
    """[K, sum N_i] image of the row-concatenated Linear weights (the input-gradient operator dX = dY W), cached on the first."""
    weights = list(weights) if isinstance(weights, (list, tuple)) else [weights]
    key = (tuple(_tag(w) for w in weights), dtype)
    cache = getattr(weights[0], "_gvd_gemm_t", None)
    if cache is None or cache[0] != key:
        W = torch.cat([w.detach().reshape(w.shape[0], -1) for w in weights], dim=0).to(dtype)
        Wt = W.t().contiguous()
        pad = (-Wt.shape[1]) % 8
        if pad:
            Wt = F.pad(Wt, (0, pad))
        cache = (key, Wt)
        try:
            weights[0]._gvd_gemm_t = cache
        except AttributeError:
            pass
    return cache[1]


class _FusedLinearFn(torch.autograd.Function):
    """[geglu](LayerNorm?(x) [W_0; W_1; ...]^T + b) [+ residual] as ONE forward launch under autograd; gradients w.r.t. x and the
    residual only (the guided sampler differentiates with frozen weights).  The backward needs nothing the forward would have to
    store beyond x: dY W is one GEMM against the transposed (unfolded) weights, the LayerNorm input gradient is the row kernel
    that recomputes the statistics from x, and for GEGLU the pre-activation is RECOMPUTED by one more LayerNorm-folded GEMM
    instead of being written (1.2 GB per block at 576x1024) and kept."""

    @staticmethod
    def forward(ctx, x2, r2, ln_w, ln_b, cfg, *wb):
        n = len(wb) // 2
        weights, biases = list(wb[:n]), list(wb[n:])
        ln, geglu = cfg[:2]
        # GradCells (ops.GradCell): grad_add -- taken in backward and added to the input gradient; res_to / in_to -- the residual's /
        # the input's gradient is put there instead of being returned to autograd (only into a cell its taker has armed)
        grad_add, res_to, in_to = cfg[2] if len(cfg) > 2 else (None, None, None)
        if grad_add is not None:
            grad_add.arm()
        res_to = res_to if (res_to is not None and res_to.armed and r2 is not None) else None
        in_to = in_to if (in_to is not None and in_to.armed) else None
        Wd, c, s = _prepared(weights, biases, ln, geglu, x2.dtype)
        st = None if ln is None else row_stats(x2, ln.eps)
        ctx.save_for_backward(*([x2] if (ln is not None or geglu) else []))   # (x is only needed by the LayerNorm backward / the gate recompute)
        ctx.cfg = (weights, biases, ln, geglu, r2 is not None, grad_add, res_to, in_to)
        return gemm_nt(x2, Wd, bias=c, row_stats=st, col_sum=s, residual=r2, geglu=geglu)

    @staticmethod
    def backward(ctx, gy):
        weights, biases, ln, geglu, has_res, grad_add, res_to, in_to = ctx.cfg
        x2 = ctx.saved_tensors[0] if (ln is not None or geglu) else None
        gy = gy if gy.stride(-1) == 1 else gy.contiguous()
        g = gy
        if geglu:   # recompute h = LayerNorm(x) W^T + b (natural column order), then the gate's backward row kernel
            Wd, c, s = _prepared(weights, biases, ln, False, x2.dtype)
            st = None if ln is None else row_stats(x2, ln.eps)
            h = gemm_nt(x2, Wd, bias=c, row_stats=st, col_sum=s)
            C = h.shape[-1] // 2
            g = torch.empty_like(h)
            with ops._on(h.device):
                ops._check(ops.lib().gvd_geglu_bwd(_P(h.data_ptr()), _P(gy.contiguous().data_ptr()), _P(g.data_ptr()),
                                                   _LL(h.shape[0]), C, 1 if h.dtype == torch.bfloat16 else 0, _P(ops._stream())))
        Wt = _transposed(weights, g.dtype)                       # [K, N]
        pad = Wt.shape[1] - g.shape[1]
        if pad:
            g = F.pad(g, (0, pad))
        extra = None if grad_add is None else grad_add.take()      # gradient of the same tensor along the residual branch, [M, K] rows
        if extra is not None:
            extra = extra.reshape(-1, extra.shape[-1])
        if ln is None:
            dh = gemm_nt(g, Wt, residual=extra)                     # `+ extra` in the dgrad GEMM's epilogue
        else:
            dh = gemm_nt(g, Wt)
            gx = torch.empty_like(dh)
            C = dh.shape[-1]
            x2c = x2 if x2.is_contiguous() else x2.contiguous()
            if extra is not None and not extra.is_contiguous():
                extra = extra.contiguous()
            with ops._on(dh.device):
                ops._check(ops.lib().gvd_layer_norm_bwd_add(_P(x2c.data_ptr()), _P(dh.data_ptr()), _P(ln.weight.data_ptr()),
                                                            _P(None if extra is None else extra.data_ptr()), _P(gx.data_ptr()),
                                                            _LL(dh.shape[0]), C, ctypes.c_float(ln.eps),
                                                            1 if dh.dtype == torch.bfloat16 else 0, _P(ops._stream())))
            dh = gx
        g_res = gy if has_res else None
        if g_res is not None and res_to is not None and res_to.put(g_res):
            g_res = None
        if in_to is not None and in_to.put(dh):
            dh = None
        return (dh, g_res, None, None, None) + (None,) * (2 * len(weights))


def _gate_gemm(x2, w, y, aux, mode, *, bias=None, row_stats=None, col_sum=None):
    """gvd_gemm_nt_gate on 2-D row tensors: mode 2 -- y [M, N / 2] = gate(x2 w^T ...), aux [M, N] = the projection (kernel block order);
    mode 3 -- the product is d/d(gated output) [M, N], aux [M, 2 N] the saved projection, y [M, 2 N] its gradient."""
    M, K = x2.shape
    N = w.shape[0]
    p = lambda t: _P(None if t is None else t.data_ptr())
    with ops._on(x2.device):
        rc = _lib().gvd_gemm_nt_gate(p(x2), x2.stride(0) if M > 1 else K, 0, p(w), w.stride(0), 0, p(y), y.stride(0) if M > 1 else y.shape[1], 0,
                                     M, N, K, 1, 1.0, p(bias), p(row_stats), p(col_sum), int(mode), p(aux), aux.stride(0) if M > 1 else aux.shape[1], 0,
                                     1 if x2.dtype == torch.bfloat16 else 0, _P(ops._stream()))
    ops._check(rc)
    return y


def _transposed_gate_order(weight, dtype):
    """[K, 2C] image of a GEGLU projection [2C, K] for the input-gradient GEMM behind the fused gate backward: columns in the kernel's
    [16 value | 16 gate] block order (the order gvd_gemm_nt_gate's mode 3 writes d/d(projection) in).  Cached on the weight."""
    key = (_tag(weight), dtype)
    cache = getattr(weight, "_gvd_gemm_tg", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            W = weight.detach().reshape(weight.shape[0], -1).to(dtype)
            Wt = W[_geglu_perm(W.shape[0], W.device)].t().contiguous()
        cache = (key, Wt)
        try:
            weight._gvd_gemm_tg = cache
        except AttributeError:
            pass
    return cache[1]


class _FeedForwardFn(torch.autograd.Function):
    """The transformer block's feed-forward under autograd with frozen weights (attention.py:415-450, :241-244; the guided sampler's
    differentiable U-Net evaluation) as FOUR GEMM launches and no gate row kernel:
        forward    g = geglu(LayerNorm(x) W1^T + b1)   one launch: LayerNorm fold, gate, and the projection saved for the backward (mode 2)
                   y = g W2^T + b2 [+ x]               one launch (residual in the epilogue)
        backward   dh = gate'(gy W2)                   one launch: the gate's backward in the epilogue of its producer (mode 3)
                   dx = LayerNorm'(dh W1) [+ gy]       one launch + the LayerNorm row kernel (which adds the residual branch's gradient)
    The unfused pair wrote the projection, read it for the gate (k_geglu), and in the backward wrote d/d(gated) and read it again next to the
    projection (k_geglu_bwd): 4.5 + 2.7 ms of a 220 ms guided step at 320x448.  Same operand roundings as the unfused kernels."""

    @staticmethod
    def forward(ctx, x2, ln_w, ln_b, cfg, w1, b1, w2, b2):
        ln, res_is_x = cfg
        Wd, c, s = _prepared([w1], [b1], ln, True, x2.dtype)
        st = None if ln is None else row_stats(x2, ln.eps)
        M, C2 = x2.shape[0], Wd.shape[0]
        h = torch.empty((M, C2), dtype=x2.dtype, device=x2.device)
        g = torch.empty((M, C2 // 2), dtype=x2.dtype, device=x2.device)
        _gate_gemm(x2, Wd, g, h, 2, bias=c, row_stats=st, col_sum=s)
        W2d, c2, _ = _prepared([w2], [b2], None, False, x2.dtype)
        y = gemm_nt(g, W2d, bias=c2, residual=x2 if res_is_x else None)
        ctx.save_for_backward(x2, h)
        ctx.cfg = (ln, res_is_x, w1, w2)
        return y

    @staticmethod
    def backward(ctx, gy):
        ln, res_is_x, w1, w2 = ctx.cfg
        x2, h = ctx.saved_tensors
        gy = gy if gy.stride(-1) == 1 else gy.contiguous()
        Wt2 = _transposed([w2], gy.dtype)                            # [C, dim]: d/d(gated) = gy W2
        dh = torch.empty_like(h)
        _gate_gemm(gy, Wt2, dh, h, 3)
        Wt1 = _transposed_gate_order(w1, gy.dtype)                   # [dim, 2C], columns in dh's block order
        extra = gy if res_is_x else None
        if ln is None:
            dx = gemm_nt(dh, Wt1, residual=extra)
        else:
            dxn = gemm_nt(dh, Wt1)
            dx = torch.empty_like(dxn)
            C = dxn.shape[-1]
            x2c = x2 if x2.is_contiguous() else x2.contiguous()
            ex = None if extra is None else (extra if extra.is_contiguous() else extra.contiguous())
            with ops._on(dxn.device):
                ops._check(ops.lib().gvd_layer_norm_bwd_add(_P(x2c.data_ptr()), _P(dxn.data_ptr()), _P(ln.weight.data_ptr()),
                                                            _P(None if ex is None else ex.data_ptr()), _P(dx.data_ptr()),
                                                            _LL(dxn.shape[0]), C, ctypes.c_float(ln.eps),
                                                            1 if dxn.dtype == torch.bfloat16 else 0, _P(ops._stream())))
        return dx, None, None, None, None, None, None, None


def feed_forward(x, w1, b1, w2, b2, *, ln=None, residual=None):
    """y = geglu(LayerNorm?(x) w1^T + b1) w2^T + b2 [+ residual] (attention.py:415-450).  Under autograd with frozen weights and the fused
    kernels available this is _FeedForwardFn (no gate row kernels); otherwise the two `linear` calls it stands for."""
    fused = (torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)
             and not any(t is not None and t.requires_grad for t in (w1, b1, w2, b2))
             and x.shape[-1] % 8 == 0 and w1.shape[0] % 32 == 0 and w2.shape[0] % 8 == 0 and w1.shape[0] == 2 * w2.shape[1]
             and (ln is None or _ln_kernel_ok(x, ln)) and (residual is None or residual is x)
             and hasattr(ops.lib(), "gvd_gemm_nt_gate") and os.environ.get("GVD_NO_FUSED_FF") is None)
    if not fused:
        return None
    x2 = _rows(x)
    y = _FeedForwardFn.apply(x2, None if ln is None else ln.weight, None if ln is None else ln.bias, (ln, residual is x), w1, b1, w2, b2)
    return y.reshape(*x.shape[:-1], y.shape[-1])


def _rows(x):
    """[..., K] -> a 2-D [M, K] view with a uniform row stride (a copy only if no such view exists)."""
    K = x.shape[-1]
    if x.dim() == 2:
        return x if x.stride(-1) == 1 else x.contiguous()
    try:
        x2 = x.view(-1, K)
    except RuntimeError:
        x2 = x.reshape(-1, K)
    return x2 if x2.stride(-1) == 1 else x2.contiguous()


def _ln_kernel_ok(x, ln):
    """The LayerNorm input-gradient row kernel covers 16-bit affine parameters of the activations' type, C <= 2048."""
    return (ln.weight is not None and ln.bias is not None and ln.weight.dtype == x.dtype and ln.bias.dtype == x.dtype
            and x.shape[-1] <= 2048 and not ln.weight.requires_grad and not ln.bias.requires_grad)


def _padded8(weight, bias):
    """Zero-padded [N8, K8] image (and bias) of a Linear / 1x1 convolution whose K or N is not a multiple of the GEMM's granule
    of 8 -- the KL-VAE's `post_quant_conv` / `quant_conv` (4 -> 4 and 8 -> 8 channels, autoencoder.py:97-107).  Cached on the
    weight until it is modified."""
    tag = (_tag(weight), _tag(bias))
    hit = getattr(weight, "_gvd_pad8", None)
    if hit is None or hit[0] != tag:
        with torch.no_grad():
            W = weight.detach().reshape(weight.shape[0], -1)
            W8 = F.pad(W, (0, (-W.shape[1]) % 8, 0, (-W.shape[0]) % 8)).contiguous()
            b8 = None if bias is None else F.pad(bias.detach(), (0, (-bias.shape[0]) % 8)).contiguous()
        hit = (tag, W8, b8)
        try:
            weight._gvd_pad8 = hit
        except AttributeError:
            pass
    return hit[1], hit[2]


def _hip_ok(x, weights):
    on_dev = ops._require_device(x, "linear")
    ok = on_dev and x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] % 8 == 0 and all(w.shape[0] % 8 == 0 for w in weights)
    if on_dev and not ok:
        ops._torch_form("linear", f"dtype {x.dtype}, K = {x.shape[-1]}, N = {[w.shape[0] for w in weights]} (the MFMA GEMM covers 16-bit rows with K, N multiples of 8)")
    return ok


def linear(x, weight, bias=None, *, ln=None, residual=None, geglu=False, grad_add=None, res_grad_to=None, grad_to=None):
    """[geglu](LayerNorm?(x) W^T + b) [+ residual] over the last dim of x.  weight [N, K] (or a 1x1 conv weight [N, K, 1(, 1)]);
    ln: an nn.LayerNorm-like module (weight, bias, eps) applied to x first; residual: tensor of the output's shape; geglu: the
    projection's two halves are value | gate (attention.py:415-423).  grad_add / res_grad_to / grad_to: ops.GradCell hand-overs
    under autograd (add the cell's gradient to d/dx in the backward kernel; deliver d/d residual, or d/dx, into a cell instead of
    to autograd's accumulation) -- ignored without autograd and on the torch-form branches."""
    cells = (grad_add, res_grad_to, grad_to)
    K = x.shape[-1]
    N = weight.shape[0]
    No = N // 2 if geglu else N
    lead = x.shape[:-1]
    if (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and (K % 8 or N % 8) and ln is None and residual is None and not geglu
            and not weight.requires_grad and (bias is None or not bias.requires_grad)):
        # narrow projections (K or N not a multiple of 8): zero-padded operands on the same MFMA GEMM; exact (the padding
        # contributes zeros), and the only library convolution left on the guided path (MIOpen igemm for the VAE's 4 -> 4
        # post_quant_conv) is gone with it
        W8, b8 = _padded8(weight, bias)
        xp = F.pad(x, (0, (-K) % 8)) if K % 8 else x
        return linear(xp, W8, b8)[..., :N]
    if not _hip_ok(x, [weight]) or (geglu and N % 32):
        h = x if ln is None else F.layer_norm(x, (K,), ln.weight, ln.bias, ln.eps)
        y = F.linear(h, weight.reshape(N, -1).to(h.dtype), None if bias is None else bias.to(h.dtype))
        if geglu:
            y = ops.geglu(y) if y.is_cuda and y.dtype in (torch.float16, torch.bfloat16) else ops.geglu_math(y)
        return y if residual is None else y + residual
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad))
    if torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad)):
        raise RuntimeError("lvdm_amd.gemm.linear: only the input gradient is implemented (freeze the weights)")
    if needs_grad:   # guided sampler: the same fused forward, input gradients through _FusedLinearFn
        if ln is not None and not _ln_kernel_ok(x, ln):
            h = ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
            return linear(h, weight, bias, residual=residual, geglu=geglu, res_grad_to=res_grad_to)
        if geglu:
            # the gate stays a separate row kernel under autograd: its backward needs the pre-activation, and recomputing it
            # (measured: guided step 1331 -> 1355 ms) costs more than writing it once (the projection itself is still ONE launch
            # with the LayerNorm folded in)
            h = _FusedLinearFn.apply(_rows(x), None, None if ln is None else ln.weight, None if ln is None else ln.bias,
                                     (ln, False, (grad_add, None, grad_to)), weight, bias)
            y = ops.geglu(h).reshape(*lead, No)
            return y if residual is None else y + residual
        r2 = None if residual is None else _rows(residual)
        y = _FusedLinearFn.apply(_rows(x), r2, None if ln is None else ln.weight, None if ln is None else ln.bias, (ln, False, cells), weight, bias)
        return y.reshape(*lead, No)
    x2 = _rows(x)
    Wd, c, s = _prepared([weight], [bias], ln, geglu, x.dtype)
    st = None if ln is None else row_stats(x2, ln.eps)
    r2 = None if residual is None else _rows(residual)
    return gemm_nt(x2, Wd, bias=c, row_stats=st, col_sum=s, residual=r2, geglu=geglu).reshape(*lead, No)


def linear_cat(x, weights, biases=None, *, ln=None, grad_add=None):
    """Several Linears of ONE input as one launch: returns [..., sum N_i]; column block i is Linear_i(LayerNorm?(x)).  The caller
    slices views (the attention kernels read them in place through their row strides)."""
    biases = [None] * len(weights) if biases is None else biases
    K = x.shape[-1]
    if not _hip_ok(x, weights):
        h = x if ln is None else (ops.layer_norm(x, ln.weight, ln.bias, ln.eps) if x.is_cuda else F.layer_norm(x, (K,), ln.weight, ln.bias, ln.eps))
        return torch.cat([linear(h, w, b) for w, b in zip(weights, biases)], dim=-1)
    if torch.is_grad_enabled() and x.requires_grad:
        if any(w.requires_grad for w in weights):
            raise RuntimeError("lvdm_amd.gemm.linear_cat: only the input gradient is implemented (freeze the weights)")
        if ln is not None and not _ln_kernel_ok(x, ln):
            h = ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
            return linear_cat(h, weights, biases)
        y = _FusedLinearFn.apply(_rows(x), None, None if ln is None else ln.weight, None if ln is None else ln.bias,
                                 (ln, False, (grad_add, None, None)), *weights, *biases)
        return y.reshape(*x.shape[:-1], y.shape[-1])
    x2 = _rows(x)
    Wd, c, s = _prepared(list(weights), list(biases), ln, False, x.dtype)
    st = None if ln is None else row_stats(x2, ln.eps)
    return gemm_nt(x2, Wd, bias=c, row_stats=st, col_sum=s).reshape(*x.shape[:-1], Wd.shape[0])
