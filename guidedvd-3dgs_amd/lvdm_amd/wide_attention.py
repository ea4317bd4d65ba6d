"""Single-head attention with a WIDE head (the VAE's mid block: d = 512 over h*w tokens, ae_modules.py:26-78) as chunked GEMMs.

The U-Net's d = 64 heads run the flash kernels (ops.attention).  A 512-wide head needs 512 accumulator columns per query row -- a
different kernel -- and costs 3 % of a guided step, so it is built from the parts that exist: for a chunk of queries the scores
S = q_c k^T / sqrt(d) are ONE MFMA GEMM (gvd_gemm_nt) into a 16-bit buffer sized to stay in the 256 MB Infinity Cache, a row kernel
turns them into P in place (fp32 softmax, log-sum-exp kept), and o_c = P v is the second GEMM.  No N x N matrix in HBM-sized
fp32, no library GEMM, no torch softmax.  The backward (guided sampler: 25 decodes per step) recomputes S per chunk like the flash
backward does: a pass over query chunks gives dq, a pass over key chunks gives dk and dv (no accumulation across chunks, every
output rounded once).  Scores are rounded to 16 bit before the softmax -- exactly what the reference's autocast `torch.bmm` does.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import gemm, ops

_P, _LL = ctypes.c_void_p, ctypes.c_longlong
# Per score buffer (the backward holds two).  Round 3 sized it to keep both in the 256 MB Infinity Cache (96 MB: 768-row chunks of
# the 25 x 2240-token mid block); measured in round 4 (tests/scripts/r4_wide_attn_chunks.py, forward + backward): 25 x 2240 tokens
# 5.2 ms at 96 MB, 3.44 ms unchunked (256 MB+); 5 x 9216 tokens 15.5 ms at 96 MB, 8.0 at 512 MB, 7.7 at 1 GB -- fewer, better
# filled GEMM launches beat cache residency of the scores.
SCORE_BYTES = 512 << 20


def _chunk_rows(B, other):
    rows = SCORE_BYTES // (2 * B * other)
    return max(256, rows // 256 * 256)


def _softmax_rows_(S, n_valid, want_lse):
    rows, N = S.numel() // S.shape[-1], S.shape[-1]
    lse = torch.empty(rows, dtype=torch.float32, device=S.device) if want_lse else None
    with ops._on(S.device):
        ops._check(ops.lib().gvd_softmax_rows(_P(S.data_ptr()), _LL(N), _LL(rows), N, int(n_valid), _P(None if lse is None else lse.data_ptr()),
                                              1 if S.dtype == torch.bfloat16 else 0, _P(ops._stream())))
    return lse


def _ds_(S, dP, lse, delta, n_valid, rows_per_batch, by_col):
    rows, N = S.numel() // S.shape[-1], S.shape[-1]
    with ops._on(S.device):
        ops._check(ops.lib().gvd_attn_ds(_P(S.data_ptr()), _P(dP.data_ptr()), _P(lse.data_ptr()), _P(delta.data_ptr()), _LL(rows), N,
                                         int(n_valid), _LL(rows_per_batch), int(by_col), 1 if S.dtype == torch.bfloat16 else 0,
                                         _P(ops._stream())))


def _pad_tokens(t):
    """Token count -> a multiple of 8 (the GEMM's N / K granule) with zero rows; they are masked out by the row kernels."""
    pad = (-t.shape[1]) % 8
    return t if pad == 0 else F.pad(t, (0, 0, 0, pad))


def _inplace_ok(t):
    return t.stride(-1) == 1 and t.stride(1) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0


MAX_ROW = 16384          # keys per softmax row (gvd_softmax_rows keeps a row in registers); longer rows go in key chunks


def _forward(q, k, v, want_lse, scale=None):
    """q [B, Nq, d], k / v [B, Nk, d] (token counts already multiples of 8 where it matters); nk_valid keys are real."""
    B, Nq, d = q.shape
    scale = d ** -0.5 if scale is None else float(scale)
    if k.shape[1] > MAX_ROW:
        # more keys than one softmax row holds (a 1024 x 1024 image is exactly 16384 tokens and still takes the single-row path;
        # anything above runs per key chunk and the chunks' normalised outputs are merged with their log-sum-exps in fp32)
        outs, lses = [], []
        for n0 in range(0, k.shape[1], MAX_ROW):
            o_c, l_c = _forward(q, k[:, n0:n0 + MAX_ROW], v[:, n0:n0 + MAX_ROW], True, scale)
            outs.append(o_c.float())
            lses.append(l_c)
        L = torch.stack(lses)                                       # [chunks, B, Nq]
        lse = torch.logsumexp(L, dim=0)
        w = torch.exp(L - lse).unsqueeze(-1)
        out = sum(w[i] * outs[i] for i in range(len(outs))).to(q.dtype)
        return out, (lse if want_lse else None)
    nk_valid = k.shape[1]
    k, v = _pad_tokens(k), _pad_tokens(v)
    Nk = k.shape[1]
    vT = v.transpose(1, 2).contiguous()                    # [B, d, Nk]: the second product's W operand (K-contiguous)
    out = torch.empty((B, Nq, d), dtype=q.dtype, device=q.device)
    lse = torch.empty(B, Nq, dtype=torch.float32, device=q.device) if want_lse else None
    step = _chunk_rows(B, Nk)
    for m0 in range(0, Nq, step):
        m1 = min(Nq, m0 + step)
        S = gemm.gemm_nt(q[:, m0:m1], k, alpha=scale)       # [B, Mc, Nk] scaled scores, 16 bit
        l = _softmax_rows_(S, nk_valid, want_lse)           # -> P in place
        if want_lse:
            lse[:, m0:m1] = l.view(B, m1 - m0)
        gemm.gemm_nt(S, vT, out=out[:, m0:m1])
    return out, lse


class _WideAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale=None):
        q, k, v = (t if _inplace_ok(t) else t.contiguous() for t in (q, k, v))
        out, lse = _forward(q, k, v, True, scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        B, nq_valid, d = q.shape
        nk_valid = k.shape[1]
        scale = d ** -0.5 if ctx.scale is None else float(ctx.scale)
        delta = (g.float() * out.float()).sum(-1)           # [B, Nq]
        # token counts to multiples of 8: padded keys are masked by n_valid, padded queries by lse = +inf (P = 0 there)
        q, g, k, v = _pad_tokens(q), _pad_tokens(g.contiguous()), _pad_tokens(k), _pad_tokens(v)
        Nq, Nk = q.shape[1], k.shape[1]
        if Nq != nq_valid:
            lse = F.pad(lse, (0, Nq - nq_valid), value=float("inf"))
            delta = F.pad(delta, (0, Nq - nq_valid))
        lse, delta = lse.contiguous(), delta.contiguous()
        kT, qT, gT = (t.transpose(1, 2).contiguous() for t in (k, q, g))
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        step = _chunk_rows(B, Nk)
        for m0 in range(0, Nq, step):                       # pass over query chunks: dq
            m1 = min(Nq, m0 + step)
            S = gemm.gemm_nt(q[:, m0:m1], k, alpha=scale)
            dP = gemm.gemm_nt(g[:, m0:m1], v)               # dO v^T  [B, Mc, Nk]
            _ds_(S, dP, lse[:, m0:m1].contiguous(), delta[:, m0:m1].contiguous(), nk_valid, m1 - m0, False)
            gemm.gemm_nt(dP, kT, alpha=scale, out=dq[:, m0:m1])
        step = _chunk_rows(B, Nq)
        for n0 in range(0, Nk, step):                       # pass over key chunks: dk, dv (transposed scores)
            n1 = min(Nk, n0 + step)
            St = gemm.gemm_nt(k[:, n0:n1], q, alpha=scale)  # [B, Nc, Nq]
            dPt = gemm.gemm_nt(v[:, n0:n1], g)
            _ds_(St, dPt, lse, delta, Nq, n1 - n0, True)
            gemm.gemm_nt(St, gT, out=dv[:, n0:n1])          # P^T dO
            gemm.gemm_nt(dPt, qT, alpha=scale, out=dk[:, n0:n1])
        return dq[:, :nq_valid], dk[:, :nk_valid], dv[:, :nk_valid], None


def supported(q, k, v):
    """Any token count (rows longer than MAX_ROW keys run in key chunks; the backward's element kernel has no row limit)."""
    d = q.shape[-1]
    return (q.is_cuda and q.dtype in (torch.float16, torch.bfloat16) and k.dtype == q.dtype and v.dtype == q.dtype and q.dim() == 3
            and d % 8 == 0)


def attention(q, k, v, scale=None):
    """softmax(q k^T * scale) v, one head (scale: 1 / sqrt(d) by default).  q [B, Nq, d], k / v [B, Nk, d], 16-bit on a ROCm
    device; d a multiple of 8."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _WideAttention.apply(q, k, v, scale)
    q, k, v = (t if _inplace_ok(t) else t.contiguous() for t in (q, k, v))
    return _forward(q, k, v, False, scale)[0]


def attention_heads(q, k, v, heads, frame_major=False):
    """Every 16-bit shape the d = 64 flash kernels do not cover: `heads` heads of any width (the reference's `num_heads`-style
    U-Net configurations have d = 40 / 80 / 160, openaimodel3d.py:404-412) as `B * heads` single-head problems on the chunked
    GEMM path -- head channels split out (one copy per operand), padded with zero channels to the GEMM's K granule of 8 (the
    softmax scale stays 1 / sqrt(d) of the true width).  q [B, Nq, h*d], k / v [B or 1, Nk, h*d]; frame_major: [N, B, h*d]."""
    if frame_major:
        q, k, v = (t.transpose(0, 1) for t in (q, k, v))
    B, Nq, C = q.shape
    d = C // heads
    if k.shape[0] != B:
        k, v = k.expand(B, -1, -1), v.expand(B, -1, -1)
    if heads == 1 and d % 8 == 0:
        o = attention(q, k, v)
    else:
        pad = (-d) % 8

        def split(t):
            t = t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(B * heads, t.shape[1], d)
            return F.pad(t, (0, pad)) if pad else t.contiguous()
        o = attention(split(q), split(k), split(v), scale=d ** -0.5)[..., :d]
        o = o.reshape(B, heads, Nq, d).permute(0, 2, 1, 3).reshape(B, Nq, C)
    return o.transpose(0, 1) if frame_major else o
