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

from . import gemm, ops

_P, _LL = ctypes.c_void_p, ctypes.c_longlong
SCORE_BYTES = 96 << 20   # per score buffer (the backward holds two): both stay in the Infinity Cache


def _chunk_rows(B, other):
    rows = SCORE_BYTES // (2 * B * other)
    return max(256, rows // 256 * 256)


def _softmax_rows_(S, want_lse):
    rows, N = S.numel() // S.shape[-1], S.shape[-1]
    lse = torch.empty(rows, dtype=torch.float32, device=S.device) if want_lse else None
    with ops._on(S.device):
        ops._check(ops.lib().gvd_softmax_rows(_P(S.data_ptr()), _LL(N), _LL(rows), N, _P(None if lse is None else lse.data_ptr()),
                                              1 if S.dtype == torch.bfloat16 else 0, _P(ops._stream())))
    return lse


def _ds_(S, dP, lse, delta, rows_per_batch, by_col):
    rows, N = S.numel() // S.shape[-1], S.shape[-1]
    with ops._on(S.device):
        ops._check(ops.lib().gvd_attn_ds(_P(S.data_ptr()), _P(dP.data_ptr()), _P(lse.data_ptr()), _P(delta.data_ptr()), _LL(rows), N,
                                         _LL(rows_per_batch), int(by_col), 1 if S.dtype == torch.bfloat16 else 0, _P(ops._stream())))


def _forward(q, k, v, want_lse):
    B, Nq, d = q.shape
    Nk = k.shape[1]
    scale = d ** -0.5
    vT = v.transpose(1, 2).contiguous()                    # [B, d, Nk]: the second product's W operand (K-contiguous)
    out = torch.empty_like(q)
    lse = torch.empty(B, Nq, dtype=torch.float32, device=q.device) if want_lse else None
    step = _chunk_rows(B, Nk)
    for m0 in range(0, Nq, step):
        m1 = min(Nq, m0 + step)
        S = gemm.gemm_nt(q[:, m0:m1], k, alpha=scale)       # [B, Mc, Nk] scaled scores, 16 bit
        l = _softmax_rows_(S, want_lse)                     # -> P in place
        if want_lse:
            lse[:, m0:m1] = l.view(B, m1 - m0)
        gemm.gemm_nt(S, vT, out=out[:, m0:m1])
    return out, lse


class _WideAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = (t if (t.stride(-1) == 1 and t.stride(1) % 8 == 0) else t.contiguous() for t in (q, k, v))
        out, lse = _forward(q, k, v, True)
        ctx.save_for_backward(q, k, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        g = g.contiguous()
        B, Nq, d = q.shape
        Nk = k.shape[1]
        scale = d ** -0.5
        delta = (g.float() * out.float()).sum(-1)           # [B, Nq]
        kT, qT, gT = (t.transpose(1, 2).contiguous() for t in (k, q, g))
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        step = _chunk_rows(B, Nk)
        for m0 in range(0, Nq, step):                       # pass over query chunks: dq
            m1 = min(Nq, m0 + step)
            S = gemm.gemm_nt(q[:, m0:m1], k, alpha=scale)
            dP = gemm.gemm_nt(g[:, m0:m1], v)               # dO v^T  [B, Mc, Nk]
            _ds_(S, dP, lse[:, m0:m1].contiguous(), delta[:, m0:m1].contiguous(), m1 - m0, False)
            gemm.gemm_nt(dP, kT, alpha=scale, out=dq[:, m0:m1])
        step = _chunk_rows(B, Nq)
        for n0 in range(0, Nk, step):                       # pass over key chunks: dk, dv (transposed scores)
            n1 = min(Nk, n0 + step)
            St = gemm.gemm_nt(k[:, n0:n1], q, alpha=scale)  # [B, Nc, Nq]
            dPt = gemm.gemm_nt(v[:, n0:n1], g)
            _ds_(St, dPt, lse, delta, n1 - n0, True)
            gemm.gemm_nt(St, gT, out=dv[:, n0:n1])          # P^T dO
            gemm.gemm_nt(dPt, qT, alpha=scale, out=dk[:, n0:n1])
        return dq, dk, dv


def supported(q, k, v):
    d = q.shape[-1]
    return (q.is_cuda and q.dtype in (torch.float16, torch.bfloat16) and k.dtype == q.dtype and v.dtype == q.dtype and q.dim() == 3
            and d % 8 == 0 and k.shape[1] % 8 == 0 and q.shape[1] % 8 == 0 and k.shape[1] <= 16384 and q.shape[1] <= 16384)


def attention(q, k, v):
    """softmax(q k^T / sqrt(d)) v, one head.  q [B, Nq, d], k / v [B, Nk, d], 16-bit on a ROCm device; Nq, Nk, d multiples of 8."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _WideAttention.apply(q, k, v)
    q, k, v = (t if (t.stride(-1) == 1 and t.stride(1) % 8 == 0) else t.contiguous() for t in (q, k, v))
    return _forward(q, k, v, False)[0]
