"""Hot operators of the diffusion path.

On a ROCm device the fp16/bf16 (autocast) path runs the hand-written HIP kernels of
csrc/diffusion_kernels.hip through the C-ABI library lib/libgvd_diffusion.so; there is no silent CPU
fallback: CPU tensors raise unless a test has explicitly enabled the reference math with
`use_reference_math(True)` (the `-m "not gpu"` tests of the host logic / module structure do that).

    attention(q, k, v, heads)          softmax(q k^T / sqrt(d)) v per head, q [B,Nq,h*d], k/v [B,Nk,h*d]
    group_norm(x, groups, w, b, eps, silu)   fp32 statistics (lvdm/basics.py:76-78), optional fused SiLU
    ddim_step(...)                     the whole no-grad DDIM update of ddim.py:208-280 in one kernel
"""
import ctypes
import os
import threading

import torch
import torch.nn.functional as F

_REFERENCE_MATH = False
_LIB = None
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GVD_DIFFUSION_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libgvd_diffusion.so")  # env: A/B builds


def use_reference_math(flag):
    """Tests and bench.py's cpu_baseline leg ONLY: allow CPU tensors through plain torch math (never enabled by the product)."""
    global _REFERENCE_MATH
    _REFERENCE_MATH = bool(flag)


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOOP = _Noop()


def _on(dev):
    """Device guard that is free when `dev` is already the current device (torch.cuda.device() costs ~25 us per use)."""
    idx = dev.index
    return _NOOP if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(dev)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: build the HIP extensions first "
                               f"(python -c 'import __graft_entry__ as g; g.build()')")
        L = ctypes.CDLL(_LIB_PATH)
        L.gvd_diff_last_error.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def _require_device(t, what):
    if t.is_cuda:
        return True
    if _REFERENCE_MATH:
        return False
    raise RuntimeError(f"lvdm_amd.ops.{what}: tensor on {t.device}; this build has no CPU path "
                       f"(tests may call ops.use_reference_math(True))")


_WARNED = set()
_FALLBACK = threading.local()


def torch_fallback_policy():
    """'error' (the default) or 'warn': what happens when a DEVICE tensor is about to take the plain-torch form of an op (torch's library
    kernels) instead of this package's hand-written one -- fp32 activations, a shape no kernel covers.  The thread's innermost
    `allow_torch_fallback(...)` context decides, then GVD_TORCH_FALLBACK in the environment, then 'error'."""
    v = getattr(_FALLBACK, "stack", None)
    if v:
        return v[-1]
    return "warn" if os.environ.get("GVD_TORCH_FALLBACK", "error").lower() in ("warn", "allow", "1") else "error"


class allow_torch_fallback:
    """`with lvdm_amd.allow_torch_fallback():` -- inside, device tensors this package's kernels do not cover (fp32 activations of a parity
    run; the fp32 Resampler / CLIP towers of a caller who did not `.half()` them) run torch's own kernels with one RuntimeWarning per
    (op, reason).  Outside, the same call RAISES: a drop-in user cannot end up on a second backend unknowingly (verdict r5, item 8).
    `allow_torch_fallback(False)` restores the strict default inside an allowing context.  Per thread, re-entrant."""

    def __init__(self, allow=True):
        self.mode = "warn" if allow else "error"

    def __enter__(self):
        st = getattr(_FALLBACK, "stack", None)
        if st is None:
            st = _FALLBACK.stack = []
        st.append(self.mode)
        return self

    def __exit__(self, *exc):
        _FALLBACK.stack.pop()
        return False


def checkpoint_dtype_tower(forward):
    """Decorator for the forward of the ONCE-PER-VIDEO conditioning modules that use this package's operators (the Resampler and the
    image projection; the CLIP towers of clip.py are plain torch modules throughout and never reach the gate).  The reference runs them in the
    checkpoint's fp32 under autocast and never `.half()`s them; a drop-in user does not either, and the reference's own
    utils_vc/diffusion_utils.py:140-160 calls them directly -- there is no place for the user to opt in.  So these two modules are the NAMED exception
    to the strict default: with fp32 parameters on a device they run torch's kernels (one RuntimeWarning per (op, reason)); in 16 bit
    they run this package's kernels like everything else.  An explicit `allow_torch_fallback(False)` around the call still wins.  The hot loop -- U-Net,
    VAE, samplers -- never passes through here."""
    import functools

    @functools.wraps(forward)
    def wrapped(self, *args, **kwargs):
        if getattr(_FALLBACK, "stack", None):          # the caller stated a policy: keep it
            return forward(self, *args, **kwargs)
        p = next(self.parameters(), None)
        if p is not None and p.is_cuda and p.dtype == torch.float32:
            with allow_torch_fallback():
                return forward(self, *args, **kwargs)
        return forward(self, *args, **kwargs)
    return wrapped


def _torch_form(op, why):
    """A DEVICE tensor is about to take the plain-torch form of `op` instead of the hand-written kernel (fp32 activations of a
    parity run, a head size the MFMA kernel does not cover, ...).  Never silent and, by default, not allowed: RuntimeError unless the
    caller opted in (`allow_torch_fallback()` / GVD_TORCH_FALLBACK=warn), then one RuntimeWarning per (op, reason) and process."""
    if torch_fallback_policy() == "error":
        raise RuntimeError(f"lvdm_amd.ops.{op}: {why} -- no hand-written kernel covers this call, and the plain-torch form (torch's library "
                           f"kernels) is not entered silently.  Run the module in 16 bit (.half() / .bfloat16(), as the reference's autocast "
                           f"does), or opt in: `with lvdm_amd.allow_torch_fallback(): ...` / GVD_TORCH_FALLBACK=warn")
    key = (op, why)
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(f"lvdm_amd.ops.{op}: {why} -> plain torch form, not the HIP kernel", RuntimeWarning, stacklevel=3)


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"gvd_diffusion error {rc}: {lib().gvd_diff_last_error().decode()}")


GRAD_CELLS = os.environ.get("GVD_GRAD_CELLS", "1") == "1"   # 0: leave every fan-in sum to autograd (A/B runs, tests)


class GradCell:
    """Hand-over of a gradient between two autograd nodes that both received the SAME tensor x: one as the input of a
    normalisation (x -> LayerNorm / GroupNorm -> ...), one as a residual (`... + x`).  Autograd would sum their two gradients with
    an elementwise kernel where the branches rejoin -- 3 passes over the activation per fork, ~170 forks per differentiable U-Net
    evaluation, ~5 % of a guided DDIM step.  Instead the residual-side node PUTS its gradient into the cell (and reports no
    gradient to autograd) and the norm-side node, whose backward necessarily runs later, TAKES it and adds it inside its last
    kernel (gvd_layer_norm_bwd_add, gvd_group_norm_bwd_apply_add, the dgrad GEMM's residual epilogue): fp32 sum, one rounding.

    Protocol (never loses a gradient): the taker ARMS the cell in its forward -- only when it is on the kernel path that will
    take in backward; a putter that finds the cell unarmed at forward time, or already taken at backward time (a second backward
    pass over a retained graph), returns its gradient to autograd as usual."""
    __slots__ = ("g", "armed", "taken")

    def __init__(self):
        self.g, self.armed, self.taken = None, False, False

    def arm(self):
        self.armed = GRAD_CELLS
        return self

    def put(self, g):
        if not self.armed or self.taken:
            return False
        self.g = g if self.g is None else self.g + g
        return True

    def take(self, like=None):
        self.taken = True
        g, self.g = self.g, None
        if g is not None and like is not None:
            g = g.reshape(like.shape)
            g = g if g.is_contiguous() else g.contiguous()
        return g


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw hipStream_t of torch's current stream on the current device (the private fast path costs ~0.3 us, the public
    torch.cuda.current_stream() object ~9 us)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# --------------------------------------------------------------------------------------------------
def attention_math(q, k, v, heads, frame_major=False):
    """Explicit form (the reference's einsum path, attention.py:101-135), fp32 softmax."""
    if frame_major:  # [N, B, C] -> [B, N, C] and back
        return attention_math(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), heads).transpose(0, 1).contiguous()
    B, Nq, C = q.shape
    d = C // heads
    qh = q.reshape(B, Nq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, k.shape[1], heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, v.shape[1], heads, d).permute(0, 2, 1, 3)
    sim = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
    p = sim.float().softmax(dim=-1).to(vh.dtype)
    return torch.matmul(p, vh).permute(0, 2, 1, 3).reshape(B, Nq, C)


class _FlashAttention(torch.autograd.Function):
    """fp16/bf16 MFMA flash attention, forward and backward both hand-written kernels (csrc/diffusion_kernels.hip,
    csrc/attention_backward.hip).  The backward runs in the guided sampler only."""

    @staticmethod
    def forward(ctx, q, k, v, heads, frame_major):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, lse = _hip_attention_fwd(q, k, v, heads, frame_major, want_lse=True)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.heads, ctx.frame_major = heads, frame_major
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        # keys / values without a gradient (the cross-attention's frame-invariant context): dQ only -- the dK / dV kernel walks ALL
        # queries of a batch entry per 128 keys, which is what made one-context-per-sample batches expensive
        need_kv = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        return _hip_attention_bwd(q, k, v, out, g, lse, ctx.heads, ctx.frame_major, need_kv=need_kv) + (None, None)


class _PackedSelfAttention(torch.autograd.Function):
    """Self-attention on a packed projection qkv [..., 3 C] (q | k | v column blocks, the output of ONE fused GEMM) under
    autograd: the forward reads the three blocks in place, the backward kernels write dq / dk / dv straight into the packed
    gradient -- no slice copies going in, no zero-fill + scatter coming back (they were ~55 ms of a guided step).
    A frame-major input may carry a leading sample dimension ([b, T, pixels, 3 C], the temporal transformer of a batched CFG pair):
    the samples are separate launches INSIDE this node (a frame-major view cannot fold b into the pixel dimension), so the graph
    has no per-sample select / stack nodes -- each of those was a zero-fill + copy of the whole packed gradient plus an accumulation."""

    @staticmethod
    def forward(ctx, qkv, heads, frame_major):
        qkv = qkv if (qkv.is_contiguous()) else qkv.contiguous()
        C = qkv.shape[-1] // 3
        samples = [qkv[i] for i in range(qkv.shape[0])] if qkv.dim() == 4 else [qkv]
        out = torch.empty(qkv.shape[:-1] + (C,), dtype=qkv.dtype, device=qkv.device)
        lses = []
        for i, s3 in enumerate(samples):
            _, lse = _hip_attention_fwd(s3[..., :C], s3[..., C:2 * C], s3[..., 2 * C:], heads, frame_major, want_lse=True,
                                        out=out[i] if qkv.dim() == 4 else out)
            lses.append(lse)
        ctx.save_for_backward(qkv, out, *lses)
        ctx.cfg = (heads, frame_major)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, out, *lses = ctx.saved_tensors
        heads, frame_major = ctx.cfg
        g = g.contiguous()
        C = qkv.shape[-1] // 3
        dqkv = torch.empty_like(qkv)
        four = qkv.dim() == 4
        LL, P = ctypes.c_longlong, ctypes.c_void_p
        for i, lse in enumerate(lses):
            s3, d3, o3, g3 = (qkv[i], dqkv[i], out[i], g[i]) if four else (qkv, dqkv, out, g)
            q, k, v = s3[..., :C], s3[..., C:2 * C], s3[..., 2 * C:]
            dq, dk, dv = d3[..., :C], d3[..., C:2 * C], d3[..., 2 * C:]
            B = q.shape[1] if frame_major else q.shape[0]
            Nq = q.shape[0] if frame_major else q.shape[1]
            d = C // heads
            q_bs, q_rs = _view_strides(q, frame_major, B)
            o_bs, o_rs = _view_strides(o3, frame_major, B)
            delta = torch.empty_like(lse)
            with _on(q.device):
                rc = lib().gvd_attention_bwd_ex(P(q.data_ptr()), P(k.data_ptr()), P(v.data_ptr()), P(o3.data_ptr()), P(g3.data_ptr()),
                                                P(lse.data_ptr()), P(delta.data_ptr()), P(dq.data_ptr()), P(dk.data_ptr()), P(dv.data_ptr()),
                                                B, heads, Nq, Nq, d, ctypes.c_float(d ** -0.5), LL(q_bs), LL(q_rs), LL(q_bs), LL(q_rs),
                                                LL(o_bs), LL(o_rs), 1 if q.dtype == torch.bfloat16 else 0, P(_stream()))
            _check(rc)
        return dqkv, None, None


def self_attention_packed(qkv, heads, frame_major=False):
    """softmax(q k^T / sqrt(d)) v for q | k | v given as the column blocks of one tensor [B, N, 3 h d] (frame_major: [N, B, 3 h d], or
    [b, N, B, 3 h d] -- b samples of a frame-major problem, e.g. the temporal attention of a batched CFG pair)."""
    C = qkv.shape[-1] // 3
    on_dev = _require_device(qkv, "attention")
    hip = on_dev and qkv.dtype in (torch.float16, torch.bfloat16) and C // heads == 64
    if hip and torch.is_grad_enabled() and qkv.requires_grad:
        return _PackedSelfAttention.apply(qkv, heads, frame_major)
    if qkv.dim() == 4:
        if hip:
            qc = qkv if qkv.is_contiguous() else qkv.contiguous()
            out = torch.empty(qc.shape[:-1] + (C,), dtype=qc.dtype, device=qc.device)
            for i in range(qc.shape[0]):
                _hip_attention_fwd(qc[i][..., :C], qc[i][..., C:2 * C], qc[i][..., 2 * C:], heads, frame_major, out=out[i])
            return out
        return torch.stack([attention(qkv[i][..., :C], qkv[i][..., C:2 * C], qkv[i][..., 2 * C:], heads, frame_major)
                            for i in range(qkv.shape[0])], 0)
    return attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, frame_major)


def _attn_geometry(q, k, heads, frame_major):
    C = q.shape[-1]
    if frame_major:
        Nq, B, Nk = q.shape[0], q.shape[1], k.shape[0]
        return B, Nq, Nk, C // heads, (C, B * C, C, B * C)
    B, Nq, Nk = q.shape[0], q.shape[1], k.shape[1]
    return B, Nq, Nk, C // heads, (Nq * C, C, Nk * C, C)


def _view_strides(t, frame_major, B):
    """(batch stride, row stride) of a [B, N, C] (or frame-major [N, B, C]) tensor read in place; a size-1 batch dim of K / V
    broadcasts (stride 0).  The channel dim must be contiguous and the strides multiples of 8 elements."""
    bd, rd = (1, 0) if frame_major else (0, 1)
    bs = 0 if (t.shape[bd] == 1 and B > 1) else t.stride(bd)
    rs = t.stride(rd)
    if t.shape[rd] == 1:
        rs = t.shape[-1]
    if t.shape[bd] == 1 and B == 1:
        bs = 0
    return bs, rs


def _readable_in_place(t):
    return t.stride(-1) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0


def _hip_attention_fwd(q, k, v, heads, frame_major=False, want_lse=False, accum=None, accum_scale=1.0, out=None):
    """q [B,Nq,C], k/v [B,Nk,C] (k/v with B = 1 are shared by all batch entries); with frame_major=True the tensors are
    [N, B, C] (sequence outermost: the T frames of B pixels).  q / k / v are read IN PLACE through their strides -- column blocks
    of a packed [.., q | k | v] projection, broadcast contexts -- as long as the channel dim is contiguous; k and v must share
    their strides.  accum: out = accum + accum_scale * attention (same shape as the dense output)."""
    q, k, v = (t if _readable_in_place(t) else t.contiguous() for t in (q, k, v))
    B = q.shape[1] if frame_major else q.shape[0]
    Nq = q.shape[0] if frame_major else q.shape[1]
    Nk = k.shape[0] if frame_major else k.shape[1]
    C = q.shape[-1]
    d = C // heads
    q_bs, q_rs = _view_strides(q, frame_major, B)
    kv_bs, kv_rs = _view_strides(k, frame_major, B)
    if _view_strides(v, frame_major, B) != (kv_bs, kv_rs):
        v = v.contiguous()
        k = k.contiguous()
        kv_bs, kv_rs = _view_strides(k, frame_major, B)
    if out is None:
        out = torch.empty(q.shape, dtype=q.dtype, device=q.device)   # (out: a caller's dense [like q] slot, e.g. one sample of a batch)
    o_bs, o_rs = _view_strides(out, frame_major, B)
    if accum is not None:
        accum = accum.contiguous()
        if accum.shape != out.shape or accum.dtype != out.dtype:
            raise ValueError("attention: accum must have the output's shape and dtype")
    lse = torch.empty(B * heads * Nq, dtype=torch.float32, device=q.device) if want_lse else None
    is_bf16 = 1 if q.dtype == torch.bfloat16 else 0
    LL, P = ctypes.c_longlong, ctypes.c_void_p
    with _on(q.device):
        rc = lib().gvd_attention_fwd_ex(P(q.data_ptr()), P(k.data_ptr()), P(v.data_ptr()), P(out.data_ptr()),
                                        B, heads, Nq, Nk, d, ctypes.c_float(d ** -0.5), LL(q_bs), LL(q_rs), LL(kv_bs), LL(kv_rs),
                                        LL(o_bs), LL(o_rs), P(None if accum is None else accum.data_ptr()), ctypes.c_float(accum_scale),
                                        P(lse.data_ptr() if want_lse else None), is_bf16, P(_stream()))
    _check(rc)
    return (out, lse) if want_lse else out


def _hip_attention_bwd(q, k, v, out, g, lse, heads, frame_major=False, need_kv=True):
    g = g.contiguous()
    B, Nq, Nk, d, strides = _attn_geometry(q, k, heads, frame_major)
    dq = torch.empty_like(q)
    dk, dv = (torch.empty_like(k), torch.empty_like(v)) if need_kv else (None, None)
    delta = torch.empty_like(lse)
    LL, P = ctypes.c_longlong, ctypes.c_void_p
    with _on(q.device):
        rc = lib().gvd_attention_bwd_strided(P(q.data_ptr()), P(k.data_ptr()), P(v.data_ptr()), P(out.data_ptr()),
                                             P(g.data_ptr()), P(lse.data_ptr()), P(delta.data_ptr()), P(dq.data_ptr()),
                                             P(None if dk is None else dk.data_ptr()), P(None if dv is None else dv.data_ptr()), B, heads, Nq, Nk, d,
                                             ctypes.c_float(d ** -0.5), *(LL(s) for s in strides),
                                             1 if q.dtype == torch.bfloat16 else 0, P(_stream()))
    _check(rc)
    return dq, dk, dv


def attention(q, k, v, heads, frame_major=False, accum=None, accum_scale=1.0):
    """softmax(q k^T / sqrt(d)) v per head.  q [B,Nq,h*d], k/v [B,Nk,h*d] (or [1,Nk,h*d]: shared by the batch); frame_major:
    [N,B,h*d] (see above).  accum / accum_scale: returns accum + accum_scale * attention (fused in the kernel's epilogue)."""
    on_dev = _require_device(q, "attention")
    d = q.shape[-1] // heads
    if on_dev and q.dtype in (torch.float16, torch.bfloat16) and d == 64 and k.dtype == q.dtype and v.dtype == q.dtype:
        if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad or (accum is not None and accum.requires_grad)):
            B = q.shape[1] if frame_major else q.shape[0]
            bd = 1 if frame_major else 0
            if k.shape[bd] != B:   # broadcast context: the backward kernels address one K / V per batch entry
                k, v = k.expand(*[B if i == bd else -1 for i in range(3)]), v.expand(*[B if i == bd else -1 for i in range(3)])
            o = _FlashAttention.apply(q, k, v, heads, frame_major)
            return o if accum is None else accum + accum_scale * o
        return _hip_attention_fwd(q, k, v, heads, frame_major, accum=accum, accum_scale=accum_scale)
    if on_dev and q.dtype in (torch.float16, torch.bfloat16) and k.dtype == q.dtype and v.dtype == q.dtype and not _REFERENCE_MATH:
        # every other 16-bit shape: one wide head (the VAE's d = 512 mid block) or heads of a width other than 64 (the reference's
        # num_heads-style U-Net configurations: d = 40 / 80 / 160) -- chunked MFMA GEMMs + row kernels, any token count
        from . import wide_attention
        o = wide_attention.attention_heads(q, k, v, heads, frame_major)
        return o if accum is None else accum + accum_scale * o
    # fp32 tensors (the parity runs): the explicit form
    if on_dev:
        _torch_form("attention", f"dtype {q.dtype}, head dim {d} (the MFMA kernels cover 16-bit inputs)")
    if k.shape[0 if not frame_major else 1] != q.shape[0 if not frame_major else 1]:
        bd = 1 if frame_major else 0
        k = k.expand(*[q.shape[bd] if i == bd else -1 for i in range(3)])
        v = v.expand(*[q.shape[bd] if i == bd else -1 for i in range(3)])
    o = attention_math(q, k, v, heads, frame_major)
    return o if accum is None else accum + accum_scale * o


# --------------------------------------------------------------------------------------------------
def group_norm_math(x, groups, weight, bias, eps=1e-5, silu=False, channels_last=False, group=None):
    """Eager form: statistics and affine in fp32 (GroupNormSpecific, lvdm/basics.py:76-86), result cast back.
    `group`: process group whose ranks each hold a slice of every (sample, group)'s elements -- statistics are
    summed across it (differentiably), see parallel.py."""
    xc = x.movedim(-1, 1) if channels_last else x   # [N, ..., C] -> [N, C, ...]
    if group is None:
        y = F.group_norm(xc.float(), groups, None if weight is None else weight.float(),
                         None if bias is None else bias.float(), eps)
    else:
        import torch.distributed.nn.functional as dist_fn
        N, C = xc.shape[0], xc.shape[1]
        xg = xc.float().reshape(N, groups, -1)
        part = torch.stack([xg.sum(-1), (xg * xg).sum(-1), torch.full((N, groups), float(xg.shape[-1]), device=x.device)])
        s1, s2, cnt = dist_fn.all_reduce(part, group=group)
        mean = s1 / cnt
        rstd = torch.rsqrt((s2 / cnt - mean * mean).clamp_min(0) + eps)
        y = ((xg - mean[..., None]) * rstd[..., None]).reshape(xc.shape)
        shp = (1, C) + (1,) * (xc.dim() - 2)
        if weight is not None:
            y = y * weight.float().reshape(shp)
        if bias is not None:
            y = y + bias.float().reshape(shp)
    if silu:
        y = F.silu(y)
    y = y.to(x.dtype)
    return y.movedim(1, -1).contiguous() if channels_last else y


def _f32_param(p):
    """fp32 copy of a (frozen) 16-bit norm parameter, cached ON the parameter object until it is modified in place
    (two tiny cast launches per GroupNorm call otherwise: ~4000 per U-Net forward)."""
    if p.dtype == torch.float32:
        return p.detach().contiguous()
    if p.requires_grad:
        return p.float().contiguous()
    hit = getattr(p, "_gvd_f32", None)
    if hit is None or hit[0] != p._version or hit[1] != p.data_ptr() or hit[2].device != p.device:
        hit = (p._version, p.data_ptr(), p.detach().float().contiguous())
        try:
            p._gvd_f32 = hit
        except AttributeError:
            pass
    return hit[2]


def _gn_dims(x, channels_last):
    N = x.shape[0]
    C = x.shape[-1] if channels_last else x.shape[1]
    return N, C, x.numel() // (N * C)


def _global_count(S, group, device):
    """Elements per (sample, channel) over the whole shard group."""
    import torch.distributed as dist
    t = torch.tensor([S], dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return int(t.item())


def _hip_group_norm(x, groups, weight, bias, eps, silu, channels_last, keep=False, group=None, S_total=None):
    x = x.contiguous()
    N, C, S = _gn_dims(x, channels_last)
    y = torch.empty_like(x)
    stats = torch.empty(2 * N * groups + N * C, dtype=torch.float64, device=x.device)  # group sums + (a, b) per (n, c)
    g = _f32_param(weight)
    b = _f32_param(bias)
    P, LL = ctypes.c_void_p, ctypes.c_longlong
    bf = 1 if x.dtype == torch.bfloat16 else 0
    with _on(x.device):
        if group is None:
            rc = lib().gvd_group_norm(P(x.data_ptr()), P(y.data_ptr()), P(g.data_ptr()), P(b.data_ptr()), P(stats.data_ptr()),
                                      N, C, LL(S), groups, ctypes.c_float(eps), int(bool(silu)), int(bool(channels_last)), bf,
                                      P(_stream()))
            S_total = S
        else:
            import torch.distributed as dist
            _check(lib().gvd_group_norm_stats(P(x.data_ptr()), P(stats.data_ptr()), N, C, LL(S), groups,
                                              int(bool(channels_last)), bf, P(_stream())))
            dist.all_reduce(stats[:2 * N * groups], group=group)
            if S_total is None:
                S_total = _global_count(S, group, x.device)
            rc = lib().gvd_group_norm_apply(P(x.data_ptr()), P(y.data_ptr()), P(g.data_ptr()), P(b.data_ptr()),
                                            P(stats.data_ptr()), N, C, LL(S), LL(S_total), groups, ctypes.c_float(eps),
                                            int(bool(silu)), int(bool(channels_last)), bf, P(_stream()))
    _check(rc)
    return (y, x, g, stats, S_total) if keep else y


def _hip_group_norm_bwd(x, gy, gamma32, stats, groups, eps, silu, channels_last, group=None, S_total=None, add=None):
    """add: a gradient of x's shape summed into the result inside the apply kernel (GradCell)."""
    gy = gy.contiguous()
    N, C, S = _gn_dims(x, channels_last)
    gx = torch.empty_like(x)
    scratch = torch.empty(2 * N * groups + N * C, dtype=torch.float64, device=x.device)
    P, LL = ctypes.c_void_p, ctypes.c_longlong
    bf = 1 if x.dtype == torch.bfloat16 else 0
    with _on(x.device):
        _check(lib().gvd_group_norm_bwd_stats(P(x.data_ptr()), P(gy.data_ptr()), P(gamma32.data_ptr()), P(stats.data_ptr()),
                                              P(scratch.data_ptr()), N, C, LL(S), groups, int(bool(silu)),
                                              int(bool(channels_last)), bf, P(_stream())))
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(scratch[:2 * N * groups], group=group)
        rc = lib().gvd_group_norm_bwd_apply_add(P(x.data_ptr()), P(gy.data_ptr()), P(None if add is None else add.data_ptr()),
                                                P(gx.data_ptr()), P(stats.data_ptr()), P(scratch.data_ptr()), N, C, LL(S),
                                                LL(S if group is None else S_total), groups, ctypes.c_float(eps),
                                                int(bool(silu)), int(bool(channels_last)), bf, P(_stream()))
    _check(rc)
    return gx


class _GroupNormFn(torch.autograd.Function):
    """Fused GroupNorm(+SiLU) kernels, forward and input-gradient.  Weight gradients are not produced: the only
    autograd user is the guided sampler, which differentiates w.r.t. x_t with frozen weights."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu, channels_last, group, S_total, grad_add=None):
        y, xc, g32, stats, S_total = _hip_group_norm(x, groups, weight, bias, eps, silu, channels_last, keep=True,
                                                     group=group, S_total=S_total)
        ctx.save_for_backward(xc, g32, stats)
        ctx.cfg = (groups, eps, silu, channels_last, group, S_total, None if grad_add is None else grad_add.arm())
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g32, stats = ctx.saved_tensors
        groups, eps, silu, cl, group, S_total, cell = ctx.cfg
        add = None if cell is None else cell.take(like=x)
        return (_hip_group_norm_bwd(x, gy, g32, stats, groups, eps, silu, cl, group, S_total, add=add),) + (None,) * 9


def group_norm(x, groups, weight, bias, eps=1e-5, silu=False, channels_last=False, group=None, S_total=None, grad_add=None):
    """GroupNorm with fp32 statistics + optional fused SiLU.  channels_last: x is [N, ..., C].
    group / S_total: statistics span the slices held by the ranks of `group` (S_total = global elements per
    (sample, channel); computed with one tiny all-reduce when omitted).  grad_add: a GradCell whose content (the gradient x
    receives along a residual branch) is added to the input gradient inside the backward kernel."""
    on_dev = _require_device(x, "group_norm")
    C = x.shape[-1] if channels_last else x.shape[1]
    if (on_dev and x.dtype in (torch.float16, torch.bfloat16) and weight is not None and bias is not None
            and (not channels_last or C % 8 == 0)):
        if torch.is_grad_enabled() and x.requires_grad:
            if weight.requires_grad or bias.requires_grad:
                raise RuntimeError("lvdm_amd.ops.group_norm: only the input gradient is implemented (freeze the weights)")
            return _GroupNormFn.apply(x, weight, bias, groups, eps, silu, channels_last, group, S_total, grad_add)
        return _hip_group_norm(x, groups, weight, bias, eps, silu, channels_last, group=group, S_total=S_total)
    if on_dev:
        _torch_form("group_norm", f"dtype {x.dtype}, C = {C} (the kernels cover 16-bit inputs with affine parameters, C % 8 == 0 token-major)")
    return group_norm_math(x, groups, weight, bias, eps, silu, channels_last, group)


# --------------------------------------------------------------------------------------------------
def _half_pair(x, *params):
    return x.dtype in (torch.float16, torch.bfloat16) and all(p is not None and p.dtype == x.dtype for p in params)


def _hip_layer_norm(x, weight, bias, eps):
    x = x.contiguous()
    C = x.shape[-1]
    y = torch.empty_like(x)
    with _on(x.device):
        rc = lib().gvd_layer_norm(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                  ctypes.c_void_p(weight.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                                  ctypes.c_longlong(x.numel() // C), C, ctypes.c_float(eps),
                                  1 if x.dtype == torch.bfloat16 else 0, ctypes.c_void_p(_stream()))
    _check(rc)
    return y


class _LayerNormFn(torch.autograd.Function):
    """Row LayerNorm kernels, forward and input gradient (weights frozen: guided sampler)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xc = x.contiguous()
        ctx.save_for_backward(xc, weight)
        ctx.eps = eps
        return _hip_layer_norm(xc, weight, bias, eps)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        C = x.shape[-1]
        gx = torch.empty_like(x)
        P = ctypes.c_void_p
        with _on(x.device):
            _check(lib().gvd_layer_norm_bwd(P(x.data_ptr()), P(gy.data_ptr()), P(weight.data_ptr()), P(gx.data_ptr()),
                                            ctypes.c_longlong(x.numel() // C), C, ctypes.c_float(ctx.eps),
                                            1 if x.dtype == torch.bfloat16 else 0, P(_stream())))
        return gx, None, None, None


class _GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        hc = h.contiguous()
        ctx.save_for_backward(hc)
        return _hip_geglu(hc)

    @staticmethod
    def backward(ctx, gy):
        (h,) = ctx.saved_tensors
        gy = gy.contiguous()
        C = h.shape[-1] // 2
        gh = torch.empty_like(h)
        P = ctypes.c_void_p
        with _on(h.device):
            _check(lib().gvd_geglu_bwd(P(h.data_ptr()), P(gy.data_ptr()), P(gh.data_ptr()), ctypes.c_longlong(h.numel() // (2 * C)),
                                       C, 1 if h.dtype == torch.bfloat16 else 0, P(_stream())))
        return gh


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dim (attention.py:283-285).  HIP kernel for no-grad 16-bit activations with 16-bit
    affine; everything else (fp32 parity runs, autocast with fp32 weights, the guided sampler's autograd pass) takes
    F.layer_norm, which is what the reference calls."""
    on_dev = _require_device(x, "layer_norm")
    C = x.shape[-1]
    if on_dev and _half_pair(x, weight, bias) and C % 8 == 0 and C <= 2048:
        grad = torch.is_grad_enabled()
        if not (grad and (x.requires_grad or weight.requires_grad or bias.requires_grad)):
            return _hip_layer_norm(x, weight, bias, eps)
        if not (weight.requires_grad or bias.requires_grad):   # guided sampler: input gradient only
            return _LayerNormFn.apply(x, weight, bias, eps)
    if on_dev:
        _torch_form("layer_norm", f"dtype {x.dtype} / weight {None if weight is None else weight.dtype}, C = {C}, or trainable affine")
    return F.layer_norm(x, (C,), weight, bias, eps)


def geglu_math(h):
    a, gate = h.chunk(2, dim=-1)
    return a * F.gelu(gate)


def geglu(h):
    """x * gelu(gate) on the two halves of the GEGLU projection (attention.py:420-423)."""
    on_dev = _require_device(h, "geglu")
    C = h.shape[-1] // 2
    if on_dev and h.dtype in (torch.float16, torch.bfloat16) and C % 8 == 0:
        if torch.is_grad_enabled() and h.requires_grad:
            return _GegluFn.apply(h)
        return _hip_geglu(h.contiguous())
    if on_dev:
        _torch_form("geglu", f"dtype {h.dtype}, C = {C}")
    return geglu_math(h)


def _hip_geglu(h):
    C = h.shape[-1] // 2
    y = torch.empty(h.shape[:-1] + (C,), dtype=h.dtype, device=h.device)
    with _on(h.device):
        rc = lib().gvd_geglu(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                             ctypes.c_longlong(h.numel() // (2 * C)), C,
                             1 if h.dtype == torch.bfloat16 else 0, ctypes.c_void_p(_stream()))
    _check(rc)
    return y


# --------------------------------------------------------------------------------------------------
def ddim_step(x, e_cond, e_uncond, noise, *, cfg_scale, guidance_rescale, sqrt_ac_t, sqrt_1mac_t, sqrt_a_prev, dir_coef,
              sigma_t, x0_rescale, temperature=1.0):
    """One no-grad DDIM update for the v-parameterisation (ddim.py:208-280):
        v      = e_uncond + s (e_cond - e_uncond);  v <- rescale_noise_cfg(v, e_cond, phi)  (if phi > 0)
        eps    = sqrt_ac_t v + sqrt_1mac_t x ;  x0 = (sqrt_ac_t x - sqrt_1mac_t v) * x0_rescale
        x_prev = sqrt_a_prev x0 + dir_coef eps + sigma * temperature * noise   (dir_coef = sqrt(1 - a_prev - sigma^2), fp32)
    Returns (x_prev, x0).  All per-batch statistics (std over non-batch dims) are fp32."""
    on_dev = _require_device(x, "ddim_step")
    if on_dev and x.dtype == torch.float32 and x.shape[0] == 1 and e_uncond is not None:
        x, e_cond, e_uncond, noise = (t.contiguous() for t in (x, e_cond, e_uncond, noise))
        x_prev, x0 = torch.empty_like(x), torch.empty_like(x)
        ws = torch.empty(8, dtype=torch.float64, device=x.device)
        with _on(x.device):
            rc = lib().gvd_ddim_step(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(e_cond.data_ptr()),
                                     ctypes.c_void_p(e_uncond.data_ptr()), ctypes.c_void_p(noise.data_ptr()),
                                     ctypes.c_void_p(x_prev.data_ptr()), ctypes.c_void_p(x0.data_ptr()),
                                     ctypes.c_void_p(ws.data_ptr()), ctypes.c_longlong(x.numel()),
                                     ctypes.c_float(cfg_scale), ctypes.c_float(guidance_rescale),
                                     ctypes.c_float(sqrt_ac_t), ctypes.c_float(sqrt_1mac_t), ctypes.c_float(sqrt_a_prev),
                                     ctypes.c_float(dir_coef), ctypes.c_float(sigma_t), ctypes.c_float(x0_rescale),
                                     ctypes.c_float(temperature),
                                     ctypes.c_void_p(_stream()))
        _check(rc)
        return x_prev, x0
    return ddim_step_math(x, e_cond, e_uncond, noise, cfg_scale=cfg_scale, guidance_rescale=guidance_rescale,
                          sqrt_ac_t=sqrt_ac_t, sqrt_1mac_t=sqrt_1mac_t, sqrt_a_prev=sqrt_a_prev, dir_coef=dir_coef,
                          sigma_t=sigma_t, x0_rescale=x0_rescale, temperature=temperature)


def ddim_step_math(x, e_cond, e_uncond, noise, *, cfg_scale, guidance_rescale, sqrt_ac_t, sqrt_1mac_t, sqrt_a_prev, dir_coef,
                   sigma_t, x0_rescale, temperature=1.0):
    from .schedule import rescale_noise_cfg
    if e_uncond is None:
        v = e_cond
    else:
        v = e_uncond + cfg_scale * (e_cond - e_uncond)
        if guidance_rescale > 0.0:
            v = rescale_noise_cfg(v, e_cond, guidance_rescale)
    eps = sqrt_ac_t * v + sqrt_1mac_t * x
    x0 = (sqrt_ac_t * x - sqrt_1mac_t * v) * x0_rescale
    x_prev = sqrt_a_prev * x0 + dir_coef * eps + sigma_t * temperature * noise
    return x_prev, x0
