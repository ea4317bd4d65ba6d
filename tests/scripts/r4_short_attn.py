"""Short-row (temporal) attention forward: the wave-per-item kernel against the general flash kernel run the old way
(GVD_ATTN_NO_SHORT=1), same inputs -- error of both against the fp32 form, lse agreement, and time / bandwidth at the temporal
shapes of the two bench resolutions.  (dev tool)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd import ops

DEV = "cuda:0"


def run(qkv, heads, short):
    if short:
        os.environ.pop("GVD_ATTN_NO_SHORT", None)
    else:
        os.environ["GVD_ATTN_NO_SHORT"] = "1"
    C = qkv.shape[-1] // 3
    return ops._hip_attention_fwd(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, True, want_lse=True)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for T, P, C in ((25, 9216, 320), (25, 2304, 640), (25, 576, 1280), (25, 144, 1280), (25, 2240, 320), (25, 560, 640), (25, 140, 1280), (16, 2240, 320), (32, 300, 320), (7, 99, 128)):
    heads = C // 64
    g = torch.Generator(device=DEV).manual_seed(P + T)
    qkv = torch.randn(T, P, 3 * C, device=DEV, generator=g).half()
    qkv[..., :C] *= 1.7
    o_new, lse_new = run(qkv, heads, True)
    o_old, lse_old = run(qkv, heads, False)
    ref = ops.attention_math(qkv[..., :C].float(), qkv[..., C:2 * C].float(), qkv[..., 2 * C:].float(), heads, True)
    en = float((o_new.float() - ref).abs().max() / ref.abs().max())
    eo = float((o_old.float() - ref).abs().max() / ref.abs().max())
    el = float((lse_new - lse_old).abs().max())
    t_new = bench(lambda: run(qkv, heads, True))
    t_old = bench(lambda: run(qkv, heads, False))
    gb = 4 * T * P * C * 2 / 1e9
    print(f"T {T:2d} P {P:5d} C {C:4d}: err new {en:.1e} old {eo:.1e}  lse diff {el:.1e}   new {t_new:7.1f} us ({gb / t_new * 1e3:5.2f} TB/s)   old {t_old:7.1f} us ({gb / t_old * 1e3:5.2f} TB/s)", flush=True)

print("-- backward (dq, dk, dv): one wave-per-item kernel against delta + dkv + dq", flush=True)
for T, P, C in ((25, 2240, 320), (25, 560, 640), (25, 140, 1280), (25, 9216, 320)):
    heads = C // 64
    g = torch.Generator(device=DEV).manual_seed(P + T)
    q, k, v = (torch.randn(T, P, C, device=DEV, generator=g).half().requires_grad_(True) for _ in range(3))
    go = torch.randn(T, P, C, device=DEV, generator=g).half()

    def grads(short):
        if short:
            os.environ.pop("GVD_ATTN_NO_SHORT", None)
        else:
            os.environ["GVD_ATTN_NO_SHORT"] = "1"
        o = ops.attention(q, k, v, heads, frame_major=True)
        return o, o.grad_fn

    res = {}
    for short in (True, False):
        o, _ = grads(short)
        gs = torch.autograd.grad(o, (q, k, v), go, retain_graph=True)
        t = bench(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
        res[short] = (gs, t)
    qf, kf, vf = (t_.detach().float().requires_grad_(True) for t_ in (q, k, v))
    gr = torch.autograd.grad(ops.attention_math(qf, kf, vf, heads, True), (qf, kf, vf), go.float())
    errs = [max(float((a.float() - r).abs().max() / r.abs().max()) for a, r in zip(res[s][0], gr)) for s in (True, False)]
    gb = 7 * T * P * C * 2 / 1e9
    print(f"T {T:2d} P {P:5d} C {C:4d}: err new {errs[0]:.1e} old {errs[1]:.1e}   new {res[True][1]:7.1f} us ({gb / res[True][1] * 1e3:5.2f} TB/s on 7 tensor passes)   old {res[False][1]:7.1f} us", flush=True)
