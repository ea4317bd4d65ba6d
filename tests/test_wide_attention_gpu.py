"""GPU tests (-m gpu) of the wide single-head attention (lvdm_amd/wide_attention.py: the VAE's d = 512 mid block, ae_modules.py:26-78)
-- chunked MFMA GEMMs + the softmax / dS row kernels -- against the explicit fp32 form, forward and all three input gradients,
with the chunking forced to several query / key chunks.  Tolerances: the scores are rounded to 16 bit before the softmax (as the
reference's autocast bmm does): 4e-3 of the largest entry forward, 1e-2 for the gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, k, v):
    s = torch.matmul(q, k.transpose(1, 2)) * q.shape[-1] ** -0.5
    return torch.matmul(s.softmax(-1), v)


def _rel(a, b):
    return float((a.float() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("B,Nq,Nk,d,dtype", [(2, 1000, 1000, 512, torch.float16), (1, 2304, 2304, 512, torch.float16), (3, 520, 776, 128, torch.float16), (2, 140, 140, 512, torch.float16), (1, 35, 35, 128, torch.float16),
                                             (2, 1000, 1000, 512, torch.bfloat16)])
def test_forward_and_gradients_match_fp32(B, Nq, Nk, d, dtype, monkeypatch):
    from lvdm_amd import ops, wide_attention
    monkeypatch.setattr(wide_attention, "SCORE_BYTES", 2 * B * 512 * max(Nq, Nk))     # 512-row chunks: several per pass
    g = torch.Generator(device=DEV).manual_seed(Nq + d)
    q, k, v = (torch.randn(B, n, d, device=DEV, generator=g).to(dtype).requires_grad_(True) for n in (Nq, Nk, Nk))
    o = ops.attention(q, k, v, heads=1)
    probe = torch.randn(B, Nq, d, device=DEV, generator=g).to(dtype)
    gq, gk, gv = torch.autograd.grad(o, (q, k, v), probe)
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    of = _ref(qf, kf, vf)
    rq, rk, rv = torch.autograd.grad(of, (qf, kf, vf), probe.float())
    tol_f, tol_g = (4e-3, 1e-2) if dtype == torch.float16 else (3e-2, 6e-2)
    assert _rel(o.detach(), of.detach()) < tol_f, _rel(o.detach(), of.detach())
    for a, b, name in ((gq, rq, "dq"), (gk, rk, "dk"), (gv, rv, "dv")):
        assert _rel(a, b) < tol_g, (name, _rel(a, b))
    with torch.no_grad():                                  # the no-grad entry (sampler's final decode) is the same forward
        assert torch.equal(ops.attention(q.detach(), k.detach(), v.detach(), heads=1), o.detach())


def test_peaked_scores_and_packed_qkv_views():
    """Rows whose softmax is nearly one-hot (large score range) and q / k / v given as column blocks of one packed projection."""
    from lvdm_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(2, 640, 3 * 512, device=DEV, generator=g).half()
    qkv[:, :, :512] *= 3.0
    q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
    o = ops.attention(q, k, v, heads=1)
    assert _rel(o, _ref(q.float(), k.float(), v.float())) < 8e-3


def test_sixteen_bit_inputs_never_take_the_torch_form():
    from lvdm_amd import ops
    x = torch.randn(1, 12, 40, device=DEV).half()          # 5 heads of 8 channels: no kernel family covers it
    with pytest.raises(RuntimeError):
        ops.attention(x, x, x, heads=5)
