"""GPU tests (-m gpu) of the hand-written MFMA GEMM (csrc/gemm_mfma.hip via lvdm_amd.gemm) against plain fp32 torch:
shapes of the U-Net / VAE / attention-chunk call sites, ragged M / N / K edges, strided and batched operands, and every epilogue
(LayerNorm fold, bias, GEGLU, residual).  Tolerance: the 16-bit output rounding (2^-11 relative for f16, 2^-8 for bf16) on top of
fp32 accumulation of 16-bit products -- 1.5e-3 / 1.2e-2 of the largest reference entry."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, ref):
    return float((a.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def _mk(g, *shape, dtype=torch.float16, scale=1.0):
    return (torch.randn(*shape, device=DEV, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("M,N,K", [(1000, 320, 320), (257, 640, 1280), (77, 2560, 1024), (3000, 512, 192), (513, 328, 72),
                                   (25, 1280, 320), (1, 1280, 320), (2304, 1280, 5120), (600, 8, 64), (300, 9216 // 8, 512)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_plain_product_with_bias(M, N, K, dtype):
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x, w = _mk(g, M, K, dtype=dtype), _mk(g, N, K, dtype=dtype, scale=K ** -0.5)
    b = torch.randn(N, device=DEV, generator=g)
    y = gemm.gemm_nt(x, w, bias=b)
    ref = x.float() @ w.float().t() + b
    assert y.shape == (M, N) and y.dtype == dtype
    assert _rel(y, ref) < (1.5e-3 if dtype == torch.float16 else 1.2e-2), _rel(y, ref)


@pytest.mark.parametrize("M,N,K", [(1, 1280, 1280), (25, 320, 1280), (77, 640, 1024), (129, 72, 40), (256, 2560, 1024), (200, 1000, 1032), (33, 8, 8),
                                   (64, 328, 5120)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_skinny_problems_run_on_the_wave_per_piece_kernel(M, N, K, dtype, monkeypatch):
    """M <= 256 rows without a LayerNorm fold / residual / gate (context projections, embedding Linears) take `k_gemm_skinny`: against
    fp32 torch and against the persistent kernel on the same operands (GVD_GEMM_NO_SKINNY is read once per process, so the second
    half of the comparison is the batched entry, which never takes the skinny form): ragged N (not a multiple of 32), K (not of 16 or
    64), strided row views, alpha and bias."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N + K)
    xw = _mk(g, M, K + 16, dtype=dtype)
    xw[:, :8] = float("nan")                                         # whatever surrounds the operand must not reach the product: the lanes
    xw[:, 8 + K:] = float("inf")                                     # past K read (and discard) in-matrix dummies
    x = xw[:, 8:8 + K]                                               # a column view: row stride K + 16
    w = _mk(g, N, K, dtype=dtype, scale=K ** -0.5)
    b = torch.randn(N, device=DEV, generator=g)
    y = gemm.gemm_nt(x, w, bias=b, alpha=0.7)
    ref = 0.7 * (x.float() @ w.float().t()) + b
    tol = 1.5e-3 if dtype == torch.float16 else 1.2e-2
    assert y.shape == (M, N) and y.dtype == dtype and _rel(y, ref) < tol, _rel(y, ref)
    y2 = gemm.gemm_nt(torch.stack([x, x]), w, bias=b, alpha=0.7)     # batch 2: the persistent kernel
    assert _rel(y2[0], ref) < tol and _rel(y, y2[0].float()) < tol


def test_strided_views_batches_and_scale():
    """Operands read in place: column slices of a packed [M, 3C] tensor, a broadcast (batch-stride 0) W, a batched W, alpha."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = _mk(g, 3, 704, 3 * 512)
    q, k = qkv[:, :, :512], qkv[:, :, 512:1024]
    s = gemm.gemm_nt(q, k, alpha=512 ** -0.5)                       # [3, 704, 704] = q k^T / sqrt(d): batched x batched
    ref = torch.matmul(q.float(), k.float().transpose(1, 2)) * 512 ** -0.5
    assert s.shape == (3, 704, 704) and _rel(s, ref) < 1.5e-3
    w = _mk(g, 320, 512, scale=0.05)
    y = gemm.gemm_nt(q, w)                                          # batched x shared
    assert _rel(y, q.float() @ w.float().t()) < 1.5e-3
    out = torch.zeros(3, 704, 640, dtype=torch.float16, device=DEV)
    gemm.gemm_nt(q, w, out=out[:, :, 320:])                          # write into a column slice
    assert float(out[:, :, :320].abs().max()) == 0.0 and _rel(out[:, :, 320:], q.float() @ w.float().t()) < 1.5e-3


@pytest.mark.parametrize("C,N", [(320, 320), (640, 1920), (1280, 1280), (1024, 640), (64, 192)])
def test_layernorm_fold_equals_layernorm_then_linear(C, N):
    """attention.py:283-285 + the projections: rstd (x W'^T - mean s) + c against F.layer_norm -> F.linear in fp32, incl. rows
    with a large mean (the cancellation the fold must survive)."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(C + N)
    M = 1500
    x = _mk(g, M, C)
    x[:100] += 6.0                                                   # mean >> std on some rows
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(C, device=DEV, generator=g))
        ln.bias.copy_(0.2 * torch.randn(C, device=DEV, generator=g))
    lin = torch.nn.Linear(C, N).to(DEV)
    ln.half().requires_grad_(False), lin.half().requires_grad_(False)
    with torch.no_grad():
        y = gemm.linear(x, lin.weight, lin.bias, ln=ln)
        ref = F.linear(F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps), lin.weight.float(), lin.bias.float())
    assert _rel(y, ref) < 2.5e-3, _rel(y, ref)
    st = gemm.row_stats(x, ln.eps)
    assert torch.allclose(st[:, 0], x.float().mean(1), atol=2e-5)
    assert torch.allclose(st[:, 1], torch.rsqrt(x.float().var(1, unbiased=False) + ln.eps), rtol=2e-5)


@pytest.mark.parametrize("C,M", [(8, 5), (64, 33), (320, 1501), (328, 97), (512, 64), (520, 31), (1024, 130), (1280, 77), (2048, 9), (4096, 3)])
def test_row_stats_every_group_width_and_ragged_row_counts(C, M):
    """gvd_row_stats: 8 / 16 / 32 / 64 lanes per row (C <= 512 / 1024 / 2048 / 4096), chunk counts that do not fill the last lane,
    row counts that do not fill the last wave; a strided view (ld > C) reads the same rows."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(C * 7 + M)
    wide = _mk(g, M, C + 24)
    wide[: M // 3] += 4.0
    for x in (wide[:, :C].contiguous(), wide[:, :C]):
        st = gemm.row_stats(x, 1e-5)
        assert st.shape == (M, 2)
        assert torch.allclose(st[:, 0], x.float().mean(1), atol=2e-5)
        assert torch.allclose(st[:, 1], torch.rsqrt(x.float().var(1, unbiased=False) + 1e-5), rtol=3e-5)


@pytest.mark.parametrize("C", [320, 640, 64])
def test_feed_forward_pair_geglu_and_residual(C):
    """attention.py:415-442 + the block's `+ x`: GEGLU(LN(x) W1^T + b1) as ONE launch, then W2 with the residual in its epilogue."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(C)
    M = 2000
    x = _mk(g, 5, M // 5, C)
    ln = torch.nn.LayerNorm(C).to(DEV).half().requires_grad_(False)
    p1 = torch.nn.Linear(C, 8 * C).to(DEV).half().requires_grad_(False)
    p2 = torch.nn.Linear(4 * C, C).to(DEV).half().requires_grad_(False)
    with torch.no_grad():
        h = gemm.linear(x, p1.weight, p1.bias, ln=ln, geglu=True)
        y = gemm.linear(h, p2.weight, p2.bias, residual=x)
        hf = F.linear(F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps), p1.weight.float(), p1.bias.float())
        a, gate = hf.chunk(2, dim=-1)
        href = a * F.gelu(gate)
        yref = F.linear(href, p2.weight.float(), p2.bias.float()) + x.float()
    assert h.shape == (5, M // 5, 4 * C) and _rel(h, href) < 3e-3, _rel(h, href)
    assert _rel(y, yref) < 3e-3, _rel(y, yref)


def test_linear_cat_and_input_gradient():
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(9)
    C = 320
    x = _mk(g, 2, 300, C)
    ws = [torch.nn.Linear(C, n, bias=False).to(DEV).half().requires_grad_(False).weight for n in (320, 320, 320)]
    ln = torch.nn.LayerNorm(C).to(DEV).half().requires_grad_(False)
    with torch.no_grad():
        qkv = gemm.linear_cat(x, ws, ln=ln)
    assert qkv.shape == (2, 300, 960)
    hn = F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps)
    for i, w in enumerate(ws):
        assert _rel(qkv[..., 320 * i:320 * (i + 1)], hn @ w.float().t()) < 2.5e-3
    # guided path: d/dx of x + Linear(LN(x)) through the MFMA GEMM (dX = dY W) and the LayerNorm row kernel
    lin = torch.nn.Linear(C, C).to(DEV).half().requires_grad_(False)
    xg = x.clone().requires_grad_(True)
    y = gemm.linear(xg, lin.weight, lin.bias, ln=ln, residual=xg)
    probe = _mk(g, *y.shape)
    (gx,) = torch.autograd.grad(y, xg, probe)
    xf = x.float().requires_grad_(True)
    yf = F.linear(F.layer_norm(xf, (C,), ln.weight.float(), ln.bias.float(), ln.eps), lin.weight.float(), lin.bias.float()) + xf
    (gref,) = torch.autograd.grad(yf, xf, probe.float())
    assert _rel(y.detach(), yf.detach()) < 2.5e-3 and _rel(gx, gref) < 4e-3, (_rel(y.detach(), yf.detach()), _rel(gx, gref))


def test_no_cpu_path():
    from lvdm_amd import gemm
    with pytest.raises(RuntimeError):
        gemm.linear(torch.randn(4, 64), torch.randn(8, 64))


@pytest.mark.parametrize("geglu", [False, True])
def test_many_tiles_per_persistent_workgroup(geglu):
    """More tiles than resident workgroups (each persistent workgroup walks several tiles, its waves drift apart between one
    tile's epilogue and the next tile's start): every epilogue feature on, full comparison.  A race between one wave's epilogue
    staging and another wave's next-tile set-up produced non-finite values at the 25-frame U-Net size only (round 3)."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(17)
    M, C, N = 60000, 320, 2560
    x = _mk(g, M, C)
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(C, device=DEV, generator=g))
        ln.bias.copy_(0.2 * torch.randn(C, device=DEV, generator=g))
    lin = torch.nn.Linear(C, N).to(DEV)
    ln.half().requires_grad_(False), lin.half().requires_grad_(False)
    res = _mk(g, M, N // 2 if geglu else N)
    with torch.no_grad():
        for _ in range(3):                                   # (a race is timing dependent: several launches)
            y = gemm.linear(x, lin.weight, lin.bias, ln=ln, geglu=geglu, residual=res)
            assert torch.isfinite(y).all()
        ref = F.linear(F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps), lin.weight.float(), lin.bias.float())
        if geglu:
            a, gate = ref.chunk(2, dim=-1)
            ref = a * F.gelu(gate)
        ref = ref + res.float()
    assert _rel(y, ref) < 3e-3, _rel(y, ref)


def test_fused_forms_under_autograd_match_fp32():
    """Guided-sampler path: the fused forward launches (LayerNorm fold + q|k|v concat; LayerNorm fold + GEGLU; residual) keep
    their fusion under autograd (`_FusedLinearFn`; only the GEGLU gate stays a separate row kernel there)."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(23)
    C, M = 320, 1100
    x = _mk(g, 2, M // 2, C).requires_grad_(True)
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(C, device=DEV, generator=g))
        ln.bias.copy_(0.2 * torch.randn(C, device=DEV, generator=g))
    ln.half().requires_grad_(False)
    ws = [torch.nn.Linear(C, C, bias=False).to(DEV).half().requires_grad_(False).weight for _ in range(3)]
    p1 = torch.nn.Linear(C, 8 * C).to(DEV).half().requires_grad_(False)
    p2 = torch.nn.Linear(4 * C, C).to(DEV).half().requires_grad_(False)
    calls = {"n": 0}
    orig = gemm.gemm_nt

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    gemm.gemm_nt = counted
    try:
        qkv = gemm.linear_cat(x, ws, ln=ln)                                   # 1 launch
        h = gemm.linear(x, p1.weight, p1.bias, ln=ln, geglu=True)             # 1 launch (+ the gate's row kernel under autograd)
        y = gemm.linear(h, p2.weight, p2.bias, residual=x)                    # 1 launch
        assert calls["n"] == 3
        probe_q, probe_y = _mk(g, *qkv.shape), _mk(g, *y.shape)
        (gx,) = torch.autograd.grad([qkv, y], [x], [probe_q, probe_y])
        assert calls["n"] == 3 + 3                                            # dqkv W_qkv; dy W2; dh W1
    finally:
        gemm.gemm_nt = orig
    xf = x.detach().float().requires_grad_(True)
    hn = F.layer_norm(xf, (C,), ln.weight.float(), ln.bias.float(), ln.eps)
    qkv_r = torch.cat([hn @ w.float().t() for w in ws], dim=-1)
    a, gate = F.linear(hn, p1.weight.float(), p1.bias.float()).chunk(2, dim=-1)
    y_r = F.linear(a * F.gelu(gate), p2.weight.float(), p2.bias.float()) + xf
    (g_r,) = torch.autograd.grad([qkv_r, y_r], [xf], [probe_q.float(), probe_y.float()])
    assert _rel(qkv.detach(), qkv_r.detach()) < 2.5e-3 and _rel(y.detach(), y_r.detach()) < 3e-3
    assert _rel(gx, g_r) < 6e-3, _rel(gx, g_r)


@pytest.mark.parametrize("C,M,with_ln,res", [(320, 1100, True, True), (640, 777, True, True), (128, 300, False, True), (320, 513, True, False)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_feed_forward_equals_the_unfused_pair(C, M, with_ln, res, dtype, monkeypatch):
    """attention.py:415-450 under autograd (guided sampler): `gemm.feed_forward` -- gate and the saved projection in the producing GEMM's
    epilogue, the gate's backward in the epilogue of ITS producer (gvd_gemm_nt_gate modes 2 / 3) -- against the two `linear` calls with the
    gate row kernels between them.  The forward is bit-identical (same operand roundings); the input gradient differs only by the
    accumulation order of the last product (its K runs over the kernel's [16 value | 16 gate] block order), and both sit at the 16-bit
    rounding level from the fp32 form."""
    from lvdm_amd import gemm
    g = torch.Generator(device=DEV).manual_seed(5)
    ln = None
    if with_ln:
        ln = torch.nn.LayerNorm(C).to(DEV)
        with torch.no_grad():
            ln.weight.copy_(1 + 0.3 * torch.randn(C, device=DEV, generator=g))
            ln.bias.copy_(0.2 * torch.randn(C, device=DEV, generator=g))
        ln.to(dtype).requires_grad_(False)
    p1 = torch.nn.Linear(C, 8 * C).to(DEV).to(dtype).requires_grad_(False)
    p2 = torch.nn.Linear(4 * C, C).to(DEV).to(dtype).requires_grad_(False)
    x0 = _mk(g, 2, M, C, dtype=dtype)
    probe = _mk(g, 2, M, C, dtype=dtype)

    def run(fused):
        if fused:
            monkeypatch.delenv("GVD_NO_FUSED_FF", raising=False)
        else:
            monkeypatch.setenv("GVD_NO_FUSED_FF", "1")
        x = x0.clone().requires_grad_(True)
        y = gemm.feed_forward(x, p1.weight, p1.bias, p2.weight, p2.bias, ln=ln, residual=x if res else None)
        if not fused:
            assert y is None
            h = gemm.linear(x, p1.weight, p1.bias, ln=ln, geglu=True)
            y = gemm.linear(h, p2.weight, p2.bias, residual=x if res else None)
        (gx,) = torch.autograd.grad(y, x, probe)
        return y.detach(), gx

    y_f, g_f = run(True)
    y_u, g_u = run(False)
    assert torch.equal(y_f, y_u)
    tol = 3e-3 if dtype == torch.float16 else 2.5e-2
    assert _rel(g_f, g_u.float()) < tol, _rel(g_f, g_u.float())
    xf = x0.float().requires_grad_(True)
    hn = xf if ln is None else F.layer_norm(xf, (C,), ln.weight.float(), ln.bias.float(), ln.eps)
    a, gate = F.linear(hn, p1.weight.float(), p1.bias.float()).chunk(2, dim=-1)
    y_r = F.linear(a * F.gelu(gate), p2.weight.float(), p2.bias.float()) + (xf if res else 0)
    (g_r,) = torch.autograd.grad(y_r, xf, probe.float())
    assert _rel(y_f, y_r.detach()) < (3e-3 if dtype == torch.float16 else 2.5e-2)
    assert _rel(g_f, g_r) < (6e-3 if dtype == torch.float16 else 4e-2), _rel(g_f, g_r)
