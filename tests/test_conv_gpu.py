"""GPU tests (-m gpu) of the hand-written MFMA convolutions (csrc/conv_mfma.hip, through the C-ABI gvd_conv_mfma) against a
plain fp32 torch form of the same operator: 3x3 convolution with every tile configuration / tile width the launcher can
pick, the fused GroupNorm(+SiLU) prologue, the bias / per-frame add / residual epilogue, the statistics it leaves for
the next norm, nearest-x2 upsampling on the fly, the temporal (3,1,1) form, and the input gradient.

Tolerance: operands are fp16, accumulation fp32, one rounding of the result to fp16 -> |err| <= 2^-11 |y| + accumulated
product rounding; the tests allow 2e-3 of the largest |y| (measured ~5e-4)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6))


def _conv_module(cin, cout, seed, three_d=False):
    g = torch.Generator().manual_seed(seed)
    m = nn.Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0)) if three_d else nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (cin * m.weight[0, 0].numel())) ** 0.5)
        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.3)
    m = m.to(DEV).half()
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def _ref_spatial(x, m, gn=None, silu=False, add_nc=None, residual=None, upsample=False):
    xf = x.float()
    if gn is not None:
        xf = F.group_norm(xf.permute(0, 3, 1, 2), gn.num_groups, gn.weight.float(), gn.bias.float(), gn.eps)
        if silu:
            xf = F.silu(xf)
        xf = xf.half().float()       # the product path rounds the activated operand to fp16 before the MFMA
    else:
        xf = xf.permute(0, 3, 1, 2)
    if upsample:
        xf = F.interpolate(xf, scale_factor=2, mode="nearest")
    torch.backends.cudnn.enabled = False   # native fp32 convolution: an independent implementation, not MIOpen's
    try:
        y = F.conv2d(xf, m.weight.float(), m.bias.float(), padding=1).permute(0, 2, 3, 1)
    finally:
        torch.backends.cudnn.enabled = True
    if add_nc is not None:
        y = y + add_nc.float()[:, None, None, :]
    if residual is not None:
        y = y + residual.float()
    return y


# (N, H, W, Cin, Cout): chosen so that the launcher takes each (tile configuration, tile width) once, with ragged edges
SHAPES = [
    (5, 72, 128, 64, 320),    # 160 x 256 px, 32-wide tiles, exact tiling
    (5, 177, 48, 40, 320),    # 160 x 256 px, 16-wide tiles, ragged rows, Cin not a multiple of 32
    (2, 18, 32, 96, 320),     # 320 x 128 px, 32-wide
    (3, 9, 16, 160, 640),     # 320 x 128 px, 16-wide, two cout tiles
    (1, 40, 72, 64, 256),     # 128 x 256 px, 16-wide (VAE class)
    (1, 40, 64, 128, 128),    # 128 x 256 px, 32-wide
    (2, 24, 32, 320, 4),      #  32 x 256 px (U-Net `out`), Cout not a multiple of 8
    (2, 21, 37, 8, 320),      # Cin = 8 (U-Net input conv), odd sizes
    (1, 16, 16, 4, 512),      # Cin = 4 (VAE conv_in): padded to 8 by the host wrapper
    (1, 33, 20, 72, 200),     # Cout neither a multiple of 160 nor 128 -> masked tile
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", SHAPES)
def test_conv3x3_matches_fp32(N, H, W, Cin, Cout, monkeypatch):
    from lvdm_amd import conv as C
    monkeypatch.setattr(C, "SHEETS", False)   # every (tile configuration, tile width) on the per-frame path; frame sheets have their own test
    g = torch.Generator(device=DEV).manual_seed(N * 1000 + H + W + Cin + Cout)
    x = torch.randn(N, H, W, Cin, device=DEV, generator=g).half()
    m = _conv_module(Cin, Cout, Cin + Cout)
    y, part = C.fused_conv(x, m)
    assert part is None and y.shape == (N, H, W, Cout) and y.dtype == torch.float16
    ref = _ref_spatial(x, m)
    assert _rel(y, ref) < 2e-3, _rel(y, ref)


@pytest.mark.parametrize("N,H,W,Cin,Cout,G", [(5, 72, 128, 320, 320, 32), (2, 18, 32, 640, 1280, 32), (1, 40, 72, 256, 128, 32),
                                              (3, 10, 16, 1920, 640, 32)])
def test_conv3x3_fused_prologue_epilogue_and_statistics(N, H, W, Cin, Cout, G):
    from lvdm_amd import conv as C
    g = torch.Generator(device=DEV).manual_seed(Cin * 3 + Cout)
    x = (torch.randn(N, H, W, Cin, device=DEV, generator=g) * 1.7 + 0.4).half()
    res = torch.randn(N, H, W, Cout, device=DEV, generator=g).half()
    add = torch.randn(N, Cout, device=DEV, generator=g).half()
    m = _conv_module(Cin, Cout, 7)
    gn = nn.GroupNorm(G, Cin, eps=1e-5).to(DEV).half()
    with torch.no_grad():
        gn.weight.copy_(torch.rand(Cin, device=DEV, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cin, device=DEV, generator=g) * 0.2)
    for p in gn.parameters():
        p.requires_grad_(False)
    y, part = C.fused_conv(x, m, gn=gn, silu=True, add_nc=add, residual=res, stats_groups=G)
    ref = _ref_spatial(x, m, gn, True, add, res)
    assert _rel(y, ref) < 2.5e-3, _rel(y, ref)
    # statistics of the rounded outputs, per (sample, group): exactly what a statistics pass over y would produce
    sums = part.sums.sum(0)                                             # [N, G, 2]
    yg = y.double().reshape(N, H * W, G, Cout // G)
    want = torch.stack([yg.sum(dim=(1, 3)), (yg * yg).sum(dim=(1, 3))], -1)
    assert torch.allclose(sums, want, rtol=2e-5, atol=1e-3), float((sums - want).abs().max())
    # ... and they drive the next norm: the state built from them equals the state from a pass over y
    gn2 = nn.GroupNorm(G, Cout, eps=1e-5).to(DEV).half()
    a = C.norm_state(gn2, partial=part)
    b = C.norm_state(gn2, x=y, n_stat=N)
    ca = a.buf[2 * N * G:].view(torch.float32)
    cb = b.buf[2 * N * G:].view(torch.float32)
    assert torch.allclose(ca, cb, rtol=1e-4, atol=1e-5)
    # without SiLU (VAE attention norm style) and without statistics
    y2, _ = C.fused_conv(x, m, gn=gn, silu=False)
    assert _rel(y2, _ref_spatial(x, m, gn, False)) < 2.5e-3


@pytest.mark.parametrize("N,h,w,Cin,Cout", [(3, 9, 16, 320, 320), (1, 36, 64, 128, 128), (2, 5, 7, 64, 640)])
def test_conv3x3_nearest_upsample_on_the_fly(N, h, w, Cin, Cout):
    from lvdm_amd import conv as C
    g = torch.Generator(device=DEV).manual_seed(h * w)
    x = torch.randn(N, h, w, Cin, device=DEV, generator=g).half()
    m = _conv_module(Cin, Cout, 11)
    y, _ = C.fused_conv(x, m, upsample=True)
    assert y.shape == (N, 2 * h, 2 * w, Cout)
    assert _rel(y, _ref_spatial(x, m, upsample=True)) < 2e-3


@pytest.mark.parametrize("T,Pp,Cc", [(25, 77, 320), (16, 64, 640), (25, 9216, 320), (3, 40, 1280), (25, 144, 1280)])
def test_temporal_conv_matches_conv3d(T, Pp, Cc):
    from lvdm_amd import conv as C
    g = torch.Generator(device=DEV).manual_seed(T * Pp)
    x = (torch.randn(T, Pp, Cc, device=DEV, generator=g) + 0.3).half()
    res = torch.randn(T, Pp, Cc, device=DEV, generator=g).half()
    m = _conv_module(Cc, Cc, 5, three_d=True)
    gn = nn.GroupNorm(32, Cc).to(DEV).half()
    with torch.no_grad():
        gn.weight.copy_(torch.rand(Cc, device=DEV, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cc, device=DEV, generator=g) * 0.2)
    for p in gn.parameters():
        p.requires_grad_(False)
    y, part = C.fused_conv(x, m, mode=C.TEMPORAL, gn=gn, silu=True, residual=res, stats_groups=32)
    xa = F.silu(F.group_norm(x.float().permute(2, 0, 1)[None], 32, gn.weight.float(), gn.bias.float(), gn.eps)).half().float()  # [1, C, T, P]
    ref = F.conv3d(xa[..., None], m.weight.float(), m.bias.float(), padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0) + res.float()
    assert _rel(y, ref) < 2.5e-3, _rel(y, ref)
    sums = part.sums.sum(0)[0]                                          # [G, 2], one sample = the whole video
    yg = y.double().reshape(T * Pp, 32, Cc // 32)
    want = torch.stack([yg.sum(dim=(0, 2)), (yg * yg).sum(dim=(0, 2))], -1)
    assert torch.allclose(sums, want, rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("N,Hin,Win,Cin,Cout,mode", [(3, 72, 128, 320, 320, "unet"), (2, 18, 32, 1280, 1280, "unet"), (2, 9, 15, 64, 640, "unet"),
                                                     (1, 64, 96, 128, 128, "vae"), (1, 33, 47, 256, 256, "vae"), (2, 40, 56, 320, 320, "unet")])
def test_stride2_downsample_convolutions(N, Hin, Win, Cin, Cout, mode):
    """U-Net Downsample (3x3, stride 2, pad 1: openaimodel3d.py:51-77) and VAE-encoder Downsample (pad (0,1,0,1), stride 2:
    ae_modules.py:90-109), forward; and for the U-Net form the input gradient (zero-stuffed stride-1 form)."""
    from lvdm_amd import conv as C
    g = torch.Generator(device=DEV).manual_seed(Hin * Win + Cin)
    x = torch.randn(N, Hin, Win, Cin, device=DEV, generator=g).half().requires_grad_(mode == "unet")
    m = _conv_module(Cin, Cout, 13)
    y, _ = C.fused_conv(x, m, mode=C.STRIDE2 if mode == "unet" else C.STRIDE2_PAD_HI)
    xf = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    torch.backends.cudnn.enabled = False
    try:
        if mode == "unet":
            ref = F.conv2d(xf, m.weight.float(), m.bias.float(), stride=2, padding=1)
        else:
            ref = F.conv2d(F.pad(xf, (0, 1, 0, 1)), m.weight.float(), m.bias.float(), stride=2)
    finally:
        torch.backends.cudnn.enabled = True
    assert y.shape == tuple(ref.permute(0, 2, 3, 1).shape)
    assert _rel(y, ref.permute(0, 2, 3, 1)) < 2e-3, _rel(y, ref.permute(0, 2, 3, 1))
    if mode == "unet":
        gy = torch.randn(y.shape, device=DEV, generator=g).half()
        (gx,) = torch.autograd.grad(y, [x], gy)
        torch.backends.cudnn.enabled = False
        try:
            (gr,) = torch.autograd.grad(ref, [xf], gy.float().permute(0, 3, 1, 2))
        finally:
            torch.backends.cudnn.enabled = True
        assert _rel(gx, gr.permute(0, 2, 3, 1)) < 3e-3, _rel(gx, gr.permute(0, 2, 3, 1))


def test_conv_input_gradients_match_autograd_of_the_fp32_form():
    from lvdm_amd import conv as C
    g = torch.Generator(device=DEV).manual_seed(99)
    N, H, W, Cin, Cout = 2, 18, 32, 320, 640
    x = (torch.randn(N, H, W, Cin, device=DEV, generator=g) + 0.2).half().requires_grad_(True)
    res = torch.randn(N, H, W, Cout, device=DEV, generator=g).half().requires_grad_(True)
    m = _conv_module(Cin, Cout, 3)
    gn = nn.GroupNorm(32, Cin).to(DEV).half()
    for p in gn.parameters():
        p.requires_grad_(False)
    gy = torch.randn(N, H, W, Cout, device=DEV, generator=g).half()
    y, _ = C.fused_conv(x, m, gn=gn, silu=True, residual=res)
    gx, gr = torch.autograd.grad(y, [x, res], gy)
    xf = x.detach().float().requires_grad_(True)
    rf = res.detach().float().requires_grad_(True)
    act = F.silu(F.group_norm(xf.permute(0, 3, 1, 2), 32, gn.weight.float(), gn.bias.float(), gn.eps))
    yr = F.conv2d(act, m.weight.float(), m.bias.float(), padding=1).permute(0, 2, 3, 1) + rf
    gxr, grr = torch.autograd.grad(yr, [xf, rf], gy.float())
    assert _rel(gx, gxr) < 6e-3, _rel(gx, gxr)   # two fp16 roundings (d_act, dx) + the fp16 GroupNorm backward kernel
    assert torch.equal(gr, gy)
    # plain convolution, upsampled input, Cout not a multiple of 8 (U-Net `out` / VAE conv_out backward)
    m2 = _conv_module(64, 4, 4)
    x2 = torch.randn(2, 6, 8, 64, device=DEV, generator=g).half().requires_grad_(True)
    y2, _ = C.fused_conv(x2, m2, upsample=True)
    gy2 = torch.randn_like(y2)
    (gx2,) = torch.autograd.grad(y2, [x2], gy2)
    x2f = x2.detach().float().requires_grad_(True)
    y2r = F.conv2d(F.interpolate(x2f.permute(0, 3, 1, 2), scale_factor=2, mode="nearest"), m2.weight.float(), m2.bias.float(), padding=1)
    (gx2r,) = torch.autograd.grad(y2r, [x2f], gy2.float().permute(0, 3, 1, 2))
    assert _rel(gx2, gx2r) < 4e-3
    # temporal
    m3 = _conv_module(320, 320, 8, three_d=True)
    x3 = torch.randn(25, 50, 320, device=DEV, generator=g).half().requires_grad_(True)
    y3, _ = C.fused_conv(x3, m3, mode=C.TEMPORAL)
    gy3 = torch.randn_like(y3)
    (gx3,) = torch.autograd.grad(y3, [x3], gy3)
    x3f = x3.detach().float().requires_grad_(True)
    y3r = F.conv3d(x3f.permute(2, 0, 1)[None, ..., None], m3.weight.float(), m3.bias.float(), padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0)
    (gx3r,) = torch.autograd.grad(y3r, [x3f], gy3.float())
    assert _rel(gx3, gx3r) < 4e-3


@pytest.mark.parametrize("mode,shape,C,Cout,silu,dtype", [
    ("spatial", (3, 18, 32), 320, 640, True, torch.float16), ("spatial", (2, 9, 16), 128, 128, False, torch.float16),
    ("spatial", (2, 20, 24), 160, 320, True, torch.bfloat16), ("temporal", (25, 50), 320, 320, True, torch.float16),
    ("temporal", (7, 33), 64, 128, False, torch.float16)])
def test_norm_backward_statistics_from_the_dgrad_epilogue(mode, shape, C, Cout, silu, dtype, monkeypatch):
    """gvd_conv_mfma_norm_bwd: the GroupNorm-backward sums accumulated in the input-gradient convolution's epilogue give the
    same input gradient as the separate statistics pass (k_gn_bwd_stats_*) -- both sum the same rounded d_act in fp32 partials
    and fp64 totals, so they agree to the 16-bit rounding of dx -- and both match fp32 autograd of the same expression."""
    from lvdm_amd import conv as C_
    monkeypatch.setattr(C_, "SHEETS", False)   # (small maps would otherwise take the frame-sheet path, whose norm backward is always two passes)
    g = torch.Generator(device=DEV).manual_seed(314)
    x = (torch.randn(*shape, C, device=DEV, generator=g) * 1.3 + 0.3).to(dtype).requires_grad_(True)
    m = _conv_module(C, Cout, 5, three_d=(mode == "temporal"))
    gn = nn.GroupNorm(32, C).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, device=DEV, generator=g) * 0.5 + 1.0)
        gn.bias.copy_(torch.randn(C, device=DEV, generator=g) * 0.3)
    for p in gn.parameters():
        p.requires_grad_(False)
    cmode = C_.TEMPORAL if mode == "temporal" else C_.SPATIAL
    outs = {}
    for fuse in (True, False):
        C_.FUSE_NORM_BACKWARD_STATS = fuse
        try:
            y, _ = C_.fused_conv(x, m, mode=cmode, gn=gn, silu=silu)
            gy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7)).to(dtype)
            (outs[fuse],) = torch.autograd.grad(y, [x], gy)
        finally:
            C_.FUSE_NORM_BACKWARD_STATS = True
    tol = 2.5e-2 if dtype == torch.bfloat16 else 3e-3
    assert _rel(outs[True], outs[False]) < tol, _rel(outs[True], outs[False])
    xf = x.detach().float().requires_grad_(True)
    n_stat = 1 if mode == "temporal" else shape[0]
    yr = C_._reference(xf, m.weight.float(), m.bias.float(), cmode, False, gn, silu, None, None, n_stat)
    (gr,) = torch.autograd.grad(yr, [xf], gy.float())
    assert _rel(outs[True], gr) < (4e-2 if dtype == torch.bfloat16 else 6e-3), _rel(outs[True], gr)


@pytest.mark.parametrize("N,H,W,Cin,Cout,G,dtype", [(50, 5, 7, 64, 128, 32, torch.float16), (25, 9, 16, 160, 320, 32, torch.float16),
                                                    (13, 10, 14, 40, 72, 8, torch.float16), (7, 4, 4, 32, 64, 4, torch.bfloat16),
                                                    (2, 9, 16, 64, 64, 32, torch.float16)])
def test_small_maps_run_as_one_frame_sheet(N, H, W, Cin, Cout, G, dtype, monkeypatch):
    """Maps smaller than a convolution tile (the 5 x 7 / 10 x 14 / 9 x 16 latents of the U-Net's deepest level) run as ONE image -- the
    frames side by side and below each other with a zero row / column between them (conv.py: _sheet_plan, k_sheet_in / k_sheet_out):
    same operator (GroupNorm + SiLU in front, bias, per-frame add, residual, statistics for the next norm) and same input gradient as
    the per-frame launch and as fp32 torch.  Tolerance: the sheet rounds the activated input and the convolution result to 16 bit
    before the per-frame adds (one rounding more than the fused epilogue)."""
    from lvdm_amd import conv as C
    assert C._sheet_plan(N, H, W) > 0
    g = torch.Generator(device=DEV).manual_seed(N * 100 + H * W + Cin)
    x = (torch.randn(N, H, W, Cin, device=DEV, generator=g) * 1.7 + 0.4).to(dtype).requires_grad_(True)
    res = torch.randn(N, H, W, Cout, device=DEV, generator=g).to(dtype).requires_grad_(True)
    add = torch.randn(N, Cout, device=DEV, generator=g).to(dtype)
    m = _conv_module(Cin, Cout, 11).to(dtype)
    gn = nn.GroupNorm(G, Cin, eps=1e-5).to(DEV).to(dtype)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(Cin, device=DEV, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cin, device=DEV, generator=g) * 0.2)
    for p in gn.parameters():
        p.requires_grad_(False)
    gy = torch.randn(N, H, W, Cout, device=DEV, generator=g).to(dtype)
    seen = []
    orig = C._sheet_conv
    monkeypatch.setattr(C, "_sheet_conv", lambda *a, **k: (seen.append(a[2]), orig(*a, **k))[1])
    outs = {}
    for sheets in (True, False):
        monkeypatch.setattr(C, "SHEETS", sheets)
        y, part = C.fused_conv(x, m, gn=gn, silu=True, add_nc=add, residual=res, stats_groups=G)
        gx, gr = torch.autograd.grad(y, [x, res], gy)
        with torch.no_grad():
            y_ng, part_ng = C.fused_conv(x.detach(), m, gn=gn, silu=True, add_nc=add, residual=res.detach(), stats_groups=G)
        assert torch.equal(y_ng, y.detach()) and torch.equal(gr, gy)
        outs[sheets] = (y.detach(), gx, part.sums.sum(0), part_ng.sums.sum(0))
    assert seen == [False, True, False], seen          # forward, input gradient, no-grad forward went through the sheet -- and only with SHEETS on
    f16 = dtype == torch.float16
    ref = _ref_spatial(x.detach(), m, gn, True, add, res.detach()) if f16 else None
    tol_y, tol_g = (3e-3, 8e-3) if f16 else (2.5e-2, 5e-2)
    assert _rel(outs[True][0], outs[False][0]) < tol_y, _rel(outs[True][0], outs[False][0])
    assert _rel(outs[True][1], outs[False][1]) < tol_g, _rel(outs[True][1], outs[False][1])
    if f16:
        assert _rel(outs[True][0], ref) < 3e-3, _rel(outs[True][0], ref)
    # the statistics are those of the tensor that was returned
    yg = outs[True][0].double().reshape(N, H * W, G, Cout // G)
    want = torch.stack([yg.sum(dim=(1, 3)), (yg * yg).sum(dim=(1, 3))], -1)
    for got in outs[True][2:]:
        assert torch.allclose(got, want, rtol=2e-5, atol=1e-3), float((got - want).abs().max())


@pytest.mark.parametrize("mode,shape,Cin,Cout,dtype", [("spatial", (25, 5, 7), 640, 512, torch.float16), ("spatial", (3, 20, 24), 512, 128, torch.float16),
                                                       ("temporal", (25, 40), 640, 512, torch.float16), ("temporal", (2, 16, 33), 512, 128, torch.float16),
                                                       ("spatial", (4, 9, 16), 544, 64, torch.bfloat16)])
def test_split_k_launches_match_the_single_pass(mode, shape, Cin, Cout, dtype, monkeypatch):
    """Launches with fewer workgroups than CU slots and a long reduction are cut along the input channels (gvd_conv_mfma_splitk, the
    slices summed in fp32 by gvd_conv_sum_slices / gvd_conv_sheet_out): same result as the single launch up to the 16-bit rounding of
    the partial sums (<= 8 slices), with the GroupNorm + SiLU prologue, residual, statistics, and the input gradient."""
    from lvdm_amd import conv as C
    cmode = C.TEMPORAL if mode == "temporal" else C.SPATIAL
    g = torch.Generator(device=DEV).manual_seed(Cin + Cout)
    x = (torch.randn(*shape, Cin, device=DEV, generator=g) * 1.3 + 0.3).to(dtype).requires_grad_(True)
    m = _conv_module(Cin, Cout, 21, three_d=(mode == "temporal")).to(dtype)
    gn = nn.GroupNorm(32, Cin).to(DEV).to(dtype)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(Cin, device=DEV, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cin, device=DEV, generator=g) * 0.2)
    for p in gn.parameters():
        p.requires_grad_(False)
    res = torch.randn(*shape, Cout, device=DEV, generator=g).to(dtype)
    gy = torch.randn(*shape, Cout, device=DEV, generator=g).to(dtype)
    n_split = []
    orig = C._launch_split
    monkeypatch.setattr(C, "_launch_split", lambda *a, **k: (lambda r: (n_split.append(r[1]), r)[1])(orig(*a, **k)))
    outs = {}
    for split in (True, False):
        monkeypatch.setattr(C, "SPLITK", split)
        y, part = C.fused_conv(x, m, mode=cmode, gn=gn, silu=True, residual=res, stats_groups=32)
        (gx,) = torch.autograd.grad(y, [x], gy)
        outs[split] = (y.detach(), gx, part.sums.sum(0))
    # the forward -- and the input gradient, whose reduction runs over Cout -- were split, and only with SPLITK on
    assert len(n_split) == (2 if Cout >= 512 else 1) and min(n_split) >= 2, n_split
    f16 = dtype == torch.float16
    assert _rel(outs[True][0], outs[False][0]) < (2e-3 if f16 else 1.6e-2), _rel(outs[True][0], outs[False][0])
    assert _rel(outs[True][1], outs[False][1]) < (6e-3 if f16 else 5e-2), _rel(outs[True][1], outs[False][1])
    n_stat = (shape[0] if len(shape) == 3 and mode == "temporal" else 1) if mode == "temporal" else shape[0]
    yg = outs[True][0].double().reshape(n_stat, -1, 32, Cout // 32)
    want = torch.stack([yg.sum(dim=(1, 3)), (yg * yg).sum(dim=(1, 3))], -1)
    assert torch.allclose(outs[True][2], want, rtol=2e-5, atol=1e-3)
    xf = x.detach().float()
    ref = C._reference(xf, m.weight.float(), m.bias.float(), cmode, False, gn.float(), True, None, res.float(), n_stat)
    assert _rel(outs[True][0], ref) < (3e-3 if f16 else 2e-2), _rel(outs[True][0], ref)


@pytest.mark.parametrize("N,h,w,Cin,Cout,dtype", [(2, 18, 32, 320, 320, torch.float16), (1, 40, 56, 256, 128, torch.float16), (3, 9, 28, 64, 640, torch.float16),
                                                  (1, 33, 47, 40, 72, torch.float16), (2, 20, 32, 128, 128, torch.bfloat16)])
def test_nearest_upsample_as_four_phase_convolutions(N, h, w, Cin, Cout, dtype, monkeypatch):
    """conv3x3(nearest_x2(x)) as four 2x2 convolutions of the input map, one per output phase, with the taps that land on the same
    input pixel summed on the host (kernel mode 4; conv.packed(..., "up2")): against the on-the-fly upsampled 9-tap form and fp32
    torch, with the GroupNorm + SiLU prologue (applied once per INPUT pixel here), per-frame add, residual at the output resolution,
    next-norm statistics; ragged tiles (input maps 9 x 28, 33 x 47); and the input gradient as one 2x2-per-phase convolution over the
    four phase images of the output-resolution gradient (kernel mode 5) against the 9-tap gradient + 2x2 sum."""
    from lvdm_amd import conv as C
    g = torch.Generator(device=DEV).manual_seed(h * w + Cin)
    x = (torch.randn(N, h, w, Cin, device=DEV, generator=g) * 1.5 + 0.3).to(dtype).requires_grad_(True)
    res = torch.randn(N, 2 * h, 2 * w, Cout, device=DEV, generator=g).to(dtype)
    add = torch.randn(N, Cout, device=DEV, generator=g).to(dtype)
    m = _conv_module(Cin, Cout, 31).to(dtype)
    G = 8
    gn = nn.GroupNorm(G, Cin, eps=1e-5).to(DEV).to(dtype)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(Cin, device=DEV, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cin, device=DEV, generator=g) * 0.2)
    for p in gn.parameters():
        p.requires_grad_(False)
    gy = torch.randn(N, 2 * h, 2 * w, Cout, device=DEV, generator=g).to(dtype)
    modes = []
    orig = C._launch
    monkeypatch.setattr(C, "_launch", lambda x_, w_, co, mode, *a, **k: (modes.append(mode), orig(x_, w_, co, mode, *a, **k))[1])
    outs = {}
    for phases in (True, False):
        monkeypatch.setattr(C, "UP2_PHASES", phases)
        y, part = C.fused_conv(x, m, upsample=True, gn=gn, silu=True, add_nc=add, residual=res, stats_groups=G)
        (gx,) = torch.autograd.grad(y, [x], gy)
        y2, _ = C.fused_conv(x.detach(), m, upsample=True)                      # plain: no prologue, no epilogue terms
        outs[phases] = (y.detach(), gx, part.sums.sum(0), y2)
    assert modes[0] == C.UP2 and modes.count(C.UP2) == 2 and modes.count(C.UP2_BWD) == 1, modes   # the two forwards and the input gradient of the first round, nothing else
    f16 = dtype == torch.float16
    tol = 2.5e-3 if f16 else 2e-2
    assert _rel(outs[True][0], outs[False][0]) < tol, _rel(outs[True][0], outs[False][0])
    assert _rel(outs[True][3], outs[False][3]) < tol
    assert _rel(outs[True][1], outs[False][1]) < (6e-3 if f16 else 5e-2)
    if f16:
        assert _rel(outs[True][0], _ref_spatial(x.detach(), m, gn, True, add, res, upsample=True)) < 3e-3
        assert _rel(outs[True][3], _ref_spatial(x.detach(), m, upsample=True)) < 2.5e-3
    yg = outs[True][0].double().reshape(N, 4 * h * w, G, Cout // G)
    want = torch.stack([yg.sum(dim=(1, 3)), (yg * yg).sum(dim=(1, 3))], -1)
    assert torch.allclose(outs[True][2], want, rtol=2e-5, atol=2e-3), float((outs[True][2] - want).abs().max())


def test_conv_bf16_and_rejections():
    from lvdm_amd import conv as C
    m = _conv_module(64, 64, 1)
    with pytest.raises(RuntimeError):
        C.fused_conv(torch.randn(1, 8, 8, 64), m)                       # CPU tensor: no CPU path
    # bf16 operands (8 significant bits): same kernel family, tolerance of the coarser rounding
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(2, 20, 32, 320, device=DEV, generator=g).bfloat16()
    mb = _conv_module(320, 320, 2).bfloat16()
    gn = nn.GroupNorm(32, 320).to(DEV).bfloat16()
    for p in gn.parameters():
        p.requires_grad_(False)
    y, part = C.fused_conv(x, mb, gn=gn, silu=True, stats_groups=32)
    xa = F.silu(F.group_norm(x.float().permute(0, 3, 1, 2), 32, gn.weight.float(), gn.bias.float(), gn.eps)).bfloat16().float()
    ref = F.conv2d(xa, mb.weight.float(), mb.bias.float(), padding=1).permute(0, 2, 3, 1)
    assert y.dtype == torch.bfloat16 and _rel(y, ref) < 1.5e-2, _rel(y, ref)
    xt = torch.randn(25, 40, 320, device=DEV, generator=g).bfloat16()
    m3 = _conv_module(320, 320, 9, three_d=True).bfloat16()
    yt, _ = C.fused_conv(xt, m3, mode=C.TEMPORAL)
    reft = F.conv3d(xt.float().permute(2, 0, 1)[None, ..., None], m3.weight.float(), m3.bias.float(), padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0)
    assert _rel(yt, reft) < 1.5e-2


_XCD_PROBE = r"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "guidedvd-3dgs_amd"))
import torch, torch.nn as nn
from lvdm_amd import conv as C
dev = "cuda:0"
torch.manual_seed(0)                       # (the modules' default initialisation of biases / the Conv3d)
g = torch.Generator(device=dev).manual_seed(3)
out = []
# several channel tiles per pixel tile (inputs first / weights first differ), a ragged map, a temporal launch with two samples, an
# upsampling launch (four phases), a small map that is split along K, and the input gradient of the first
for (N, H, W, Cin, Cout, kw) in [(3, 40, 56, 320, 640, {}), (2, 21, 37, 128, 384, {}), (2, 32, 48, 256, 256, {"upsample": True}), (5, 10, 14, 1280, 1280, {})]:
    m = nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half().requires_grad_(False)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, device=dev, generator=g) * 0.02)
    x = torch.randn(N, (H // 2) if kw else H, (W // 2) if kw else W, Cin, device=dev, generator=g).half().requires_grad_(not kw)
    y, _ = C.fused_conv(x, m, **kw)
    out.append(y.detach())
    if not kw and Cin == 320:
        (gx,) = torch.autograd.grad(y, [x], torch.ones_like(y))
        out.append(gx)
m3 = nn.Conv3d(640, 640, (3, 1, 1), padding=(1, 0, 0)).to(dev).half().requires_grad_(False)
xt = torch.randn(2, 25, 70, 640, device=dev, generator=g).half()
out.append(C.fused_conv(xt, m3, mode=C.TEMPORAL)[0])
h = hashlib.sha256()
for t in out:
    h.update(t.float().cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
"""


def test_xcd_aware_grid_readings_are_permutations_of_the_same_work():
    """csrc/diffusion_common.h xcd_conv_ids: the plain grid reading, inputs-first and weights-first (GVD_CONV_XCD_MAP = 0 / 1 / 2, read
    once per process) and the per-launch rule must produce bit-identical outputs -- they only change WHICH workgroup does which tile."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for tag, val in (("plain", "0"), ("inputs", "1"), ("weights", "2"), ("rule", None)):
        env = dict(os.environ)
        env.pop("GVD_CONV_XCD_MAP", None)
        if val is not None:
            env["GVD_CONV_XCD_MAP"] = val
        r = subprocess.run([sys.executable, "-c", _XCD_PROBE, root], env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")]
        assert r.returncode == 0 and lines, (tag, r.stdout[-500:], r.stderr[-2000:])
        digests[tag] = lines[-1]
    assert len(set(digests.values())) == 1, digests


def test_fused_skip_concatenation_with_group_norm_statistics():
    """conv.cat_with_stats (gvd_cat2_group_norm_stats): the decoder's `torch.cat([h, skip], channel)` and the statistics of the ResBlock's
    first GroupNorm in one pass.  The concatenated rows must be the copy's bits; the (a, b) affine must equal the one a statistics pass over
    the concatenated tensor gives (same per-lane partial sums, fp64 accumulation: equal to fp32 rounding), for equal and unequal halves,
    groups that straddle no boundary and groups that do, f16 and bf16; and the U-Net must give the same output with the fused
    concatenation as with torch.cat (GVD_NO_FUSED_CAT=1)."""
    import os
    from lvdm_amd import conv as mconv
    g = torch.Generator(device=DEV).manual_seed(9)
    for dtype in (torch.float16, torch.bfloat16):
        for (N, H, W, Ca, Cb) in ((3, 9, 16, 320, 320), (2, 18, 33, 640, 320), (5, 5, 7, 1280, 1280), (1, 40, 56, 64, 32), (2, 7, 5, 8, 24)):
            a = (torch.randn(N, H, W, Ca, device=DEV, generator=g) * 1.5 + 0.3).to(dtype)
            b = (torch.randn(N, H, W, Cb, device=DEV, generator=g) * 0.7 - 0.2).to(dtype)
            gn = torch.nn.GroupNorm(32 if (Ca + Cb) % 32 == 0 else 8, Ca + Cb).to(DEV)
            gn.weight.data = 1 + 0.2 * torch.randn(Ca + Cb, device=DEV, generator=g)
            gn.bias.data = 0.2 * torch.randn(Ca + Cb, device=DEV, generator=g)
            assert mconv.cat_with_stats_ok(a, b, gn)
            out, ns = mconv.cat_with_stats(a, b, gn)
            ref = torch.cat([a, b], dim=-1)
            assert torch.equal(out, ref)
            ns_ref = mconv.norm_state(gn, x=ref, n_stat=N)
            k = 2 * N * gn.num_groups
            assert torch.allclose(ns.buf[:k], ns_ref.buf[:k], rtol=1e-12, atol=0.0)            # fp64 sums: atomics order only
            co, co_ref = ns.buf[k:].view(torch.float32), ns_ref.buf[k:].view(torch.float32)
            assert torch.allclose(co, co_ref, rtol=1e-6, atol=1e-7), float((co - co_ref).abs().max())
    # with a gradient wanted the fused form steps aside
    assert not mconv.cat_with_stats_ok(a.requires_grad_(True), b, gn)
    # the network: same output either way
    from test_diffusion_goldens_gpu import UNET64
    from lvdm_amd.unet import UNetModel
    from fill_by_name import fill_by_name
    unet = fill_by_name(UNetModel(**UNET64)).half().eval().to(DEV).to_token_major().requires_grad_(False)
    x = torch.randn(1, 8, 4, 16, 24, device=DEV, generator=g).half()
    ctx = torch.randn(1, 93, 64, device=DEV, generator=g).half()
    calls = {"n": 0}
    orig = mconv.cat_with_stats

    def counted(*a_, **k_):
        calls["n"] += 1
        return orig(*a_, **k_)
    mconv.cat_with_stats = counted
    try:
        with torch.no_grad():
            y_fused = unet(x, torch.tensor([300], device=DEV), context=ctx, fs=torch.tensor([10], device=DEV))
            n_fused = calls["n"]
            os.environ["GVD_NO_FUSED_CAT"] = "1"
            y_plain = unet(x, torch.tensor([300], device=DEV), context=ctx, fs=torch.tensor([10], device=DEV))
    finally:
        os.environ.pop("GVD_NO_FUSED_CAT", None)
        mconv.cat_with_stats = orig
    assert n_fused == len(unet.output_blocks) and calls["n"] == n_fused, (n_fused, calls["n"])
    assert float((y_fused.float() - y_plain.float()).abs().max()) <= 2e-3 * float(y_plain.float().abs().max())
