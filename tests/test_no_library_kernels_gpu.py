"""The 16-bit product path of the diffusion loop launches NO vendor-library GEMM or convolution (round-4 verdict, weak #9 / next #7):
one guided DDIM step (two U-Net evaluations with their input gradients, the VAE decode with its input gradient, the masked-L2
guidance, the fused DDIM update) and one plain step run under the profiler on the fp16 token-major model, and no kernel name may
look like hipBLASLt / rocBLAS (`Cijk_*`), MIOpen (`igemm`, `miopen`, `naive_conv`, `Conv*`) or a BLAS gemv / gemm.  This is what
allowed the recorded TunableOp / MIOpen find-db files and `lvdm_amd.configure_tuning()` to be removed in round 5: nothing on this path
reads them.  The kernel list must contain this package's own kernels (a silent torch-form fallback would show up as a missing name)."""
import re

import pytest
import torch

from fill_by_name import fill_by_name
from test_ddim_parallel_gloo import HL, SMALL_UNET, SMALL_VAE, T, WL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LIBRARY = re.compile(r"Cijk_|igemm|miopen|MIOpen|naive_conv|rocblas|hipblaslt|gemv|gemm_kernel|ck::|ck_tile", re.I)
OWN = {"guided": ("k_gemm_nt", "k_conv_mfma", "k_attn", "k_gn_bwd"), "plain": ("k_gemm_nt", "k_conv_mfma", "k_attn", "k_ddim")}


def _kernel_names(fn):
    from torch.profiler import ProfilerActivity, profile
    fn()                                    # warm-up outside the profile (lazy initialisation, packed weights)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = set()
    for ev in prof.events():
        if getattr(ev, "device_type", None) is not None and str(ev.device_type).endswith("CUDA"):
            names.add(ev.name)
    return names


def test_guided_and_plain_ddim_steps_launch_no_library_gemm_or_convolution():
    from lvdm_amd.guidance import LossGuidance
    from lvdm_amd.model import LatentDiffusion
    from lvdm_amd.samplers import DDIMSampler, DDIMSamplerGuidance
    ld = LatentDiffusion(SMALL_UNET, SMALL_VAE).eval()
    fill_by_name(ld.model, std=0.08)
    fill_by_name(ld.first_stage_model, std=0.08)
    ld = ld.to(DEV)
    ld.model.diffusion_model.half().to_token_major()
    ld.first_stage_model.half().to_token_major()
    ld.requires_grad_(False)
    am, dc = ld.apply_model, ld.decode_core
    ld.apply_model = lambda x, t, c, **kw: am(x.half(), t, {k: [v.half() for v in vs] for k, vs in c.items()}, **kw)
    ld.decode_core = lambda z, **kw: dc(z.half(), **kw)
    g = torch.Generator().manual_seed(5)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    cond = {"c_crossattn": [mk(1, 93, 64)], "c_concat": [mk(1, 4, T, HL, WL) * 0.2]}
    uc = {"c_crossattn": [mk(1, 93, 64)], "c_concat": cond["c_concat"]}
    x, n0, n1 = mk(1, 4, T, HL, WL), mk(1, 4, T, HL, WL), mk(1, 4, T, HL, WL)
    fs = torch.tensor([10], device=DEV)

    sg = DDIMSamplerGuidance(ld)
    sg.make_schedule(50, "uniform_trailing", 1.0)
    lg = LossGuidance(ddim_steps=50, recur_steps=1, device=DEV)
    lg.set_hw(2 * HL, 2 * WL)
    lg.set_guidance_images(torch.rand(T, 3, 2 * HL, 2 * WL, generator=g).to(DEV))
    lg.set_guidance_masks((torch.rand(T, 1, 2 * HL, 2 * WL, generator=g) > 0.3).float().to(DEV))
    tg = torch.full((1,), int(sg.ddim_timesteps[30]), dtype=torch.long, device=DEV)

    def guided():
        sg.p_sample_ddim(x, cond, tg, index=30, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, guidance_rescale=0.7,
                         fs=fs, loss_guidance_fn=lg, noise=n0, renoise=n1)

    sp = DDIMSampler(ld)
    sp.make_schedule(50, "uniform_trailing", 1.0)

    def plain():
        with torch.no_grad():
            sp.p_sample_ddim(x, cond, tg, index=30, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, guidance_rescale=0.7,
                             fs=fs, noise=n0)

    for label, fn in (("guided", guided), ("plain", plain)):
        names = _kernel_names(fn)
        assert names, "the profiler saw no device kernels"
        bad = sorted(n for n in names if LIBRARY.search(n))
        assert not bad, (label, bad)
        for own in OWN[label]:
            assert any(own in n for n in names), (label, own, sorted(names)[:40])


def test_raster_step_and_knn_launch_only_this_packages_kernels():
    """The rasterizer's forward + backward and `distCUDA2` launch nothing but `gvd::k_*` / `k_*` kernels of this package: no rocPRIM / hipCUB
    scan or sort (the reference leans on cub for both, `rasterizer_impl.cu:278,304`, `simple_knn.cu:208`), no library memset."""
    import math
    import numpy as np
    import synthetic as syn
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(3)
    P, W, H = 4000, 320, 240
    xyz = rng.normal(size=(P, 3)).astype(np.float32)
    xyz[:, 2] = np.abs(xyz[:, 2]) + 1.0
    cam = syn.make_camera(syn.look_at((0.0, 0.0, -3.0), (0.0, 0.0, 1.0)), math.radians(70), math.radians(55), W, H)
    t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a, np.float32), device=DEV, requires_grad=rg)
    st = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], t([0, 0, 0]), 1.0, t(cam["viewmatrix"]), t(cam["projmatrix"]),
                                       3, t(cam["campos"]), False, False, torch.ones(P, 1, device=DEV))
    lv = dict(means3D=t(xyz, True), means2D=torch.zeros(P, 3, device=DEV, requires_grad=True), opacities=t(rng.uniform(0.1, 0.9, (P, 1)), True),
              shs=t(rng.normal(0, 0.3, (P, 16, 3)), True), scales=t(np.full((P, 3), 0.03), True),
              rotations=t(np.tile([1.0, 0, 0, 0], (P, 1)), True))
    pts = t(xyz)

    def step():
        c, r, d, a = GaussianRasterizer(st)(**lv)
        (c.sum() + d.sum()).backward()
        distCUDA2(pts)

    names = _kernel_names(step)
    own = [n for n in names if re.search(r"\bk_[a-z_0-9]+", n)]
    for k in ("k_preprocess", "k_scatter", "k_render_fwd", "k_render_bwd", "k_gather_bwd", "k_radix_scatter", "k_knn"):
        assert any(k in n for n in own), (k, sorted(names))
    bad = sorted(n for n in names if re.search(r"rocprim|hipcub|cub::|thrust|radix_sort|DeviceScan", n))
    assert not bad, bad
