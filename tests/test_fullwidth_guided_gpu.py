"""Full-width guided-step anchor (`-m gpu`; round-4 verdict, missing #3 / next #4).

Every other golden of part B runs a 64-wide miniature or one module at the shipped width.  Here ONE guided DDIM step
(ddim_guidance.py:205-363: two U-Net evaluations with their input gradients, the per-frame VAE decode with its input gradient, the
masked-L2 guidance, CFG 7.5, guidance_rescale 0.7, eta 1) runs on the fp16 HIP path AT THE SHIPPED WIDTHS -- the 1.44 B-parameter
U-Net (model_channels 320, [1, 2, 4, 4]) and the KL-VAE decoder (ch 128, [1, 2, 4, 4]) -- on a 40 x 56 latent (320 x 448 video,
the size train_guidedvd.py runs), 3 frames, against tests/golden/fullwidth_guided_ref.npz: the REFERENCE's sampler around the
REFERENCE's modules with the same name-derived weights and the same seeded inputs, evaluated in fp32 in the build container
(tests/golden/make_golden_fullwidth_guided.py), which also measured the reference's OWN error under fp16 autocast.  The bar is that
error:

      err(HIP fp16 vs reference fp32)  <=  K x err(reference fp16-autocast vs reference fp32),   K = 1.5

for x_prev, pred_x0 and the guidance term x_prev - x_prev(without guidance) = -rho d(loss)/dx, the latter also by direction (cosine).
The step is the one that is 98 % of train_guidedvd.py's diffusion time; at these shapes the machinery that the miniatures cannot
reach runs IN COMBINATION, and the test asserts that it did: frame sheets (5 x 7 and 10 x 14 maps), split-K convolutions, the
phase-decomposed upsampling convolution and its mode-5 input gradient, GradCell hand-overs, the wide-head (d = 512) VAE attention."""
import os

import numpy as np
import pytest
import torch

from fill_by_name import fill_by_name
from fullwidth_inputs import HL, INDEX, STD, T, WL, inputs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"
K = 1.5


def _count_calls(monkeypatch, module, name, counts):
    fn = getattr(module, name)

    def wrapped(*a, **kw):
        counts[name] = counts.get(name, 0) + 1
        return fn(*a, **kw)
    monkeypatch.setattr(module, name, wrapped)


def test_guided_step_at_the_shipped_widths_within_the_references_own_fp16_error(monkeypatch):
    from lvdm_amd import conv, ops, wide_attention
    from lvdm_amd.guidance import LossGuidance
    from lvdm_amd.model import VIEWCRAFTER_UNET, VIEWCRAFTER_VAE, LatentDiffusion
    from lvdm_amd.samplers import DDIMSamplerGuidance
    R = np.load(os.path.join(HERE, "golden", "fullwidth_guided_ref.npz"))
    ld = LatentDiffusion(VIEWCRAFTER_UNET, VIEWCRAFTER_VAE).eval()
    fill_by_name(ld.model.diffusion_model, std=STD)                 # == the reference U-Net's keys
    fill_by_name(ld.first_stage_model.decoder, std=STD)
    fill_by_name(ld.first_stage_model.post_quant_conv, std=0.5)
    ld = ld.to(DEV)
    ld.model.diffusion_model.half().to_token_major()
    ld.first_stage_model.half().to_token_major()
    ld.requires_grad_(False)
    am, dc = ld.apply_model, ld.decode_core
    ld.apply_model = lambda x, t, c, **kw: am(x.half(), t, {k: [v.half() for v in vs] for k, vs in c.items()}, **kw)
    ld.decode_core = lambda z, **kw: dc(z.half(), **kw)
    d = {k: v.to(DEV) for k, v in inputs().items()}
    cond = {"c_crossattn": [d["ctx_c"]], "c_concat": [d["concat"]]}
    uc = {"c_crossattn": [d["ctx_uc"]], "c_concat": [d["concat"]]}
    fs = torch.tensor([10], device=DEV)

    counts = {}
    for name in ("_sheet_conv", "_split_conv"):
        _count_calls(monkeypatch, conv, name, counts)
    puts = {"n": 0}
    put0 = ops.GradCell.put

    def put(self, g):
        ok = put0(self, g)
        puts["n"] += int(ok)
        return ok
    monkeypatch.setattr(ops.GradCell, "put", put)
    modes = set()
    launch0 = conv._launch

    def launch(x, wpk, Cout, mode, *a, **kw):
        modes.add(int(mode))
        return launch0(x, wpk, Cout, mode, *a, **kw)
    monkeypatch.setattr(conv, "_launch", launch)
    wide = {"n": 0}
    wa0 = wide_attention.attention_heads

    def wa(*a, **kw):
        wide["n"] += 1
        return wa0(*a, **kw)
    monkeypatch.setattr(wide_attention, "attention_heads", wa)

    def step(guided, prescale=True):
        s = DDIMSamplerGuidance(ld)
        s.scale_guidance_gradient = prescale
        s.make_schedule(50, "uniform_trailing", 1.0)
        t = torch.full((1,), int(s.ddim_timesteps[INDEX]), dtype=torch.long, device=DEV)
        kw = dict(index=INDEX, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, guidance_rescale=0.7, fs=fs, noise=d["noise0"])
        if not guided:   # the step without its guidance term: the sampler's plain branch (same x_prev formula, same noise)
            with torch.no_grad():
                xp, p0 = s.p_sample_ddim(d["x"], cond, t, **kw)
            return xp.float().cpu(), p0.float().cpu()
        lg = LossGuidance(ddim_steps=50, recur_steps=1, device=DEV)
        lg.set_hw(8 * HL, 8 * WL)
        lg.set_guidance_images(d["guide_imgs"])
        lg.set_guidance_masks(d["guide_masks"])
        xp, p0 = s.p_sample_ddim(d["x"], cond, t, loss_guidance_fn=lg, renoise=d["noise1"], **kw)
        return xp.float().cpu(), p0.float().cpu()

    xp, p0 = step(True)
    xpp, _ = step(False)
    assert torch.isfinite(xp).all() and torch.isfinite(xpp).all()
    # the shapes of this step reach what the miniatures cannot -- and did
    assert counts.get("_sheet_conv", 0) > 0 and counts.get("_split_conv", 0) > 0, counts
    assert conv.UP2 in modes and conv.UP2_BWD in modes, modes
    assert puts["n"] > 100, puts
    assert wide["n"] > 0, wide

    ref = lambda k: torch.tensor(R[k])
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    e_x, e_p = rel(xp, ref("x_prev32")), rel(p0, ref("pred_x032"))
    g_hip, g_ref = xp - xpp, ref("x_prev32") - ref("x_prev32_plain")
    e_g = rel(g_hip, g_ref)
    cos = float(torch.nn.functional.cosine_similarity(g_hip.flatten(), g_ref.flatten(), dim=0))
    print(f"full-width guided step, HIP fp16 vs reference fp32: x_prev {e_x:.3e} (reference fp16 {float(R['e16_x_prev']):.3e}), "
          f"pred_x0 {e_p:.3e} ({float(R['e16_pred_x0']):.3e}), guidance term {e_g:.3e} ({float(R['e16_guidance']):.3e}), "
          f"cosine {cos:.5f} ({float(R['e16_guidance_cos']):.5f})")
    # ... and the same step with the reference's unscaled hand-over of d(loss)/d(pred_x0) to the 16-bit U-Net backward (fp16 subnormals):
    xp_raw, _ = step(True, prescale=False)
    g_raw = xp_raw - xpp
    cos_raw = float(torch.nn.functional.cosine_similarity(g_raw.flatten(), g_ref.flatten(), dim=0))
    print(f"   without the power-of-two pre-scaling of the guidance gradient: x_prev {rel(xp_raw, ref('x_prev32')):.3e}, cosine {cos_raw:.5f}")
    assert e_p <= K * float(R["e16_pred_x0"]), (e_p, float(R["e16_pred_x0"]))
    # absolute bars next to the relative ones (the reference's fp16 guidance term is uncorrelated with its fp32 one on these weights,
    # so K x its error would allow anything): measured 8.5e-3 / 0.9978
    assert e_x <= 3e-2 and cos >= 0.99, (e_x, cos)
    assert e_x <= K * float(R["e16_x_prev"]), (e_x, float(R["e16_x_prev"]))
    assert e_g <= K * float(R["e16_guidance"]), (e_g, float(R["e16_guidance"]))
    assert 1.0 - cos <= K * (1.0 - float(R["e16_guidance_cos"])) + 1e-4, (cos, float(R["e16_guidance_cos"]))
