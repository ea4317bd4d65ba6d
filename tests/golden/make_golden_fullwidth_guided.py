"""Full-width guided-step anchor (round-4 verdict, missing #3): ONE guided DDIM step of the REFERENCE's sampler
(lvdm/models/samplers/ddim_guidance.py:205-363) around the REFERENCE's U-Net and VAE decoder AT THE SHIPPED WIDTHS
(configs/inference_pvd_1024.yaml:33-87: model_channels 320 / [1, 2, 4, 4], VAE ch 128 / [1, 2, 4, 4]) on a 40 x 56 latent
(320 x 448 video, the size train_guidedvd.py runs), 3 frames so that the fp32 evaluation fits the build container.
Imports the reference's Python (build container only); stores ARRAYS only:

  * `x_prev32`, `pred_x032`: the step in fp32 (the anchor), CFG 7.5, guidance_rescale 0.7, masked-L2 guidance, eta 1;
  * `x_prev32_plain`: the same step with a zero-gradient loss (rho = 0): x_prev32 - x_prev32_plain is the
    guidance term -rho * d(loss)/dx of Algorithm 1, L12-13;
  * `e16_*`: the error of the SAME reference modules run the way the reference runs them (fp32 weights under fp16 autocast,
    viewcrafter.py:104) against the fp32 anchor -- the bar for the fp16 HIP path, measured, on the same weights and inputs.

Weights by state-dict key name (tests/fill_by_name.py, std 0.02) and inputs from seeded CPU generators: the test regenerates both.
~4 minutes on 8 cores.  python tests/golden/make_golden_fullwidth_guided.py
"""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/third_party/ViewCrafter")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fill_by_name import fill_by_name  # noqa: E402

import lvdm.models.samplers.ddim_guidance as ddg_mod  # noqa: E402
import lvdm.models.utils_diffusion as ud  # noqa: E402
from lvdm.models.samplers.ddim_guidance import DDIMSamplerGuidance  # noqa: E402
from lvdm.modules.networks.ae_modules import Decoder  # noqa: E402
from lvdm.modules.networks.openaimodel3d import UNetModel  # noqa: E402

UNET = dict(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
            channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=1024, use_linear=True,
            use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
            use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
            image_cross_attention=True, default_fs=10, fs_condition=True)
VAE = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
           attn_resolutions=[], dropout=0.0)
from fullwidth_inputs import HL, INDEX, STD, T, WL, inputs  # noqa: E402,F401  (shared with the GPU test)


class Duck(torch.nn.Module):
    """What the sampler reads of LatentDiffusion, around the reference's full-width U-Net and decoder."""

    def __init__(self):
        super().__init__()
        betas = ud.rescale_zero_terminal_snr(ud.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012))
        ac = np.cumprod(1. - betas, axis=0)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.num_timesteps, self.parameterization, self.use_dynamic_rescale = 1000, "v", True
        self.betas, self.alphas_cumprod = f32(betas), f32(ac)
        self.alphas_cumprod_prev = f32(np.append(1., ac[:-1]))
        self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(ac)), f32(np.sqrt(1. - ac))
        self.scale_arr = f32(np.concatenate((np.linspace(1.0, 0.3, 400), np.full(600, 0.3))))   # ddpm3d.py:99-105, base_scale 0.3
        self.device = torch.device("cpu")
        self.model = fill_by_name(UNetModel(**UNET), std=STD).eval()               # keys == lvdm_amd's model.diffusion_model.*
        self.first_stage_model = torch.nn.Module()
        self.first_stage_model.decoder = fill_by_name(Decoder(**VAE), std=STD).eval()
        self.first_stage_model.post_quant_conv = fill_by_name(torch.nn.Conv2d(4, 4, 1), std=0.5)
        self.scale_factor = 0.18215

    def apply_model(self, x, t, c, **kw):   # ddpm3d.py:723-738 + DiffusionWrapper 'hybrid' (ddpm3d.py:1060-1064)
        xc = torch.cat([x] + c["c_concat"], dim=1)
        return self.model(xc, t, context=torch.cat(c["c_crossattn"], 1), fs=kw.get("fs"))

    def predict_start_from_z_and_v(self, x_t, t, v):
        e = lambda a: a.gather(-1, t).reshape(t.shape[0], 1, 1, 1, 1)
        return e(self.sqrt_alphas_cumprod) * x_t - e(self.sqrt_one_minus_alphas_cumprod) * v

    def predict_eps_from_z_and_v(self, x_t, t, v):
        e = lambda a: a.gather(-1, t).reshape(t.shape[0], 1, 1, 1, 1)
        return e(self.sqrt_alphas_cumprod) * v + e(self.sqrt_one_minus_alphas_cumprod) * x_t

    def differentiable_decode_first_stage(self, z):   # ddpm3d.py:646-675, perframe_ae: frame by frame
        outs = []
        for i in range(z.shape[2]):
            zi = 1. / self.scale_factor * z[:, :, i]
            outs.append(self.first_stage_model.decoder(self.first_stage_model.post_quant_conv(zi)))
        return torch.stack(outs, dim=2)


class CPUGuided(DDIMSamplerGuidance):
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


class LG:  # the 'recon' term of LossGuidance (utils/viewcrafter_wrapper.py:123-148)
    recur_steps, verbose, mean_loss, scale_guidance_weight = 1, False, False, False

    def __init__(self, imgs, masks):
        self.g, self.m = imgs, masks

    def __call__(self, D, idx, a, b):
        D = ((D.permute(1, 0, 2, 3) + 1.) / 2.).clamp(0, 1)
        m = self.m[a:b].expand_as(D)
        return {"recon": (0.5 * torch.square(D - self.g[a:b]) * m).sum()}, m.sum()

    def save_pred_x0(self, *a):
        pass


class LGOff(LG):   # a loss with a zero gradient (numel 1, not 0 / 0): the step without its guidance term
    def __call__(self, D, idx, a, b):
        return {"recon": (D * 0.0).sum()}, torch.tensor(1.0)


def step(duck, d, guided, autocast):
    s = CPUGuided(duck)
    s.make_schedule(50, "uniform_trailing", 1.0, verbose=False)
    t = torch.full((1,), int(s.ddim_timesteps[INDEX]), dtype=torch.long)
    it = iter([d["noise0"], d["noise1"]])
    ddg_mod.noise_like = lambda shape, device, repeat=False: next(it)
    ddg_mod.torch.cuda.empty_cache = lambda: None
    cond = {"c_crossattn": [d["ctx_c"]], "c_concat": [d["concat"]]}
    uc = {"c_crossattn": [d["ctx_uc"]], "c_concat": [d["concat"]]}
    with torch.autocast("cpu", dtype=torch.float16, enabled=autocast):
        xp, p0 = s.p_sample_ddim(d["x"], cond, t, index=INDEX, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                 guidance_rescale=0.7, fs=torch.tensor([10]),
                                 loss_guidance_fn=(LG if guided else LGOff)(d["guide_imgs"], d["guide_masks"]))
    return xp.float(), p0.float()


def main():
    torch.manual_seed(0)
    d = inputs()
    t0 = time.time()
    duck = Duck()
    print(f"reference modules built and filled in {time.time() - t0:.0f} s", flush=True)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    out = {}
    xp32, p032 = step(duck, d, True, False)
    print(f"fp32 guided step: {time.time() - t0:.0f} s", flush=True)
    xp32p, _ = step(duck, d, False, False)
    print(f"fp32 unguided step: {time.time() - t0:.0f} s", flush=True)
    out["x_prev32"], out["pred_x032"], out["x_prev32_plain"] = xp32.numpy(), p032.numpy(), xp32p.numpy()
    np.savez_compressed(os.path.join(HERE, "fullwidth_guided_ref.npz"), **out)     # (kept if the slow pass below is interrupted)
    xp16, p016 = step(duck, d, True, True)
    print(f"fp16-autocast guided step: {time.time() - t0:.0f} s", flush=True)
    xp16p, _ = step(duck, d, False, True)
    g32, g16 = xp32 - xp32p, xp16 - xp16p
    out["e16_x_prev"], out["e16_pred_x0"] = np.float64(rel(xp16, xp32)), np.float64(rel(p016, p032))
    out["e16_guidance"] = np.float64(rel(g16, g32))
    out["e16_guidance_cos"] = np.float64(torch.nn.functional.cosine_similarity(g16.flatten(), g32.flatten(), dim=0))
    np.savez_compressed(os.path.join(HERE, "fullwidth_guided_ref.npz"), **out)
    print({k: (v.shape if getattr(v, "ndim", 0) else float(v)) for k, v in out.items()})
    print("wrote", os.path.getsize(os.path.join(HERE, "fullwidth_guided_ref.npz")), "bytes in", f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
