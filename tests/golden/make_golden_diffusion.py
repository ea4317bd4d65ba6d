"""Generates tests/golden/diffusion_ref.npz by IMPORTING THE REFERENCE'S OWN PYTHON (build container only):

  G1 schedule tables           lvdm/models/utils_diffusion.py + the formulas of ddpm3d.py:123-186,522-527
  G2 timestep_embedding        utils_diffusion.py:8-28
  G3-G6 UNetModel              lvdm/modules/networks/openaimodel3d.py (tiny config, every zero-init module
                               re-randomised with a fixed seed; weights are stored so the rebuilt model loads them)
  G7 plain DDIM step           lvdm/models/samplers/ddim.py::p_sample_ddim against a duck-typed model
  G8 guided DDIM step          lvdm/models/samplers/ddim_guidance.py::p_sample_ddim (tiny stand-in decoder)
  G9 VAE Decoder               lvdm/modules/networks/ae_modules.py::Decoder (tiny config)
Only arrays (inputs, weights, outputs) are stored -- no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VC = "/root/reference/third_party/ViewCrafter"
sys.path.insert(0, VC)
for stub in ("cv2",):
    sys.modules.setdefault(stub, types.ModuleType(stub))


sys.path.insert(0, os.path.join(ROOT, "tests"))
from fill_by_name import fill_by_name  # deterministic weights from parameter NAMES (no weight tensors stored)


def main():
    out = {}
    from lvdm.models import utils_diffusion as ud
    # ---------------- G1: schedule ----------------
    betas = ud.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    betas = ud.rescale_zero_terminal_snr(betas)
    ac = np.cumprod(1. - betas, axis=0)
    out["g1_betas"], out["g1_alphas_cumprod"] = betas, ac
    for name, method in (("trailing", "uniform_trailing"), ("uniform", "uniform")):
        ts = ud.make_ddim_timesteps(method, 50, 1000, verbose=False)
        sig, a, ap = ud.make_ddim_sampling_parameters(torch.tensor(ac, dtype=torch.float32).numpy(), ts, 1.0, verbose=False)
        out[f"g1_ts_{name}"], out[f"g1_sig_{name}"], out[f"g1_a_{name}"], out[f"g1_aprev_{name}"] = ts, sig, a, ap
    # ---------------- G2 ----------------
    tt = torch.tensor([0, 19, 500, 999])
    out["g2_t"], out["g2_emb"] = tt.numpy(), ud.timestep_embedding(tt, 320).numpy()
    # ---------------- G3-G6: U-Net ----------------
    from lvdm.modules.networks.openaimodel3d import UNetModel
    cfg = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1,
               channel_mult=[1, 2], dropout=0.1, num_head_channels=32, transformer_depth=1, context_dim=48,
               use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True,
               temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False, temporal_length=16,
               addition_attention=True, image_cross_attention=True, default_fs=10, fs_condition=True)
    torch.manual_seed(0)
    unet = fill_by_name(UNetModel(**cfg)).eval()
    out["unet_keys"] = np.array(sorted(unet.state_dict().keys()))
    g = torch.Generator().manual_seed(2)
    for tag, T, L in (("shared", 3, 77 + 20), ("perframe", 2, 77 + 2 * 16)):
        x = torch.randn(1, 8, T, 8, 8, generator=g)
        ctx = torch.randn(1, L, 48, generator=g)
        tstep = torch.tensor([400])
        fs = torch.tensor([10])
        x.requires_grad_(True)
        y = unet(x, tstep, context=ctx, fs=fs)
        gy = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad(y, x, gy)
        out[f"unet_{tag}_x"], out[f"unet_{tag}_ctx"], out[f"unet_{tag}_y"] = x.detach().numpy(), ctx.numpy(), y.detach().numpy()
        out[f"unet_{tag}_gy"], out[f"unet_{tag}_gx"] = gy.numpy(), gx.numpy()
    # ---------------- G9: VAE decoder ----------------
    from lvdm.modules.networks.ae_modules import Decoder
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2],
              num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    torch.manual_seed(3)
    dec = fill_by_name(Decoder(**dd)).eval()
    out["dec_keys"] = np.array(sorted(dec.state_dict().keys()))
    z = torch.randn(1, 4, 6, 8, generator=g).requires_grad_(True)
    img = dec(z)
    gi = torch.randn(img.shape, generator=g)
    (gz,) = torch.autograd.grad(img, z, gi)
    out["dec_z"], out["dec_img"], out["dec_gi"], out["dec_gz"] = z.detach().numpy(), img.detach().numpy(), gi.numpy(), gz.numpy()

    # ---------------- G7 / G8: sampler steps against a duck-typed model ----------------
    from lvdm.models.samplers.ddim import DDIMSampler
    from lvdm.models.samplers.ddim_guidance import DDIMSamplerGuidance
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)

    class Duck(torch.nn.Module):
        """Exposes what the samplers read; the 'U-Net' is a tiny differentiable stand-in."""

        def __init__(self):
            super().__init__()
            self.num_timesteps = 1000
            self.parameterization = "v"
            self.use_dynamic_rescale = True
            self.betas, self.alphas_cumprod = f32(betas), f32(ac)
            self.alphas_cumprod_prev = f32(np.append(1., ac[:-1]))
            self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(ac)), f32(np.sqrt(1. - ac))
            self.scale_arr = f32(np.concatenate((np.linspace(1.0, 0.3, 400), np.full(1000, 0.3))))
            self.device = torch.device("cpu")
            self.model = torch.nn.Conv3d(4, 4, 1)          # "U-Net": v = conv(x) * (1 + mean(c))
            self.first_stage_model = torch.nn.Conv2d(4, 3, 1)  # "decoder"
            fill_by_name(self, std=0.5)

        def apply_model(self, x, t, c, **kw):
            return self.model(x) * (1 + c["c_crossattn"][0].mean())

        def predict_start_from_z_and_v(self, x_t, t, v):
            e = lambda a: a.gather(-1, t).reshape(t.shape[0], 1, 1, 1, 1)
            return e(self.sqrt_alphas_cumprod) * x_t - e(self.sqrt_one_minus_alphas_cumprod) * v

        def predict_eps_from_z_and_v(self, x_t, t, v):
            e = lambda a: a.gather(-1, t).reshape(t.shape[0], 1, 1, 1, 1)
            return e(self.sqrt_alphas_cumprod) * v + e(self.sqrt_one_minus_alphas_cumprod) * x_t

        def differentiable_decode_first_stage(self, z):
            return torch.tanh(self.first_stage_model(z[:, :, 0]))[:, :, None]

    class CPUSampler(DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    class CPUGuided(DDIMSamplerGuidance):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    duck = Duck()
    x = torch.randn(1, 4, 5, 6, 7, generator=g)
    cond = {"c_crossattn": [torch.randn(1, 3, 8, generator=g)]}
    uc = {"c_crossattn": [torch.randn(1, 3, 8, generator=g)]}
    out["step_x"], out["step_c"], out["step_uc"] = x.numpy(), cond["c_crossattn"][0].numpy(), uc["c_crossattn"][0].numpy()
    import lvdm.models.samplers.ddim as ddim_mod
    import lvdm.models.samplers.ddim_guidance as ddg_mod
    noises = [torch.randn(1, 4, 5, 6, 7, generator=g) for _ in range(2)]
    out["step_noise0"], out["step_noise1"] = noises[0].numpy(), noises[1].numpy()
    for index in (49, 30, 0):
        s = CPUSampler(duck)
        s.make_schedule(50, "uniform_trailing", 1.0, verbose=False)
        t = torch.full((1,), int(s.ddim_timesteps[index]), dtype=torch.long)
        ddim_mod.noise_like = lambda shape, device, repeat=False: noises[0]
        xp, p0 = s.p_sample_ddim(x, cond, t, index=index, unconditional_guidance_scale=7.5,
                                 unconditional_conditioning=uc, guidance_rescale=0.7)
        out[f"plain{index}_xprev"], out[f"plain{index}_x0"] = xp.numpy(), p0.numpy()

    class LG:  # the 'recon' term of LossGuidance (viewcrafter_wrapper.py:123-148), restated for the duck decoder
        recur_steps, verbose, mean_loss, scale_guidance_weight = 1, False, False, False

        def __init__(self, imgs, masks):
            self.g, self.m = imgs, masks

        def __call__(self, D, idx, a, b):
            D = ((D.permute(1, 0, 2, 3) + 1.) / 2.).clamp(0, 1)
            m = self.m[a:b].expand_as(D)
            return {"recon": (0.5 * torch.square(D - self.g[a:b]) * m).sum()}, m.sum()

        def save_pred_x0(self, *a):
            pass

    gi_, gm_ = torch.rand(5, 3, 6, 7, generator=g), (torch.rand(5, 1, 6, 7, generator=g) > 0.3).float()
    out["guide_imgs"], out["guide_masks"] = gi_.numpy(), gm_.numpy()
    for index in (40, 3):
        s = CPUGuided(duck)
        s.make_schedule(50, "uniform_trailing", 1.0, verbose=False)
        t = torch.full((1,), int(s.ddim_timesteps[index]), dtype=torch.long)
        it = iter(noises)
        ddg_mod.noise_like = lambda shape, device, repeat=False: next(it)
        ddg_mod.torch.cuda.empty_cache = lambda: None
        xp, p0 = s.p_sample_ddim(x, cond, t, index=index, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                 guidance_rescale=0.7, loss_guidance_fn=LG(gi_, gm_))
        out[f"guided{index}_xprev"], out[f"guided{index}_x0"] = xp.numpy(), p0.numpy()
    # G8b (round 5): the same step with recur_steps = 2 (viewcrafter_wrapper.py:51 default of LossGuidance; train_guidedvd.py passes
    # opt.guidance_recur_steps = 1): two passes of the loop, four noise draws (sigma noise + re-noise per pass, ddim_guidance.py:287,360).
    # The two extra noises come from a generator of their own so that every array above keeps its bits.
    class LG2(LG):
        recur_steps = 2
    g3 = torch.Generator().manual_seed(333)
    noises4 = noises + [torch.randn(1, 4, 5, 6, 7, generator=g3) for _ in range(2)]
    out["step_noise2"], out["step_noise3"] = noises4[2].numpy(), noises4[3].numpy()
    s = CPUGuided(duck)
    s.make_schedule(50, "uniform_trailing", 1.0, verbose=False)
    t = torch.full((1,), int(s.ddim_timesteps[40]), dtype=torch.long)
    it = iter(noises4)
    ddg_mod.noise_like = lambda shape, device, repeat=False: next(it)
    xp, p0 = s.p_sample_ddim(x, cond, t, index=40, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                             guidance_rescale=0.7, loss_guidance_fn=LG2(gi_, gm_))
    out["guided40_recur2_xprev"], out["guided40_recur2_x0"] = xp.numpy(), p0.numpy()
    # ---------------- G6b: a U-Net whose heads are 64 wide (the ViewCrafter head size) on a 16x24 latent: the case the
    # `-m gpu` tests push through the fp16 HIP path (MFMA attention needs d = 64; the convolution tiles need > 8x8) ----
    cfg64 = dict(cfg, num_head_channels=64, context_dim=64)
    unet64 = fill_by_name(UNetModel(**cfg64)).eval()
    g2 = torch.Generator().manual_seed(64)
    for tag, T, L in (("shared", 4, 77 + 24), ("perframe", 2, 77 + 2 * 16)):
        x = torch.randn(1, 8, T, 16, 24, generator=g2)
        ctx = torch.randn(1, L, 64, generator=g2)
        x.requires_grad_(True)
        y = unet64(x, torch.tensor([250]), context=ctx, fs=torch.tensor([10]))
        gy = torch.randn(y.shape, generator=g2)
        (gx,) = torch.autograd.grad(y, x, gy)
        out[f"unet64_{tag}_x"], out[f"unet64_{tag}_ctx"], out[f"unet64_{tag}_y"] = x.detach().numpy(), ctx.numpy(), y.detach().numpy()
        out[f"unet64_{tag}_gy"], out[f"unet64_{tag}_gx"] = gy.numpy(), gx.numpy()
    # ---------------- G9b: VAE decoder with a 64-channel single-head mid attention on a 12x20 latent ----
    dd64 = dict(dd, ch=32, ch_mult=[1, 2])
    dec64 = fill_by_name(Decoder(**dd64)).eval()
    z = torch.randn(1, 4, 12, 20, generator=g2).requires_grad_(True)
    img = dec64(z)
    gi = torch.randn(img.shape, generator=g2)
    (gz,) = torch.autograd.grad(img, z, gi)
    out["dec64_z"], out["dec64_img"], out["dec64_gi"], out["dec64_gz"] = z.detach().numpy(), img.detach().numpy(), gi.numpy(), gz.numpy()
    path = os.path.join(HERE, "diffusion_ref.npz")
    if os.path.exists(path):   # regenerating must not move an array that is already pinned
        old = np.load(path)
        moved = [k for k in old.files if k not in out or not np.array_equal(old[k], out[k])]
        print("arrays that differ from the existing fixture:", moved or "none")
    np.savez_compressed(path, **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "diffusion_ref.npz")), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
