"""Generates tests/golden/multicond_ref.npz by IMPORTING THE REFERENCE'S OWN PYTHON (build container only):

  lvdm/models/samplers/ddim_multiplecond.py::DDIMSampler.p_sample_ddim -- the three-way classifier-free guidance (text x image) that
  utils_vc/diffusion_utils.py:123-125,177-183 selects with `multiple_cond_cfg` -- against the duck-typed model, inputs and noise draw of
  make_golden_diffusion.py (read back from diffusion_ref.npz, so the two fixtures share them), at three step indices, for cfg_img given
  and defaulted, with and without guidance rescale; plus a whole 6-step `sample()` trajectory (seeded draws injected).
Only arrays are stored -- no reference source text.
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
from fill_by_name import fill_by_name  # noqa: E402


def main():
    G = np.load(os.path.join(HERE, "diffusion_ref.npz"))
    betas, ac = G["g1_betas"], G["g1_alphas_cumprod"]
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    import lvdm.models.samplers.ddim_multiplecond as mc_mod
    from lvdm.models.samplers.ddim_multiplecond import DDIMSampler

    class Duck(torch.nn.Module):
        """The stand-in of make_golden_diffusion.py (same attribute surface, same name-derived weights)."""

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
            self.model = torch.nn.Conv3d(4, 4, 1)
            self.first_stage_model = torch.nn.Conv2d(4, 3, 1)
            fill_by_name(self, std=0.5)

        def apply_model(self, x, t, c, **kw):
            return self.model(x) * (1 + c["c_crossattn"][0].mean())

        def predict_start_from_z_and_v(self, x_t, t, v):
            e = lambda a: a.gather(-1, t).reshape(t.shape[0], 1, 1, 1, 1)
            return e(self.sqrt_alphas_cumprod) * x_t - e(self.sqrt_one_minus_alphas_cumprod) * v

        def predict_eps_from_z_and_v(self, x_t, t, v):
            e = lambda a: a.gather(-1, t).reshape(t.shape[0], 1, 1, 1, 1)
            return e(self.sqrt_alphas_cumprod) * v + e(self.sqrt_one_minus_alphas_cumprod) * x_t

    class CPUSampler(DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    duck = Duck()
    x = torch.tensor(G["step_x"])
    cond = {"c_crossattn": [torch.tensor(G["step_c"])]}
    uc = {"c_crossattn": [torch.tensor(G["step_uc"])]}
    g = torch.Generator().manual_seed(77)
    uc_img = {"c_crossattn": [torch.randn(1, 3, 8, generator=g)]}          # image = yes, text = "" (diffusion_utils.py:177-181)
    noise0 = torch.tensor(G["step_noise0"])
    out = {"mc_uc_img": uc_img["c_crossattn"][0].numpy()}
    mc_mod.noise_like = lambda shape, device, repeat=False: noise0
    for index in (49, 30, 0):
        for tag, cfg_img, resc in (("a", 3.0, 0.7), ("b", None, 0.7), ("c", 1.5, 0.0)):
            s = CPUSampler(duck)
            s.make_schedule(50, "uniform_trailing", 1.0, verbose=False)
            t = torch.full((1,), int(s.ddim_timesteps[index]), dtype=torch.long)
            xp, p0 = s.p_sample_ddim(x, cond, t, index=index, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                     cfg_img=cfg_img, guidance_rescale=resc, unconditional_conditioning_img_nonetext=uc_img)
            out[f"mc{index}{tag}_xprev"], out[f"mc{index}{tag}_x0"] = xp.numpy(), p0.numpy()
    # a whole trajectory through sample(): x_T given, the per-step draws injected in order
    draws = [torch.randn(1, 4, 5, 6, 7, generator=g) for _ in range(6)]
    it = iter(draws)
    mc_mod.noise_like = lambda shape, device, repeat=False: next(it)
    s = CPUSampler(duck)
    xT = torch.randn(1, 4, 5, 6, 7, generator=g)
    samples, inter = s.sample(S=6, batch_size=1, shape=(4, 5, 6, 7), conditioning=cond, verbose=False, unconditional_guidance_scale=7.5,
                              unconditional_conditioning=uc, eta=1.0, cfg_img=2.5, x_T=xT, timestep_spacing="uniform_trailing",
                              guidance_rescale=0.7, unconditional_conditioning_img_nonetext=uc_img, fs=None)
    out["mc_traj_xT"], out["mc_traj_draws"], out["mc_traj_samples"] = xT.numpy(), torch.stack(draws).numpy(), samples.numpy()
    # the plain sampler's sample() with the mask / x0 blending of ddim.py:175-182 (q_sample of x0, or x0 itself with clean_cond) and the
    # intermediates bookkeeping (log_every_t) -- unused by the guidedvd drivers (they pass mask=None) but part of the sampler's surface
    import lvdm.models.samplers.ddim as ddim_mod
    from lvdm.models.samplers.ddim import DDIMSampler as PlainSampler

    class CPUPlain(PlainSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    def q_sample(x0, t, noise=None):
        e = lambda a: a.gather(-1, t).reshape(t.shape[0], 1, 1, 1, 1)
        noise = torch.full_like(x0, 0.25) if noise is None else noise          # (deterministic stand-in draw: the reference draws randn here)
        return e(duck.sqrt_alphas_cumprod) * x0 + e(duck.sqrt_one_minus_alphas_cumprod) * noise
    duck.q_sample = q_sample
    mask = (torch.rand(1, 1, 5, 6, 7, generator=g) > 0.5).float()
    x0 = torch.randn(1, 4, 5, 6, 7, generator=g)
    out["mask"], out["mask_x0"] = mask.numpy(), x0.numpy()
    for tag, clean in (("blend", False), ("clean", True)):
        it2 = iter(draws)
        ddim_mod.noise_like = lambda shape, device, repeat=False: next(it2)
        s = CPUPlain(duck)
        samples, inter = s.sample(S=6, batch_size=1, shape=(4, 5, 6, 7), conditioning=cond, verbose=False, unconditional_guidance_scale=7.5,
                                  unconditional_conditioning=uc, eta=1.0, x_T=xT, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                  mask=mask, x0=x0, log_every_t=2, fs=None, **({"clean_cond": True} if clean else {}))
        out[f"mask_{tag}_samples"] = samples.numpy()
        out[f"mask_{tag}_n_inter"] = np.array([len(inter["x_inter"]), len(inter["pred_x0"])])
        out[f"mask_{tag}_last_pred_x0"] = inter["pred_x0"][-1].numpy()
    # the guided sampler OUTSIDE its index window (ddim_guidance.py:234-235,304,329: `start = 101`, `end = -1`; guidance is applied only for
    # start > index >= end, i.e. never skipped by the 50-step runs of the drivers, skipped for the early steps of a run with more than 101 steps):
    # the step then is the plain update, twice re-noised with recur_steps = 2, and the loss object is never called
    import lvdm.models.samplers.ddim_guidance as ddg_mod
    from lvdm.models.samplers.ddim_guidance import DDIMSamplerGuidance

    class CPUGuided(DDIMSamplerGuidance):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    class NeverCalled:
        verbose, scale_guidance_weight, save_dir, mean_loss = False, False, None, False

        def __init__(self, recur):
            self.recur_steps = recur

        def __call__(self, *a, **k):
            raise AssertionError("the guidance loss must not be evaluated outside the index window")

    duck.first_stage_model.requires_grad_(True)
    for recur in (1, 2):
        it3 = iter(draws)
        ddg_mod.noise_like = lambda shape, device, repeat=False: next(it3)
        s = CPUGuided(duck)
        s.make_schedule(120, "uniform_trailing", 1.0, verbose=False)
        index = 110
        t = torch.full((1,), int(s.ddim_timesteps[index]), dtype=torch.long)
        xp, p0 = s.p_sample_ddim(x, cond, t, index=index, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                 guidance_rescale=0.7, loss_guidance_fn=NeverCalled(recur))
        out[f"outside_r{recur}_xprev"], out[f"outside_r{recur}_x0"] = xp.numpy(), p0.numpy()
    assert all(np.isfinite(v).all() for v in out.values())
    np.savez_compressed(os.path.join(HERE, "multicond_ref.npz"), **out)
    print("wrote multicond_ref.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
