"""Noise-schedule tables and v-parameterisation helpers.

Restates, in float64 numpy -> float32 buffers exactly like the reference:
  make_beta_schedule('linear')            lvdm/models/utils_diffusion.py:31-53
  rescale_zero_terminal_snr               utils_diffusion.py:112-144
  DDPM.register_schedule (sampling subset) lvdm/models/ddpm3d.py:123-186
  scale_arr (dynamic rescale)             ddpm3d.py:522-527
  make_ddim_timesteps                     utils_diffusion.py:56-76
  make_ddim_sampling_parameters           utils_diffusion.py:79-91
  predict_start/eps_from_z_and_v          ddpm3d.py:239-251
  timestep_embedding                      utils_diffusion.py:8-28
"""
import math

import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":  # "linear" is linear in sqrt(beta); torch.linspace (not numpy's) to match the reference's fp64 bits
        return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64, device="cpu") ** 2).numpy()
    if schedule == "sqrt_linear":
        return torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64, device="cpu").numpy()
    if schedule == "sqrt":
        return (torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64, device="cpu") ** 0.5).numpy()
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        a = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        a = a / a[0]
        return np.clip(1 - a[1:] / a[:-1], 0, 0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")


def rescale_zero_terminal_snr(betas):
    """Algorithm 1 of arXiv:2305.08891: shift/scale sqrt(alpha_bar) so that the last step has zero SNR."""
    abar_sqrt = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    first, last = abar_sqrt[0].copy(), abar_sqrt[-1].copy()
    abar_sqrt = (abar_sqrt - last) * (first / (first - last))
    abar = abar_sqrt ** 2
    alphas = np.concatenate([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


def make_ddim_timesteps(method, num_ddim, num_ddpm):
    if method == "uniform":
        c = num_ddpm // num_ddim
        return np.asarray(list(range(0, num_ddpm, c))) + 1
    if method == "uniform_trailing":
        c = num_ddpm / num_ddim
        return np.flip(np.round(np.arange(num_ddpm, 0, -c))).astype(np.int64) - 1
    if method == "quad":
        return ((np.linspace(0, np.sqrt(num_ddpm * .8), num_ddim)) ** 2).astype(int) + 1
    raise NotImplementedError(f'There is no ddim discretization method called "{method}"')


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    alphacums = np.asarray(alphacums)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


_FREQS = {}


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    key = (half, max_period, timesteps.device)
    freqs = _FREQS.get(key)
    if freqs is None:  # formed on the CPU exactly like the reference (utils_diffusion.py:18-20), moved to the device ONCE
        freqs = _FREQS[key] = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class DiffusionSchedule(torch.nn.Module):
    """The fp32 buffers a sampler reads from the LatentDiffusion object (ViewCrafter yaml:
    linear 0.00085->0.012, 1000 steps, zero-terminal-SNR, v-parameterisation, dynamic rescale to 0.3)."""

    def __init__(self, timesteps=1000, linear_start=0.00085, linear_end=0.012, beta_schedule="linear",
                 rescale_betas_zero_snr=True, parameterization="v", use_dynamic_rescale=True, base_scale=0.3,
                 turning_step=400, cosine_s=8e-3, v_posterior=0., full_tables=False):
        super().__init__()
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end, cosine_s)
        if rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        ac = np.cumprod(1. - betas, axis=0)
        ac_prev = np.append(1., ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        self.parameterization = parameterization
        self.use_dynamic_rescale = use_dynamic_rescale
        f32 = lambda a: torch.tensor(a, dtype=torch.float32, device="cpu")   # host tables (also under a torch.device(...) context)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(ac))
        self.register_buffer("alphas_cumprod_prev", f32(ac_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1. - ac)))
        if full_tables:
            # the remaining persistent buffers of DDPM.register_schedule (ddpm3d.py:152-171): not read by the samplers, but
            # part of the checkpoint's state dict, so a strict load needs them
            with np.errstate(divide="ignore", invalid="ignore"):
                self.register_buffer("log_one_minus_alphas_cumprod", f32(np.log(1. - ac)))
                if parameterization != "v":
                    self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1. / ac)))
                    self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1. / ac - 1)))
                else:
                    self.register_buffer("sqrt_recip_alphas_cumprod", torch.zeros(self.num_timesteps, device="cpu"))
                    self.register_buffer("sqrt_recipm1_alphas_cumprod", torch.zeros(self.num_timesteps, device="cpu"))
                pv = (1 - v_posterior) * betas * (1. - ac_prev) / (1. - ac) + v_posterior * betas
                self.register_buffer("posterior_variance", f32(pv))
                self.register_buffer("posterior_log_variance_clipped", f32(np.log(np.maximum(pv, 1e-20))))
                self.register_buffer("posterior_mean_coef1", f32(betas * np.sqrt(ac_prev) / (1. - ac)))
                self.register_buffer("posterior_mean_coef2", f32((1. - ac_prev) * np.sqrt(1. - betas) / (1. - ac)))
        if use_dynamic_rescale:
            self.register_buffer("scale_arr", f32(np.concatenate((np.linspace(1.0, base_scale, turning_step),
                                                                  np.full(self.num_timesteps, base_scale)))))

    @staticmethod
    def _gather(a, t, ndim):
        return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))

    def predict_start_from_z_and_v(self, x_t, t, v):
        return (self._gather(self.sqrt_alphas_cumprod, t, x_t.dim()) * x_t
                - self._gather(self.sqrt_one_minus_alphas_cumprod, t, x_t.dim()) * v)

    def predict_eps_from_z_and_v(self, x_t, t, v):
        return (self._gather(self.sqrt_alphas_cumprod, t, x_t.dim()) * v
                + self._gather(self.sqrt_one_minus_alphas_cumprod, t, x_t.dim()) * x_t)

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return (self._gather(self.sqrt_alphas_cumprod, t, x_start.dim()) * x_start
                + self._gather(self.sqrt_one_minus_alphas_cumprod, t, x_start.dim()) * noise)


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """utils_diffusion.py:147-158 (std over all non-batch dims, unbiased)."""
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg
