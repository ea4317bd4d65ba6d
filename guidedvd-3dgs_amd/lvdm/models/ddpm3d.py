"""lvdm.models.ddpm3d (reference: lvdm/models/ddpm3d.py:40-1491), the sampling half.

`DDPM` / `LatentDiffusion` / `LatentVisualDiffusion` / `VIPLatentDiffusion` / `DiffusionWrapper` take the keyword set of
`configs/inference_pvd_1024.yaml:5-110` (training-only keywords are accepted and ignored), build their sub-models from the
yaml's `target:` nodes, and expose what `utils_vc/diffusion_utils.py:118-223`, `viewcrafter.py:315-335` and the samplers
touch: the schedule buffers (same persistent set -> strict checkpoint load), `apply_model`, `get_learned_conditioning`,
`encode_first_stage` / `decode_first_stage` / `differentiable_decode_first_stage`, `embedder`, `image_proj_model`,
`cond_stage_model`, `model.conditioning_key`, `model.diffusion_model`, `uncond_type`, `perframe_ae`, `device`.
No pytorch-lightning: plain nn.Module.

MI355X specifics: on a ROCm device the U-Net and the VAE are converted ONCE, at the first `apply_model` / decode after the
checkpoint is loaded, to fp16 token-major form (`half().to_token_major()`): what autocast (viewcrafter.py:104) would cast
per call, cast once; GroupNorm statistics and the sampler arithmetic stay fp32.
"""
import torch
import torch.nn as nn

from lvdm_amd.model import DiffusionWrapper as _Wrapper, _cfg_get, _plain, instantiate_from_config
from lvdm_amd.schedule import DiffusionSchedule
from lvdm_amd.vae import DiagonalGaussianDistribution


def disabled_train(self, mode=True):
    return self


def _node(config):
    """yaml node -> {'target', 'params'} with plain-python params."""
    return {"target": _cfg_get(config, "target"), "params": _plain(_cfg_get(config, "params") or {})}


class DiffusionWrapper(_Wrapper):
    def __init__(self, diff_model_config, conditioning_key):
        super().__init__(diff_model_config if isinstance(diff_model_config, nn.Module)
                         else instantiate_from_config(_node(diff_model_config)), conditioning_key)


class DDPM(DiffusionSchedule):
    """ddpm3d.py:40-186 (schedule + wrapper); everything about losses / EMA / logging is training-side and absent."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None, ignore_keys=(),
                 load_only_unet=False, monitor=None, use_ema=True, first_stage_key="image", image_size=256, channels=3,
                 log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, given_betas=None,
                 original_elbo_weight=0., v_posterior=0., l_simple_weight=1., conditioning_key=None, parameterization="eps",
                 scheduler_config=None, use_positional_encodings=False, learn_logvar=False, logvar_init=0.,
                 rescale_betas_zero_snr=False, use_dynamic_rescale=False, base_scale=0.7, turning_step=400):
        assert parameterization in ["eps", "x0", "v"], 'currently only supporting "eps" and "x0" and "v"'
        if given_betas is not None or use_ema:
            raise NotImplementedError("given_betas / use_ema are training-side options (the shipped yaml sets use_ema: False)")
        super().__init__(timesteps=timesteps, linear_start=linear_start, linear_end=linear_end, beta_schedule=beta_schedule,
                         rescale_betas_zero_snr=rescale_betas_zero_snr, parameterization=parameterization,
                         use_dynamic_rescale=use_dynamic_rescale, base_scale=base_scale, turning_step=turning_step,
                         cosine_s=cosine_s, v_posterior=v_posterior, full_tables=True)
        self.cond_stage_model = None
        self.clip_denoised, self.log_every_t, self.first_stage_key, self.channels = clip_denoised, log_every_t, first_stage_key, channels
        self.temporal_length = _cfg_get(_cfg_get(unet_config, "params"), "temporal_length")
        self.image_size = [image_size, image_size] if isinstance(image_size, int) else list(image_size)
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = False
        self.rescale_betas_zero_snr = rescale_betas_zero_snr
        self.logvar = torch.full(fill_value=logvar_init, size=(self.num_timesteps,))
        self._native_ready = False

    @property
    def device(self):
        return self.betas.device


class LatentDiffusion(DDPM):
    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="caption",
                 cond_stage_trainable=False, cond_stage_forward=None, conditioning_key=None, uncond_prob=0.2,
                 uncond_type="empty_seq", scale_factor=1.0, scale_by_std=False, encoder_type="2d", only_model=False,
                 noise_strength=0, use_dynamic_rescale=False, base_scale=0.7, turning_step=400, loop_video=False,
                 fps_condition_type='fs', perframe_ae=False, logdir=None, rand_cond_frame=False,
                 en_and_decode_n_samples_a_time=None, *args, **kwargs):
        kwargs.pop("ckpt_path", None)
        kwargs.pop("ignore_keys", None)
        self.num_timesteps_cond = 1 if num_timesteps_cond is None else num_timesteps_cond
        if scale_by_std:
            raise NotImplementedError("scale_by_std is a training-side option")
        super().__init__(*args, conditioning_key=conditioning_key or "crossattn", use_dynamic_rescale=use_dynamic_rescale,
                         base_scale=base_scale, turning_step=turning_step, **kwargs)
        self.cond_stage_trainable, self.cond_stage_key, self.cond_stage_forward = cond_stage_trainable, cond_stage_key, cond_stage_forward
        self.noise_strength, self.loop_video, self.fps_condition_type = noise_strength, loop_video, fps_condition_type
        self.perframe_ae, self.logdir, self.rand_cond_frame = perframe_ae, logdir, rand_cond_frame
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        self.ae_frames_per_call = en_and_decode_n_samples_a_time   # frames per VAE call under perframe_ae; None = all (vae.py)
        self.scale_factor = scale_factor
        self.first_stage_model = self._frozen(instantiate_from_config(_node(first_stage_config)))
        self.cond_stage_model = self._frozen(instantiate_from_config(_node(cond_stage_config)))
        self.first_stage_config, self.cond_stage_config = first_stage_config, cond_stage_config
        self.clip_denoised = False
        self.encoder_type = encoder_type
        assert encoder_type in ["2d", "3d"]
        assert uncond_type in ["zero_embed", "empty_seq"]
        self.uncond_prob, self.uncond_type = uncond_prob, uncond_type
        self.classifier_free_guidance = uncond_prob > 0

    @staticmethod
    def _frozen(model):
        if model is None:
            return None
        model = model.eval()
        model.train = disabled_train.__get__(model)
        for p in model.parameters():
            p.requires_grad = False
        return model

    # ---- conditioning (once per video) ----
    def get_learned_conditioning(self, c):   # ddpm3d.py:598-609
        if self.cond_stage_forward is None:
            if hasattr(self.cond_stage_model, "encode") and callable(self.cond_stage_model.encode):
                c = self.cond_stage_model.encode(c)
                if isinstance(c, DiagonalGaussianDistribution):
                    c = c.mode()
            else:
                c = self.cond_stage_model(c)
        else:
            c = getattr(self.cond_stage_model, self.cond_stage_forward)(c)
        return c

    # ---- MI355X-native form of the heavy sub-models ----
    def _prepare_native(self):
        if self._native_ready or self.device.type != "cuda":
            return
        unet = self.model.diffusion_model
        if next(unet.parameters()).dtype == torch.float32:
            unet.half()
        unet.eval().to_token_major()
        for p in unet.parameters():
            p.requires_grad_(False)
        if next(self.first_stage_model.parameters()).dtype == torch.float32:
            self.first_stage_model.half()
        self._native_ready = True

    def apply_model(self, x_noisy, t, cond, **kwargs):   # ddpm3d.py:723-738
        self._prepare_native()
        if not isinstance(cond, dict):
            cond = {"c_concat" if self.model.conditioning_key == "concat" else "c_crossattn": cond if isinstance(cond, list) else [cond]}
        wdtype = next(self.model.diffusion_model.parameters()).dtype
        cond = {k: [v.to(wdtype) for v in vs] for k, vs in cond.items()}
        fwd_kw = {k: v for k, v in kwargs.items() if k in ("fs", "features_adapter")}
        out = self.model(x_noisy.to(wdtype), t, **cond, **fwd_kw)
        out = out[0] if isinstance(out, tuple) else out
        return out.to(x_noisy.dtype)

    # ---- VAE ----
    def get_first_stage_encoding(self, encoder_posterior, noise=None):   # ddpm3d.py:611-619
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample(noise=noise)
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    def _fs_dtype(self):
        return next(self.first_stage_model.parameters()).dtype

    @torch.no_grad()
    def encode_first_stage(self, x):   # ddpm3d.py:621-644
        self._prepare_native()
        reshape_back = self.encoder_type == "2d" and x.dim() == 5
        if reshape_back:
            b, _, t, _, _ = x.shape
            x = x.transpose(1, 2).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])
        xd = x.to(self._fs_dtype())
        # perframe_ae=False (one call for all b*t frames in the reference) takes the chunked route too: every VAE norm is per
        # sample, so the values are the same, and the MFMA convolutions index with 32-bit offsets (25 x 576 x 1024 x 256
        # elements in one call would exceed them); only perframe_ae=True honours the caller's frames-per-call bound
        res = self.first_stage_model.perframe(lambda xx: self.get_first_stage_encoding(self.first_stage_model.encode(xx)).detach(),
                                              xd, self.ae_frames_per_call if self.perframe_ae else None, latent=False)
        res = res.to(x.dtype)
        if reshape_back:
            res = res.reshape(b, t, *res.shape[1:]).transpose(1, 2)
        return res

    def decode_core(self, z, **kwargs):   # ddpm3d.py:646-667
        self._prepare_native()
        reshape_back = self.encoder_type == "2d" and z.dim() == 5
        if reshape_back:
            b, _, t, _, _ = z.shape
            z = z.transpose(1, 2).reshape(b * t, z.shape[1], z.shape[3], z.shape[4])
        zd = (1. / self.scale_factor * z).to(self._fs_dtype())
        res = self.first_stage_model.perframe(lambda zz: self.first_stage_model.decode(zz, **kwargs), zd,
                                              self.ae_frames_per_call if self.perframe_ae else None)   # (see encode_first_stage)
        res = res.to(z.dtype)
        if reshape_back:
            res = res.reshape(b, t, *res.shape[1:]).transpose(1, 2)
        return res

    @torch.no_grad()
    def decode_first_stage(self, z, **kwargs):
        return self.decode_core(z, **kwargs)

    def differentiable_decode_first_stage(self, z, **kwargs):
        return self.decode_core(z, **kwargs)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training forward (p_losses) is not part of the sampling hot path")


class LatentVisualDiffusion(LatentDiffusion):
    def __init__(self, img_cond_stage_config, image_proj_stage_config, freeze_embedder=True, image_proj_model_trainable=True,
                 fix_temporal=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.image_proj_model_trainable = image_proj_model_trainable
        self.embedder = instantiate_from_config(_node(img_cond_stage_config))
        if freeze_embedder:
            self.embedder = self._frozen(self.embedder)
        self.image_proj_model = instantiate_from_config(_node(image_proj_stage_config))
        if not image_proj_model_trainable:
            self.image_proj_model = self._frozen(self.image_proj_model)
        self.fix_temporal = fix_temporal


class VIPLatentDiffusion(LatentVisualDiffusion):
    """ddpm3d.py:1250: differs from its parent only in training-batch preparation."""
