"""The LatentDiffusion-shaped object the samplers duck-type against.

Provides exactly what `DDIMSampler` / `DDIMSamplerGuidance` read from `model` in the reference
(SURVEY 8b "Sampler API"): num_timesteps, betas, alphas_cumprod(_prev), device, use_dynamic_rescale, scale_arr,
parameterization, apply_model, predict_*_from_z_and_v, q_sample, differentiable_decode_first_stage,
decode_first_stage, .model (DiffusionWrapper with .diffusion_model), .first_stage_model.
"""
import importlib

import torch
import torch.nn as nn

from .schedule import DiffusionSchedule
from .unet import UNetModel
from .vae import AutoencoderKLDecoder

# configs/inference_pvd_1024.yaml:33-64 / :66-87
VIEWCRAFTER_UNET = dict(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                        num_res_blocks=2, channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64,
                        transformer_depth=1, context_dim=1024, use_linear=True, use_checkpoint=False,
                        temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
                        use_relative_position=False, use_causal_attention=False, temporal_length=16,
                        addition_attention=True, image_cross_attention=True, default_fs=10, fs_condition=True)
VIEWCRAFTER_VAE = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                       ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def viewcrafter_yaml_node(clip_cfg=None, unet_over=None, vae_over=None):
    """configs/inference_pvd_1024.yaml:4-110 as the nested mapping OmegaConf hands to instantiate_from_config."""
    unet = dict(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=1024,
                use_linear=True, use_checkpoint=True, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
                use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
                image_cross_attention=True, default_fs=10, fs_condition=True)
    unet.update(unet_over or {})
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    dd.update(vae_over or {})
    extra = {} if clip_cfg is None else {"model_cfg": clip_cfg}
    return {"target": "lvdm.models.ddpm3d.VIPLatentDiffusion", "params": dict(
        rescale_betas_zero_snr=True, parameterization="v", linear_start=0.00085, linear_end=0.012, num_timesteps_cond=1,
        log_every_t=200, timesteps=1000, first_stage_key="video", cond_stage_key="caption", cond_stage_trainable=False,
        image_proj_model_trainable=False, conditioning_key="hybrid", image_size=[72, 128], channels=4, scale_by_std=False,
        scale_factor=0.18215, use_ema=False, uncond_prob=0.05, uncond_type="empty_seq", rand_cond_frame=True,
        use_dynamic_rescale=True, base_scale=0.3, fps_condition_type="fps", perframe_ae=True, loop_video="Flase",
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": unet},
        first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                            "params": dict(embed_dim=4, monitor="val/rec_loss", ddconfig=dd, lossconfig={"target": "torch.nn.Identity"})},
        cond_stage_config={"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder",
                           "params": dict(freeze=True, layer="penultimate", **extra)},
        img_cond_stage_config={"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2",
                               "params": dict(freeze=True, **({} if clip_cfg is None else {"model_cfg": clip_cfg}))},
        image_proj_stage_config={"target": "lvdm.modules.encoders.resampler.Resampler",
                                 "params": dict(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280,
                                                output_dim=1024, ff_mult=4, video_length=16)})}


def _cfg_get(config, key, default=None):
    """Item access that works for dicts, OmegaConf nodes and attribute namespaces."""
    if hasattr(config, "get"):
        return config.get(key, default)
    return getattr(config, key, default)


def instantiate_from_config(config):
    """utils_vc/diffusion_utils.py:32-47: build `config["target"]` with `config["params"]`."""
    target = _cfg_get(config, "target")
    if target is None:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    module, cls = target.rsplit(".", 1)
    params = _cfg_get(config, "params") or {}
    return getattr(importlib.import_module(module), cls)(**{k: params[k] for k in params})


def _plain(node):
    """OmegaConf / nested containers -> plain python (lists stay lists)."""
    if hasattr(node, "items"):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)) or type(node).__name__ == "ListConfig":
        return [_plain(v) for v in node]
    return node


class DiffusionWrapper(nn.Module):
    """ddpm3d.py:1420-1491, 'hybrid' conditioning: channel-concat c_concat, token-concat c_crossattn."""

    def __init__(self, unet, conditioning_key="hybrid"):
        super().__init__()
        if conditioning_key != "hybrid":
            raise NotImplementedError("ViewCrafter uses conditioning_key='hybrid'")
        if not isinstance(unet, nn.Module):   # the reference passes the yaml node (ddpm3d.py:1421-1424)
            unet = instantiate_from_config(unet)
        self.diffusion_model = unet
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat=None, c_crossattn=None, **kwargs):
        xc = torch.cat([x] + list(c_concat), dim=1)
        cc = torch.cat(list(c_crossattn), 1)
        return self.diffusion_model(xc, t, context=cc, **kwargs)


class LatentDiffusion(DiffusionSchedule):
    def __init__(self, unet_config=None, first_stage_config=None, scale_factor=0.18215, perframe_ae=True, **schedule_kw):
        super().__init__(**schedule_kw)
        self.model = DiffusionWrapper(UNetModel(**(unet_config or VIEWCRAFTER_UNET)))
        self.first_stage_model = AutoencoderKLDecoder(first_stage_config or VIEWCRAFTER_VAE)
        self.scale_factor = scale_factor
        self.perframe_ae = perframe_ae
        self.ae_frames_per_call = None   # frames per VAE call under perframe_ae (vae.py: perframe); None = all

    @property
    def device(self):
        return self.betas.device

    def apply_model(self, x_noisy, t, cond, **kwargs):  # ddpm3d.py:723-738
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        # sampler-only kwargs the reference's UNet swallows in **kwargs (openaimodel3d.py:548)
        fwd_kw = {k: v for k, v in kwargs.items() if k in ("fs", "features_adapter")}
        out = self.model(x_noisy, t, **cond, **fwd_kw)
        return out[0] if isinstance(out, tuple) else out

    def decode_core(self, z, **kwargs):  # ddpm3d.py:646-667
        reshape_back = z.dim() == 5
        if reshape_back:
            b, _, t, _, _ = z.shape
            z = z.transpose(1, 2).reshape(b * t, z.shape[1], z.shape[3], z.shape[4])
        if not self.perframe_ae:
            res = self.first_stage_model.decode(1. / self.scale_factor * z, **kwargs)
        else:
            res = self.first_stage_model.perframe(lambda zz: self.first_stage_model.decode(1. / self.scale_factor * zz, **kwargs),
                                                  z, self.ae_frames_per_call)
        if reshape_back:
            res = res.reshape(b, t, *res.shape[1:]).transpose(1, 2)
        return res

    @torch.no_grad()
    def encode_first_stage(self, x):  # ddpm3d.py:611-644: per-frame posterior SAMPLE times scale_factor
        reshape_back = x.dim() == 5
        if reshape_back:
            b, _, t, _, _ = x.shape
            x = x.transpose(1, 2).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])
        if not self.perframe_ae:
            res = self.scale_factor * self.first_stage_model.encode(x).sample()
        else:
            res = self.first_stage_model.perframe(lambda xx: self.scale_factor * self.first_stage_model.encode(xx).sample(),
                                                  x, self.ae_frames_per_call, latent=False)
        if reshape_back:
            res = res.reshape(b, t, *res.shape[1:]).transpose(1, 2)
        return res.detach()

    @torch.no_grad()
    def decode_first_stage(self, z, **kwargs):
        return self.decode_core(z, **kwargs)

    def differentiable_decode_first_stage(self, z, **kwargs):
        return self.decode_core(z, **kwargs)
