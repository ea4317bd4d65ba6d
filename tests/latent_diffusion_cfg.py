"""The miniature LatentDiffusion configuration that tests/golden/make_golden_latent_diffusion.py builds the REFERENCE classes from and
tests/test_lvdm_dropin.py builds the drop-in from (the shipped yaml's keyword set, configs/inference_pvd_1024.yaml:5-110, at small widths)."""
UNET = dict(target="lvdm.modules.networks.openaimodel3d.UNetModel",
            params=dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1, channel_mult=[1, 2], dropout=0.1,
                        num_head_channels=32, transformer_depth=1, context_dim=48, use_linear=True, use_checkpoint=False, temporal_conv=True,
                        temporal_attention=True, temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False, temporal_length=16,
                        addition_attention=True, image_cross_attention=True, default_fs=10, fs_condition=True))
VAE = dict(target="lvdm.models.autoencoder.AutoencoderKL",
           params=dict(embed_dim=4, monitor="val/rec_loss",
                       ddconfig=dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2], num_res_blocks=1,
                                     attn_resolutions=[], dropout=0.0),
                       lossconfig=dict(target="torch.nn.Identity")))
LD_KW = dict(timesteps=1000, linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", parameterization="v", rescale_betas_zero_snr=True,
             use_dynamic_rescale=True, base_scale=0.7, turning_step=400, uncond_type="empty_seq", scale_factor=0.18215, perframe_ae=True, use_ema=False,
             first_stage_key="video", cond_stage_key="caption", channels=4, image_size=[8, 8], log_every_t=200)
