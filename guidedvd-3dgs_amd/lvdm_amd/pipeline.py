"""Entry of the diffusion hot path: what `ViewCrafterWrapper.run_video_diffusion` ->
`ViewCrafter.run_diffusion` -> `image_guided_synthesis` do around the DDIM loop, on top of lvdm_amd's samplers.

    utils/viewcrafter_wrapper.py:550-573      run_video_diffusion : guidance tensors -> loss fn, [-1,1] -> [0,1], NHWC -> NCHW
    third_party/ViewCrafter/viewcrafter.py:92-112   run_diffusion  : [T,H,W,3] in [0,1] -> [1,3,T,H,W] in [-1,1], autocast, clamp
    utils_vc/diffusion_utils.py:111-223       get_latent_z / image_guided_synthesis : cond / uncond dicts, sampler.sample, decode

The modules that produce the conditioning (CLIP image embedder, Resampler `image_proj_model`, text encoder, VAE
*encoder*) run once per video, outside the loop, and are SURVEY "next" row N2: they are reached through the same
duck-typed attributes of `model` as in the reference (`embedder`, `image_proj_model`, `get_learned_conditioning`,
`encode_first_stage`, `decode_first_stage`, `uncond_type`, `model.conditioning_key`), so the reference's own modules
plug in unchanged.
"""
import torch

from .samplers import DDIMSampler, DDIMSamplerGuidance, DDIMSamplerMultiCond


def get_latent_z(model, videos):
    """[b, c, t, h, w] video -> per-frame VAE latents [b, 4, t, h/8, w/8] (diffusion_utils.py:111-116)."""
    b, c, t, h, w = videos.shape
    x = videos.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    z = model.encode_first_stage(x)
    return z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)


def image_guided_synthesis(model, prompts, videos, noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.,
                           unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, timestep_spacing='uniform', guidance_rescale=0.0,
                           condition_index=None, loss_guidance_fn=None, no_guidance=False, **kwargs):
    """Same signature and return ([batch, n_samples, c, t, h, w]) as diffusion_utils.py:118-223."""
    if multiple_cond_cfg:     # diffusion_utils.py:123-125: the three-way sampler replaces BOTH the plain and the guided one
        sampler = DDIMSamplerMultiCond(model)
    else:
        sampler = DDIMSampler(model) if no_guidance else DDIMSamplerGuidance(model)
    batch_size = noise_shape[0]
    fs = torch.tensor([fs] * batch_size, dtype=torch.long, device=model.device)
    if not text_input:
        prompts = [""] * batch_size
    assert condition_index is not None, "Error: condition index is None!"
    img = videos[:, :, condition_index[0]]  # b c h w

    with torch.no_grad():
        img_emb = model.image_proj_model(model.embedder(img))       # b l c
        cond_emb = model.get_learned_conditioning(prompts)
    cond = {"c_crossattn": [torch.cat([cond_emb, img_emb], dim=1)]}
    hybrid = model.model.conditioning_key == 'hybrid'
    if hybrid:
        with torch.no_grad():
            img_cat_cond = get_latent_z(model, videos)              # b c t h w : every rendered frame conditions
        cond["c_concat"] = [img_cat_cond]

    uc = None
    if unconditional_guidance_scale != 1.0:
        with torch.no_grad():
            if model.uncond_type == "empty_seq":
                uc_emb = model.get_learned_conditioning(batch_size * [""])
            elif model.uncond_type == "zero_embed":
                uc_emb = torch.zeros_like(cond_emb)
            else:
                raise ValueError(f"unknown uncond_type {model.uncond_type!r}")
            uc_img_emb = model.image_proj_model(model.embedder(torch.zeros_like(img)))
        uc = {"c_crossattn": [torch.cat([uc_emb, uc_img_emb], dim=1)]}
        if hybrid:
            uc["c_concat"] = [img_cat_cond]
    if multiple_cond_cfg and cfg_img != 1.0:    # one more unconditional: image = yes, text = "" (diffusion_utils.py:176-183)
        if uc is None:
            raise ValueError("multiple_cond_cfg needs unconditional_guidance_scale != 1.0 (the reference reads uc_emb here)")
        uc_2 = {"c_crossattn": [torch.cat([uc_emb, img_emb], dim=1)]}
        if hybrid:
            uc_2["c_concat"] = [img_cat_cond]
        kwargs.update({"unconditional_conditioning_img_nonetext": uc_2})
    else:
        kwargs.update({"unconditional_conditioning_img_nonetext": None})
    if loss_guidance_fn is not None:
        kwargs.update({"loss_guidance_fn": loss_guidance_fn})

    variants = []
    for _ in range(n_samples):
        samples, _ = sampler.sample(S=ddim_steps, conditioning=cond, batch_size=batch_size, shape=noise_shape[1:],
                                    verbose=False, unconditional_guidance_scale=unconditional_guidance_scale,
                                    unconditional_conditioning=uc, eta=ddim_eta, cfg_img=cfg_img, mask=None, x0=None,
                                    fs=fs, timestep_spacing=timestep_spacing, guidance_rescale=guidance_rescale, **kwargs)
        variants.append(model.decode_first_stage(samples))          # latent -> pixel space
    return torch.stack(variants).permute(1, 0, 2, 3, 4, 5)           # batch, variants, c, t, h, w


def run_diffusion(model, renderings, noise_shape, opts, loss_guidance_fn=None, no_guidance=False, autocast=True):
    """viewcrafter.py:92-112.  renderings [T, H, W, 3] in [0, 1] (the point-cloud renders that condition the video)
    -> [T, H, W, 3] in [-1, 1].  `opts`: the reference's option namespace (prompt, n_samples, ddim_steps, ddim_eta,
    unconditional_guidance_scale, cfg_img, frame_stride, text_input, multiple_cond_cfg, timestep_spacing,
    guidance_rescale)."""
    videos = (renderings * 2. - 1.).permute(3, 0, 1, 2).unsqueeze(0).to(model.device)   # [1, 3, T, H, W] in [-1, 1]
    on_gpu = torch.device(model.device).type == "cuda"
    with torch.autocast("cuda", enabled=bool(autocast and on_gpu)):
        batch = image_guided_synthesis(model, [opts.prompt], videos, noise_shape, opts.n_samples, opts.ddim_steps,
                                       opts.ddim_eta, opts.unconditional_guidance_scale, opts.cfg_img, opts.frame_stride,
                                       opts.text_input, opts.multiple_cond_cfg, opts.timestep_spacing,
                                       opts.guidance_rescale, [0], loss_guidance_fn, no_guidance)
    return torch.clamp(batch[0][0].permute(1, 2, 3, 0), -1., 1.)


def run_video_diffusion(model, point_cloud_render_results, noise_shape, opts, loss_guidance_fn=None, guidance_images=None,
                        guidance_masks=None, guidance_depths=None, no_guidance=False):
    """viewcrafter_wrapper.py:550-573: hand the 3DGS renders to the guidance loss, run the diffusion, return
    [T, 3, H, W] in [0, 1] (the per-run mp4 dump of :567 is left to the caller)."""
    if loss_guidance_fn is not None:
        loss_guidance_fn.set_guidance_images(guidance_images)
        if guidance_masks is not None:
            loss_guidance_fn.set_guidance_masks(guidance_masks)
        if guidance_depths is not None:
            loss_guidance_fn.set_guidance_depths(guidance_depths)
    res = run_diffusion(model, point_cloud_render_results, noise_shape, opts, loss_guidance_fn, no_guidance)
    return ((res + 1.0) / 2.0).permute(0, 3, 1, 2)
