"""The diffusion side is a drop-in under the REFERENCE'S OWN module paths: with `guidedvd-3dgs_amd/` on sys.path every import
string and every yaml `target:` string the guidedvd drivers use resolves to the MI355X-native implementation, and the model
object `viewcrafter.py:315-335` builds from `configs/inference_pvd_1024.yaml` has the attribute / state-dict surface the
callers and the checkpoint need.  (Strings below are transcribed from the reference; the yaml text itself is not copied.)

    utils_vc/diffusion_utils.py:8-10      from lvdm.models.samplers.{ddim, ddim_guidance, ddim_multiplecond} import ...
    configs/inference_pvd_1024.yaml:5     target: lvdm.models.ddpm3d.VIPLatentDiffusion
    :34 UNetModel  :67 AutoencoderKL  :90 FrozenOpenCLIPEmbedder  :96 FrozenOpenCLIPImageEmbedderV2  :101 Resampler
    lvdm/models/ddpm3d.py:24-35           lvdm.ema / distributions / utils_diffusion / basics / common names
"""
import importlib
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

IMPORTS = [
    ("lvdm.models.samplers.ddim", "DDIMSampler"),
    ("lvdm.models.samplers.ddim_guidance", "DDIMSamplerGuidance"),
    ("lvdm.models.samplers.ddim_multiplecond", "DDIMSampler"),
    ("lvdm.models.ddpm3d", "VIPLatentDiffusion"),
    ("lvdm.models.ddpm3d", "LatentVisualDiffusion"),
    ("lvdm.models.ddpm3d", "LatentDiffusion"),
    ("lvdm.models.ddpm3d", "DDPM"),
    ("lvdm.models.ddpm3d", "DiffusionWrapper"),
    ("lvdm.models.autoencoder", "AutoencoderKL"),
    ("lvdm.modules.networks.openaimodel3d", "UNetModel"),
    ("lvdm.modules.networks.ae_modules", "Encoder"),
    ("lvdm.modules.networks.ae_modules", "Decoder"),
    ("lvdm.modules.attention", "SpatialTransformer"),
    ("lvdm.modules.attention", "TemporalTransformer"),
    ("lvdm.modules.encoders.resampler", "Resampler"),
    ("lvdm.modules.encoders.condition", "FrozenOpenCLIPEmbedder"),
    ("lvdm.modules.encoders.condition", "FrozenOpenCLIPImageEmbedderV2"),
    ("lvdm.distributions", "DiagonalGaussianDistribution"),
    ("lvdm.models.utils_diffusion", "make_beta_schedule"),
    ("lvdm.models.utils_diffusion", "rescale_zero_terminal_snr"),
    ("lvdm.models.utils_diffusion", "make_ddim_timesteps"),
    ("lvdm.models.utils_diffusion", "make_ddim_sampling_parameters"),
    ("lvdm.models.utils_diffusion", "rescale_noise_cfg"),
    ("lvdm.models.utils_diffusion", "timestep_embedding"),
    ("lvdm.basics", "disabled_train"),
    ("lvdm.basics", "zero_module"),
    ("lvdm.basics", "normalization"),
    ("lvdm.common", "extract_into_tensor"),
    ("lvdm.common", "noise_like"),
    ("lvdm.common", "default"),
    ("lvdm.common", "exists"),
]


@pytest.mark.parametrize("module,name", IMPORTS)
def test_reference_import_strings_resolve_to_the_native_package(module, name):
    obj = getattr(importlib.import_module(module), name)
    src = getattr(obj, "__module__", "")
    assert src.startswith(("lvdm_amd", "lvdm.")), (module, name, src)
    pkg = importlib.import_module("lvdm")
    assert os.path.realpath(pkg.__file__).startswith(os.path.realpath(os.path.join(HERE, "..", "guidedvd-3dgs_amd")))


from lvdm_amd.model import viewcrafter_yaml_node as _yaml_model_node  # noqa: E402  (the yaml's `model:` node as a plain mapping)


# persistent buffers of DDPM.register_schedule + LatentDiffusion (ddpm3d.py:145-171,527): part of the checkpoint's state dict
SCHEDULE_BUFFERS = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
                    "scale_arr"]


def test_full_size_model_builds_from_the_yaml_node_and_has_the_checkpoint_key_surface():
    """The whole 2.6 B-parameter object (1.44 B U-Net, VAE, both ViT-H/14 towers, Resampler) on the meta device: every
    `target:` resolves, the constructor keyword set of the yaml is accepted, and the state dict has the reference's layout."""
    from lvdm_amd.model import instantiate_from_config
    with torch.device("meta"):
        model = instantiate_from_config(_yaml_model_node())
    sd = model.state_dict()
    tops = sorted({k.split(".")[0] for k in sd})
    assert tops == sorted(SCHEDULE_BUFFERS + ["model", "first_stage_model", "cond_stage_model", "embedder", "image_proj_model"]), tops
    n = lambda prefix: sum(v.numel() for k, v in sd.items() if k.startswith(prefix))
    assert abs(n("model.diffusion_model.") / 1e6 - 1438.855) < 0.01            # SURVEY 8c probe of the reference U-Net
    assert n("first_stage_model.") == 83653863                                  # SD-VAE kl-f8 parameter count
    assert 353e6 < n("cond_stage_model.model.") < 355e6 and 631e6 < n("embedder.model.visual.") < 633e6   # ViT-H/14 text / vision
    for k in ("model.diffusion_model.input_blocks.1.0.temopral_conv.conv4.3.weight",
              "model.diffusion_model.init_attn.0.transformer_blocks.0.attn1.to_q.weight",
              "model.diffusion_model.output_blocks.11.1.transformer_blocks.0.attn2.to_k_ip.weight",
              "model.diffusion_model.fps_embedding.2.bias",
              "first_stage_model.encoder.down.3.block.1.norm2.weight", "first_stage_model.decoder.mid.attn_1.proj_out.bias",
              "first_stage_model.quant_conv.weight", "first_stage_model.post_quant_conv.bias",
              "cond_stage_model.model.transformer.resblocks.23.attn.in_proj_weight", "cond_stage_model.model.token_embedding.weight",
              "cond_stage_model.model.ln_final.bias", "cond_stage_model.model.text_projection", "cond_stage_model.model.logit_scale",
              "embedder.model.visual.transformer.resblocks.31.mlp.c_fc.weight", "embedder.model.visual.class_embedding",
              "embedder.model.visual.proj", "embedder.model.visual.conv1.weight", "embedder.model.positional_embedding",
              "image_proj_model.latents", "image_proj_model.layers.3.0.to_kv.weight", "image_proj_model.proj_out.weight"):
        assert k in sd, k
    # the attribute surface viewcrafter.py:315-335 and diffusion_utils.py:118-223 touch
    assert model.model.conditioning_key == "hybrid" and model.model.diffusion_model.out_channels == 4
    assert model.uncond_type == "empty_seq" and model.perframe_ae is True and model.scale_factor == 0.18215
    assert model.num_timesteps == 1000 and model.parameterization == "v" and model.use_dynamic_rescale
    assert model.cond_stage_model.device == "cuda"    # viewcrafter.py:323 assigns it
    for name in ("apply_model", "get_learned_conditioning", "encode_first_stage", "decode_first_stage",
                 "differentiable_decode_first_stage", "embedder", "image_proj_model", "predict_start_from_z_and_v",
                 "predict_eps_from_z_and_v", "q_sample"):
        assert hasattr(model, name), name
    assert not any(p.requires_grad for m in (model.first_stage_model, model.cond_stage_model, model.embedder, model.image_proj_model)
                   for p in m.parameters())


def test_sub_model_state_dict_keys_equal_the_reference_modules():
    """Key sets recorded from the reference's own modules by the golden generators (tiny configurations, same code path)."""
    from lvdm.models.autoencoder import AutoencoderKL
    from lvdm.modules.encoders.resampler import Resampler
    from lvdm.modules.networks.openaimodel3d import UNetModel
    G = np.load(os.path.join(HERE, "golden", "diffusion_ref.npz"), allow_pickle=False)
    E = np.load(os.path.join(HERE, "golden", "vae_encoder_ref.npz"), allow_pickle=False)
    R = np.load(os.path.join(HERE, "golden", "resampler_ref.npz"), allow_pickle=False)
    unet = UNetModel(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1,
                     channel_mult=[1, 2], dropout=0.1, num_head_channels=32, transformer_depth=1, context_dim=48, use_linear=True,
                     use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
                     use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
                     image_cross_attention=True, default_fs=10, fs_condition=True)
    assert sorted(unet.state_dict()) == list(G["unet_keys"])
    ae = AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2],
                                     num_res_blocks=1, attn_resolutions=[], dropout=0.0), lossconfig={"target": "torch.nn.Identity"},
                       embed_dim=4, monitor="val/rec_loss")
    assert sorted(k[len("decoder."):] for k in ae.state_dict() if k.startswith("decoder.")) == list(G["dec_keys"])
    ae2 = AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4],
                                      num_res_blocks=2, attn_resolutions=[], dropout=0.0), lossconfig=None, embed_dim=4)
    assert sorted(k[len("encoder."):] for k in ae2.state_dict() if k.startswith("encoder.")) == list(E["keys"])
    assert {k.split(".")[0] for k in ae2.state_dict()} == {"encoder", "decoder", "quant_conv", "post_quant_conv"}
    rs = Resampler(dim=128, depth=2, dim_head=64, heads=2, num_queries=4, embedding_dim=96, output_dim=80, ff_mult=4, video_length=3)
    assert sorted(rs.state_dict()) == list(R["keys"])


def test_small_model_runs_the_reference_call_sequence_end_to_end():
    """A miniature of the yaml (same class tree, tiny widths) driven exactly like viewcrafter.py:315-335 + diffusion_utils.py:
    instantiate -> load_state_dict(strict) of its own state dict -> eval -> image_guided_synthesis with the lvdm samplers."""
    from lvdm_amd import ops, pipeline
    from lvdm_amd.model import instantiate_from_config
    clip = dict(embed_dim=32, text_width=32, text_layers=2, text_heads=2, vocab=64, ctx=77,
                vision_cfg=dict(width=48, layers=2, heads=2, patch=8, image=32))
    node = _yaml_model_node(clip, unet_over=dict(model_channels=32, num_head_channels=32, context_dim=32, channel_mult=[1, 2],
                                                 attention_resolutions=[2, 1], num_res_blocks=1, use_checkpoint=False),
                            vae_over=dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, resolution=32))
    node["params"]["image_proj_stage_config"]["params"].update(dim=32, depth=1, dim_head=16, heads=2, num_queries=2, embedding_dim=48,
                                                              output_dim=32, video_length=2)
    torch.manual_seed(0)
    model = instantiate_from_config(node)
    from fill_by_name import fill_by_name
    fill_by_name(model.model, std=0.05)          # a freshly initialised U-Net is degenerate (zero-init output convolution)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    model.eval()
    assert str(model.device) == "cpu"
    T, H, W = 2, 16, 16
    videos = torch.rand(1, 3, T, H, W) * 2 - 1
    tokens = torch.randint(0, 64, (1, 77))
    model.get_learned_conditioning = lambda prompts, _f=model.get_learned_conditioning: _f(tokens.expand(len(prompts), -1))  # no BPE vocabulary here
    ops.use_reference_math(True)
    try:
        out = pipeline.image_guided_synthesis(model, ["Rotating view of a scene"], videos, [1, 4, T, H // 2, W // 2], 1, 2, 1.0, 7.5, None, 10,
                                              True, False, "uniform_trailing", 0.7, [0], None, True)
    finally:
        ops.use_reference_math(False)
    assert out.shape == (1, 1, 3, T, H, W) and torch.isfinite(out).all()


@pytest.mark.gpu
def test_dropin_plumbing_on_the_device_agrees_with_the_packages_own_cpu_form():
    """PLUMBING, not parity: a self-comparison of this package's two execution forms (the parity of the fp16 kernels against the
    REFERENCE is tests/test_diffusion_parity_bars_gpu.py, test_diffusion_goldens_gpu.py and, over ten DDIM steps,
    test_diffusion_trajectory_gpu.py).  What it pins: the drop-in class tree converts itself for the device (`_prepare_native`),
    the guided sampler, the grouped decode and the batch-2 CFG pair are wired as on the CPU, and nothing is lost on the way.
    The same miniature class tree, built once, driven through `image_guided_synthesis` with the GUIDED lvdm sampler twice:
    on the CPU with the reference formulation of every operator (fp32) and on the GPU through the product path
    (`_prepare_native`: fp16 token-major U-Net and VAE, MFMA convolutions with fused norms, flash attention at the 64-wide heads,
    grouped VAE decode + backward inside the step).  Same noise (drawn on the CPU generator), so the two videos agree to the
    accumulated fp16 rounding of 3 guided steps."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from lvdm_amd import ops, pipeline, samplers
    from lvdm_amd.guidance import LossGuidance
    from lvdm_amd.model import instantiate_from_config
    from fill_by_name import fill_by_name
    clip = dict(embed_dim=32, text_width=32, text_layers=2, text_heads=2, vocab=64, ctx=77,
                vision_cfg=dict(width=48, layers=2, heads=2, patch=8, image=32))
    node = _yaml_model_node(clip, unet_over=dict(model_channels=64, num_head_channels=64, context_dim=32, channel_mult=[1, 2],
                                                 attention_resolutions=[2, 1], num_res_blocks=1, use_checkpoint=False),
                            vae_over=dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, resolution=32))
    node["params"]["image_proj_stage_config"]["params"].update(dim=32, depth=1, dim_head=16, heads=2, num_queries=2, embedding_dim=48,
                                                              output_dim=32, video_length=3)
    torch.manual_seed(0)
    model = instantiate_from_config(node)
    fill_by_name(model.model, std=0.05)
    fill_by_name(model.first_stage_model, std=0.05)
    model.eval()
    T, H, W = 3, 32, 48
    g = torch.Generator().manual_seed(11)
    videos = torch.rand(1, 3, T, H, W, generator=g) * 2 - 1
    tokens = torch.randint(0, 64, (1, 77), generator=g)
    glc = model.get_learned_conditioning
    model.get_learned_conditioning = lambda prompts: glc(tokens.to(model.device).expand(len(prompts), -1))

    def run(dev):
        model.to(dev)
        lg = LossGuidance(ddim_steps=3, recur_steps=1)
        lg.set_hw(H, W)
        lg.set_guidance_images((videos[0].permute(1, 0, 2, 3) * 0.5).to(dev))      # [T, 3, H, W]
        lg.set_guidance_masks(torch.ones(T, 1, H, W, device=dev))
        torch.manual_seed(5)
        samplers.DDIMSampler.noise_device = "cpu"   # the draws of both runs come from the CPU generator
        try:
            return pipeline.image_guided_synthesis(model, ["a room"], videos.to(dev), [1, 4, T, H // 2, W // 2], 1, 3, 1.0, 7.5, None,
                                                   10, True, False, "uniform_trailing", 0.7, [0], lg, False).float().cpu()
        finally:
            samplers.DDIMSampler.noise_device = "device"

    ops.use_reference_math(True)
    try:
        ref = run("cpu")
    finally:
        ops.use_reference_math(False)
    batches = []
    apply = model.apply_model
    model.apply_model = lambda x, *a_, **k_: (batches.append(x.shape[0]), apply(x, *a_, **k_))[1]
    out = run("cuda:0")
    assert next(model.model.diffusion_model.parameters()).dtype == torch.float16      # converted once by _prepare_native
    assert out.shape == ref.shape == (1, 1, 3, T, H, W) and torch.isfinite(out).all()
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < 5e-2, err
    # the CFG pair went through the U-Net as ONE batch-2 call per step (samplers.BATCH_CFG_MAX_PIXELS: small latents on a device,
    # this package's U-Net); forced back to the reference's two sequential calls the video is the same up to fp16 rounding
    assert batches == [2, 2, 2], batches
    del batches[:]
    samplers.DDIMSamplerGuidance.batch_cfg = False
    try:
        out_seq = run("cuda:0")
    finally:
        del samplers.DDIMSamplerGuidance.batch_cfg
        model.apply_model = apply
    assert batches == [1] * 6, batches
    # (one evaluation agrees to 1.3e-3 forward / 1.6e-3 input gradient -- tests/scripts/r3_pair_probe.py: rounding level, the batch-2
    #  launches tile differently -- and three guided steps at CFG 7.5 amplify that like they amplify the fp16-vs-fp32 difference above)
    assert float((out - out_seq).abs().max() / out_seq.abs().max()) < 5e-2


def test_latent_diffusion_matches_the_references_own_classes():
    """SURVEY rows B5 / B13 glue against lvdm/models/ddpm3d.py::LatentDiffusion + autoencoder.py::AutoencoderKL THEMSELVES
    (tests/golden/make_golden_latent_diffusion.py imports them with a two-name placeholder for pytorch_lightning; until round 6 the comparison was key
    counts and shapes).  The drop-in built from the same configuration has the SAME state-dict keys and shapes -- so name-derived weights land
    identically -- the same schedule buffers, and reproduces apply_model with hybrid conditioning, the v-parameterisation helpers, q_sample, decode
    (per frame and batched) with its input gradient, and encode (posterior sample, same generator order)."""
    from latent_diffusion_cfg import LD_KW, UNET, VAE
    from fill_by_name import fill_by_name
    from lvdm.models.ddpm3d import LatentDiffusion
    from lvdm_amd import ops
    R = np.load(os.path.join(HERE, "golden", "latent_diffusion_ref.npz"), allow_pickle=False)
    ld = LatentDiffusion(first_stage_config=VAE, cond_stage_config={"target": "torch.nn.Identity"}, unet_config=UNET, **LD_KW).eval()
    sd = ld.state_dict()
    assert sorted(sd) == list(R["keys"])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(R["shapes"])
    for k in sd:
        if not k.startswith(("model.", "first_stage_model.")):
            np.testing.assert_allclose(sd[k].numpy(), R["buf_" + k], rtol=1e-6, atol=1e-12, err_msg=k)
    fill_by_name(ld.model)
    fill_by_name(ld.first_stage_model)
    t = lambda k: torch.tensor(R[k])
    x, tt, fs = t("x"), torch.tensor([400]), torch.tensor([10])
    cond = {"c_crossattn": [t("c_text"), t("c_img")], "c_concat": [t("c_concat")]}
    ops.use_reference_math(True)
    try:
        with torch.no_grad():
            v = ld.apply_model(x, tt, cond, fs=fs)
            np.testing.assert_allclose(v.numpy(), R["v"], rtol=2e-4, atol=2e-5 * np.abs(R["v"]).max())
            vr = t("v")
            np.testing.assert_allclose(ld.predict_start_from_z_and_v(x, tt, vr).numpy(), R["x0_from_v"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(ld.predict_eps_from_z_and_v(x, tt, vr).numpy(), R["eps_from_v"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(ld.q_sample(x, tt, noise=t("q_noise")).numpy(), R["q_sample"], rtol=1e-5, atol=1e-6)
        z = t("z")
        for tag, pf in (("perframe", True), ("batched", False)):
            ld.perframe_ae = pf
            with torch.no_grad():
                img = ld.decode_first_stage(z)
            np.testing.assert_allclose(img.numpy(), R[f"decode_{tag}"], rtol=2e-4, atol=2e-5 * np.abs(R[f"decode_{tag}"]).max())
            zr = z.clone().requires_grad_(True)
            (gz,) = torch.autograd.grad(ld.differentiable_decode_first_stage(zr), zr, t("decode_gi"))
            np.testing.assert_allclose(gz.numpy(), R[f"decode_grad_{tag}"], rtol=5e-4, atol=5e-5 * np.abs(R[f"decode_grad_{tag}"]).max())
            torch.manual_seed(5)
            enc = ld.encode_first_stage(t("video"))
            np.testing.assert_allclose(enc.numpy(), R[f"encode_{tag}"], rtol=2e-4, atol=2e-5 * np.abs(R[f"encode_{tag}"]).max())
    finally:
        ops.use_reference_math(False)


def test_full_size_keys_and_shapes_equal_the_references_classes_at_the_shipped_yaml():
    """The drop-in's strict-checkpoint-load claim against lvdm/models/ddpm3d.py::VIPLatentDiffusion ITSELF, built in the build container from the
    reference's configs/inference_pvd_1024.yaml (tests/golden/make_golden_full_keys.py; the CLIP nodes swapped for Identity there -- open_clip is absent --
    so their keys stay with the count / name checks above): every key of the U-Net (1.44 B parameters), the KL-VAE and the Resampler with its shape,
    the schedule buffers' names AND values, the scalar attributes the callers read; and the yaml transcription in lvdm_amd.model.viewcrafter_yaml_node()
    hashes to the yaml's own `params` mapping."""
    import hashlib
    import json
    from lvdm_amd.model import instantiate_from_config
    from lvdm_amd.schedule import DiffusionSchedule
    R = np.load(os.path.join(HERE, "golden", "full_keys_ref.npz"), allow_pickle=False)
    node = _yaml_model_node()
    assert hashlib.sha256(json.dumps(node["params"], sort_keys=True, separators=(",", ":")).encode()).hexdigest() == str(R["yaml_params_sha256"])
    with torch.device("meta"):
        model = instantiate_from_config(node)
    sd = model.state_dict()
    ref = dict(zip(R["keys"].tolist(), R["shapes"].tolist()))
    pinned = ("model.", "first_stage_model.", "image_proj_model.")
    ours = {k: str(tuple(v.shape)) for k, v in sd.items() if k.startswith(pinned)}
    theirs = {k: s for k, s in ref.items() if k.startswith(pinned)}
    assert len(theirs) > 1700 and sorted(ours) == sorted(theirs)
    assert ours == theirs
    bufs = sorted(k for k in ref if k.split(".")[0] not in ("model", "first_stage_model", "image_proj_model", "cond_stage_model", "embedder"))
    assert bufs == sorted(k for k in sd if k.split(".")[0] not in ("model", "first_stage_model", "image_proj_model", "cond_stage_model", "embedder"))
    p = node["params"]
    sched = DiffusionSchedule(timesteps=p["timesteps"], linear_start=p["linear_start"], linear_end=p["linear_end"], rescale_betas_zero_snr=p["rescale_betas_zero_snr"],
                              parameterization=p["parameterization"], use_dynamic_rescale=p["use_dynamic_rescale"], base_scale=p["base_scale"], full_tables=True)
    for k in bufs:
        np.testing.assert_allclose(getattr(sched, k).numpy(), R["buf_" + k], rtol=1e-6, atol=1e-12, err_msg=k)
    sc = json.loads(str(R["scalars"]))
    assert (float(model.scale_factor), model.uncond_type, bool(model.perframe_ae), int(model.num_timesteps), model.parameterization,
            bool(model.use_dynamic_rescale), model.model.conditioning_key, list(model.image_size), int(model.channels), int(model.temporal_length)) == \
           (sc["scale_factor"], sc["uncond_type"], sc["perframe_ae"], sc["num_timesteps"], sc["parameterization"], sc["use_dynamic_rescale"],
            sc["conditioning_key"], sc["image_size"], sc["channels"], sc["temporal_length"])
