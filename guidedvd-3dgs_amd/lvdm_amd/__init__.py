"""MI355X-native rebuild of the ViewCrafter DDIM denoise loop.

    schedule   B4   beta / zero-terminal-SNR / DDIM tables, v-parameterisation helpers, timestep embedding
    samplers   B2/B3 DDIMSampler, DDIMSamplerGuidance (same .sample() API as lvdm.models.samplers.*)
    unet       B6-B11 UNetModel (state-dict compatible with the ViewCrafter checkpoint)
    vae        B13  KL-VAE decoder / encoder
    conv       the hand-written MFMA convolutions (3x3, upsample, temporal) behind unet / vae
    guidance   B12  LossGuidance
    model      B1/B5 LatentDiffusion-shaped wrapper the samplers duck-type against (apply_model, decode)
    ops        hot operators; HIP kernels (csrc/*.hip) on ROCm devices, no silent CPU fallback

Importing this package changes no process-wide state, and nothing in it configures a vendor library: since round 4 no hipBLASLt /
rocBLAS GEMM and no MIOpen convolution is left on the 16-bit product path of the U-Net and the VAE (tests/test_no_library_kernels_gpu.py
holds that), so the recorded TunableOp / MIOpen find-db choices of rounds 1-3 and `configure_tuning()` were removed in round 5.
"""


def allow_torch_fallback(allow=True):
    """`with lvdm_amd.allow_torch_fallback():` -- see ops.allow_torch_fallback.  Device tensors no hand-written kernel covers (fp32
    activations, mostly) RAISE by default instead of quietly running torch's library kernels; inside this context they run them with
    one RuntimeWarning per (op, reason)."""
    from .ops import allow_torch_fallback as _ctx
    return _ctx(allow)
