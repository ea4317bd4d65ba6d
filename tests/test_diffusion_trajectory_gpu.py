"""`north_star` names *denoised latents*: a MULTI-STEP parity bar for the fp16 HIP path (`-m gpu`; round-3 verdict, missing #2).

tests/golden/make_golden_trajectory.py drove the REFERENCE's `DDIMSampler.p_sample_ddim` (ddim.py:208-280) over ten DDIM steps
(CFG 7.5 -- which amplifies every rounding of the two U-Net evaluations -- `rescale_noise_cfg` 0.7, v-parameterisation, dynamic
rescale, eta = 1 with the ten noise draws stored) around the reference's U-Net, once in float64 (the golden trajectory) and once the
way the reference runs it (fp32 weights under fp16 autocast), and stored the latter's error after every step.  Here the same x_T,
conditioning and noise go through THIS package's sampler (`k_ddim_stats` / `k_ddim_apply`) around the fp16 token-major U-Net on
the HIP kernels (MFMA convolutions, GEMMs, flash attention, fused norms), and after every step

        err(HIP path vs float64 trajectory)  <=  K x err(reference under fp16 autocast vs float64 trajectory),     K = 1.5

for x_{t-1} and pred_x0, in the max norm (relative to the step's largest float64 entry) and as an RMS ratio, with an absolute floor
for the first steps where both errors are ~1e-4 and their ratio is noise.  Both forms of the CFG pair the sampler can take (two
sequential calls: ddim.py:222-223; one batch-2 call: samplers.BATCH_CFG_MAX_PIXELS) are held to the bar.
"""
import os

import numpy as np
import pytest
import torch

from fill_by_name import fill_by_name

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TR = np.load(os.path.join(HERE, "golden", "trajectory_fp64.npz"), allow_pickle=False)
DEV = "cuda:0"
K, FLOOR = 1.5, 1e-3
STEPS, CFG, RESCALE, ETA = 10, 7.5, 0.7, 1.0


def _model():
    from lvdm_amd.model import LatentDiffusion
    from test_diffusion_goldens_gpu import UNET64
    tiny_vae = dict(double_z=True, z_channels=4, resolution=16, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2], num_res_blocks=1,
                    attn_resolutions=[], dropout=0.0)
    ld = LatentDiffusion(UNET64, tiny_vae).eval()
    fill_by_name(ld.model.diffusion_model)            # the same name-derived weights the golden's reference U-Net carries
    ld = ld.to(DEV)
    ld.model.diffusion_model.half().to_token_major()
    ld.requires_grad_(False)
    am = ld.apply_model
    ld.apply_model = lambda x, t, c, **kw: am(x.half(), t, {k: [v.half() for v in vs] for k, vs in c.items()}, **kw)
    return ld


def _err(a, ref):
    a, ref = a.detach().double().cpu(), torch.as_tensor(ref).double()
    return float((a - ref).abs().max() / ref.abs().max()), float(((a - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())


@pytest.mark.parametrize("batch_cfg", [False, True])
def test_ten_step_ddim_trajectory_stays_within_the_references_own_fp16_error(batch_cfg):
    from lvdm_amd.samplers import DDIMSampler
    ld = _model()
    t = lambda k: torch.tensor(TR[k], device=DEV)
    cond = {"c_crossattn": [t("traj_ctx_c")], "c_concat": [t("traj_concat")]}
    uc = {"c_crossattn": [t("traj_ctx_uc")], "c_concat": [t("traj_concat")]}
    s = DDIMSampler(ld)
    s.batch_cfg = batch_cfg
    s.make_schedule(STEPS, "uniform_trailing", ETA)
    assert [int(v) for v in np.flip(s.ddim_timesteps)] == TR["traj_steps"].tolist()
    x = t("traj_xT")
    noise = t("traj_noise")
    fs = torch.tensor([10], device=DEV)
    rows, worst = [], 0.0
    with torch.no_grad():
        for i, step in enumerate(np.flip(s.ddim_timesteps)):
            index = STEPS - i - 1
            ts = torch.full((1,), int(step), device=DEV, dtype=torch.long)
            x, p0 = s.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=CFG, unconditional_conditioning=uc,
                                    guidance_rescale=RESCALE, fs=fs, noise=noise[i])
            assert x.dtype == torch.float32 and torch.isfinite(x).all()
            (ex, rx), (ep, rp) = _err(x, TR["traj_x64"][i]), _err(p0, TR["traj_p064"][i])
            rows.append((i, ex, float(TR["traj_e16_x"][i]), rx, float(TR["traj_r16_x"][i]), ep, float(TR["traj_e16_p0"][i]),
                         rp, float(TR["traj_r16_p0"][i])))
    print(f"\n10-step DDIM trajectory, CFG pair as {'one batch-2 call' if batch_cfg else 'two calls'}: HIP fp16 / reference fp16-autocast, vs float64")
    for i, ex, e16x, rx, r16x, ep, e16p, rp, r16p in rows:
        print(f"  step {i}: x max {ex:.2e} / {e16x:.2e}  rms {rx:.2e} / {r16x:.2e}   pred_x0 max {ep:.2e} / {e16p:.2e}  rms {rp:.2e} / {r16p:.2e}")
        for got, ref in ((ex, e16x), (rx, r16x), (ep, e16p), (rp, r16p)):
            worst = max(worst, got / max(ref, FLOOR / K))
            assert got <= max(K * ref, FLOOR), (i, got, ref)
    print(f"  worst ratio {worst:.2f} (bar {K})")
