"""Soak of the guided DDIM step (miniature U-Net + VAE on the fp16 HIP path): 400 guided steps and 400 plain steps interleaved; the torch
allocator's `allocated` figure and the host RSS must be flat between the first and the last third (GradCells, norm states, packed-weight and
decode-group caches, the zero arena and the autograd graphs of the guided step must all be released)."""
import os, sys, resource, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch
from fill_by_name import fill_by_name
from test_ddim_parallel_gloo import HL, SMALL_UNET, SMALL_VAE, T, WL
from lvdm_amd.guidance import LossGuidance
from lvdm_amd.model import LatentDiffusion
from lvdm_amd.samplers import DDIMSampler, DDIMSamplerGuidance
DEV = "cuda:0"
ld = LatentDiffusion(SMALL_UNET, SMALL_VAE).eval()
fill_by_name(ld.model, std=0.02); fill_by_name(ld.first_stage_model, std=0.02)
ld = ld.to(DEV)
ld.model.diffusion_model.half().to_token_major(); ld.first_stage_model.half().to_token_major(); ld.requires_grad_(False)
am, dc = ld.apply_model, ld.decode_core
ld.apply_model = lambda x, t, c, **kw: am(x.half(), t, {k: [v.half() for v in vs] for k, vs in c.items()}, **kw)
ld.decode_core = lambda z, **kw: dc(z.half(), **kw)
g = torch.Generator().manual_seed(5)
mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
cond = {"c_crossattn": [mk(1, 93, 64)], "c_concat": [mk(1, 4, T, HL, WL) * 0.2]}
uc = {"c_crossattn": [mk(1, 93, 64)], "c_concat": cond["c_concat"]}
fs = torch.tensor([10], device=DEV)
sg, sp = DDIMSamplerGuidance(ld), DDIMSampler(ld)
sg.make_schedule(50, "uniform_trailing", 1.0); sp.make_schedule(50, "uniform_trailing", 1.0)
lg = LossGuidance(ddim_steps=50, recur_steps=1, device=DEV)
lg.set_hw(2 * HL, 2 * WL)
lg.set_guidance_images(torch.rand(T, 3, 2 * HL, 2 * WL, generator=g).to(DEV))
lg.set_guidance_masks((torch.rand(T, 1, 2 * HL, 2 * WL, generator=g) > 0.3).float().to(DEV))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
x = mk(1, 4, T, HL, WL)
marks = []
t0 = time.time()
for it in range(N):
    idx = 49 - it % 50
    tt = torch.full((1,), int(sg.ddim_timesteps[idx]), dtype=torch.long, device=DEV)
    xg, _ = sg.p_sample_ddim(x, cond, tt, index=idx, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, guidance_rescale=0.7, fs=fs, loss_guidance_fn=lg)
    with torch.no_grad():
        xp, _ = sp.p_sample_ddim(x, cond, tt, index=idx, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, guidance_rescale=0.7, fs=fs)
    assert torch.isfinite(xg).all() and torch.isfinite(xp).all()
    if it % (N // 6) == N // 6 - 1:
        torch.cuda.synchronize()
        marks.append((it, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3))
for m in marks: print("step %5d: device allocated %.1f MB, reserved %.1f MB, host max RSS %.1f MB" % m)
print(f"{N} guided + {N} plain steps in {time.time() - t0:.1f} s")
assert marks[-1][1] <= marks[1][1] * 1.02 + 1 and marks[-1][3] <= marks[1][3] * 1.02 + 16, "memory grows"
print("guided soak ok")
