"""Stand-in LatentDiffusion for the pipeline (row B1) goldens and tests: every attribute `image_guided_synthesis`
and the samplers read, tiny deterministic modules with weights derived from parameter names.  Test infrastructure."""
import numpy as np
import torch
import torch.nn.functional as F

from fill_by_name import fill_by_name


class PipeDuck(torch.nn.Module):
    uncond_type = "empty_seq"

    def __init__(self, sched):
        super().__init__()
        object.__setattr__(self, "_sched", sched)
        self.model = torch.nn.Conv3d(8, 4, 1)              # "U-Net" over cat([x, c_concat])
        self.model.conditioning_key = "hybrid"
        self.first_stage_model = torch.nn.Conv2d(4, 3, 1)  # "decoder"
        self.first_stage_enc = torch.nn.Conv2d(3, 4, 1)    # "encoder"
        self.txt = torch.nn.Embedding(4, 8)
        fill_by_name(self, std=0.5)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(object.__getattribute__(self, "_sched"), name)

    @property
    def device(self):
        return self.txt.weight.device

    # -- conditioning side (N2 modules in the real model) --
    def embedder(self, img):                               # [b,3,h,w] -> [b,4,8]
        m = img.mean((2, 3))
        return torch.stack([m.repeat(1, 3)[:, :8] * (k + 1) for k in range(4)], 1)

    def image_proj_model(self, e):
        return torch.tanh(e) * 0.5 + 0.1

    def get_learned_conditioning(self, prompts):
        ids = torch.tensor([[len(p) % 4, (len(p) + 1) % 4, 3] for p in prompts], device=self.device)
        return self.txt(ids)

    def encode_first_stage(self, x):                       # [(b t),3,H,W] -> [(b t),4,H/2,W/2]
        return F.avg_pool2d(self.first_stage_enc(x), 2) * 0.18215

    def decode_first_stage(self, z):                       # [b,4,t,h,w] -> [b,3,t,2h,2w]
        b, c, t, h, w = z.shape
        y = torch.tanh(0.25 * self.first_stage_model(z.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)))
        y = F.interpolate(y, scale_factor=2.0, mode="nearest")
        return y.reshape(b, t, 3, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)

    def differentiable_decode_first_stage(self, z):
        return self.decode_first_stage(z)

    # -- what the samplers call --
    def apply_model(self, x, t, c, **kw):
        xc = torch.cat([x] + list(c["c_concat"]), 1)
        return self.model(xc) * (1 + c["c_crossattn"][0].mean()) * (1 + 1e-3 * t.float().mean() / 1000)


def inputs(seed=9, T=4, H=12, W=16):
    g = torch.Generator().manual_seed(seed)
    renderings = torch.rand(T, H, W, 3, generator=g)                      # [T,H,W,3] in [0,1]
    guide = torch.rand(T, 3, H, W, generator=g)
    masks = (torch.rand(T, 1, H, W, generator=g) > 0.3).float()
    return renderings, guide, masks, [1, 4, T, H // 2, W // 2]


class Opts:
    prompt = "Rotating view of a scene"
    n_samples, ddim_steps, ddim_eta = 1, 4, 1.0
    unconditional_guidance_scale, cfg_img, frame_stride = 7.5, None, 10
    text_input, multiple_cond_cfg, timestep_spacing, guidance_rescale = True, False, "uniform_trailing", 0.7
