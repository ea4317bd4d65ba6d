"""Golden for the multiple-condition branch of SURVEY row B1: the REFERENCE's image_guided_synthesis (utils_vc/diffusion_utils.py:118-223) with
`multiple_cond_cfg=True, cfg_img=3.0` -- its DDIMSampler_multicond (:123-125), the third conditioning built at :176-183 -- on the stand-in model of
tests/pipeline_duck.py (CPU).  Build container only (needs /root/reference).  Output: tests/golden/pipeline_multicond_ref.npz (arrays only)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VC = "/root/reference/third_party/ViewCrafter"
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, VC)
# The reference FIRST, before guidedvd-3dgs_amd/ is importable at all: that directory also holds an `lvdm` -- the drop-in that shadows the
# reference's in production, and does so from ANY position of sys.path, because the reference's lvdm/ has no __init__.py (a namespace
# package loses to a regular one).  With the old import order the reference's image_guided_synthesis would drive THIS repository's
# samplers.  (pipeline_ref.npz was generated before the drop-in existed; regenerated with this order in round 6: bit-identical.)
from utils_vc import diffusion_utils as du  # the reference
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import inspect
for _cls in (du.DDIMSampler, du.DDIMSamplerGuidance, du.DDIMSampler_multicond):
    assert inspect.getsourcefile(_cls).startswith("/root/reference/"), inspect.getsourcefile(_cls)
from lvdm_amd import ops
from lvdm_amd.schedule import DiffusionSchedule
import pipeline_duck as pd

ops.use_reference_math(True)


class CPUMulti(du.DDIMSampler_multicond):   # the reference registers its buffers on "cuda" (ddim_multiplecond.py:18-22)
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


du.DDIMSampler_multicond = CPUMulti
out = {}
duck = pd.PipeDuck(DiffusionSchedule())
renderings, guide, masks, noise_shape = pd.inputs()
o = pd.Opts
videos = (renderings * 2. - 1.).permute(3, 0, 1, 2).unsqueeze(0)
for tag, cfg_img in (("cfg3", 3.0), ("cfg1", 1.0)):
    torch.manual_seed(123)
    try:
        res = du.image_guided_synthesis(duck, [o.prompt], videos, noise_shape, o.n_samples, o.ddim_steps, o.ddim_eta,
                                        o.unconditional_guidance_scale, cfg_img, o.frame_stride, o.text_input, True,
                                        o.timestep_spacing, o.guidance_rescale, [0], None, True)
        out[f"{tag}_video"] = res.detach().numpy()
        print(tag, res.shape, float(res.abs().mean()))
    except Exception as e:   # cfg_img == 1.0: the reference hands None to apply_model as the third conditioning
        print(tag, "reference raised", type(e).__name__, str(e)[:120])
# branches of image_guided_synthesis the drivers' settings do not take (diffusion_utils.py:131-133 no text input, :163-169 uncond_type
# "zero_embed", :194 n_samples > 1, :161 / 171-172 classifier-free guidance off), with the reference's plain sampler
class CPUPlain(du.DDIMSampler):
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


du.DDIMSampler = CPUPlain
for tag, kw in (("notext", dict(text_input=False)), ("zero_embed", dict(uncond_type="zero_embed")), ("two_samples", dict(n_samples=2)),
                ("cfg_off", dict(scale=1.0))):
    duck.uncond_type = kw.get("uncond_type", "empty_seq")
    torch.manual_seed(123)
    prompt = "A room" if tag == "notext" else o.prompt      # (a prompt whose stand-in embedding differs from the empty one's: it must be ignored)
    res = du.image_guided_synthesis(duck, [prompt], videos, noise_shape, kw.get("n_samples", o.n_samples), o.ddim_steps, o.ddim_eta,
                                    kw.get("scale", o.unconditional_guidance_scale), o.cfg_img, o.frame_stride, kw.get("text_input", o.text_input), False,
                                    o.timestep_spacing, o.guidance_rescale, [0], None, True)
    out[f"{tag}_video"] = res.detach().numpy()
    print(tag, res.shape, float(res.abs().mean()))
duck.uncond_type = "empty_seq"
np.savez_compressed(os.path.join(HERE, "pipeline_multicond_ref.npz"), **out)
