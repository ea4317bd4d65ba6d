"""Golden for SURVEY row B1: runs the REFERENCE's image_guided_synthesis (utils_vc/diffusion_utils.py:118-223) with its
own DDIMSampler / DDIMSamplerGuidance on the stand-in model of tests/pipeline_duck.py (CPU) and stores the decoded
videos.  Build container only (needs /root/reference).  Output: tests/golden/pipeline_ref.npz (arrays only)."""
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
from lvdm_amd.guidance import LossGuidance
from lvdm_amd.schedule import DiffusionSchedule
import pipeline_duck as pd

ops.use_reference_math(True)


class CPUSampler(du.DDIMSampler):          # the reference registers its buffers on "cuda" (ddim.py:20-24)
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


class CPUGuided(du.DDIMSamplerGuidance):
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


du.DDIMSampler, du.DDIMSamplerGuidance = CPUSampler, CPUGuided
import lvdm.models.samplers.ddim_guidance as ddg
ddg.torch.cuda.empty_cache = lambda: None  # no-op on CPU anyway

out = {}
duck = pd.PipeDuck(DiffusionSchedule())
renderings, guide, masks, noise_shape = pd.inputs()
o = pd.Opts
videos = (renderings * 2. - 1.).permute(3, 0, 1, 2).unsqueeze(0)
for tag, no_guidance in (("plain", True), ("guided", False)):
    lg = None
    if not no_guidance:
        lg = LossGuidance(ddim_steps=o.ddim_steps, recur_steps=1, device="cpu")
        lg.set_hw(renderings.shape[1], renderings.shape[2])
        lg.set_guidance_images(guide)
        lg.set_guidance_masks(masks)
    torch.manual_seed(123)
    res = du.image_guided_synthesis(duck, [o.prompt], videos, noise_shape, o.n_samples, o.ddim_steps, o.ddim_eta,
                                    o.unconditional_guidance_scale, o.cfg_img, o.frame_stride, o.text_input, o.multiple_cond_cfg,
                                    o.timestep_spacing, o.guidance_rescale, [0], lg, no_guidance)
    out[f"{tag}_video"] = res.detach().numpy()
    print(tag, res.shape, float(res.abs().mean()))
np.savez_compressed(os.path.join(HERE, "pipeline_ref.npz"), **out)
