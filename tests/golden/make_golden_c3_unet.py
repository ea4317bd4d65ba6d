"""BASELINE configs[2] (C3) resolution anchor for the U-Net alone: ONE forward of the REFERENCE's `UNetModel`
(lvdm/modules/networks/openaimodel3d.py:282-603) AT THE SHIPPED WIDTH (configs/inference_pvd_1024.yaml:33-62: model_channels 320,
[1, 2, 4, 4], 1.44 B parameters) on a 72 x 128 latent (the 576 x 1024 video BASELINE quotes), 2 frames so that the fp32 evaluation
fits the build container (the 9216 x 9216 level-0 attention of a frame is 340 MB per head in fp32).  Frames interact only through the
temporal layers, so a 2-frame video is its own problem at the full spatial size: the tile configurations, grid readings and attention
shapes of the level-0 .. level-3 kernels are the ones of the 25-frame run.

Imports the reference's Python (build container only); stores ARRAYS only:
  * `y32`: the fp32 output [1, 4, 2, 72, 128] (the anchor);
  * `e16`: max |y16 - y32| / max |y32| of the SAME module under fp16 autocast (how the reference runs it, viewcrafter.py:104) -- the
    bar for the fp16 HIP path, measured on the same weights and inputs.
Weights by state-dict key name (tests/fill_by_name.py, std 0.02), inputs from seeded CPU generators (tests/c3_inputs.py): the GPU test
regenerates both.  ~15 minutes on 8 cores.  python tests/golden/make_golden_c3_unet.py
"""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/third_party/ViewCrafter")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from c3_inputs import STD, inputs  # noqa: E402  (shared with the GPU test)
from fill_by_name import fill_by_name  # noqa: E402

from lvdm.modules.networks.openaimodel3d import UNetModel  # noqa: E402

UNET = dict(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
            channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=1024, use_linear=True,
            use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
            use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
            image_cross_attention=True, default_fs=10, fs_condition=True)


def main():
    t0 = time.time()
    d = inputs()
    unet = fill_by_name(UNetModel(**UNET), std=STD).eval()
    print(f"reference U-Net built and filled in {time.time() - t0:.0f} s", flush=True)
    with torch.no_grad():
        y32 = unet(d["x"], d["t"], context=d["ctx"], fs=d["fs"]).float()
        print(f"fp32 forward: {time.time() - t0:.0f} s", flush=True)
        with torch.autocast("cpu", dtype=torch.float16):
            y16 = unet(d["x"], d["t"], context=d["ctx"], fs=d["fs"]).float()
        print(f"fp16-autocast forward: {time.time() - t0:.0f} s", flush=True)
    e16 = float((y16 - y32).abs().max() / y32.abs().max())
    e16_rms = float((y16 - y32).pow(2).mean().sqrt() / y32.pow(2).mean().sqrt())
    print(f"reference fp16 autocast vs fp32: max {e16:.3e}, rms {e16_rms:.3e}; |y32| max {float(y32.abs().max()):.3f}, std {float(y32.std()):.3f}")
    np.savez_compressed(os.path.join(HERE, "c3_unet_ref.npz"), y32=y32.numpy(), e16=np.float64(e16), e16_rms=np.float64(e16_rms))


if __name__ == "__main__":
    main()
