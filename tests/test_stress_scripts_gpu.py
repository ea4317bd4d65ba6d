"""The two randomised stress / fuzz scripts as `-m gpu` tests (each in a process of its own; ~20 s together on an MI355X):

  tests/scripts/r5_raster_stress.py    60 random scene / image / SH-degree configurations of the rasterizer operator (each twice, images and
                                       gradients bit-identical), one scene under 100 cameras from inside the cloud to far away (two passes with different
                                       speculation state, bit-identical), then sizes past the bench's: 2 M and 5 M Gaussians, tile lists past 16 384
                                       entries, blanket splats, 3840 x 2160, a side stream, strided inputs.  (Found the capacity-0 decode of a
                                       one-instance binning chunk and the unsorted-list walk of a mis-guessed speculative forward in round 5.)
  tests/scripts/r5_raster_threads.py   two host threads on two streams rendering + back-propagating concurrently through the compiled operator (GIL
                                       released in the native call): bit-identical to the single-threaded results.
  tests/scripts/r5_raster_soak.py      12 000 training iterations with densification-like changes of the point count: device and host memory flat.
  tests/scripts/r5_diffusion_fuzz.py   random shapes through the MFMA GEMM (+ Linear with the LayerNorm fold), flash attention forward and
                                       backward, the implicit-GEMM convolution (all forms) forward and input gradient, against fp32 torch math.
  tests/scripts/r5_guided_soak.py      240 guided + 240 plain DDIM steps on the miniature: allocator and host memory flat (GradCells, norm states,
                                       caches, autograd graphs all released).
  tests/scripts/r5_unet_shape_fuzz.py  a three-level miniature of the ViewCrafter U-Net on random (batch, frames, height, width, context length),
                                       fp16 HIP path against the fp32 torch form, forward and input gradient.
  tests/scripts/r5_vae_shape_fuzz.py   the same for a three-level VAE decoder miniature (512-wide mid attention head) on random (frames, height,
                                       width, frames per call)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,args,marker", [("r5_raster_stress.py", [], "strided inputs: ok"),
                                                ("r5_raster_threads.py", [], "bit-identical to the single-threaded run"),
                                                ("r5_raster_soak.py", ["12000"], "raster soak ok"),
                                                ("r5_diffusion_fuzz.py", ["11"], "diffusion fuzz ok"),
                                                ("r5_diffusion_fuzz.py", ["23"], "diffusion fuzz ok"),
                                                ("r5_guided_soak.py", ["240"], "guided soak ok"),
                                                ("r5_unet_shape_fuzz.py", ["3", "16"], "unet shape fuzz ok"),
                                                ("r5_vae_shape_fuzz.py", ["5", "12"], "vae shape fuzz ok")])
def test_stress_script(script, args, marker):
    r = subprocess.run([sys.executable, os.path.join(HERE, "scripts", script)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and marker in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
