"""Golden for the VAE encoder: the reference's lvdm/modules/networks/ae_modules.py::Encoder and
lvdm/distributions.py::DiagonalGaussianDistribution on a tiny config with name-derived weights.  Build container only."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/third_party/ViewCrafter")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fill_by_name import fill_by_name
from lvdm.distributions import DiagonalGaussianDistribution
from lvdm.modules.networks.ae_modules import Encoder

cfg = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4], num_res_blocks=2,
           attn_resolutions=[], dropout=0.0)
enc = fill_by_name(Encoder(**cfg), std=0.05).eval()
quant = fill_by_name(torch.nn.Conv2d(8, 8, 1), std=0.3)
g = torch.Generator().manual_seed(31)
x = torch.rand(2, 3, 20, 28, generator=g) * 2 - 1
with torch.no_grad():
    h = enc(x)
    post = DiagonalGaussianDistribution(quant(h))
    noise = torch.randn(post.mean.shape, generator=g)
    z = post.sample(noise=noise)
np.savez_compressed(os.path.join(HERE, "vae_encoder_ref.npz"), x=x.numpy(), h=h.numpy(), mean=post.mean.numpy(), std=post.std.numpy(),
                    noise=noise.numpy(), z=z.numpy(), keys=np.array(sorted(enc.state_dict().keys())),
                    quant_w=quant.weight.detach().numpy(), quant_b=quant.bias.detach().numpy())
print(h.shape, z.shape, float(z.std()))
