"""Golden for the image-conditioning projector: the reference's lvdm/modules/encoders/resampler.py
Resampler (frame-wise queries) and ImageProjModel on small configs with name-derived weights, forward and input gradient.
Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/third_party/ViewCrafter")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fill_by_name import fill_by_name
from lvdm.modules.encoders.resampler import ImageProjModel, Resampler

CFG = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=4, embedding_dim=96, output_dim=80, ff_mult=4, video_length=3)
PROJ = dict(cross_attention_dim=64, clip_embeddings_dim=48, clip_extra_context_tokens=4)
if __name__ == "__main__":
    g = torch.Generator().manual_seed(77)
    rs = fill_by_name(Resampler(**CFG), std=0.08).eval()
    x = torch.randn(2, 37, CFG["embedding_dim"], generator=g).requires_grad_(True)
    y = rs(x)
    probe = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad((y * probe).sum(), x)
    pm = fill_by_name(ImageProjModel(**PROJ), std=0.1).eval()
    e = torch.randn(3, PROJ["clip_embeddings_dim"], generator=g)
    with torch.no_grad():
        t = pm(e)
    np.savez_compressed(os.path.join(HERE, "resampler_ref.npz"), x=x.detach().numpy(), y=y.detach().numpy(), probe=probe.numpy(),
                        gx=gx.numpy(), keys=np.array(sorted(rs.state_dict().keys())), e=e.numpy(), t=t.numpy(),
                        proj_keys=np.array(sorted(pm.state_dict().keys())))
    print(y.shape, float(y.std()), float(gx.abs().max()), t.shape)
