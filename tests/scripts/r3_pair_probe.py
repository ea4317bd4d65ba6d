"""Batched CFG pair vs two sequential U-Net calls on the device (fp16 token-major miniature): forward and input gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from fill_by_name import fill_by_name
from lvdm_amd.model import DiffusionWrapper
from lvdm_amd.unet import UNetModel
dev = "cuda:0"
cfg = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[1, 2], num_res_blocks=1,
           channel_mult=[1, 2], dropout=0.0, num_head_channels=64, transformer_depth=1, context_dim=64, use_linear=True,
           use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
           use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
           image_cross_attention=True, default_fs=10, fs_condition=True)
unet = fill_by_name(UNetModel(**cfg), std=0.08).half().eval().to(dev).to_token_major()
unet.requires_grad_(False)
w = DiffusionWrapper(unet)
g = torch.Generator(device=dev).manual_seed(3)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
T, H, W = 5, 16, 24
x = mk(1, 4, T, H, W)
c = {"c_crossattn": [mk(1, 93, 64).half()], "c_concat": [(mk(1, 4, T, H, W) * 0.2).half()]}
uc = {"c_crossattn": [mk(1, 93, 64).half()], "c_concat": c["c_concat"]}
t = torch.tensor([500], device=dev)
fs = torch.tensor([10], device=dev)
probe = mk(1, 4, T, H, W)
def seq(x):
    return w(x.half(), t, **c, fs=fs), w(x.half(), t, **uc, fs=fs)
def pair(x):
    cc = {k: [torch.cat([a, b]) for a, b in zip(c[k], uc[k])] for k in c}
    e = w(torch.cat([x, x]).half(), torch.cat([t, t]), **cc, fs=torch.cat([fs, fs]))
    return e.chunk(2)
rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
res = {}
for name, fn in (("seq", seq), ("pair", pair)):
    xs = x.clone().requires_grad_(True)
    e1, e2 = fn(xs)
    (gx,) = torch.autograd.grad(((e1.float() * probe).sum() + 0.7 * (e2.float() * probe).sum()), xs)
    res[name] = (e1.detach(), e2.detach(), gx.detach())
print("forward cond  pair vs seq:", rel(res["pair"][0], res["seq"][0]), " uncond:", rel(res["pair"][1], res["seq"][1]), " input gradient:", rel(res["pair"][2], res["seq"][2]))
