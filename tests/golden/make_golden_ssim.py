"""Golden vectors for the image losses: imports the REFERENCE's own utils/loss_utils.py (pure torch, runs on CPU) and
records ssim / l1 values and d ssim / d img1 on seeded inputs.  Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_ssim.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/utils/loss_utils.py"
spec = importlib.util.spec_from_file_location("ref_loss_utils", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
g = torch.Generator().manual_seed(20)
cases = {"a": (3, 40, 56), "b": (3, 33, 17), "c": (1, 16, 16), "d": (2, 3, 24, 31)}
for tag, shp in cases.items():
    x = torch.rand(shp, generator=g, dtype=torch.float32).requires_grad_(True)
    y = (x.detach() + 0.15 * torch.randn(shp, generator=g)).clamp(0, 1)
    out[f"{tag}_x"], out[f"{tag}_y"] = x.detach().numpy(), y.numpy()
    v = ref.ssim(x, y)
    (gx,) = torch.autograd.grad(v, x)
    out[f"{tag}_ssim"], out[f"{tag}_grad"] = v.detach().numpy(), gx.numpy()
    out[f"{tag}_l1"] = ref.l1_loss(x.detach(), y).numpy()
    if len(shp) == 3:
        mask = (torch.rand((1,) + shp[1:], generator=g) > 0.4).float()
        out[f"{tag}_mask"] = mask.numpy()
        vm = ref.ssim(x, y, mask)
        (gm,) = torch.autograd.grad(vm, x)
        out[f"{tag}_ssim_masked"], out[f"{tag}_grad_masked"] = vm.detach().numpy(), gm.numpy()
        out[f"{tag}_l1_masked"] = ref.l1_loss_mask(x.detach(), y, mask).numpy()
    else:
        out[f"{tag}_ssim_per_batch"] = ref.ssim(x.detach(), y, size_average=False).numpy()
# LossGuidance with ssim_guidance (utils/viewcrafter_wrapper.py:145-155), from the reference's ssim_noavg
x = (torch.rand((3, 1, 24, 40), generator=g) * 2 - 1).requires_grad_(True)       # decoded frame [3,1,H,W] in [-1,1]
G_img = torch.rand((1, 3, 24, 40), generator=g)
mask = (torch.rand((1, 1, 24, 40), generator=g) > 0.35).float()
D = ((x.permute(1, 0, 2, 3) + 1.) / 2.).clamp(0, 1)
gm = mask.expand_as(D)
recon = (0.5 * torch.square(D - G_img) * gm).sum()
loss = 0.8 * recon + 0.2 * (1.0 - ref.ssim_noavg(D.float(), G_img.float(), mask=gm)).sum()
(gx,) = torch.autograd.grad(loss, x)
out["lg_x"], out["lg_G"], out["lg_mask"] = x.detach().numpy(), G_img.numpy(), mask.numpy()
out["lg_loss"], out["lg_grad"], out["lg_numel"] = loss.detach().numpy(), gx.numpy(), gm.sum().numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ssim_ref.npz"), **out)
print({k: v.shape for k, v in out.items() if "ssim" in k})
