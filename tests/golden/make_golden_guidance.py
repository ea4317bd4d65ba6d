"""Golden for SURVEY row B12 from THE REFERENCE'S OWN CLASS: utils/viewcrafter_wrapper.py::LossGuidance (:47-192), imported in the build container
with the absent third-party packages of that module's import block (dust3r, pytorch3d, lpipsPyTorch, torchmetrics, MiDaS, cv2, ...) replaced by
inert placeholders -- none of them is touched by LossGuidance (torch, F, utils/loss_utils.py::ssim_noavg and the module's own learning_rate_decay
are).  Until round 6 the class was only RESTATED inside the sampler goldens' generators; this pins lvdm_amd.guidance.LossGuidance to it directly:
set_hw / set_guidance_images (bilinear resize + clamp) / set_guidance_masks (nearest), __call__ per frame with and without masks, the
`ssim_guidance` mix, the `scale_guidance_weight` schedule.  (`lpips_guidance` needs torchvision's VGG: pinned separately by make_golden_vgg.py.)
Output: tests/golden/guidance_ref.npz (arrays only)."""
import importlib
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
placeholders = []
for name in ("cv2", "lpipsPyTorch", "torchmetrics", "torchmetrics.functional", "torchmetrics.functional.regression", "utils.midas_depth_estimator",
             "dust3r", "dust3r.inference", "dust3r.utils", "dust3r.utils.image", "dust3r.image_pairs", "dust3r.cloud_opt", "dust3r.utils.device",
             "third_party", "third_party.ViewCrafter", "third_party.ViewCrafter.viewcrafter", "third_party.ViewCrafter.configs",
             "third_party.ViewCrafter.configs.infer_config", "third_party.ViewCrafter.utils_vc", "third_party.ViewCrafter.utils_vc.pvd_utils",
             "pytorch3d", "pytorch3d.renderer", "torchvision", "torchvision.models", "plyfile", "open3d"):
    try:
        importlib.import_module(name)
    except Exception:  # noqa: BLE001 -- absent (or unimportable here): an inert placeholder, recorded below
        sys.modules[name] = MagicMock(name=name)
        placeholders.append(name)
vw = importlib.import_module("utils.viewcrafter_wrapper")
assert vw.__file__.startswith(REF + "/"), vw.__file__
import inspect
assert inspect.getsourcefile(vw.LossGuidance).startswith(REF + "/")


def main():
    g = torch.Generator().manual_seed(41)
    F_, H, W = 5, 24, 32                       # decoded frames at H x W; the 3DGS renders arrive at another size and are resized
    imgs = torch.rand(F_, 3, 37, 50, generator=g) * 1.2 - 0.1          # a little outside [0, 1]: the clamp matters
    masks = (torch.rand(F_, 1, 37, 50, generator=g) > 0.35).float()
    D = (torch.randn(3, F_, H, W, generator=g) * 0.7)                  # decoded x0 frames, roughly [-1, 1] with tails (the clamp matters)
    out = {"imgs": imgs.numpy(), "masks": masks.numpy(), "D": D.numpy()}
    for tag, kw, use_masks in (("l2_masked", {}, True), ("l2_nomask", {}, False), ("ssim_masked", {"ssim_guidance": True}, True),
                               ("ssim_nomask", {"ssim_guidance": True}, False), ("w02", {"w_recon_loss": 0.2}, True)):
        lg = vw.LossGuidance(ddim_steps=50, recur_steps=1, device="cpu", **kw)
        lg.set_hw(H, W)
        lg.set_guidance_images(imgs)
        if use_masks:
            lg.set_guidance_masks(masks)
        if tag == "l2_masked":
            out["resized_imgs"], out["resized_masks"] = lg.guidance_images.numpy(), lg.guidance_masks.numpy()
        Dr = D.clone().requires_grad_(True)
        losses, numels, total = [], [], None
        for j in range(F_):
            ld, n = lg(Dr[:, j:j + 1], 10, j, j + 1)
            losses.append(float(ld["recon"]))
            numels.append(float(n))
            total = ld["recon"] if total is None else total + ld["recon"]
        (gD,) = torch.autograd.grad(total, Dr)
        out[f"{tag}_loss"], out[f"{tag}_numel"], out[f"{tag}_grad"] = np.array(losses), np.array(numels), gD.numpy()
    lg = vw.LossGuidance(ddim_steps=50, recur_steps=2, device="cpu", scale_guidance_weight=True)
    steps = np.array([0, 1, 100, 1250, 2499, 2500, 9999])
    out["weight_steps"], out["weight_values"] = steps, np.array([lg.guidance_weight_fn(int(s)) for s in steps])
    out["placeholders"] = np.array(placeholders)
    np.savez_compressed(os.path.join(HERE, "guidance_ref.npz"), **out)
    print("placeholders for absent packages:", placeholders)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith(("_loss", "_numel"))})
    print("weights", out["weight_values"])


if __name__ == "__main__":
    main()
