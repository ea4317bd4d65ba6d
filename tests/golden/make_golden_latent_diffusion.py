"""Golden for SURVEY rows B5 (apply_model / hybrid conditioning), the VAE glue of B13 / N2 (encode / decode through LatentDiffusion) and the drop-in's
state-dict claim, from THE REFERENCE'S OWN CLASSES: lvdm/models/ddpm3d.py::LatentDiffusion (DDPM, DiffusionWrapper) and
lvdm/models/autoencoder.py::AutoencoderKL, imported in the build container.  Until round 6 these were compared by key counts on the meta device and by
shape (tests/test_lvdm_dropin.py, test_latent_diffusion_wrapper_shapes_and_hybrid_conditioning); the classes themselves could not be imported because
they derive from pytorch_lightning.LightningModule, which the image lacks.  Here `pytorch_lightning` is a two-name placeholder -- LightningModule = an
nn.Module subclass with a `device` property, rank_zero_only = identity -- which is all the SAMPLING side of these classes uses of it (the training side,
which needs the real package, is out of scope: SURVEY 2); cv2 / torchvision / kornia / open_clip are inert placeholders, untouched by what is run.
Weights come from parameter NAMES (tests/fill_by_name.py), so the rebuilt drop-in gets identical weights iff its state-dict keys are the reference's.
Output: tests/golden/latent_diffusion_ref.npz (arrays + key-name strings only)."""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VC = "/root/reference/third_party/ViewCrafter"
sys.path.insert(0, VC)
pl = types.ModuleType("pytorch_lightning")


class LightningModule(torch.nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device


pl.LightningModule = LightningModule
plu = types.ModuleType("pytorch_lightning.utilities")
plu.rank_zero_only = lambda f: f
pl.utilities = plu
sys.modules["pytorch_lightning"], sys.modules["pytorch_lightning.utilities"] = pl, plu
for n in ("cv2", "torchvision", "torchvision.utils", "kornia", "open_clip"):
    try:
        importlib.import_module(n)
    except Exception:  # noqa: BLE001
        sys.modules[n] = MagicMock(name=n)
from lvdm.models import ddpm3d  # noqa: E402  (the reference; guidedvd-3dgs_amd/ is not on sys.path here)
import inspect  # noqa: E402
assert inspect.getsourcefile(ddpm3d.LatentDiffusion).startswith("/root/reference/")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fill_by_name import fill_by_name  # noqa: E402
from latent_diffusion_cfg import UNET, VAE, LD_KW  # noqa: E402  (the configuration both sides build from)


class AD(dict):
    __getattr__ = dict.__getitem__


def ad(d):
    return AD({k: ad(v) if isinstance(v, dict) else v for k, v in d.items()})


def main():
    ld = ddpm3d.LatentDiffusion(first_stage_config=ad(VAE), cond_stage_config=ad({"target": "torch.nn.Identity"}), unet_config=ad(UNET), **LD_KW).eval()
    fill_by_name(ld.model)
    fill_by_name(ld.first_stage_model)
    sd = ld.state_dict()
    out = {"keys": np.array(sorted(sd.keys())), "shapes": np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])}
    for k in sorted(sd):
        if not k.startswith(("model.", "first_stage_model.")):
            out["buf_" + k] = sd[k].numpy()
    g = torch.Generator().manual_seed(17)
    T = 2
    x = torch.randn(1, 4, T, 8, 8, generator=g)
    cond = {"c_crossattn": [torch.randn(1, 77, 48, generator=g), torch.randn(1, 20, 48, generator=g)], "c_concat": [torch.randn(1, 4, T, 8, 8, generator=g)]}
    t = torch.tensor([400])
    fs = torch.tensor([10])
    with torch.no_grad():
        v = ld.apply_model(x, t, cond, fs=fs)
        out["x"], out["c_text"], out["c_img"], out["c_concat"], out["v"] = x.numpy(), cond["c_crossattn"][0].numpy(), cond["c_crossattn"][1].numpy(), cond["c_concat"][0].numpy(), v.numpy()
        out["x0_from_v"] = ld.predict_start_from_z_and_v(x, t, v).numpy()
        out["eps_from_v"] = ld.predict_eps_from_z_and_v(x, t, v).numpy()
        noise = torch.randn(x.shape, generator=g)
        out["q_noise"], out["q_sample"] = noise.numpy(), ld.q_sample(x, t, noise=noise).numpy()
    z = torch.randn(1, 4, T, 8, 8, generator=g)
    out["z"] = z.numpy()
    for tag, pf in (("perframe", True), ("batched", False)):
        ld.perframe_ae = pf
        with torch.no_grad():
            out[f"decode_{tag}"] = ld.decode_first_stage(z).numpy()
        zr = z.clone().requires_grad_(True)
        img = ld.differentiable_decode_first_stage(zr)
        gi = torch.randn(img.shape, generator=torch.Generator().manual_seed(3))
        (gz,) = torch.autograd.grad(img, zr, gi)
        out[f"decode_grad_{tag}"] = gz.numpy()
        out["decode_gi"] = gi.numpy()
        video = torch.rand(1, 3, T, 16, 16, generator=torch.Generator().manual_seed(4)) * 2 - 1
        out["video"] = video.numpy()
        torch.manual_seed(5)
        out[f"encode_{tag}"] = ld.encode_first_stage(video).numpy()
    np.savez_compressed(os.path.join(HERE, "latent_diffusion_ref.npz"), **out)
    print(len(out["keys"]), "state-dict keys;", {k: v.shape for k, v in out.items() if k in ("v", "decode_perframe", "encode_perframe")})
    print("decode perframe == batched:", np.abs(out["decode_perframe"] - out["decode_batched"]).max(), " encode:", np.abs(out["encode_perframe"] - out["encode_batched"]).max())


if __name__ == "__main__":
    main()
