"""Golden for the drop-in's FULL-SIZE state-dict surface ("the same state-dict keys, so load_model_checkpoint loads the ViewCrafter checkpoint
strict") from THE REFERENCE'S OWN CLASSES at THE SHIPPED CONFIGURATION: lvdm/models/ddpm3d.py::VIPLatentDiffusion built from the reference's
configs/inference_pvd_1024.yaml (read with PyYAML here; not stored) -- U-Net 1.44 B parameters, KL-VAE, Resampler, schedule buffers -- with the two CLIP
towers' nodes replaced by torch.nn.Identity (open_clip is absent: their keys stay pinned by counts and names in tests/test_lvdm_dropin.py) and
pytorch_lightning as the two-name placeholder of make_golden_latent_diffusion.py.  Stored: every key with its shape (names + shape strings), the values of
the schedule buffers, a few scalar attributes, and a SHA-256 of the canonical JSON of the yaml's `params` mapping -- the test recomputes it from
lvdm_amd.model.viewcrafter_yaml_node(), which pins that transcription of the yaml without storing the yaml.
Output: tests/golden/full_keys_ref.npz."""
import hashlib
import importlib
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
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
from lvdm.models import ddpm3d  # noqa: E402
import inspect  # noqa: E402
assert inspect.getsourcefile(ddpm3d.VIPLatentDiffusion).startswith("/root/reference/")


class AD(dict):
    __getattr__ = dict.__getitem__


def ad(d):
    return AD({k: ad(v) if isinstance(v, dict) else v for k, v in d.items()})


def canonical(params):
    """The yaml's params with the two CLIP nodes reduced to their target + params (what both sides can state), as sorted JSON."""
    return json.dumps(params, sort_keys=True, separators=(",", ":"))


def main():
    with open(os.path.join(VC, "configs", "inference_pvd_1024.yaml")) as fh:
        node = yaml.safe_load(fh)["model"]
    assert node["target"] == "lvdm.models.ddpm3d.VIPLatentDiffusion"
    params = node["params"]
    digest = hashlib.sha256(canonical(params).encode()).hexdigest()
    build = json.loads(json.dumps(params))
    build["cond_stage_config"] = {"target": "torch.nn.Identity"}
    build["img_cond_stage_config"] = {"target": "torch.nn.Identity"}
    model = ddpm3d.VIPLatentDiffusion(**ad(build))
    sd = model.state_dict()
    keys = sorted(sd)
    out = {"keys": np.array(keys), "shapes": np.array([str(tuple(sd[k].shape)) for k in keys]), "yaml_params_sha256": np.array(digest)}
    for k in keys:
        if k.split(".")[0] not in ("model", "first_stage_model", "image_proj_model", "cond_stage_model", "embedder"):
            out["buf_" + k] = sd[k].numpy()
    out["scalars"] = np.array(json.dumps({"scale_factor": float(model.scale_factor), "uncond_type": model.uncond_type, "perframe_ae": bool(model.perframe_ae),
                                          "num_timesteps": int(model.num_timesteps), "parameterization": model.parameterization,
                                          "use_dynamic_rescale": bool(model.use_dynamic_rescale), "conditioning_key": model.model.conditioning_key,
                                          "image_size": list(model.image_size), "channels": int(model.channels), "temporal_length": int(model.temporal_length)}))
    np.savez_compressed(os.path.join(HERE, "full_keys_ref.npz"), **out)
    n = lambda p: sum(v.numel() for k, v in sd.items() if k.startswith(p))
    print(len(keys), "keys;  U-Net", n("model.diffusion_model.") / 1e6, "M  VAE", n("first_stage_model."), " Resampler", n("image_proj_model."), " sha", digest[:16])


if __name__ == "__main__":
    main()
