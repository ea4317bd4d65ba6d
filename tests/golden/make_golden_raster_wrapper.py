"""Golden for SURVEY rows A1-A4 / A12 (the PYTHON half of the rasterizer: settings tuple, GaussianRasterizer.forward / markVisible, the autograd
operator's argument slots, returned-gradient order and confidence scaling) from THE REFERENCE'S OWN WRAPPER:
submodules/diff-gaussian-rasterization-confidence/diff_gaussian_rasterization/__init__.py, imported in the build container with its compiled `_C`
replaced by the deterministic stand-in of tests/raster_fake_backend.py (the CUDA extension cannot be built here).  What is recorded: the descriptor of
every argument the wrapper hands the native entry points, slot by slot, the outputs, and the gradients that reach the inputs.
Output: tests/golden/raster_wrapper_ref.npz (arrays + descriptor strings only)."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PKG = "/root/reference/submodules/diff-gaussian-rasterization-confidence/diff_gaussian_rasterization"
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raster_fake_backend as fb  # noqa: E402

backend = fb.FakeBackend()
fake = types.ModuleType("ref_dgr._C")
fake.rasterize_gaussians, fake.rasterize_gaussians_backward, fake.mark_visible = backend.rasterize_gaussians, backend.rasterize_gaussians_backward, backend.mark_visible
spec = importlib.util.spec_from_file_location("ref_dgr", os.path.join(PKG, "__init__.py"), submodule_search_locations=[PKG])
ref = importlib.util.module_from_spec(spec)
sys.modules["ref_dgr"], sys.modules["ref_dgr._C"] = ref, fake
spec.loader.exec_module(ref)
assert ref.__file__.startswith("/root/reference/")

out = {"settings_fields": np.array(ref.GaussianRasterizationSettings._fields)}
sc = fb.scene()
for case in ("sh", "pre"):
    backend.calls.clear()
    outs, grads = fb.drive(ref, backend, sc, case)
    out[f"{case}_calls"] = np.array(json.dumps(backend.calls))
    for k, v in outs.items():
        out[f"{case}_out_{k}"] = v.numpy()
    for k, v in grads.items():
        out[f"{case}_grad_{k}"] = v.numpy()
backend.calls.clear()
S = ref.GaussianRasterizationSettings(6, 8, 0.7, 0.6, sc["bg"], 1.25, sc["view"], sc["proj"], 3, sc["campos"], False, False, sc["confidence"])
vis = ref.GaussianRasterizer(S).markVisible(sc["means3D"])
out["vis"], out["vis_calls"] = vis.numpy(), np.array(json.dumps(backend.calls))
msgs = []
for kw in (dict(), dict(shs=sc["shs"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"]),
           dict(shs=sc["shs"]), dict(shs=sc["shs"], scales=sc["scales"]), dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=sc["cov3D"])):
    try:
        ref.GaussianRasterizer(S)(means3D=sc["means3D"], means2D=sc["means2D"], opacities=sc["opacities"], **kw)
        msgs.append("")
    except Exception as e:  # noqa: BLE001
        msgs.append(f"{type(e).__name__}: {e}")
out["error_messages"] = np.array(msgs)
np.savez_compressed(os.path.join(HERE, "raster_wrapper_ref.npz"), **out)
print(out["settings_fields"]); print(msgs); print({k: v.shape for k, v in out.items() if "grad" in k})
