"""Generates tests/golden/raster_camera_ref.npz by IMPORTING THE REFERENCE'S OWN PYTHON (build container only):

  scene/cameras.py:72-93        PseudoCamera: world_view_transform / projection_matrix / full_proj_transform / camera_center
  utils/graphics_utils.py:38-75 getWorld2View2, getProjectionMatrix (the non-standard P with P[2,2] = P[3,2] = 1)
  gaussian_renderer/__init__.py:66-88 (restated with the reference's own functions, `pipe.compute_cov3D_python` /
      `pipe.convert_SHs_python` branches): cov3D_precomp = strip_symmetric(L L^T)  (scene/gaussian_model.py:30-34),
      colors_precomp = clamp_min(eval_sh(deg, shs_view, normalize(xyz - camera_center)) + 0.5, 0)

for the cameras of the synthetic raster scenes (C1: 1 camera, C2: 6 ring cameras) and the Gaussians of scene C1.
These pin `synthetic.make_camera` (what every raster test / bench feeds the rasterizer) and the precomputed-colour /
precomputed-covariance operator inputs to code the reference ships.  Only arrays are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))


def load_ref_module(relpath, name):
    for stub in ("cv2", "matplotlib", "matplotlib.pyplot", "matplotlib.cm"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    import synthetic as syn
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference hard-codes .cuda(); this process has no GPU
    sys.path.insert(0, REF)
    cameras = load_ref_module("scene/cameras.py", "ref_cameras")
    sh_utils = load_ref_module("utils/sh_utils.py", "ref_sh_utils")
    gen_utils = load_ref_module("utils/general_utils.py", "ref_general_utils")
    _zeros = torch.zeros
    gen_utils.torch.zeros = lambda *a, **k: _zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})

    out = {}
    c1, c2 = syn.scene_c1(), syn.scene_c2(P=16)
    cams = [("c1_0", c1["cameras"][0])] + [(f"c2_{i}", c) for i, c in enumerate(c2["cameras"])]
    for tag, cam in cams:
        w2v = cam["viewmatrix"].T.astype(np.float64)        # synthetic stores the transposed world-to-view
        R, T = w2v[:3, :3].T.copy(), w2v[:3, 3].copy()      # reference convention: R = camera-to-world rotation, T = w2v translation
        pc = cameras.PseudoCamera(R, T, cam["FoVx"], cam["FoVy"], cam["image_width"], cam["image_height"])
        out[f"{tag}_R"], out[f"{tag}_T"] = R, T
        out[f"{tag}_fov"] = np.array([cam["FoVx"], cam["FoVy"]])
        out[f"{tag}_world_view_transform"] = pc.world_view_transform.numpy()
        out[f"{tag}_projection_matrix"] = pc.projection_matrix.numpy()
        out[f"{tag}_full_proj_transform"] = pc.full_proj_transform.numpy()
        out[f"{tag}_camera_center"] = pc.camera_center.numpy()
        if tag == "c1_0":
            center = pc.camera_center
    # python-side SH -> colour and covariance for the Gaussians of scene C1 (float32, as the reference would run them)
    xyz = torch.tensor(c1["means3D"])
    feats = torch.tensor(c1["shs"])                          # [P, 16, 3] = pc.get_features
    shs_view = feats.transpose(1, 2).view(-1, 3, 16)
    dir_pp = xyz - center.repeat(feats.shape[0], 1)
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    for deg in (0, 3):
        sh2rgb = sh_utils.eval_sh(deg, shs_view, dir_pp_normalized)
        out[f"c1_colors_precomp_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()
    L = gen_utils.build_scaling_rotation(1.0 * torch.tensor(c1["scales"]), torch.tensor(c1["rotations"]))
    out["c1_cov3D_precomp"] = gen_utils.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    np.savez_compressed(os.path.join(HERE, "raster_camera_ref.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "raster_camera_ref.npz")), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
