"""Generates tests/golden/raster_ref_python.npz and raster_c1_oracle.npz.

Run ONLY in the build container (needs /root/reference):  python tests/golden/make_golden_raster.py

Part 1 imports the reference's own Python -- utils/sh_utils.py (eval_sh, :57-112) and
utils/general_utils.py (build_rotation / build_scaling_rotation / strip_symmetric, :68-114) as used by
scene/gaussian_model.py:30-34 -- and records their outputs on seeded inputs: these pin the oracle's
SH->RGB and cov3D against code the reference itself ships (SURVEY.md 8c cross-checks (i),(ii)).
Only INPUT/OUTPUT arrays are stored; no reference source text.

Part 2 stores the oracle's own outputs on scene C1 (regression pin for the oracle + fixture for GPU tests).
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
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))


def load_ref_module(relpath, name):
    # stubs for import-time-only dependencies that are absent here
    for stub in ("cv2", "matplotlib", "matplotlib.pyplot", "matplotlib.cm"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    rng = np.random.default_rng(2024)
    # ---------------- part 1: reference Python ----------------
    sh_utils = load_ref_module("utils/sh_utils.py", "ref_sh_utils")
    gen_utils = load_ref_module("utils/general_utils.py", "ref_general_utils")
    _zeros = torch.zeros

    def zeros_cpu(*a, **k):  # the reference hard-codes device="cuda"
        k.pop("device", None)
        return _zeros(*a, **k)

    gen_utils.torch.zeros = zeros_cpu
    N = 256
    sh = rng.normal(0, 0.3, size=(N, 16, 3)).astype(np.float32)
    dirs = rng.normal(size=(N, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = dirs.astype(np.float32)
    out = {"sh": sh, "dirs": dirs}
    sh_t = torch.tensor(sh, dtype=torch.float64).transpose(1, 2)  # [N,3,16] as gaussian_renderer/__init__.py:81
    for deg in range(4):
        res = sh_utils.eval_sh(deg, sh_t, torch.tensor(dirs, dtype=torch.float64))
        out[f"rgb_deg{deg}"] = torch.clamp_min(res + 0.5, 0.0).numpy()  # gaussian_renderer/__init__.py:85
    scales = np.exp(rng.normal(-2.0, 0.6, size=(N, 3))).astype(np.float32)
    rots = rng.normal(size=(N, 4)).astype(np.float32)
    rots /= np.linalg.norm(rots, axis=1, keepdims=True)  # the model always feeds normalised quaternions
    L = gen_utils.build_scaling_rotation(torch.tensor(1.7 * scales, dtype=torch.float64), torch.tensor(rots, dtype=torch.float64))
    cov = gen_utils.strip_symmetric((L @ L.transpose(1, 2)).to(torch.float32))
    torch.zeros = _zeros
    out.update(scales=scales, rots=rots, scale_modifier=np.float32(1.7), cov3D=cov.numpy().astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "raster_ref_python.npz"), **out)

    # ---------------- part 2: oracle on C1 ----------------
    from oracle import raster_oracle as ro
    import synthetic as syn
    sc = syn.scene_c1()
    cam = sc["cameras"][0]
    H, W = cam["image_height"], cam["image_width"]
    st = ro.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], sc["bg"],
                    W, H, cam["tanfovx"], cam["tanfovy"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"],
                    sh_degree=sc["sh_degree"])
    g = np.random.default_rng(99)
    grads = (g.normal(size=(3, H, W)).astype(np.float32) / (H * W), g.normal(size=(H, W)).astype(np.float32) / (H * W),
             g.normal(size=(H, W)).astype(np.float32) / (H * W))
    gr = ro.backward(st, *grads)
    keep = {k: st[k] for k in ("radii", "keys", "point_list", "ranges", "n_contrib", "color", "depth", "alpha",
                               "tiles_touched", "point_offsets")}
    keep.update({k: v for k, v in gr.items() if k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D",
                                                      "dL_dsh", "dL_dscales", "dL_drotations")})
    keep.update(gC=grads[0], gD=grads[1], gA=grads[2], R=np.int64(st["R"]))
    np.savez_compressed(os.path.join(HERE, "raster_c1_oracle.npz"), **keep)
    print("wrote goldens:", {k: os.path.getsize(os.path.join(HERE, k)) for k in os.listdir(HERE) if k.endswith(".npz")})


if __name__ == "__main__":
    main()
