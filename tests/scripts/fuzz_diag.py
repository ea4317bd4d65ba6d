import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
from raster_compare import run_hip, run_oracle, rel_to_max
from test_raster_fuzz_gpu import _scene
for seed in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 10):
    sc, cam, grads = _scene(seed)
    st_o, g_o = run_oracle(sc, cam, grads)
    _, g = run_hip(sc, cam, grads, alpha_override=st_o["alpha"])
    P = sc["means3D"].shape[0]
    print("seed", seed, "P", P, "WxH", cam["image_width"], cam["image_height"], "R", st_o["R"])
    for k in ("dL_dmeans3D", "dL_dscales", "dL_dmeans2D", "dL_dcov3D", "dL_dopacity"):
        e = np.abs(g[k] - g_o[k]).reshape(P, -1).max(1)
        i = int(e.argmax())
        print(f"  {k:12s} rel {rel_to_max(g[k], g_o[k]):.2e} worst id {i} err {e[i]:.3e} ref {np.abs(g_o[k]).reshape(P,-1)[i].max():.3e} max|ref| {np.abs(g_o[k]).max():.3e}"
              f" | z_cam {st_o['depths'][i]:.3f} radius {st_o['radii'][i]} scale {sc['scales'][i]} op {sc['opacities'][i,0]:.3f}")
