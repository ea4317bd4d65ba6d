import sys, os
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT,"guidedvd-3dgs_amd"), os.path.join(ROOT,"tests")): sys.path.insert(0,p)
import numpy as np, synthetic as syn
from raster_compare import run_oracle, run_hip, rel_to_max
sc = syn.scene_c2(); cam = sc["cameras"][2]; H,W=480,640
rng=np.random.default_rng(3)
g=(rng.normal(size=(3,H,W))/(H*W), rng.normal(size=(H,W))/(H*W), rng.normal(size=(H,W))/(H*W))
so, go = run_oracle(sc, cam, g)
sh, gh = run_hip(sc, cam, g, alpha_override=so["alpha"])
for k in sorted(set(go) & set(gh)):
    print(k, rel_to_max(gh[k], go[k]))
