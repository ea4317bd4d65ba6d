"""Small driver for rocprofv3 runs: N forward+backward iterations of the C2 workload through the
native boundary (no bench bookkeeping).  Usage: python tests/profile_raster.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import synthetic as syn
from raster_compare import run_hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc = syn.scene_c2()
for i in range(n):
    cam = sc["cameras"][i % 6]
    H, W = cam["image_height"], cam["image_width"]
    rng = np.random.default_rng(i)
    run_hip(sc, cam, (rng.normal(size=(3, H, W)) / (H * W), np.zeros((H, W)), np.zeros((H, W))))
