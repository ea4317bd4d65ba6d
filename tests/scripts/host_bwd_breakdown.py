"""Fine-grained host timing of the backward wrapper (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import _C
dev = torch.device("cuda:0")
sc = syn.scene_c2(); cam = sc["cameras"][0]
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
m3, op, scl, rot, sh = t(sc["means3D"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), t(sc["shs"])
vm, pm, cp, bg = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), t(sc["bg"])
E = torch.Tensor([])
out = _C.rasterize_gaussians(bg, m3, E, op, scl, rot, 1.0, E, vm, pm, cam["tanfovx"], cam["tanfovy"], 480, 640, sh, 3, cp, False, False)
R, color, depth, alpha, radii, gb, bb, ib = out
gC = torch.randn(3, 480, 640, device=dev)
conf = torch.ones(200000, 1, device=dev)
def call():
    return _C.rasterize_gaussians_backward(bg, m3, radii, E, scl, rot, 1.0, E, vm, pm, cam["tanfovx"], cam["tanfovy"], gC, None, None,
                                           sh, 3, cp, gb, R, bb, ib, alpha, False, confidence=conf)
for _ in range(20): call()
torch.cuda.synchronize()
n = 300; t0 = time.perf_counter()
for _ in range(n):
    call()
    if _ % 20 == 19: torch.cuda.synchronize()
print("wrapper total us/call (GPU kept shallow)", (time.perf_counter() - t0) / n * 1e6)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    call()
    if _ % 20 == 19: torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(10)
