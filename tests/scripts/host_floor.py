"""Host-side floor of one raster training iteration: sync-free forward + backward issued as fast as the host can
(the GPU queue just grows), so wall time per step == host time per step.  Dev tool."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
sc = syn.scene_c2(P=20000)   # small scene: the GPU is never the limiter
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
P = sc["means3D"].shape[0]
prm = [t(sc[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")]
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
cam = sc["cameras"][0]
s = GaussianRasterizationSettings(image_height=480, image_width=640, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(sc["bg"]),
                                  scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
                                  campos=t(cam["campos"]), prefiltered=False, debug=False, confidence=torch.ones((P, 1), device=dev))
gC = torch.randn(3, 480, 640, device=dev)
_C.set_instance_capacity(400000)
def step():
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=prm[0], means2D=m2, opacities=prm[1], shs=prm[4], scales=prm[2], rotations=prm[3])
    for p_ in prm + [m2]: p_.grad = None
    torch.autograd.backward([color], [gC])
for _ in range(50): step()
torch.cuda.synchronize()
n = 400; t0 = time.perf_counter()
for _ in range(n): step()
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
print("host time per step us", host * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
