"""Host time of one raster training iteration (GaussianRasterizer forward + autograd backward) on a scene so small that the GPU is never
the limiter (P = 2000, 128 x 128: ~60 us of kernels): what the Python / binding path costs per iteration.  Run with and without
GVD_RASTER_NO_EXT=1 to compare the compiled operator (lib/_gvd_raster_torch.so) with the ctypes + Python autograd.Function path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
sc = syn.scene_c2(P=2000)
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
P = sc["means3D"].shape[0]
prm = [t(sc[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")]
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
cam = syn.make_camera(syn.look_at((0.0, 0.0, 0.0), (0.0, 0.0, 1.0)), 1.0, 1.0, 128, 128)
s = GaussianRasterizationSettings(image_height=128, image_width=128, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(sc["bg"]),
                                  scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
                                  campos=t(cam["campos"]), prefiltered=False, debug=False, confidence=torch.ones((P, 1), device=dev))
gC = torch.randn(3, 128, 128, device=dev)
def step():
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=prm[0], means2D=m2, opacities=prm[1], shs=prm[4], scales=prm[2], rotations=prm[3])
    for p_ in prm + [m2]: p_.grad = None
    torch.autograd.backward([color], [gC])
for _ in range(100): step()
torch.cuda.synchronize()
res = []
for rep in range(5):
    n = 400; t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / n * 1e6)
print(f"{'compiled operator' if _C.ext() is not None else 'ctypes + Python autograd.Function'}: iteration {min(res):.1f} us (best of 5 x 400; all: {[round(r, 1) for r in res]})")
