"""Host-side time of the raster operator calls (C2 workload): forward / backward wall time with the GPU drained before
each call, minus the kernels' own time -> what the host adds around the forward's read-back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import cProfile, pstats
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
sc = syn.scene_c2()
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
P = sc["means3D"].shape[0]
prm = [t(sc[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")]
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
cam = sc["cameras"][0]
s = GaussianRasterizationSettings(image_height=480, image_width=640, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(sc["bg"]),
                                  scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
                                  campos=t(cam["campos"]), prefiltered=False, debug=False, confidence=torch.ones((P, 1), device=dev))
gC = torch.randn(3, 480, 640, device=dev)
def step():
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=prm[0], means2D=m2, opacities=prm[1], shs=prm[4], scales=prm[2], rotations=prm[3])
    for p_ in prm + [m2]: p_.grad = None
    torch.autograd.backward([color], [gC])
for _ in range(30): step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print("step wall us", (time.perf_counter() - t0) / n * 1e6)
fw = bw = 0.0
for _ in range(100):
    torch.cuda.synchronize(); a = time.perf_counter()
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=prm[0], means2D=m2, opacities=prm[1], shs=prm[4], scales=prm[2], rotations=prm[3])
    b = time.perf_counter()
    for p_ in prm + [m2]: p_.grad = None
    torch.cuda.synchronize(); c = time.perf_counter()
    torch.autograd.backward([color], [gC])
    d = time.perf_counter(); torch.cuda.synchronize(); e = time.perf_counter()
    fw += b - a; bw += d - c
print("forward call (incl. its sync) us", fw / 100 * 1e6, " backward call (launch only) us", bw / 100 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)

# ---- where the backward's host time goes: wrap the C entry point ----
L = _C.lib()
orig = L.gvd_raster_backward_conf
acc = {"c": 0.0, "n": 0}
class W:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); acc["c"] += time.perf_counter() - t0; acc["n"] += 1; return r
_C._lib.gvd_raster_backward_conf = W() if hasattr(_C, "_lib") else None
tot = 0.0
for _ in range(100):
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=prm[0], means2D=m2, opacities=prm[1], shs=prm[4], scales=prm[2], rotations=prm[3])
    for p_ in prm + [m2]: p_.grad = None
    torch.cuda.synchronize(); c = time.perf_counter()
    torch.autograd.backward([color], [gC])
    tot += time.perf_counter() - c
print("backward python+autograd total us", tot / 100 * 1e6, " inside C call us", acc["c"] / max(acc["n"], 1) * 1e6, "calls", acc["n"])
