"""Host time of the raster operator's Python path, by section (round 5 dev tool): the native calls are wrapped with perf_counter so that
what is Python (argument checks, allocations, autograd bookkeeping) separates from what is the C-ABI call (launches + the ticket wait)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
sc = syn.scene_c2(); c = sc["cameras"][0]
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
m3, op, scl, rot, sh = t(sc["means3D"], True), t(sc["opacities"], True), t(sc["scales"], True), t(sc["rotations"], True), t(sc["shs"], True)
m2 = torch.zeros((200000, 3), device=dev, requires_grad=True)
s = GaussianRasterizationSettings(image_height=480, image_width=640, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=t(sc["bg"]),
                                  scale_modifier=1.0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), sh_degree=3,
                                  campos=t(c["campos"]), prefiltered=False, debug=False, confidence=torch.ones((200000, 1), device=dev))
gC = torch.randn(3, 480, 640, device=dev) / (480 * 640)
L = _C.lib()
T = {k: [] for k in ("fwd_total", "fwd_native", "bwd_total", "bwd_native", "iter")}
nf, nb = L.gvd_raster_forward, L.gvd_raster_backward_conf
class W:
    def __init__(self, fn, key): self.fn, self.key = fn, key
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); T[self.key].append(time.perf_counter() - t0); return r
L.gvd_raster_forward, L.gvd_raster_backward_conf = W(nf, "fwd_native"), W(nb, "bwd_native")
f0, b0 = _C.rasterize_gaussians, _C.rasterize_gaussians_backward
def fwd(*a, **k):
    t0 = time.perf_counter(); r = f0(*a, **k); T["fwd_total"].append(time.perf_counter() - t0); return r
def bwd(*a, **k):
    t0 = time.perf_counter(); r = b0(*a, **k); T["bwd_total"].append(time.perf_counter() - t0); return r
_C.rasterize_gaussians, _C.rasterize_gaussians_backward = fwd, bwd
params = [m3, op, scl, rot, sh, m2]
def step():
    t0 = time.perf_counter()
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=m3, means2D=m2, opacities=op, shs=sh, scales=scl, rotations=rot)
    for p in params: p.grad = None
    torch.autograd.backward([color], [gC])
    T["iter"].append(time.perf_counter() - t0)
for _ in range(50): step()
torch.cuda.synchronize()
for k in T: T[k].clear()
for _ in range(400): step()
torch.cuda.synchronize()
med = lambda v: float(np.median(v)) * 1e6
print({k: round(med(v), 1) for k, v in T.items()})
print(f"python around the forward's native call {med(T['fwd_total']) - med(T['fwd_native']):.1f} us; around the backward's {med(T['bwd_total']) - med(T['bwd_native']):.1f} us; "
      f"rest of the iteration (autograd, module call, grad resets) {med(T['iter']) - med(T['fwd_total']) - med(T['bwd_total']):.1f} us")
