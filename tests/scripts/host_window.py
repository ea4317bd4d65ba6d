"""Host time between the forward's return (after its one sync) and the end of the backward wrapper (its launches): the
window the GPU covers with the speculatively queued stage 2 (scatter + sort + blend, ~100 us).  Dev tool."""
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
T = {"fwd_in": [], "fwd": [], "gap": [], "bwd": []}
f0, b0 = _C.rasterize_gaussians, _C.rasterize_gaussians_backward
state = {}
def fwd(*a, **k):
    t0 = time.perf_counter(); r = f0(*a, **k); t1 = time.perf_counter()
    T["fwd"].append(t1 - t0); state["ret"] = t1; return r
def bwd(*a, **k):
    t0 = time.perf_counter(); T["gap"].append(t0 - state["ret"]); r = b0(*a, **k); T["bwd"].append(time.perf_counter() - t0); return r
_C.rasterize_gaussians, _C.rasterize_gaussians_backward = fwd, bwd
params = [m3, op, scl, rot, sh, m2]
def step():
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=m3, means2D=m2, opacities=op, shs=sh, scales=scl, rotations=rot)
    for p in params: p.grad = None
    torch.autograd.backward([color], [gC])
for _ in range(30): step()
torch.cuda.synchronize()
for k in T: T[k].clear()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 300 * 1e6
med = lambda v: float(np.median(v)) * 1e6
print(f"iteration {tot:.1f} us | native forward call (incl. wait) {med(T['fwd']):.1f} | forward return -> backward wrapper entry {med(T['gap']):.1f} | backward wrapper {med(T['bwd']):.1f}")
