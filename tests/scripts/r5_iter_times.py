"""Per-iteration wall times of the raster loop right after a synchronise (what the driver's 20-step region sees).  Dev tool."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
sc = syn.scene_c2()
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
m3, op, scl, rot, sh = t(sc["means3D"], True), t(sc["opacities"], True), t(sc["scales"], True), t(sc["rotations"], True), t(sc["shs"], True)
m2 = torch.zeros((200000, 3), device=dev, requires_grad=True)
conf = torch.ones((200000, 1), device=dev); bg = t(sc["bg"])
cams = [GaussianRasterizationSettings(image_height=480, image_width=640, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), sh_degree=3, campos=t(c["campos"]), prefiltered=False, debug=False, confidence=conf) for c in sc["cameras"]]
gC = torch.randn(3, 480, 640, device=dev) / (480 * 640)
params = [m3, op, scl, rot, sh, m2]
def step(i):
    color, radii, depth, alpha = GaussianRasterizer(cams[i % 6])(means3D=m3, means2D=m2, opacities=op, shs=sh, scales=scl, rotations=rot)
    for p in params: p.grad = None
    torch.autograd.backward([color], [gC])
for i in range(12): step(i)
for trial in range(4):
    if trial == 2: gc.disable()
    for i in range(5): step(i)
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for i in range(20):
        step(5 + i); ts.append(time.perf_counter())
    torch.cuda.synchronize(); te = time.perf_counter()
    d = np.diff(ts) * 1e6
    print(f"trial {trial} (gc {'off' if trial >= 2 else 'on'}): total {1e6 * (te - ts[0]) / 20:.1f} us/step; host per iteration:", np.round(d).astype(int).tolist(), f"final sync {1e6 * (te - ts[-1]):.0f}")
