"""Soak of the compiled raster operator: 30 000 forward + backward iterations over six cameras with a densification-like change of the point count
every 2 000 iterations; device memory (torch allocator) and host RSS must be flat between the first and the last third (a leak of a saved tensor,
a parked gradient array or an autograd node would show as growth)."""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
sc = syn.scene_c2(P=60000)
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
def params(n):
    return [t(sc[k][:n], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")] + [torch.zeros((n, 3), device=dev, requires_grad=True)]
def settings(c, n):
    return GaussianRasterizationSettings(image_height=480, image_width=640, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=t(sc["bg"]), scale_modifier=1.0,
                                         viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), sh_degree=3, campos=t(c["campos"]), prefiltered=False,
                                         debug=False, confidence=torch.ones((n, 1), device=dev))
gC = torch.randn(3, 480, 640, device=dev) / (480 * 640)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
marks = []
n = 40000
prm, cams = params(n), [settings(c, n) for c in sc["cameras"]]
t0 = time.time()
for it in range(N):
    if it and it % 2000 == 0:
        n = 40000 + ((it // 2000) % 5) * 5000
        prm, cams = params(n), [settings(c, n) for c in sc["cameras"]]
    m3, op, scl, rot, sh, m2 = prm
    color, radii, depth, alpha = GaussianRasterizer(cams[it % 6])(means3D=m3, means2D=m2, opacities=op, shs=sh, scales=scl, rotations=rot)
    for p in prm: p.grad = None
    torch.autograd.backward([color], [gC])
    if it % (N // 6) == N // 6 - 1:
        torch.cuda.synchronize()
        marks.append((it, torch.cuda.memory_allocated(dev) / 1e6, torch.cuda.memory_reserved(dev) / 1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3))
torch.cuda.synchronize()
for m in marks: print("iteration %6d: device allocated %.1f MB, reserved %.1f MB, host max RSS %.1f MB" % m)
print(f"{N} iterations in {time.time() - t0:.1f} s ({'compiled operator' if _C.ext() is not None else 'python operator'})")
assert marks[-1][1] <= marks[1][1] * 1.02 + 1 and marks[-1][3] <= marks[1][3] * 1.02 + 8, "memory grows"
print("raster soak ok")
