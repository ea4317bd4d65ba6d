"""Raster operator stress (round 5): changing point counts (densification-like), image sizes from 17 x 17 to 2064 x 1616 (more than 12288
tiles: the global-histogram path with the separate tile scan), SH degrees, mixed no-grad renders, every configuration rendered twice --
images and gradients must be bit-identical between the two runs and finite.  Exercises the speculation hints (16-entry table), the parked
gradient arrays, the walk cut, the fused two-role tile scan with tile counts that are not multiples of 8."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)   # (the -m gpu test runs seed 7; other seeds by hand)
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=rg)

def scene(P):
    xyz = rng.normal(size=(P, 3)) * np.array([2.0, 1.5, 2.0]); xyz[:, 2] = np.abs(xyz[:, 2]) + 0.5
    scales = np.exp(rng.normal(math.log(0.03), 0.5, size=(P, 3)))
    q = rng.normal(size=(P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    op = 1 / (1 + np.exp(-rng.normal(0, 2, size=(P, 1))))
    sh = rng.normal(0, 0.3, size=(P, 16, 3)); sh[:, 0] = syn.rgb2sh(rng.uniform(0, 1, size=(P, 3)))
    return xyz, scales, q, op, sh

def run(P, W, H, deg, sc, cam, grad, gC):
    xyz, scales, q, op, sh = sc
    lv = dict(means3D=t(xyz, grad), opacities=t(op, grad), scales=t(scales, grad), rotations=t(q, grad), shs=t(sh, grad),
              means2D=torch.zeros(P, 3, device=dev, requires_grad=grad))
    s = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], t([0.1, 0.2, 0.3]), 1.0, t(cam["viewmatrix"]), t(cam["projmatrix"]),
                                      deg, t(cam["campos"]), False, bool(os.environ.get("GVD_STRESS_DEBUG")), torch.ones(P, 1, device=dev))
    if not grad:
        with torch.no_grad():
            c, r, d, a = GaussianRasterizer(s)(**lv)
        return c, None
    c, r, d, a = GaussianRasterizer(s)(**lv)
    torch.autograd.backward([c, d], [gC, gC[:1] * 0.1])
    return c.detach(), {k: v.grad for k, v in lv.items()}

n = 0
sizes = [(17, 17), (64, 48), (200, 120), (333, 257), (640, 480), (1000, 31), (2064, 1616)]
for it in range(60):
    P = int(rng.choice([1, 50, 3000, 20000, 60000]))
    W, H = sizes[int(rng.integers(0, len(sizes)))] if it % 10 else sizes[-1]
    deg = int(rng.integers(0, 4))
    sc = scene(P)
    eye = rng.normal(size=3) * 0.3 - np.array([0, 0, 3.0])
    cam = syn.make_camera(syn.look_at(tuple(eye), (0.0, 0.0, 1.0)), math.radians(rng.uniform(40, 90)), math.radians(rng.uniform(35, 80)), W, H)
    gC = torch.randn(3, H, W, device=dev) / (H * W)
    outs = []
    print(f"it {it}: P {P} {W}x{H} deg {deg}", flush=True)
    skip = it < int(os.environ.get("GVD_STRESS_FROM", "0"))   # (replaying a seed from iteration k on: same random draws, no renders before k)
    if os.environ.get("GVD_STRESS_ONLY"):
        skip = str(it) not in os.environ["GVD_STRESS_ONLY"].split(",")
    for rep in range(2):
        if rng.random() < 0.3 and not skip:
            run(P, W, H, deg, sc, cam, False, gC)            # a no-grad render in between (skips the backward preparation)
        if not skip:
            outs.append(run(P, W, H, deg, sc, cam, True, gC))
    if skip:
        continue
    (c0, g0), (c1, g1) = outs
    assert torch.isfinite(c0).all() and torch.equal(c0, c1), (it, P, W, H)
    for k in g0:
        assert torch.isfinite(g0[k]).all() and torch.equal(g0[k], g1[k]), (it, P, W, H, k)
    n += 1
torch.cuda.synchronize()
print(f"raster stress: {n} configurations x 2 runs, images and gradients finite and bit-identical between runs")

# ---- training-like section: ONE scene and image size, a different camera every iteration (what the speculation hints see in a training
#      run): distances from inside the cloud to far away make num_rendered and the longest tile list jump by orders of magnitude, so the
#      speculative stage 2 mis-guesses capacity and sort class again and again.  Two passes over the same camera sequence: the hint
#      state differs between them (exact path / speculative / re-run), the results must not ----
P, W, H = 30000, 320, 240
sc = scene(P)
crng = np.random.default_rng(int(rng.integers(1 << 30)))
cams = []
for i in range(100):
    dist = float(np.exp(crng.uniform(np.log(0.05), np.log(12.0))))
    eye = crng.normal(size=3); eye = eye / np.linalg.norm(eye) * dist + np.array([0.0, 0.0, 1.5])
    cams.append(syn.make_camera(syn.look_at(tuple(eye), (0.0, 0.0, 1.5)), math.radians(crng.uniform(25, 100)), math.radians(crng.uniform(20, 90)), W, H))
gC = torch.randn(3, H, W, device=dev) / (H * W)
passes = []
for pas in range(2):
    res = []
    for i, cam in enumerate(cams):
        if (i + pas) % 7 == 0:
            run(P, W, H, 3, sc, cam, False, gC)
        c, g_ = run(P, W, H, 3, sc, cam, True, gC)
        res.append((c, g_))
    passes.append(res)
for i, ((c0, g0), (c1, g1)) in enumerate(zip(*passes)):
    assert torch.isfinite(c0).all() and torch.equal(c0, c1), ("camera", i)
    for k in g0:
        assert torch.isfinite(g0[k]).all() and torch.equal(g0[k], g1[k]), ("camera", i, k)
torch.cuda.synchronize()
print("training-like: 100 cameras x 2 passes over one scene, bit-identical between passes")
del passes

# ---- heavy section: sizes past the bench's (each twice, bit-identical; on a side stream; strided inputs) ----
def pile(P, spread):          # every Gaussian in front of the camera inside a small disc: one or a few very long tile lists
    xyz = rng.normal(size=(P, 3)) * np.array([spread, spread, 0.3]); xyz[:, 2] = np.abs(xyz[:, 2]) + 2.0
    scales = np.exp(rng.normal(math.log(0.01), 0.3, size=(P, 3)))
    q = rng.normal(size=(P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    op = 1 / (1 + np.exp(-rng.normal(-3, 1, size=(P, 1))))     # faint: the walks do not saturate early
    sh = rng.normal(0, 0.3, size=(P, 16, 3)); sh[:, 0] = syn.rgb2sh(rng.uniform(0, 1, size=(P, 3)))
    return xyz, scales, q, op, sh

def blanket(P):               # huge splats: every Gaussian touches (nearly) every tile
    xyz, scales, q, op, sh = scene(P)
    return xyz, scales * 60.0, q, op * 0.05, sh

heavy = [("2 M Gaussians, 1600 x 1200", 2_000_000, 1600, 1200, scene),
         ("5 M Gaussians, 640 x 480", 5_000_000, 640, 480, scene),
         ("60 k Gaussians piled on a few tiles (lists > 16384: global-memory sort)", 60_000, 320, 240, lambda P: pile(P, 0.02)),
         ("200 k piled on ~60 tiles", 200_000, 640, 480, lambda P: pile(P, 0.25)),
         ("400 blanket splats, 1920 x 1080", 400, 1920, 1080, blanket),
         ("100 k Gaussians, 3840 x 2160 (32 400 tiles)", 100_000, 3840, 2160, scene)]
side = torch.cuda.Stream()
for name, P, W, H, make in heavy:
    sc = make(P)
    cam = syn.make_camera(syn.look_at((0.0, 0.0, -3.0), (0.0, 0.0, 1.0)), math.radians(70), math.radians(55), W, H)
    gC = torch.randn(3, H, W, device=dev) / (H * W)
    torch.cuda.synchronize()
    outs = []
    for rep in range(2):
        if rep:
            with torch.cuda.stream(side):      # second run on a side stream
                outs.append(run(P, W, H, 3, sc, cam, True, gC))
            side.synchronize()
        else:
            outs.append(run(P, W, H, 3, sc, cam, True, gC))
    (c0, g0), (c1, g1) = outs
    assert torch.isfinite(c0).all() and torch.equal(c0, c1), name
    for k in g0:
        assert torch.isfinite(g0[k]).all() and torch.equal(g0[k], g1[k]), (name, k)
    print(f"heavy: {name}: ok (|image| max {float(c0.abs().max()):.3f}, nonzero mean-gradient rows {int((g0['means3D'].abs().sum(1) > 0).sum())})", flush=True)
    del outs, c0, g0, c1, g1, sc
    torch.cuda.empty_cache()

# strided / non-contiguous inputs: the operator must give the same result as with packed copies
P, W, H = 5000, 320, 200
xyz, scales, q, op, sh = scene(P)
cam = syn.make_camera(syn.look_at((0.0, 0.0, -3.0), (0.0, 0.0, 1.0)), math.radians(70), math.radians(55), W, H)
st = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], t([0.1, 0.2, 0.3]), 1.0, t(cam["viewmatrix"]), t(cam["projmatrix"]),
                                   3, t(cam["campos"]), False, False, torch.ones(P, 1, device=dev))
big = torch.zeros(P, 8, device=dev); big[:, 1:4] = t(xyz)
shT = t(sh.transpose(1, 0, 2)).transpose(0, 1)            # [P, 16, 3] view of a [16, P, 3] tensor
assert not big[:, 1:4].is_contiguous() and not shT.is_contiguous()
a = GaussianRasterizer(st)(means3D=big[:, 1:4], means2D=torch.zeros(P, 3, device=dev), opacities=t(op), shs=shT, scales=t(scales), rotations=t(q))
b = GaussianRasterizer(st)(means3D=t(xyz), means2D=torch.zeros(P, 3, device=dev), opacities=t(op), shs=t(sh), scales=t(scales), rotations=t(q))
assert all(torch.equal(x, y) for x, y in zip(a, b)), "strided inputs change the result"
print("strided inputs: ok")
