"""Timeline of k_render_bwd's workgroups (round 5 experiment): needs the library built with -DGVD_RBWD_TRACE
(GVD_RASTER_LIB=.../libgvd_raster_trace.so).  One C2 view, colour gradient only; prints residency over time, per-XCD finish
times, how long waves sit outside the walk (barriers + staging) and how uneven the four quadrant walks of a tile are."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C

dev = torch.device("cuda:0")
P, W, H, D = 200000, 640, 480, 3
sc = syn.scene_c2(P=P, W=W, H=H, sh_degree=D)
t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
means3D, opac = t(sc["means3D"], True), t(sc["opacities"], True)
scales, rots, shs = t(sc["scales"], True), t(sc["rotations"], True), t(sc["shs"], True)
means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
c = sc["cameras"][int(sys.argv[1]) if len(sys.argv) > 1 else 0]
s = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=t(sc["bg"]),
                                  scale_modifier=1.0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), sh_degree=D,
                                  campos=t(c["campos"]), prefiltered=False, debug=False, confidence=torch.ones((P, 1), device=dev))
gC = torch.randn((3, H, W), device=dev) / (H * W)
for it in range(4):
    color, radii, depth, alpha = GaussianRasterizer(s)(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots)
    torch.autograd.backward([color], [gC])
torch.cuda.synchronize()
L = _C.lib()
T = ((((W + 15) // 16) * ((H + 15) // 16) + 7) // 8) * 32 + (((((W + 15) // 16) * ((H + 15) // 16) + 3) // 4 + 7) // 8 * 8) * 12   # extra segment units + (tile, quadrant) units
buf = (ctypes.c_ulonglong * (T * 8))()
rc = L.gvd_debug_rtrace_read(buf, ctypes.c_size_t(T * 8))
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(T, 8).astype(np.int64)
t0, t1 = a[:, 0], a[:, 1]
live = (t1 > 0) & (t0 > 0)
base = t0.min()
st, en = (t0 - base) * 0.01, (t1 - base) * 0.01     # us (100 MHz)
xcc = (a[:, 2] >> 32) & 0xf
hw = a[:, 2] & 0xffffffff
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7   # gfx9 HW_ID: [11:8] cu, [12] sh, [15:13] se
tile_max, nlist = a[:, 3] & 0xffffffff, a[:, 3] >> 32
ph = a[:, 4:8] / 2180.0   # wave 0's phases in us: prologue, staging sum, walk sum, tail sum (s_memtime = 100 MHz constant clock here)
dur = en - st
print(f"kernel span {en[live].max():.1f} us; workgroups {live.sum()} (+{(~live).sum()} empty)")
print("start-time quantiles (us):", np.round(np.quantile(st[live], [0, .25, .5, .64, .75, .9, 1]), 1))
print("duration quantiles (us):  ", np.round(np.quantile(dur[live], [0, .25, .5, .75, .9, 1]), 1))
for x in range(8):
    mk = live & (xcc == x)
    print(f"  xcc {x}: {mk.sum()} wgs, first start {st[mk].min():.1f}, last end {en[mk].max():.1f}, sum(dur) {dur[mk].sum():.0f} us, sum(tile_max) {tile_max[mk].sum()}")
grid = np.arange(0, en[live].max(), 5.0)
res = [(int(((st <= g) & (en > g) & live).sum())) for g in grid]
print("resident workgroups every 5 us:", res)
tot_ph = ph.sum(axis=1)
print("wave-0 phase shares of workgroup life (prologue, staging, walk, tail):", np.round(ph[live].sum(axis=0) / dur[live].sum(), 3), " (s_memtime unit check: sum phases / dur =", round(float(tot_ph[live].sum() / dur[live].sum()), 3), ")")
for lo, hi in ((1, 64), (64, 128), (128, 256), (256, 512), (512, 768), (768, 2000)):
    mk = live & (tile_max >= lo) & (tile_max < hi)
    if mk.sum():
        print(f"  tile_max in [{lo},{hi}): {mk.sum()} wgs, mean dur {dur[mk].mean():.1f} us = prologue {ph[mk,0].mean():.1f} + staging {ph[mk,1].mean():.1f} + walk {ph[mk,2].mean():.1f} + tail {ph[mk,3].mean():.1f}; walk per entry {1e3*ph[mk,2].sum()/tile_max[mk].sum():.0f} ns")
blocks = np.arange(T)
order = np.argsort(st)
print("first 10 by start:", [(int(b), int(tile_max[b]), round(float(dur[b]), 1)) for b in order[:10]])
lastend = np.argsort(-en)
print("last 10 to finish (block, tile_max, list, start, dur):", [(int(b), int(tile_max[b]), int(nlist[b]), round(float(st[b]), 1), round(float(dur[b]), 1)) for b in lastend[:10]])
print("corr(dur, tile_max) =", np.corrcoef(dur[live], tile_max[live])[0, 1])
np.save(os.path.join(ROOT, "gpurun_out", "r5_bwd_trace.npy"), a)
