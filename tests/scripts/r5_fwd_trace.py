"""Timeline of k_render_fwd's workgroups (round 5 experiment): needs the library built with -DGVD_RFWD_TRACE
(GVD_RASTER_LIB=.../libgvd_raster_trace.so).  One C2 view: residency over time, the share of the cooperative sort in a workgroup's
life, how uneven the four quadrant waves' blend phases are."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synthetic as syn
from diff_gaussian_rasterization import _C

dev = "cuda:0"
P, W, H, D = 200000, 640, 480, 3
sc = syn.scene_c2(P=P, W=W, H=H, sh_degree=D)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
c = sc["cameras"][int(sys.argv[1]) if len(sys.argv) > 1 else 0]
args = (t(sc["bg"]), t(sc["means3D"]), torch.empty(0, device=dev), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0,
        torch.empty(0, device=dev), t(c["viewmatrix"]), t(c["projmatrix"]), c["tanfovx"], c["tanfovy"], H, W, t(sc["shs"]), D,
        t(c["campos"]), False, False)
L = _C.lib()
for i in range(4):
    torch.cuda.synchronize()
    L.gvd_debug_ftrace_clear()
    _C.rasterize_gaussians(*args)
torch.cuda.synchronize()
T = ((W + 15) // 16) * ((H + 15) // 16)
buf = (ctypes.c_ulonglong * (T * 8))()
assert L.gvd_debug_ftrace_read(buf, ctypes.c_size_t(T * 8)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(T, 8).astype(np.int64)
live = (a[:, 0] > 0) & (a[:, 1] > 0)
base = a[live, 0].min()
st, en = (a[:, 0] - base) * 0.01, (a[:, 1] - base) * 0.01
dur = en - st
n = a[:, 3]
CLK = 2390.0
sort_us, blend = a[:, 2] / CLK, a[:, 4:8] / CLK
print(f"kernel span {en[live].max():.1f} us; workgroups {live.sum()}")
print("start quantiles:", np.round(np.quantile(st[live], [0, .25, .5, .75, .9, 1]), 1), " duration quantiles:", np.round(np.quantile(dur[live], [0, .25, .5, .75, .9, 1]), 1))
grid = np.arange(0, en[live].max(), 4.0)
print("resident workgroups every 4 us:", [int(((st <= g) & (en > g) & live).sum()) for g in grid])
print(f"sort share of workgroup life {sort_us[live].sum() / dur[live].sum():.3f}; slowest wave's blend share {blend[live].max(axis=1).sum() / dur[live].sum():.3f}; "
      f"mean wave blend / slowest wave blend {blend[live].mean(axis=1).sum() / blend[live].max(axis=1).sum():.3f}")
for lo, hi in ((1, 128), (128, 256), (256, 512), (512, 768), (768, 4000)):
    mk = live & (n >= lo) & (n < hi)
    if mk.sum():
        print(f"  list in [{lo},{hi}): {mk.sum()} wgs, mean dur {dur[mk].mean():.1f} us = sort {sort_us[mk].mean():.1f} + slowest blend {blend[mk].max(axis=1).mean():.1f} (mean wave {blend[mk].mean():.1f})")
last = np.argsort(-en)[:6]
print("last to finish (block, list, start, dur):", [(int(b), int(n[b]), round(float(st[b]), 1), round(float(dur[b]), 1)) for b in last])
