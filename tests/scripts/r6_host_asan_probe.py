"""Host entry points of libgvd_raster that need no device, over extreme arguments -- run under the host address sanitizer
(profiles/r06_host_asan.log): LD_PRELOAD=<libclang_rt.asan-x86_64.so> GVD_RASTER_LIB=<asan build> python tests/scripts/r6_host_asan_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "guidedvd-3dgs_amd"))
from diff_gaussian_rasterization import _C
L = _C.lib()
maps = open("/proc/self/maps").read()
print("asan runtime mapped:", "libclang_rt.asan" in maps, "| library:", [l.split()[-1] for l in maps.splitlines() if "gvd_raster" in l][:1])
# host entry points that take no device: sizes, capacity decode (both forms), argument validation, error strings
for r in (0, 1, 2, 63, 64, 65, 4096, 436438, 10_000_000, 0xfffffff0):
    b, c = L.gvd_raster_binning_bytes(r), L.gvd_raster_binning_bytes_no_backward(r)
    assert L.gvd_raster_binning_capacity(b) == max(r, 1) and L.gvd_raster_binning_capacity(c) == max(r, 1)
    assert L.gvd_raster_binning_capacity(b + 1) == 0xffffffff
lay = _C._ChunkLayout()
for (P, W, H, R) in ((0, 16, 16, 0), (1, 1, 1, 1), (200000, 640, 480, 436438), (5_000_000, 3840, 2160, 40_000_000)):
    L.gvd_raster_chunk_layout(P, W, H, R, ctypes.byref(lay))
    assert L.gvd_raster_geometry_bytes(P, W, H) > 0 and L.gvd_raster_image_bytes(W, H) > 0
for f in (0, 1):
    L.gvd_raster_expect_backward(f); L.gvd_raster_set_speculation(f); L.gvd_raster_set_backward_split(512 * f)
print("host entry points exercised: ok;", L.gvd_version().decode())
