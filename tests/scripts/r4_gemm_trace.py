"""Round 4: where a GEMM tile's time goes (s_memtime stamps of workgroup 0, EVERY wave), for a trace build of the library given as
argv[1] (-DGVD_GEMM_TRACE [-DGVD_GEMM_DBG=16: no global stores]).  Stamps: 0 top, 1 init done, 2 first barrier, 3 second barrier,
4 loop end, 5 post barrier, 6 next issued, 7 epilogue done; printed relative to wave 0's stamp 0 of the tile."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GVD_DIFFUSION_LIB"] = os.path.join(ROOT, "guidedvd-3dgs_amd", "lib", sys.argv[1])
os.environ["GVD_GEMM_VARIANT"] = os.environ.get("GVD_GEMM_VARIANT", "1")
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd import gemm, ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
print("lib", sys.argv[1])
for (M, N, K) in [(230400, 2560, 320), (230400, 320, 320), (57600, 5120, 640), (14400, 10240, 1280)]:
    x = torch.randn(M, K, device=dev, generator=g).half()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).half()
    b = torch.randn(N, device=dev, generator=g)
    for _ in range(3):
        gemm.gemm_nt(x, w, bias=b)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 1024)()
    ops.lib().gvd_gemm_trace_read(buf, 1024)
    t = list(buf)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        gemm.gemm_nt(x, w, bias=b)
    e.record()
    torch.cuda.synchronize()
    print("shape", (M, N, K), f"{a.elapsed_time(e) / 20 * 1e3:.1f} us/launch")
    nw = 8 if t[128] else 4
    for tile in (2, 3):
        base = t[tile * 8]
        for w_ in range(nw):
            r = t[w_ * 128 + tile * 8: w_ * 128 + tile * 8 + 9]
            print(f"  tile {tile} wave {w_}", [r[j] - base for j in range(8)], "next top +", r[8] - base)
