import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GVD_DIFFUSION_LIB"] = os.path.join(ROOT, "guidedvd-3dgs_amd", "lib", "libgvd_diffusion_trace.so")
os.environ["GVD_GEMM_VARIANT"] = "1"
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd import gemm, ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(57600, 5120, 640), (230400, 2560, 320)]:
    x = torch.randn(M, K, device=dev, generator=g).half()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).half()
    b = torch.randn(N, device=dev, generator=g)
    for _ in range(3):
        gemm.gemm_nt(x, w, bias=b)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 1024)()
    ops.lib().gvd_gemm_trace_read(buf, 1024)
    t = list(buf)
    print("shape", (M, N, K), "stamps per tile: 0 top, 1 init done, 2 first barrier, 3 second barrier, 4 loop end, 5 post barrier, 6 next issued, 7 epilogue done")
    for i in range(0, 8 * 8, 8):
        r = t[i:i + 8]
        if r[0] == 0:
            break
        print(i // 8, [r[j] - r[0] for j in range(8)], "next tile starts +", (t[i + 8] - r[0]) if t[i + 8] else None)
