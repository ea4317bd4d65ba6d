"""Run a test script under the red-zone guard allocator (tests/guard/guard_allocator.cpp; verdict r5 item 6):

    python tests/scripts/r6_guard_run.py tests/scripts/r5_raster_stress.py [args ...]

Every torch device allocation of the script -- the rasterizer's chunks, outputs, gradient arrays, every workspace of the diffusion kernels -- becomes
a hipMalloc of its own with poisoned red zones and a poisoned body; writes past either end are reported when the block is freed and at exit, reads
past either end return poison (the script's own bit-equality / tolerance checks then fail) or fault.  Exit code: the script's, or 97 when the script
passed but a red zone was written.  Not for scripts that capture hipGraphs or read allocator statistics (the two soaks): a pluggable allocator
has neither private pools nor memory_allocated().  `--selftest` instead of a script: allocate, scribble past both ends through the allocator's
test hook, and show that both are reported."""
import ctypes
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(ROOT, "tests", "guard", "_build", "libgvd_guard.so")


def install():
    import subprocess
    import torch
    if not os.path.exists(SO):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "guard", "build.sh")])
    alloc = torch.cuda.memory.CUDAPluggableAllocator(SO, "gvd_guard_malloc", "gvd_guard_free")
    torch.cuda.memory.change_current_allocator(alloc)      # must precede the first device allocation of the process
    G = ctypes.CDLL(SO)
    for n in ("gvd_guard_check_all", "gvd_guard_violations", "gvd_guard_allocs", "gvd_guard_frees", "gvd_guard_peak_bytes", "gvd_guard_redzone_bytes"):
        getattr(G, n).restype = ctypes.c_ulonglong
    G.gvd_guard_scribble.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_size_t]
    return G


def selftest(G):
    import torch
    a = torch.zeros(1000, device="cuda:0")                 # 4000 bytes: the rounding slack behind it belongs to the red zone
    b = torch.ones(3, 5, device="cuda:0")
    assert float((a + 1).sum()) == 1000.0 and G.gvd_guard_check_all() == 0
    fresh = torch.empty(64, device="cuda:0", dtype=torch.uint8)
    assert int(fresh.min()) == 0xFB and int(fresh.max()) == 0xFB          # bodies are poisoned: an unwritten byte cannot pass for a zero
    assert G.gvd_guard_scribble(ctypes.c_void_p(a.data_ptr()), 0, 4) == 0      # the first four bytes past the end of `a`
    assert G.gvd_guard_check_all() == 1
    assert G.gvd_guard_scribble(ctypes.c_void_p(b.data_ptr()), -8, 8) == 0     # the eight bytes in front of `b`
    n = G.gvd_guard_violations()
    del b
    torch.cuda.synchronize()
    assert G.gvd_guard_violations() == n + 1                                    # found at free time
    G.gvd_guard_check_all()                                                     # `a` is still live and still scribbled: reported again
    print(f"[gvd_guard] selftest ok: {G.gvd_guard_allocs()} allocations, violations reported {G.gvd_guard_violations()} (3 expected: one twice)")
    return 0 if G.gvd_guard_violations() == 3 else 1


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    os.environ.setdefault("GVD_TORCH_FALLBACK", "warn")     # (the fuzzers' fp32 reference legs, as under pytest)
    G = install()
    if sys.argv[1] == "--selftest":
        return selftest(G)
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    rc = 0
    try:
        runpy.run_path(script, run_name="__main__")
    except SystemExit as e:
        rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    import gc
    import torch
    gc.collect()
    torch.cuda.synchronize()
    v = G.gvd_guard_check_all()
    print(f"[gvd_guard] {os.path.basename(script)}: {G.gvd_guard_allocs()} allocations, {G.gvd_guard_frees()} freed and checked, "
          f"{G.gvd_guard_allocs() - G.gvd_guard_frees()} checked live at exit, red zones {G.gvd_guard_redzone_bytes()} B, "
          f"peak {G.gvd_guard_peak_bytes() / 2 ** 30:.2f} GiB, violations: {v}", flush=True)
    return rc if rc else (97 if v else 0)


if __name__ == "__main__":
    code = main()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code or 0)     # no interpreter finalisation: the tensors still alive would be freed -- through this allocator's device synchronise and
    #                         red-zone copies -- while the HIP runtime is being torn down; every live block was checked by gvd_guard_check_all() above
