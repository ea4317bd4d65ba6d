"""The red-zone guard allocator (tests/guard/guard_allocator.cpp, tests/scripts/r6_guard_run.py; verdict r5 item 6) as a `-m gpu` test: its
self-test (bodies poisoned, a scribble past either end of a block reported at the next check and at free time), then one randomised raster pass and
one diffusion fuzz pass under it with zero violations.  Each in a process of its own: the allocator must be installed before the first device
allocation.  The CPU half (it builds; it exports what the runner binds) runs everywhere."""
import ctypes
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
# The tool was written in round 6 while GPU access was closed to the build (no supervised first run on a device).  Its GPU tests join the suite
# once that run exists -- tests/scripts/r6_guard_all.sh writes the log, the builder triages it and commits it -- so that an untried tool cannot
# turn the round-end suite red for reasons that are not findings.  GVD_GUARD_TESTS=1 runs them regardless.
_LOG = os.path.join(os.path.dirname(HERE), "profiles", "r06_guard.log")
supervised = pytest.mark.skipif(not (os.path.exists(_LOG) or os.environ.get("GVD_GUARD_TESTS") == "1"),
                                reason="profiles/r06_guard.log absent: the guard allocator has not had its first supervised GPU run (GVD_GUARD_TESTS=1 forces)")
RUN = os.path.join(HERE, "scripts", "r6_guard_run.py")
SO = os.path.join(HERE, "guard", "_build", "libgvd_guard.so")


def test_guard_allocator_builds_and_exports_its_entry_points():
    subprocess.check_call(["bash", os.path.join(HERE, "guard", "build.sh")])
    G = ctypes.CDLL(SO)
    for n in ("gvd_guard_malloc", "gvd_guard_free", "gvd_guard_check_all", "gvd_guard_violations", "gvd_guard_allocs", "gvd_guard_frees",
              "gvd_guard_peak_bytes", "gvd_guard_redzone_bytes", "gvd_guard_scribble"):
        assert hasattr(G, n), n
    G.gvd_guard_redzone_bytes.restype = ctypes.c_ulonglong
    assert G.gvd_guard_redzone_bytes() % 256 == 0 and G.gvd_guard_redzone_bytes() >= 256


def test_guard_allocator_logic_on_the_host(tmp_path):
    """The allocator's own logic without a device: guard_allocator.cpp built with -DGVD_GUARD_HOST_FAKE (its six HIP calls mapped to malloc / memset /
    memcpy) and driven through the entry points torch would call.  Bodies and zones are poisoned, the body is 256-byte aligned, a clean free reports
    nothing, one byte written just past the end / just before the start / at the far edge of a zone is found with its offset (at the next check and at
    free time), the rounding slack behind an odd-sized body belongs to the zone, a foreign pointer is reported, counters add up."""
    so = str(tmp_path / "libgvd_guard_host.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DGVD_GUARD_HOST_FAKE", "-o", so, os.path.join(HERE, "guard", "guard_allocator.cpp")])
    G = ctypes.CDLL(so)
    G.gvd_guard_malloc.restype = ctypes.c_void_p
    G.gvd_guard_malloc.argtypes = [ctypes.c_ssize_t, ctypes.c_int, ctypes.c_void_p]
    G.gvd_guard_free.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_void_p]
    for n in ("gvd_guard_check_all", "gvd_guard_violations", "gvd_guard_allocs", "gvd_guard_frees", "gvd_guard_redzone_bytes"):
        getattr(G, n).restype = ctypes.c_ulonglong
    G.gvd_guard_scribble.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_size_t]
    rz = G.gvd_guard_redzone_bytes()
    assert rz == 4096
    a = G.gvd_guard_malloc(1000, 0, None)                     # odd size: 24 bytes of rounding slack behind the body
    assert a % 256 == 0
    assert ctypes.string_at(a, 1000) == b"\xfb" * 1000 and ctypes.string_at(a - rz, rz) == b"\xfb" * rz and ctypes.string_at(a + 1000, 24 + rz) == b"\xfb" * (24 + rz)
    ctypes.memset(a, 0, 1000)                                 # the owner writes its whole body: fine
    assert G.gvd_guard_check_all() == 0
    G.gvd_guard_free(a, 1000, 0, None)
    assert G.gvd_guard_violations() == 0 and (G.gvd_guard_allocs(), G.gvd_guard_frees()) == (1, 1)
    b = G.gvd_guard_malloc(1000, 0, None)
    ctypes.memset(b + 1000, 0x11, 1)                          # ONE byte past the end (inside the rounding slack)
    assert G.gvd_guard_check_all() == 1
    c = G.gvd_guard_malloc(4096, 0, None)
    ctypes.memset(c - 1, 0x11, 1)                             # one byte before the start
    ctypes.memset(c + 4096 + rz - 1, 0x11, 1)                 # the last byte of the zone behind
    v0 = G.gvd_guard_violations()
    G.gvd_guard_free(c, 4096, 0, None)
    assert G.gvd_guard_violations() == v0 + 2                 # both zones of c, found at free time
    assert G.gvd_guard_scribble(b, -8, 8) == 0 and G.gvd_guard_scribble(b, 10 ** 9, 1) == -2 and G.gvd_guard_scribble(c, 0, 1) == -1
    v1 = G.gvd_guard_violations()
    G.gvd_guard_free(b, 1000, 0, None)
    assert G.gvd_guard_violations() == v1 + 2                 # b: the earlier byte behind + the scribble in front
    G.gvd_guard_free(b, 1000, 0, None)                        # a pointer that is no longer (or never was) this allocator's
    assert G.gvd_guard_violations() == v1 + 3
    assert G.gvd_guard_malloc(0, 0, None) is None
    assert (G.gvd_guard_allocs(), G.gvd_guard_frees()) == (3, 3) and G.gvd_guard_check_all() == G.gvd_guard_violations()


@pytest.mark.gpu
@supervised
def test_guard_allocator_selftest():
    r = subprocess.run([sys.executable, RUN, "--selftest"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "selftest ok" in r.stdout and r.stderr.count("VIOLATION") == 3, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
@supervised
@pytest.mark.parametrize("script,args,marker", [("r5_raster_stress.py", ["7"], "strided inputs: ok"),
                                                ("r5_diffusion_fuzz.py", ["11"], "diffusion fuzz ok")])
def test_stress_scripts_write_no_red_zone(script, args, marker):
    """A written red zone = exit code 97 and a VIOLATION line naming the block; a poisoned read breaks the script's own checks."""
    r = subprocess.run([sys.executable, RUN, os.path.join(HERE, "scripts", script)] + args, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and marker in r.stdout and "violations: 0" in r.stdout, (r.stdout[-2500:], r.stderr[-3000:])
