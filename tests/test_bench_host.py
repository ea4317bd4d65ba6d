"""Host-side pieces of bench.py that need no device (round 6: the last session of the round edited the bench line without GPU access -- these
are the edited pieces as pure functions): the SURVEY 8(d) byte model against the verdict's own recomputation, the two extra HBM rooflines of the
raster line, the MFMA family roofline on executed flops, and the lookup of the newest counter file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench.py touches no device)


def test_byte_model_reproduces_the_verdicts_recomputation():
    """VERDICT.md (round 5), weak #6: 155.8 MB forward + 142.8 MB backward for the driver run's C2 view (P = 200 000, 640 x 480, R = 436 438);
    k_gather_bwd's share 115 MB."""
    fwd, bwd, pg = bench.raster_byte_model(200000, 640, 480, 436438)
    assert abs(fwd / 1e6 - 155.8) < 0.1 and abs(bwd / 1e6 - 142.8) < 0.1 and abs(pg / 1e6 - 115.0) < 0.1
    f2, b2, _ = bench.raster_byte_model(200000, 640, 480, 0)
    assert f2 < fwd and b2 < bwd


def test_raster_rooflines_of_the_round5_driver_run():
    """0.2649 ms per step -> 1.13 TB/s = 0.14 of 8 TB/s; gather_bwd 41.86 us -> 2.75 TB/s = 0.34 (the figures the verdict quotes)."""
    kern = {"render_bwd": {"avg_us": 111.28}, "gather_bwd": {"avg_us": 41.86}}
    step, gather = bench.raster_hbm_rooflines(200000, 640, 480, 436438, 0.2649e-3, kern)
    assert abs(step["frac"] - 0.141) < 0.002 and step["unit"] == "GB/s" and abs(step["alg_bytes_fwd"] + step["alg_bytes_bwd"] - 298.6e6) < 0.2e6
    assert abs(step["achieved"] - 1127.0) < 5.0
    assert abs(gather["frac"] - 0.343) < 0.005 and gather["kernel"] == "gather_bwd" and gather["alg_bytes"] == 115000000
    assert gather["traffic"] is None or gather["traffic"] > gather["alg_bytes"] * 0.5
    json.dumps([step, gather])                                    # both go into the one JSON line
    assert bench.raster_hbm_rooflines(1, 16, 16, 1, 1e-3, {}) == (None, None)
    s2, g2 = bench.raster_hbm_rooflines(1000, 64, 64, 5000, 1e-4, {"render_fwd": {"avg_us": 10.0}})
    assert s2 is not None and g2 is None


class _Ev:
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return other.t - self.t


def test_mfma_family_roofline_counts_executed_flops_and_keeps_the_reference_operators():
    ev = lambda ms, *rest: (_Ev(0.0), _Ev(ms)) + rest
    plain = [ev(1.0, 1e12, ("conv", 0, 1, 8, 8, 32, 32, 0), 1e12), ev(1.0, 1e12, ("gemm", 1, 8, 8, 8, False, False, False))]
    r = bench.mfma_family_roofline(plain, "k", 2)
    assert r["achieved"] == 1000.0 and r["frac"] == 0.4 and r["ms_per_step"] == 1.0 and "flops" not in r and r["launches"] == 2
    up2 = plain + [ev(2.0, 4e12, ("conv", 4, 1, 8, 8, 32, 32, 0), 9e12)]       # a phase convolution: 4 of 9 taps per output pixel executed
    r = bench.mfma_family_roofline(up2, "k", 2)
    assert r["achieved"] == 1500.0 and r["achieved_on_reference_operator_flops"] == 2750.0 and r["reference_operator_tflop_per_step"] == 5.5
    attn = [ev(1.0, 2e12, True, 123.0), ev(1.0, 2e12, False, 5.0)]              # attention events carry (flops, frame_major, bytes): no tuple key
    r = bench.mfma_family_roofline(attn, "a", 1)
    assert r["achieved"] == 2000.0 and "flops" not in r
    assert bench.mfma_family_roofline([], "k", 2) is None and bench.mfma_family_roofline(plain, "k", 0) is None
    json.dumps(r)


def test_newest_profile_picks_the_latest_round():
    path, name = bench.newest_profile("mfma_pmc.json")
    have = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_mfma_pmc.json") and f[0] == "r" and f[1:3].isdigit())
    assert name == "profiles/" + have[-1] and os.path.exists(path)
    keys = json.load(open(path))
    assert any(k.startswith("attn i4ELi2") for k in keys) and "gemm L0 out-proj 230400x320x320 +residual" in keys   # what bench.py looks up in it
    assert bench.newest_profile("no_such_suffix.json") == (None, None)


def test_every_profile_file_the_documents_cite_exists():
    """Evidence hygiene (verdict r5 item 7): a `profiles/...` path quoted in DESIGN.md / README.md / INTEGRATION.md / a source comment must be a committed
    file.  Glob-like mentions (`r05_*`, `rNN_...`, `{a,b}`) are patterns, not citations; the three round-6 files below are named as the OUTPUT of scripts
    that have not run (no GPU access in the session that wrote them) and are listed here so that nothing else can hide behind that excuse."""
    import re
    pending = {"profiles/r06_gemm_vs_hipblaslt.txt", "profiles/r06_guard.log"}
    texts = [os.path.join(ROOT, f) for f in ("DESIGN.md", "README.md", "INTEGRATION.md", "bench.py")]
    for d in ("guidedvd-3dgs_amd/csrc", "guidedvd-3dgs_amd/lvdm_amd", "guidedvd-3dgs_amd/diff_gaussian_rasterization"):
        texts += [os.path.join(ROOT, d, f) for f in os.listdir(os.path.join(ROOT, d)) if f.endswith((".hip", ".h", ".cpp", ".py"))]
    missing = []
    for t in texts:
        for m in re.finditer(r"profiles/[A-Za-z0-9_./*{},<>-]+", open(t, errors="replace").read()):
            p = m.group(0).rstrip(".,;:)")
            if any(c in p for c in "*{<") or "rNN" in p or p.endswith("/") or p in pending:
                continue
            if not os.path.exists(os.path.join(ROOT, p)):
                missing.append((os.path.relpath(t, ROOT), p))
    assert not missing, missing
