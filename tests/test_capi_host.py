"""CPU-only checks of the C-ABI boundary: the shared library builds for gfx950, loads, exports every
symbol include/gvd_raster.h declares, the chunk-size/layout host logic is sane, and the operator
fails LOUDLY (no silent CPU fallback) when tensors are not on a ROCm device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build_hip()
    from diff_gaussian_rasterization import _C
    return _C


def test_library_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "gvd_raster.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(gvd_[a-z_0-9]+)\s*\(", hdr))
    names -= {"gvd_alloc_fn"}
    assert {"gvd_raster_forward", "gvd_raster_backward", "gvd_raster_mark_visible", "gvd_raster_forward_capped"} <= names
    L = capi.lib()
    for n in sorted(names):
        assert hasattr(L, n), f"libgvd_raster.so does not export {n}"
    assert b"gfx950" in L.gvd_version()


def test_chunk_sizes_and_layout(capi):
    L = capi.lib()
    g1 = L.gvd_raster_geometry_bytes(1000, 128, 128)
    g2 = L.gvd_raster_geometry_bytes(200000, 640, 480)
    assert 0 < g1 < g2
    assert L.gvd_raster_binning_bytes(0) > 0
    assert L.gvd_raster_binning_bytes(1000) < L.gvd_raster_binning_bytes(1000000)
    assert L.gvd_raster_image_bytes(640, 480) >= 640 * 480 * 4 + 1200 * 8
    lay = capi._ChunkLayout()
    P, W, H, R = 1000, 128, 128, 3201
    L.gvd_raster_chunk_layout(P, W, H, R, ctypes.byref(lay))
    geom = [("depths", 4 * P), ("means2D", 8 * P), ("conic_opacity", 16 * P), ("rgbd", 16 * P), ("cov3D", 24 * P),
            ("clamped", 4 * P), ("internal_radii", 4 * P), ("tiles_touched", 4 * P), ("point_offsets", 4 * P), ("scalars", 32)]
    spans = sorted((getattr(lay, n), getattr(lay, n) + sz) for n, sz in geom)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0, "geometry sub-arrays overlap"
    assert all(getattr(lay, n) % 128 == 0 for n, _ in geom)
    assert spans[-1][1] <= L.gvd_raster_geometry_bytes(P, W, H)
    assert lay.point_list_keys + 8 * R <= lay.point_list and lay.point_list + 4 * R <= lay.bucket
    assert lay.bucket + 8 * R <= L.gvd_raster_binning_bytes(R)
    assert lay.ranges + 8 * 64 <= lay.n_contrib
    assert lay.n_contrib + 4 * W * H <= lay.tile_order and lay.tile_order + 4 * 64 <= L.gvd_raster_image_bytes(W, H)
    # the ctypes mirror must have exactly the members of the C struct (a shorter mirror would be overrun by the callee)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "gvd_raster.h")).read()
    body = hdr[hdr.index("typedef struct gvd_chunk_layout {"):hdr.index("} gvd_chunk_layout;")]
    members = re.findall(r"^\s*size_t\s+(\w+);", body, flags=re.M)
    assert members == [n for n, _ in capi._ChunkLayout._fields_]


def test_settings_and_operator_surface():
    import diff_gaussian_rasterization as d
    assert d.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug", "confidence")
    for n in ("GaussianRasterizer", "rasterize_gaussians"):
        assert hasattr(d, n)
    assert hasattr(d.GaussianRasterizer, "markVisible")


def _settings(d):
    z = torch.zeros
    return d.GaussianRasterizationSettings(16, 16, 1.0, 1.0, z(3), 1.0, torch.eye(4), torch.eye(4), 0, z(3), False, False,
                                           torch.ones(4, 1))


def test_argument_validation_matches_reference():
    import diff_gaussian_rasterization as d
    r = d.GaussianRasterizer(_settings(d))
    m, op = torch.zeros(4, 3), torch.ones(4, 1)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, op, shs=None, colors_precomp=None, scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, op, shs=torch.zeros(4, 16, 3), colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, op, shs=torch.zeros(4, 16, 3))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, op, shs=torch.zeros(4, 16, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4), cov3D_precomp=torch.zeros(4, 6))


def test_no_cpu_fallback_fails_loudly():
    import diff_gaussian_rasterization as d
    r = d.GaussianRasterizer(_settings(d))
    m, op = torch.zeros(4, 3), torch.ones(4, 1)
    with pytest.raises(RuntimeError, match="ROCm device"):
        r(m, m, op, shs=torch.zeros(4, 16, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="ROCm device"):
        r.markVisible(m)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "guidedvd-3dgs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "raster_oracle" not in src and "oracle/" not in src.replace("CPU oracle", ""), os.path.join(dp, f)


def test_binning_capacity_inverts_binning_bytes_and_never_reports_less_than_it_holds(capi):
    """gvd_raster_backward_conf decodes the capacity a binning chunk was laid out for from its byte size.  Capacities 0 and 1 share a
    layout (every sub-array holds at least one element); the decode must report 1 there -- a chunk holding exactly ONE instance was
    rejected as too small (round 5, found by tests/scripts/r5_raster_stress.py).  Sizes that are no chunk size decode to 0xffffffff."""
    L = capi.lib()
    L.gvd_raster_binning_capacity.restype = ctypes.c_uint32
    L.gvd_raster_binning_capacity.argtypes = [ctypes.c_size_t]
    prev = 0
    for r in (0, 1, 2, 3, 7, 64, 1000, 436438, 5_000_000):
        b = L.gvd_raster_binning_bytes(r)
        assert b >= prev
        prev = b
        cap = L.gvd_raster_binning_capacity(b)
        assert cap == max(r, 1), (r, cap)
        assert L.gvd_raster_binning_capacity(b + 4) == 0xffffffff


def test_no_backward_chunk_is_a_ninth_of_the_size_and_never_passes_for_a_full_one(capi):
    """The binning chunk of a forward run under gvd_raster_expect_backward(0) carries no partial records (advisor finding, round 5: 220 bytes
    per instance held per render in flight).  Its size is never the size of a full layout -- the backward tells the two apart by size alone
    and refuses the compact one."""
    L = capi.lib()
    for r in list(range(0, 70)) + [100, 1000, 4095, 4096, 4097, 436438, 436439, 5_000_000, 10_000_000]:
        full, compact = L.gvd_raster_binning_bytes(r), L.gvd_raster_binning_bytes_no_backward(r)
        assert compact < full
        assert full % 64 == 0 and compact % 4 == 2        # the size alone names the form
        assert L.gvd_raster_binning_capacity(compact) == max(r, 1)   # (keys / point_list / bucket / qmask sit at the same offsets in both forms)
        if r >= 1000:
            assert compact < 0.14 * full, (r, compact, full)
    assert L.gvd_raster_binning_bytes_no_backward(10_000_000) < 300e6 < 2.2e9 < L.gvd_raster_binning_bytes(10_000_000)
    sizes = [L.gvd_raster_binning_bytes_no_backward(r) for r in range(1, 3000)]
    assert all(b > a for a, b in zip(sizes, sizes[1:]))   # strictly increasing: the size names the capacity


def test_chunk_size_decode_over_random_capacities(capi):
    """400 random capacities up to 2^31 through both layout forms: the byte size names the capacity exactly, sizes one byte off name nothing, and the
    two forms never share a size (full: multiple of 64; without the backward's records: 2 mod 4)."""
    import random
    L = capi.lib()
    rng = random.Random(20260930)
    for _ in range(400):
        r = rng.choice((rng.randrange(0, 70), rng.randrange(0, 1 << 20), rng.randrange(0, 1 << 31)))
        full, compact = L.gvd_raster_binning_bytes(r), L.gvd_raster_binning_bytes_no_backward(r)
        assert full % 64 == 0 and compact % 4 == 2 and compact < full
        assert L.gvd_raster_binning_capacity(full) == max(r, 1) and L.gvd_raster_binning_capacity(compact) == max(r, 1)
        assert L.gvd_raster_binning_capacity(full + 1) == 0xffffffff and L.gvd_raster_binning_capacity(compact - 1) == 0xffffffff
        if r > 1:
            assert L.gvd_raster_binning_bytes(r - 1) < full and L.gvd_raster_binning_bytes_no_backward(r - 1) < compact


def test_compiled_operator_loads_and_fails_loudly_without_a_device(capi):
    """lib/_gvd_raster_torch.so (csrc/raster_torch_ext.cpp, built by __graft_entry__.build_raster_torch_ext) imports, resolves its C-ABI
    entry points from libgvd_raster.so and refuses CPU tensors with the binding's message -- no silent fallback."""
    import __graft_entry__ as g
    g.build_raster_torch_ext()
    capi._ext = None
    ext = capi.ext()
    assert ext is not None and hasattr(ext, "rasterize")
    import diff_gaussian_rasterization as dgr
    P = 5
    st = dgr.GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False,
                                           torch.ones(P, 1))
    with pytest.raises(RuntimeError, match="ROCm device"):
        dgr.GaussianRasterizer(st)(means3D=torch.zeros(P, 3), means2D=torch.zeros(P, 3), opacities=torch.ones(P, 1), shs=torch.zeros(P, 1, 3),
                                   scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    with pytest.raises(RuntimeError, match="num_points, 3"):
        dgr.GaussianRasterizer(st)(means3D=torch.zeros(P, 4), means2D=torch.zeros(P, 3), opacities=torch.ones(P, 1), shs=torch.zeros(P, 1, 3),
                                   scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
