"""simple-knn replacement: oracle checks on CPU, HIP-vs-oracle parity on the GPU.

Parity bar: integer/index results bit-exact; the fp32 mean distance bit-exact too (same contraction as the reference:
fma(dz,dz,fma(dy,dy,dx*dx)), mean = (d0+d1+d2)/3.0f) -- there is no tolerance in these comparisons."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import knn_oracle as ko  # noqa: E402


def _cloud(kind, P, seed):
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        pts = rng.normal(size=(P, 3))
    elif kind == "room":  # points on the walls of a 6x4x3 box (the C2 raster scene's distribution)
        pts = rng.uniform([-3, -2, -1.5], [3, 2, 1.5], size=(P, 3))
        ax = rng.integers(0, 3, size=P)
        side = rng.integers(0, 2, size=P) * 2 - 1
        pts[np.arange(P), ax] = side * np.array([3, 2, 1.5])[ax] + rng.normal(scale=0.02, size=P)
    elif kind == "dupes":  # densify_and_clone leaves exact duplicates: neighbours at distance 0, ties everywhere
        base = rng.normal(size=(max(P // 3, 1), 3))
        pts = base[rng.integers(0, base.shape[0], size=P)]
    elif kind == "lattice":  # many exactly equal distances
        g = np.stack(np.meshgrid(*[np.arange(int(round(P ** (1 / 3))) + 1)] * 3, indexing="ij"), -1).reshape(-1, 3)
        pts = g[:P].astype(np.float64) * 0.25 - 1.0
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(pts[:P], np.float32)


def _brute(pts, order_rank):
    """float64 distances + the reference's tie rule (earlier Morton position first)."""
    d = ((pts[:, None, :].astype(np.float64) - pts[None, :, :].astype(np.float64)) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    key = np.lexsort((np.broadcast_to(order_rank, d.shape), d), axis=1)[:, :3]
    return np.take_along_axis(d, key, 1), key


@pytest.mark.parametrize("kind,P", [("gauss", 700), ("room", 1500), ("dupes", 600), ("lattice", 512)])
def test_oracle_matches_float64_brute_force_with_the_morton_tie_rule(kind, P):
    pts = _cloud(kind, P, 1)
    means, idx = ko.dist2(pts)
    codes, order = ko.morton_order(pts)
    assert sorted(order.tolist()) == list(range(P)) and np.all(np.diff(codes[order].astype(np.int64)) >= 0)
    rank = np.empty(P, np.int64)
    rank[order] = np.arange(P)
    d3, i3 = _brute(pts, rank)
    np.testing.assert_allclose(means, d3.mean(1), rtol=2e-6, atol=1e-12)
    if kind in ("gauss", "room"):  # no exact ties in fp32 distance -> indices must agree with the float64 ranking
        agree = (i3 == idx).all(1)
        assert agree.mean() > 0.999
    # whatever the ties, the returned ids must realise the returned distances
    got = ((pts[idx].astype(np.float64) - pts[:, None, :]) ** 2).sum(-1)
    np.testing.assert_allclose(got.mean(1), means, rtol=2e-6, atol=1e-12)
    assert np.all(np.diff(got, axis=1) >= -1e-12) and np.all(idx != np.arange(P)[:, None])


def test_oracle_small_clouds_keep_flt_max_slots_like_the_reference():
    for P in (1, 2, 3):
        m, i = ko.dist2(_cloud("gauss", P, 3))
        assert np.all(np.isinf(m)) or np.all(m > 1e37)   # (FLT_MAX + ...) / 3 overflows to inf
        assert i.shape == (P, 3)
    m, i = ko.dist2(_cloud("gauss", 4, 3))
    assert np.all(np.isfinite(m))


def test_library_exports_every_declared_symbol_and_refuses_cpu_tensors():
    import __graft_entry__ as g
    g.build_knn()
    from simple_knn import _C
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gvd_knn.h")).read(), flags=re.S)
    names = set(re.findall(r"\b(gvd_[a-z_0-9]+)\s*\(", hdr))
    assert {"gvd_knn_mean_dist", "gvd_knn_workspace_bytes", "gvd_knn_last_error"} <= names
    L = _C.lib()
    for n in sorted(names):
        assert hasattr(L, n), f"libgvd_knn.so does not export {n}"
    assert 0 < L.gvd_knn_workspace_bytes(1000) < L.gvd_knn_workspace_bytes(1000000)
    with pytest.raises(RuntimeError, match="ROCm device"):
        _C.distCUDA2(torch.zeros(10, 3))


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind,P", [("gauss", 1), ("gauss", 2), ("gauss", 3), ("gauss", 4), ("gauss", 5), ("gauss", 1023),
                                    ("gauss", 1025), ("room", 5000), ("dupes", 3000), ("lattice", 4096), ("gauss", 20000),
                                    # the in-tree radix sort's 2048-key tiles and 64-key chunks: sizes around them, many equal codes
                                    ("gauss", 2047), ("gauss", 2048), ("gauss", 2049), ("dupes", 4160), ("dupes", 100_003)])
def test_hip_knn_is_bit_exact_against_the_oracle(kind, P):
    from simple_knn._C import distCUDA2
    pts = _cloud(kind, P, 7)
    m_ref, i_ref = ko.dist2(pts)
    m, i = distCUDA2(torch.tensor(pts, device="cuda:0"))
    assert m.dtype == torch.float32 and i.dtype == torch.int32 and tuple(i.shape) == (P, 3)
    assert np.array_equal(m.cpu().numpy().view(np.uint32), m_ref.view(np.uint32))
    assert np.array_equal(i.cpu().numpy(), i_ref)


@pytest.mark.gpu
def test_hip_knn_full_size_properties_and_empty_input():
    """BASELINE-size cloud (200 000 points): checked against float64 brute force on 1500 sampled queries, plus
    size-independent properties (ids valid and distinct from self, distances sorted, realised by the ids)."""
    from simple_knn._C import distCUDA2
    P = 200_000
    pts = _cloud("room", P, 11)
    tp = torch.tensor(pts, device="cuda:0")
    m, i = distCUDA2(tp)
    torch.cuda.synchronize()
    i64 = i.long()
    assert int(i64.min()) >= 0 and int(i64.max()) < P and bool((i64 != torch.arange(P, device="cuda:0")[:, None]).all())
    d = ((tp[i64].double() - tp[:, None, :].double()) ** 2).sum(-1)
    assert bool((d[:, 1:] >= d[:, :-1]).all())
    assert torch.allclose(d.mean(1).float(), m, rtol=2e-6, atol=1e-12)
    q = torch.randperm(P, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(0))[:1500]
    full = torch.cdist(tp[q].double(), tp.double()) ** 2
    full[torch.arange(1500, device="cuda:0"), q] = float("inf")
    best = full.topk(3, dim=1, largest=False).values
    assert torch.allclose(best.mean(1).float(), m[q], rtol=1e-5, atol=1e-10)
    m0, i0 = distCUDA2(torch.zeros(0, 3, device="cuda:0"))
    assert m0.shape == (0,) and i0.shape == (0, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)))
def test_hip_knn_randomized_clouds_bit_exact(seed):
    """Random sizes around the 256-query workgroup and the 1024-point box, clustered / duplicated / degenerate
    (collinear, coplanar, all-equal) clouds, large coordinate offsets."""
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(700 + seed)
    P = int(rng.choice([6, 255, 256, 257, 1023, 1024, 1025, 2049, 3333]))
    kind = rng.choice(["gauss", "clusters", "line", "plane", "same", "offset"])
    pts = rng.normal(size=(P, 3))
    if kind == "clusters":
        pts = pts * 0.01 + rng.normal(size=(8, 3))[rng.integers(0, 8, size=P)] * 5
    elif kind == "line":
        pts[:, 1:] = 0.0
    elif kind == "plane":
        pts[:, 2] = 1.5
    elif kind == "same":
        pts[:] = pts[0]
    elif kind == "offset":
        pts = pts * 0.05 + np.array([1000.0, -2000.0, 500.0])
    pts = np.ascontiguousarray(pts, np.float32)
    m_ref, i_ref = ko.dist2(pts)
    m, i = distCUDA2(torch.tensor(pts, device="cuda:0"))
    assert np.array_equal(m.cpu().numpy().view(np.uint32), m_ref.view(np.uint32)), (P, kind)
    assert np.array_equal(i.cpu().numpy(), i_ref), (P, kind)
