"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP rasterizer, called through the
C-ABI (ctypes binding `_C`) and through the drop-in operator, against the CPU oracle.

Tolerances (DESIGN.md "parity bar"):
  * integer/byte/index state -- radii, tiles_touched, point_offsets, (tile|depth) keys, sorted point
    list, tile ranges, n_contrib, and the fp32 BITS of depth / means2D / conic / cov3D / rgb:
    bit-exact (both sides follow one explicit fmaf sequence, -ffp-contract=off);
  * rendered color/depth/alpha: |a-b| <= 1e-4*max(1,|b|) at every pixel (device v_exp_f32 vs libm
    expf is the only difference);
  * gradients: max|a-b| <= 1e-4*max|b| per tensor when backward is fed the same alpha image;
    end-to-end (each side's own forward) 2e-3, because the reference recovers the final
    transmittance as 1 - out_alpha (backward.cu:463), which turns a 5e-7 forward difference into
    up to ~5e-3 relative on saturated pixels -- a property of the reference algorithm itself.
"""
import math
import os

import numpy as np
import pytest
import torch

import synthetic as syn
from raster_compare import compare, rel_elementwise, rel_to_max, run_hip, run_oracle

pytestmark = pytest.mark.gpu

EXACT = ("R_equal", "radii_equal", "depth_bits_equal", "means2D_bits_equal", "tiles_touched_equal",
         "point_offsets_equal", "ranges_equal", "conic_bits_equal", "rgb_bits_equal", "keys_equal", "point_list_equal")


def _grads(H, W, seed, depth_alpha=True):
    rng = np.random.default_rng(seed)
    gC = rng.normal(size=(3, H, W)) / (H * W)
    if depth_alpha:
        return gC, rng.normal(size=(H, W)) / (H * W), rng.normal(size=(H, W)) / (H * W)
    return gC, np.zeros((H, W)), np.zeros((H, W))


def _check(rep, exact=EXACT, grad_tol=2e-3, n_contrib_outliers=0.0, img_outliers=0.0):
    """n_contrib_outliers: fraction of pixels whose last-contributor index may differ (each one proven a near-tie below);
    img_outliers: fraction of pixels whose colour / depth / alpha may differ by more than 1e-4 -- 0 unless a test says why."""
    for k in exact:
        if k in rep:
            assert rep[k] is True, (k, rep)
    assert rep.get("n_contrib_mismatch_frac", 0.0) <= n_contrib_outliers, rep
    # n_contrib is exact except where one of the blend loop's comparisons sits within fp32 rounding of its threshold (the two
    # sides differ by <= 1 ulp in exp()); every mismatching pixel is replayed in float64 and must show such a near-tie
    assert rep.get("n_contrib_mismatch_worst_threshold_margin", 0.0) < 2e-4, rep
    for k in ("color", "depth", "alpha"):
        assert rep[k + "_outlier_frac"] <= img_outliers, (k, rep)
    for k, v in rep.items():
        if k.startswith("grad_"):
            assert v < grad_tol, (k, v)


def _tiny(seed, P=300, W=100, H=70, deg=3, spread=1.0):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.0, 1.0, size=(P, 3)) * spread
    xyz[:, 2] = xyz[:, 2] * 1.5 + 3.5
    xyz[: P // 20, 2] = -1.0
    scales = np.exp(rng.normal(math.log(0.08), 0.5, size=(P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1 / (1 + np.exp(-rng.normal(0.0, 1.5, size=(P, 1))))).astype(np.float32)
    sh = rng.normal(0, 0.25, size=(P, 16, 3)).astype(np.float32)
    sh[:, 0] = syn.rgb2sh(rng.uniform(0, 1, size=(P, 3)))
    cam = syn.make_camera(syn.look_at((0.1, -0.05, 0.0), (0.0, 0.1, 3.0)), math.radians(70), math.radians(55), W, H)
    return dict(means3D=xyz.astype(np.float32), scales=scales, rotations=q, opacities=opac, shs=sh, sh_degree=deg,
                bg=np.array([0.1, 0.4, 0.8], np.float32), cameras=[cam])


def test_c1_bit_exact_keys_and_golden_fixture():
    import os
    sc = syn.scene_c1()
    cam = sc["cameras"][0]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_c1_oracle.npz"))
    grads = (z["gC"], z["gD"], z["gA"])
    st_o, g_o = run_oracle(sc, cam, grads)
    st_h, g_h = run_hip(sc, cam, grads, debug=True)
    _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=1e-4)
    # and directly against the committed fixture (no oracle code involved)
    assert st_h["R"] == int(z["R"])
    assert np.array_equal(st_h["point_list_keys"].view(np.uint64), z["keys"])
    assert np.array_equal(st_h["point_list"].view(np.uint32), z["point_list"])
    assert np.array_equal(st_h["ranges"].view(np.uint32), z["ranges"])
    assert np.array_equal(st_h["radii"], z["radii"])
    np.testing.assert_allclose(st_h["color"], z["color"], atol=1e-5, rtol=0)
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dopacity"):
        assert rel_to_max(g_h[k], z[k]) < 1e-4, k


@pytest.mark.parametrize("cam_id", [0, 3])
def test_c2_full_size_parity(cam_id):
    sc = syn.scene_c2()
    cam = sc["cameras"][cam_id]
    H, W = cam["image_height"], cam["image_width"]
    grads = _grads(H, W, 11 + cam_id)
    st_o, g_o = run_oracle(sc, cam, grads)
    st_h, g_h = run_hip(sc, cam, grads)
    # images: no slack (measured <= 4e-7 at this size); n_contrib: <= 2e-5 of the pixels, each replayed in float64 by compare()
    _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=2e-3, n_contrib_outliers=2e-5)
    # backward kernels in isolation: same alpha image on both sides -> 1e-4 of the largest entry per tensor, and element-wise
    # 5e-3 relative on every entry above 1 % of the largest (below that floor the per-tensor bound is the operative one)
    _, g_h2 = run_hip(sc, cam, grads, alpha_override=st_o["alpha"])
    for k in g_h2:
        assert rel_to_max(g_h2[k], g_o[k]) < 1e-4, k
        assert rel_elementwise(g_h2[k], g_o[k]) < 5e-3, (k, rel_elementwise(g_h2[k], g_o[k]))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_degrees_and_ragged_image(deg):
    sc = _tiny(deg, deg=deg)  # 100x70: not a multiple of 16
    cam = sc["cameras"][0]
    grads = _grads(70, 100, deg)
    st_o, g_o = run_oracle(sc, cam, grads)
    st_h, g_h = run_hip(sc, cam, grads, debug=True)
    _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=5e-4)
    ncoef = (deg + 1) ** 2
    assert np.all(g_h["dL_dsh"][:, ncoef:] == 0)  # quirk 7


def test_precomputed_colors_and_cov3d():
    sc = _tiny(5)
    cam = sc["cameras"][0]
    st_ref, _ = run_oracle(sc, cam)
    colors = np.random.default_rng(3).uniform(0, 1, size=(sc["means3D"].shape[0], 3)).astype(np.float32)
    cov = st_ref["cov3D"].copy()
    # invisible Gaussians never had their cov3D written: give them something finite
    cov[st_ref["radii"] <= 0] = np.array([1e-2, 0, 0, 1e-2, 0, 1e-2], np.float32)
    grads = _grads(70, 100, 5)
    st_o, g_o = run_oracle(sc, cam, grads, colors_precomp=colors, cov3D_precomp=cov)
    st_h, g_h = run_hip(sc, cam, grads, colors_precomp=colors, cov3D_precomp=cov, debug=True)
    rep = compare(st_h, st_o, g_h, g_o, verbose=False)
    _check(rep, grad_tol=5e-4)
    assert g_h["dL_dsh"].shape == (sc["means3D"].shape[0], 0, 3)
    assert np.all(g_h["dL_dscales"] == 0) and np.all(g_h["dL_drotations"] == 0)
    assert np.abs(g_h["dL_dcov3D"]).max() > 0 and np.abs(g_h["dL_dcolors"]).max() > 0


def test_empty_and_fully_culled():
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    e = torch.Tensor([])
    bg = torch.tensor([0.3, 0.6, 0.9], device=dev)
    cam = syn.make_camera(syn.look_at((0, 0, 0), (0, 0, 1)), 1.0, 1.0, 40, 24)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    # P == 0: zero images, num_rendered 0 (rasterize_points.cu:81)
    out = _C.rasterize_gaussians(bg, torch.zeros((0, 3), device=dev), e, torch.zeros((0, 1), device=dev), torch.zeros((0, 3), device=dev),
                                 torch.zeros((0, 4), device=dev), 1.0, e, t(cam["viewmatrix"]), t(cam["projmatrix"]), 1.0, 1.0, 24, 40,
                                 torch.zeros((0, 16, 3), device=dev), 0, t(cam["campos"]), False, False)
    assert out[0] == 0 and float(out[1].abs().max()) == 0.0 and out[4].numel() == 0
    # every Gaussian behind the camera: background everywhere, alpha 0, radii 0
    P = 50
    m = torch.randn(P, 3, device=dev)
    m[:, 2] = -2.0
    out = _C.rasterize_gaussians(bg, m, e, torch.full((P, 1), 0.5, device=dev), torch.full((P, 3), 0.1, device=dev),
                                 torch.nn.functional.normalize(torch.randn(P, 4, device=dev)), 1.0, e, t(cam["viewmatrix"]),
                                 t(cam["projmatrix"]), 1.0, 1.0, 24, 40, torch.zeros((P, 16, 3), device=dev), 0, t(cam["campos"]), False, True)
    assert out[0] == 0
    assert torch.allclose(out[1], bg[:, None, None].expand(3, 24, 40))
    assert float(out[3].abs().max()) == 0.0 and int(out[4].abs().max()) == 0
    g = _C.rasterize_gaussians_backward(bg, m, out[4], e, torch.full((P, 3), 0.1, device=dev),
                                        torch.nn.functional.normalize(torch.randn(P, 4, device=dev)), 1.0, e, t(cam["viewmatrix"]),
                                        t(cam["projmatrix"]), 1.0, 1.0, torch.ones(3, 24, 40, device=dev), torch.ones(1, 24, 40, device=dev),
                                        torch.ones(1, 24, 40, device=dev), torch.zeros((P, 16, 3), device=dev), 0, t(cam["campos"]),
                                        out[5], out[0], out[6], out[7], out[3], True)
    assert all(float(x.abs().max()) == 0.0 for x in g if x.numel())


def test_one_instance_scene():
    """num_rendered == 1: one small Gaussian inside one tile.  Capacities 0 and 1 of the binning chunk share a byte size, and the
    backward used to decode such a chunk as capacity 0 and refuse it (round 5, tests/scripts/r5_raster_stress.py)."""
    sc = _tiny(11, P=1, W=17, H=17, deg=1)
    sc["means3D"][:] = np.array([[0.05, 0.12, 3.0]], np.float32)
    sc["scales"][:] = 0.004
    sc["opacities"][:] = 0.8
    cam = sc["cameras"][0] = syn.make_camera(syn.look_at((0.0, 0.0, 0.0), (0.0, 0.1, 3.0)), math.radians(70), math.radians(55), 17, 17)
    grads = _grads(17, 17, 3)
    st_o, g_o = run_oracle(sc, cam, grads)
    assert int(st_o["R"]) == 1
    st_h, g_h = run_hip(sc, cam, grads, debug=True)
    _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=5e-4)
    assert np.abs(g_h["dL_dmeans3D"]).max() > 0


@pytest.mark.parametrize("P,expect_max", [(3000, 2048), (18000, 16384)])
def test_long_tile_lists_exercise_lds_and_global_sort(P, expect_max):
    # many Gaussians piled into a 32x32 image: tile lists longer than the 2048-entry (LDS class 0)
    # and 16384-entry (LDS class 1) sorters; includes exact depth ties (duplicated points).
    sc = _tiny(9, P=P, W=32, H=32, deg=0, spread=0.15)
    sc["means3D"][1::2] = sc["means3D"][0::2][: len(sc["means3D"][1::2])]  # ties: resolved by id (stable sort)
    sc["opacities"][:] = 0.02
    cam = sc["cameras"][0]
    grads = _grads(32, 32, 9)
    st_o, g_o = run_oracle(sc, cam, grads)
    lens = st_o["ranges"][:, 1] - st_o["ranges"][:, 0]
    assert lens.max() > expect_max
    st_h, g_h = run_hip(sc, cam, grads, debug=True)
    # 16k-entry lists of opacity-0.02 splats: alpha sits right at the 1/255 skip threshold for a large share of the (pixel,
    # entry) pairs, so 1-ulp exp() differences do flip single contributions (|delta| <= alpha T c ~ 4e-3): the one scene built to
    # provoke it keeps an image allowance; every n_contrib mismatch is still proven a near-tie
    _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=2e-3, n_contrib_outliers=2e-3, img_outliers=2e-3)


def test_huge_splats_many_tiles():
    # a few Gaussians covering the whole 640x480 frame (1200 tiles each) + small ones
    sc = _tiny(13, P=400, W=640, H=480, deg=1)
    sc["scales"][20:26] = 3.0  # (the first P//20 are behind the camera)
    cam = sc["cameras"][0]
    grads = _grads(480, 640, 13)
    st_o, g_o = run_oracle(sc, cam, grads)
    assert st_o["tiles_touched"].max() == 1200
    st_h, g_h = run_hip(sc, cam, grads, debug=True)
    _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=2e-3, n_contrib_outliers=1e-5)


def test_operator_autograd_confidence_and_determinism():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    sc = _tiny(21, P=500)
    cam = sc["cameras"][0]
    H, W, P = 70, 100, 500
    t = lambda a, rg=False: torch.tensor(a, dtype=torch.float32, device=dev, requires_grad=rg)
    conf = torch.rand(P, 1, device=dev) * 2.0

    def run(confidence):
        leaves = dict(m=t(sc["means3D"], True), o=t(sc["opacities"], True), s=t(sc["scales"], True), r=t(sc["rotations"], True),
                      sh=t(sc["shs"], True), m2=torch.zeros(P, 3, device=dev, requires_grad=True))
        s = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], t(sc["bg"]), 1.0, t(cam["viewmatrix"]),
                                          t(cam["projmatrix"]), 3, t(cam["campos"]), False, False, confidence)
        color, radii, depth, alpha = GaussianRasterizer(s)(means3D=leaves["m"], means2D=leaves["m2"], opacities=leaves["o"],
                                                           shs=leaves["sh"], scales=leaves["s"], rotations=leaves["r"])
        assert color.shape == (3, H, W) and depth.shape == (1, H, W) and alpha.shape == (1, H, W)
        assert radii.dtype == torch.int32 and radii.shape == (P,)
        g = torch.Generator(device=dev).manual_seed(5)
        loss = (color * torch.randn(3, H, W, device=dev, generator=g)).sum() + 0.3 * depth.sum() + (alpha ** 2).sum()
        loss.backward()
        return {k: v.grad.clone() for k, v in leaves.items()}, color

    g1, c1 = run(torch.ones(P, 1, device=dev))
    gc, c2 = run(conf)
    g1b, _ = run(torch.ones(P, 1, device=dev))
    assert torch.equal(c1, c2)
    for k in g1:  # no float atomics anywhere: bitwise reproducible
        assert torch.equal(g1[k], g1b[k]), k
    assert torch.equal(gc["m2"], g1["m2"])  # screen-space gradient is NOT scaled (ref __init__.py:149)
    assert float(g1["m2"][:, 2].abs().max()) == 0.0 and float(g1["m2"][:, :2].abs().max()) > 0  # quirk 8
    for k in ("m", "o", "s", "r"):
        assert torch.allclose(gc[k], g1[k] * conf, rtol=1e-6, atol=0), k
    assert torch.allclose(gc["sh"], g1["sh"] * conf[..., None], rtol=1e-6, atol=0)


def test_backward_split_walks_are_bit_identical():
    """k_render_bwd cuts long quadrant walks into units that replay the part behind their own (round 5): a pure scheduling change.
    The C2 view has lists of 1000+ entries; every gradient must be the same BITS with the cut off, at the default 512 and at 64 (every
    walk of more than 64 entries in up to four parts) -- through the native boundary with depth / alpha gradients, and through the
    autograd operator with the colour gradient only (the kernel instantiation without the depth / alpha recurrences)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    sc = syn.scene_c2()
    cam = sc["cameras"][1]
    H, W, P = cam["image_height"], cam["image_width"], sc["means3D"].shape[0]
    rng = np.random.default_rng(9)
    grads = tuple(rng.normal(size=s_) / (H * W) for s_ in ((3, H, W), (H, W), (H, W)))
    dev = torch.device("cuda:0")
    t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
    st = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], t(sc["bg"]), 1.0, t(cam["viewmatrix"]), t(cam["projmatrix"]),
                                       sc["sh_degree"], t(cam["campos"]), False, False, torch.ones(P, 1, device=dev))
    gC = t(grads[0])

    def operator_grads():
        lv = dict(means3D=t(sc["means3D"], True), opacities=t(sc["opacities"], True), scales=t(sc["scales"], True),
                  rotations=t(sc["rotations"], True), shs=t(sc["shs"], True), means2D=torch.zeros(P, 3, device=dev, requires_grad=True))
        color, _, _, _ = GaussianRasterizer(st)(**lv)
        torch.autograd.backward([color], [gC])
        return {k: v.grad.cpu().numpy() for k, v in lv.items()}

    L = _C.lib()
    try:
        ref_n = ref_o = None
        for split in (0, 512, 64):
            L.gvd_raster_set_backward_split(split)
            _, g_n = run_hip(sc, cam, grads)
            g_o = operator_grads()
            if ref_n is None:
                ref_n, ref_o = g_n, g_o
                continue
            for k in ref_n:
                assert np.array_equal(ref_n[k], g_n[k]), (split, k)
            for k in ref_o:
                assert np.array_equal(ref_o[k], g_o[k]), (split, k)
    finally:
        L.gvd_raster_set_backward_split(512)


def test_no_grad_renders_skip_the_backward_preparation_and_do_not_disturb_training_renders():
    """gvd_raster_expect_backward (advisor finding, round 2): a no-grad render does not zero the backward's partial records; the
    images are the same bits, and a training render (forward + backward) interleaved with no-grad renders -- same thread, the
    native flag is sticky -- gives the same gradients as without them."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    sc = _tiny(31, P=800)
    cam = sc["cameras"][0]
    H, W, P = 70, 100, 800
    t = lambda a, rg=False: torch.tensor(a, dtype=torch.float32, device=dev, requires_grad=rg)
    s = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], t(sc["bg"]), 1.0, t(cam["viewmatrix"]),
                                      t(cam["projmatrix"]), 3, t(cam["campos"]), False, False, torch.ones(P, 1, device=dev))
    gC = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

    def train():
        lv = dict(means3D=t(sc["means3D"], True), opacities=t(sc["opacities"], True), scales=t(sc["scales"], True),
                  rotations=t(sc["rotations"], True), shs=t(sc["shs"], True), means2D=torch.zeros(P, 3, device=dev, requires_grad=True))
        color, _, depth, alpha = GaussianRasterizer(s)(**lv)
        torch.autograd.backward([color], [gC])
        return color.detach().clone(), {k: v.grad.clone() for k, v in lv.items()}

    def infer():
        with torch.no_grad():
            color, _, _, _ = GaussianRasterizer(s)(means3D=t(sc["means3D"]), means2D=torch.zeros(P, 3, device=dev),
                                                    opacities=t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]),
                                                    rotations=t(sc["rotations"]))
        return color.clone()

    c_ref, g_ref = train()
    for _ in range(3):
        assert torch.equal(infer(), c_ref)          # same bits without the preparation
        c, g = train()                              # ... and the next training render prepares its own chunk again
        assert torch.equal(c, c_ref)
        for k in g_ref:
            assert torch.equal(g[k], g_ref[k]), k


def test_mark_visible_matches_oracle(oracle):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    sc = syn.scene_c2(P=5000)
    cam = sc["cameras"][1]
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    s = GaussianRasterizationSettings(480, 640, cam["tanfovx"], cam["tanfovy"], t(sc["bg"]), 1.0, t(cam["viewmatrix"]),
                                      t(cam["projmatrix"]), 0, t(cam["campos"]), False, False, torch.ones(5000, 1, device=dev))
    vis = GaussianRasterizer(s).markVisible(t(sc["means3D"]))
    assert vis.dtype == torch.bool
    assert np.array_equal(vis.cpu().numpy(), oracle.mark_visible(sc["means3D"], cam["viewmatrix"], cam["projmatrix"]))


def test_capped_forward_no_sync_and_overflow_flag():
    """gvd_raster_forward_capped: same images without the host sync; overflow is flagged, not UB."""
    import ctypes
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    sc = syn.scene_c1()
    cam = sc["cameras"][0]
    st_h, _ = run_hip(sc, cam)
    L = _C.lib()
    P, W, H = 1000, 128, 128
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    m3, op, scl, rot, sh = t(sc["means3D"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), t(sc["shs"])
    vm, pm, cp, bg = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), t(sc["bg"])
    for cap, expect in ((10000, 0), (1000, -4)):
        geom = torch.empty(L.gvd_raster_geometry_bytes(P, W, H), dtype=torch.uint8, device=dev)
        img = torch.empty(L.gvd_raster_image_bytes(W, H), dtype=torch.uint8, device=dev)
        binb = torch.empty(L.gvd_raster_binning_bytes(cap), dtype=torch.uint8, device=dev)
        color, depth, alpha = (torch.empty(n, H, W, device=dev) for n in (3, 1, 1))
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        status = torch.full((1,), 123, dtype=torch.int32, device=dev)
        rc = L.gvd_raster_forward_capped(geom.data_ptr(), binb.data_ptr(), img.data_ptr(), cap, P, 3, 16, bg.data_ptr(), W, H,
                                         m3.data_ptr(), sh.data_ptr(), None, op.data_ptr(), scl.data_ptr(), 1.0, rot.data_ptr(), None,
                                         vm.data_ptr(), pm.data_ptr(), cp.data_ptr(), cam["tanfovx"], cam["tanfovy"], 0,
                                         color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), radii.data_ptr(), status.data_ptr(), 0,
                                         torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert int(status.item()) == expect
        if expect == 0:
            assert np.array_equal(color.cpu().numpy(), st_h["color"])
            assert np.array_equal(alpha.cpu().numpy(), st_h["alpha"])


def test_full_size_properties():
    """Size-independent properties at BASELINE's full size (C2): sorted keys, ranges partition [0,R),
    alpha/T bounds, and linearity of the backward in the incoming pixel gradients."""
    sc = syn.scene_c2()
    cam = sc["cameras"][5]
    H, W = 480, 640
    g1, g2 = _grads(H, W, 31), _grads(H, W, 32)
    st, ga = run_hip(sc, cam, g1)
    _, gb = run_hip(sc, cam, g2)
    mix = tuple(0.7 * a - 1.3 * b for a, b in zip(g1, g2))
    _, gm = run_hip(sc, cam, mix)
    k = st["point_list_keys"].view(np.uint64)
    assert np.all(k[1:] >= k[:-1])
    r = st["ranges"].view(np.uint32).astype(np.int64)
    lens = r[:, 1] - r[:, 0]
    assert lens.sum() == st["R"] and np.all(lens >= 0)
    nz = lens > 0
    order = np.argsort(r[nz, 0])  # non-empty tiles laid end to end in tile order
    assert np.array_equal(r[nz, 0][order][1:], r[nz, 1][order][:-1]) and r[nz, 0].min() == 0
    assert np.array_equal(order, np.arange(order.size))
    tiles = (k >> np.uint64(32)).astype(np.int64)
    assert np.array_equal(np.bincount(tiles, minlength=r.shape[0]), lens)
    assert st["alpha"].min() >= 0.0 and st["alpha"].max() <= 1.0 + 1e-5
    assert np.all(st["n_contrib"].view(np.uint32).reshape(-1) <= np.repeat(lens.reshape(30, 40), 16, 0).repeat(16, 1).reshape(-1))
    for key in ga:
        lin = 0.7 * ga[key] - 1.3 * gb[key]
        assert rel_to_max(gm[key], lin) < 1e-4, key  # fp32 rounding of the cancelling rotation/scale terms


def test_operator_sync_free_mode_matches_and_reports_overflow_late():
    """set_instance_capacity(n): the drop-in operator renders and differentiates without the host read-back; images and
    gradients are bit-identical to the default mode; an overflow raises at the next rasterize call."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    dev = torch.device("cuda:0")
    sc = syn.scene_c1()
    cam = sc["cameras"][0]
    t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
    P = sc["means3D"].shape[0]

    def run():
        prm = [t(sc[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")]
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        s = GaussianRasterizationSettings(image_height=128, image_width=128, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                          bg=t(sc["bg"]), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]),
                                          projmatrix=t(cam["projmatrix"]), sh_degree=3, campos=t(cam["campos"]),
                                          prefiltered=False, debug=False, confidence=torch.ones((P, 1), device=dev))
        color, radii, depth, alpha = GaussianRasterizer(s)(means3D=prm[0], means2D=m2, opacities=prm[1], shs=prm[4],
                                                           scales=prm[2], rotations=prm[3])
        g = torch.Generator(device=dev).manual_seed(5)
        (color * torch.randn(color.shape, device=dev, generator=g)).sum().backward()
        return [color.detach(), depth.detach(), alpha.detach(), radii] + [p.grad for p in prm] + [m2.grad]

    ref = run()
    try:
        _C.set_instance_capacity(20000)
        got = run()
        torch.cuda.synchronize()
        _C._drain_status(block=True)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        _C.set_instance_capacity(500)   # too small for this view: background image, all-zero gradients, no fault
        over = run()
        torch.cuda.synchronize()
        assert all(float(g.abs().max()) == 0.0 for g in over[4:])
        with pytest.raises(RuntimeError, match="instance capacity"):
            run()
    finally:
        _C._PENDING.clear()
        _C.set_instance_capacity(0)


def test_speculative_stage2_misprediction_falls_back_exactly():
    """Same (P, W, H) three times: small splats (first call: exact path, sets the hint), then 3x larger splats (the
    speculative capacity from the hint is too small -> the forward re-runs the exact path), then small again
    (speculation succeeds with a now generous capacity).  Every result must match the oracle bit for bit."""
    base = _tiny(41, P=400, W=96, H=80)
    cam = base["cameras"][0]
    grads = _grads(80, 96, 5)
    Rs = []
    for mult in (0.5, 3.0, 0.5, 3.0):
        sc = dict(base)
        sc["scales"] = (base["scales"] * mult).astype(np.float32)
        st_o, g_o = run_oracle(sc, cam, grads)
        st_h, g_h = run_hip(sc, cam, grads)
        _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=2e-3)
        Rs.append(int(st_o["R"]))
    assert Rs[1] > 1.5 * Rs[0], Rs   # the second render really overflows the speculative capacity


def test_speculative_forward_with_a_mis_guessed_sort_class_never_walks_an_unsorted_list():
    """The speculative stage 2 also guesses the SORT CLASS from the longest tile list seen for a (P, W, H).  When a later render of that size
    has lists past the guess (here: <= 2048 entries first, then > 2048 and > 16384 in one 32 x 32 image), the sort kernels queued
    speculatively do not cover them, so `point_list` holds whatever the allocator handed out -- and `k_render_fwd` used to walk it: a GPU
    memory fault on ids from recycled memory (round 5, tests/scripts/r5_raster_stress.py seed 203; the Python operator's heap layout had hidden it).
    The freed blocks the binning chunk will be carved from are poisoned with huge ids first; the mis-guessed forward must blend nothing from such
    a list and the exact re-run must match the oracle."""
    dev = torch.device("cuda:0")
    for P, W, expect in ((6000, 128, 2048), (40000, 256, 16384)):
        wide = _tiny(9, P=P, W=W, H=W, deg=0, spread=2.2)             # spread over the whole image: many instances, short lists
        wide["opacities"][:] = 0.02
        cam = wide["cameras"][0]
        grads = _grads(W, W, 9)
        st_o, g_o = run_oracle(wide, cam, grads)
        R_wide = int(st_o["R"])
        assert (st_o["ranges"][:, 1] - st_o["ranges"][:, 0]).max() * 1.25 <= expect      # the guessed class stops at `expect`
        st_h, g_h = run_hip(wide, cam, grads)                         # sets the hint for (P, W, W): capacity ~ R_wide, the lower sort class
        _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=2e-3, n_contrib_outliers=2e-3, img_outliers=2e-3)
        pile = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in wide.items()}
        rng = np.random.default_rng(4)
        pile["means3D"][:, :2] = (rng.normal(size=(P, 2)) * 0.01 + np.array([0.3, 0.3])).astype(np.float32)   # one small pile inside a tile
        pile["means3D"][:, 2] = (3.5 + rng.uniform(-0.5, 0.5, size=P)).astype(np.float32)
        pile["scales"] = (wide["scales"] * 0.2).astype(np.float32)
        st_o, g_o = run_oracle(pile, cam, grads)
        lens = st_o["ranges"][:, 1] - st_o["ranges"][:, 0]
        assert lens.max() > expect and int(st_o["R"]) <= R_wide, (int(lens.max()), int(st_o["R"]), R_wide)   # fits the capacity, not the class
        for _ in range(3):                                            # poison what the caching allocator will hand out next
            junk = [torch.full((n,), 0x7f7f7f7f, dtype=torch.int32, device=dev) for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24)]
            del junk
        st_h, g_h = run_hip(pile, cam, grads)                         # speculative with the stale class -> must not fault -> exact re-run
        _check(compare(st_h, st_o, g_h, g_o, verbose=False), grad_tol=2e-3, n_contrib_outliers=2e-3, img_outliers=2e-3)


def test_many_outstanding_speculative_forwards_then_backward():
    """80 renders of the same size are kept alive (all but the first laid out speculatively, every one in its own
    binning chunk) and differentiated afterwards in reverse order: each gradient must equal the one obtained from a
    render that is differentiated immediately.  Guards the per-chunk capacity bookkeeping of the speculative stage 2."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    sc = syn.scene_c1()
    cam = sc["cameras"][0]
    t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
    P = sc["means3D"].shape[0]
    base = {k: t(sc[k]) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    s = GaussianRasterizationSettings(image_height=128, image_width=128, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(sc["bg"]),
                                      scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
                                      campos=t(cam["campos"]), prefiltered=False, debug=False, confidence=torch.ones((P, 1), device=dev))
    gC = torch.randn(3, 128, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(1))

    def render(k):
        prm = {n: v.clone().requires_grad_(True) for n, v in base.items()}
        with torch.no_grad():
            prm["scales"] *= (1.0 + 0.004 * k)      # a slightly different num_rendered every time
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        color, _, _, _ = GaussianRasterizer(s)(means3D=prm["means3D"], means2D=m2, opacities=prm["opacities"], shs=prm["shs"],
                                               scales=prm["scales"], rotations=prm["rotations"])
        return color, prm

    held = [render(k) for k in range(80)]
    for k in reversed(range(80)):
        color, prm = held[k]
        color.backward(gC)
    for k in (0, 1, 37, 79):
        color, prm = render(k)
        color.backward(gC)
        for n in prm:
            assert torch.equal(prm[n].grad, held[k][1][n].grad), (k, n)


def test_tile_dispatch_order_is_a_permutation_by_image_region():
    """k_tilescan's tile_order: every tile exactly once; workgroup b (observed on XCD b % 8) gets a tile of image region
    b % 8, and inside a region the lists come longest first (256 geometric cost buckets)."""
    import torch
    from diff_gaussian_rasterization import _C
    sc = syn.scene_c2(P=20000, W=400, H=272)
    cam = sc["cameras"][1]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
    E = torch.Tensor([])
    W, H = cam["image_width"], cam["image_height"]
    out = _C.rasterize_gaussians(t(sc["bg"]), t(sc["means3D"]), E, t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, E,
                                 t(cam["viewmatrix"]), t(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, t(sc["shs"]),
                                 sc["sh_degree"], t(cam["campos"]), False, False)
    R, gb, bb, ib = out[0], out[5], out[6], out[7]
    v = _C.chunk_views(sc["means3D"].shape[0], W, H, R, gb, bb, ib)
    order = v["tile_order"].cpu().numpy().astype(np.int64)
    ranges = v["ranges"].cpu().numpy().astype(np.int64)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    assert sorted(order.tolist()) == list(range(T))
    gx = (W + 15) // 16
    region = ((order % gx) // 2 + 3 * ((order // gx) // 2)) % 8       # 2x2-tile blocks dealt round-robin (xcd_region)
    full = 8 * int(np.bincount(region, minlength=8).min())
    assert full >= T - 64 and np.all(region[:full] == np.arange(full) % 8)
    length = ranges[order, 1] - ranges[order, 0]
    for x in range(8):                      # per band: non-increasing up to the bucket resolution (1/16 of a power of two)
        ln = length[:full][x::8].astype(np.float64)
        assert np.all(ln[1:] <= ln[:-1] * 1.07 + 1.0), x


def test_colour_only_backward_equals_zero_depth_and_alpha_gradients():
    """render_bwd is specialised for the usual training step (no gradient for the depth / alpha images: NULL at the
    C-ABI): it must return exactly what explicit all-zero depth / alpha gradients give."""
    import torch
    from diff_gaussian_rasterization import _C
    sc = syn.scene_c2(P=30000, W=320, H=240)
    cam = sc["cameras"][2]
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    E = torch.Tensor([])
    W, H = cam["image_width"], cam["image_height"]
    bg, m3, op, scl, rot, sh = t(sc["bg"]), t(sc["means3D"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), t(sc["shs"])
    vm, pm, cp = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"])
    R, color, depth, alpha, radii, gb, bb, ib = _C.rasterize_gaussians(bg, m3, E, op, scl, rot, 1.0, E, vm, pm, cam["tanfovx"], cam["tanfovy"],
                                                                      H, W, sh, sc["sh_degree"], cp, False, False)
    gC = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) / (H * W)
    z = torch.zeros(1, H, W, device=dev)
    call = lambda gD, gA: _C.rasterize_gaussians_backward(bg, m3, radii, E, scl, rot, 1.0, E, vm, pm, cam["tanfovx"], cam["tanfovy"],
                                                          gC, gD, gA, sh, sc["sh_degree"], cp, gb, R, bb, ib, alpha, False)
    lean, full = call(None, None), call(z, z)
    assert float(lean[3].abs().max()) > 0
    for a, b in zip(lean, full):
        assert torch.equal(a, b)


def test_no_backward_forward_lays_out_a_compact_chunk_that_the_backward_refuses():
    """gvd_raster_expect_backward(0) (advisor finding, round 5): the forward of a no-grad render asks its binning allocator for the layout WITHOUT
    the backward's partial records (24 instead of 220 bytes per instance), renders the same bits, and a backward handed that chunk fails loudly
    instead of reading records that were never laid out."""
    import torch
    from diff_gaussian_rasterization import _C
    sc = syn.scene_c2(P=30000, W=320, H=240)
    cam = sc["cameras"][1]
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    E = torch.Tensor([])
    W, H = cam["image_width"], cam["image_height"]
    bg, m3, op, scl, rot, sh = t(sc["bg"]), t(sc["means3D"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), t(sc["shs"])
    vm, pm, cp = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"])
    fwd = lambda eb: _C.rasterize_gaussians(bg, m3, E, op, scl, rot, 1.0, E, vm, pm, cam["tanfovx"], cam["tanfovy"], H, W, sh, sc["sh_degree"], cp,
                                            False, False, expect_backward=eb)
    L = _C.lib()
    for _ in range(3):      # exact path first, then the speculative one (the capacity then exceeds num_rendered)
        full, lean = fwd(True), fwd(False)
        assert full[0] == lean[0] > 1000
        for a, b in zip(full[1:5], lean[1:5]):
            assert torch.equal(a, b)
        nf, nl = full[6].numel(), lean[6].numel()
        assert nf % 64 == 0 and nl % 4 == 2 and nl < 0.2 * nf
        cap = L.gvd_raster_binning_capacity(nl)
        assert cap >= lean[0] and nl == L.gvd_raster_binning_bytes_no_backward(cap)
        vf, vl = (_C.chunk_views(m3.shape[0], W, H, o[0], o[5], o[6], o[7]) for o in (full, lean))
        R = full[0]
        assert torch.equal(vf["point_list"][:R], vl["point_list"][:R]) and torch.equal(vf["ranges"], vl["ranges"])
    gC = torch.ones(3, H, W, device=dev)
    R, color, depth, alpha, radii, gb, bb, ib = lean
    with pytest.raises(RuntimeError, match="without the backward's partial records"):
        _C.rasterize_gaussians_backward(bg, m3, radii, E, scl, rot, 1.0, E, vm, pm, cam["tanfovx"], cam["tanfovy"], gC, None, None, sh,
                                        sc["sh_degree"], cp, gb, R, bb, ib, alpha, False)
    R, color, depth, alpha, radii, gb, bb, ib = full     # ... and the full chunk of the same render still differentiates
    g = _C.rasterize_gaussians_backward(bg, m3, radii, E, scl, rot, 1.0, E, vm, pm, cam["tanfovx"], cam["tanfovy"], gC, None, None, sh,
                                        sc["sh_degree"], cp, gb, R, bb, ib, alpha, False)
    assert float(g[3].abs().max()) > 0


def test_compiled_operator_equals_the_python_operator():
    """`rasterize_gaussians` runs the compiled torch::autograd::Function of lib/_gvd_raster_torch.so (csrc/raster_torch_ext.cpp); the
    Python `_RasterizeGaussians` over the ctypes entry points is the same operator (and carries debug dumps and the capacity mode).
    Both must return the same bits: images, radii, every gradient (confidence-scaled, screen-space one unscaled), with gradients for a
    subset of the outputs, under no_grad, on a retained graph, and with strided inputs."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings, GaussianRasterizer
    if _C.ext() is None and not os.environ.get("GVD_RASTER_NO_EXT"):     # a tree that was never built: build it now (g++, ~30 s)
        import __graft_entry__ as graft
        graft.build_raster_torch_ext()
        _C._ext = None
    assert _C.ext() is not None, "lib/_gvd_raster_torch.so is not built / did not load"
    dev = torch.device("cuda:0")
    sc = _tiny(21, P=1500, W=160, H=96, deg=2)
    cam = sc["cameras"][0]
    P = 1500
    t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=rg)
    conf = torch.rand(P, 1, device=dev) + 0.5
    st = GaussianRasterizationSettings(96, 160, cam["tanfovx"], cam["tanfovy"], t(sc["bg"]), 1.0, t(cam["viewmatrix"]), t(cam["projmatrix"]),
                                       2, t(cam["campos"]), False, False, conf)
    gC, gD, gA = (torch.randn(3, 96, 160, device=dev), torch.randn(1, 96, 160, device=dev), torch.randn(1, 96, 160, device=dev))

    def run(which, outputs):
        lv = dict(means3D=t(sc["means3D"], True), means2D=torch.zeros(P, 3, device=dev, requires_grad=True), opacities=t(sc["opacities"], True),
                  shs=t(sc["shs"], True)[:, :9].contiguous().detach().requires_grad_(True), scales=t(sc["scales"], True), rotations=t(sc["rotations"], True))
        saved = _C._ext
        if which == "python":
            _C._ext = False
        try:
            c, r, d, a = GaussianRasterizer(st)(**lv)
            outs, grads = {"c": ([c], [gC]), "cd": ([c, d], [gC, gD]), "cda": ([c, d, a], [gC, gD, gA]), "a": ([a], [gA])}[outputs]
            torch.autograd.backward(outs, grads, retain_graph=True)
            g1 = {k: v.grad.clone() for k, v in lv.items()}
            for v in lv.values():
                v.grad = None
            torch.autograd.backward(outs, grads)                      # a second backward over the retained graph
            g2 = {k: v.grad.clone() for k, v in lv.items()}
            with torch.no_grad():
                cn = GaussianRasterizer(st)(**lv)[0]
        finally:
            _C._ext = saved
        assert type(c.grad_fn).__name__ == ("_RasterizeGaussiansBackward" if which == "python" else "RasterFnBackward") or which == "ext", type(c.grad_fn)
        return (c.detach(), r, d.detach(), a.detach(), cn), g1, g2

    for outputs in ("c", "cd", "cda", "a"):
        (o_e, g_e, g_e2), (o_p, g_p, g_p2) = run("ext", outputs), run("python", outputs)
        for x, y in zip(o_e, o_p):
            assert torch.equal(x, y), outputs
        for k in g_e:
            assert torch.equal(g_e[k], g_e2[k]), (outputs, k, "compiled: second backward differs", float((g_e[k] - g_e2[k]).abs().max()))
            assert torch.equal(g_p[k], g_p2[k]), (outputs, k, "python: second backward differs", float((g_p[k] - g_p2[k]).abs().max()))
            assert torch.equal(g_e[k], g_p[k]), (outputs, k, "compiled vs python", float((g_e[k] - g_p[k]).abs().max()))
        assert float(g_e["means3D"].abs().max()) > 0
    # the compiled path refuses what the Python path refuses
    with pytest.raises(RuntimeError, match="ROCm device"):
        dgr.rasterize_gaussians(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 1, 3), torch.Tensor([]), torch.ones(4, 1), torch.ones(4, 3),
                                torch.ones(4, 4), torch.Tensor([]), st)
    with pytest.raises(RuntimeError, match="float32"):
        GaussianRasterizer(st)(means3D=t(sc["means3D"]).double(), means2D=torch.zeros(P, 3, device=dev), opacities=t(sc["opacities"]),
                               shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
