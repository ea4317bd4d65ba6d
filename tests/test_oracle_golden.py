"""CPU-only: the oracle against committed golden vectors.

raster_ref_python.npz comes from the REFERENCE'S OWN Python (eval_sh, build_scaling_rotation /
strip_symmetric; see tests/golden/make_golden_raster.py) evaluated in float64 -> the oracle's fp32
SH->RGB and cov3D must agree to fp32 rounding.  raster_c1_oracle.npz pins the oracle itself
(keys / lists / ranges bit-exact, images and gradients to 1e-6) against regressions.
"""
import ctypes
import os

import numpy as np

import synthetic as syn

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_sh_to_rgb_matches_reference_eval_sh(oracle):
    z = np.load(os.path.join(G, "raster_ref_python.npz"))
    sh, dirs = np.ascontiguousarray(z["sh"]), z["dirs"]
    N = sh.shape[0]
    campos = np.array([0.3, -0.2, 0.1], np.float32)
    pos = np.ascontiguousarray(campos[None] + 2.5 * dirs, dtype=np.float32)
    L = oracle.lib()
    for deg in range(4):
        rgb = np.zeros((N, 3), np.float32)
        cl = np.zeros((N, 3), np.uint8)
        L.gvdo_sh_to_rgb_batch(N, deg, _p(pos), _p(campos), _p(sh), _p(rgb), _p(cl))
        ref = z[f"rgb_deg{deg}"]
        # direction recomputed from fp32 positions: ~1e-7 relative perturbation of dirs
        np.testing.assert_allclose(rgb, ref, atol=3e-6, rtol=0)
        assert np.array_equal(cl.astype(bool), (ref == 0.0)) or np.abs(rgb[cl.astype(bool) != (ref == 0.0)]).max() < 3e-6


def test_cov3d_matches_reference_python(oracle):
    z = np.load(os.path.join(G, "raster_ref_python.npz"))
    scales, rots = np.ascontiguousarray(z["scales"]), np.ascontiguousarray(z["rots"])
    N = scales.shape[0]
    cov = np.zeros((N, 6), np.float32)
    oracle.lib().gvdo_cov3d_batch(N, _p(scales), ctypes.c_float(float(z["scale_modifier"])), _p(rots), _p(cov))
    ref = z["cov3D"]
    np.testing.assert_allclose(cov, ref, rtol=2e-5, atol=1e-6 * np.abs(ref).max())


def test_oracle_c1_regression(oracle):
    z = np.load(os.path.join(G, "raster_c1_oracle.npz"))
    sc = syn.scene_c1()
    cam = sc["cameras"][0]
    st = oracle.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], sc["bg"],
                        cam["image_width"], cam["image_height"], cam["tanfovx"], cam["tanfovy"], shs=sc["shs"],
                        scales=sc["scales"], rotations=sc["rotations"], sh_degree=sc["sh_degree"])
    assert st["R"] == int(z["R"])
    for k in ("radii", "keys", "point_list", "ranges", "tiles_touched", "point_offsets", "n_contrib"):
        assert np.array_equal(st[k], z[k]), k
    for k in ("color", "depth", "alpha"):
        np.testing.assert_allclose(st[k], z[k], atol=1e-6, rtol=0)
    g = oracle.backward(st, z["gC"], z["gD"], z["gA"])
    for k in g:
        if k in z.files:
            np.testing.assert_allclose(g[k], z[k], rtol=1e-5, atol=1e-7 * np.abs(z[k]).max(), err_msg=k)


def test_higher_msb_matches_reference_semantics(oracle):
    # rasterizer_impl.cu:35-50: smallest b with n < 2^b (n>0), so the sort covers 32+b key bits
    f = oracle.lib().gvdo_higher_msb
    for n in (1, 2, 3, 4, 63, 64, 65, 1200, 8160, 65535, 65536):
        b = f(ctypes.c_uint32(n))
        assert (n >> b) == 0 and (b == 0 or (n >> (b - 1)) != 0)


def test_keys_sorted_and_ranges_partition(oracle):
    sc = syn.scene_c2(P=20000, W=320, H=240)
    cam = sc["cameras"][2]
    st = oracle.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], sc["bg"],
                        cam["image_width"], cam["image_height"], cam["tanfovx"], cam["tanfovy"], shs=sc["shs"],
                        scales=sc["scales"], rotations=sc["rotations"], sh_degree=1)
    k = st["keys"]
    assert np.all(k[1:] >= k[:-1])
    r = st["ranges"]
    lens = (r[:, 1] - r[:, 0]).astype(np.int64)
    assert lens.sum() == st["R"]
    tiles = (k >> np.uint64(32)).astype(np.int64)
    assert np.array_equal(np.bincount(tiles, minlength=r.shape[0]), lens)


def test_synthetic_cameras_match_reference_camera_class():
    """synthetic.make_camera (what every raster test and bench.py feed the rasterizer) against the reference's own
    PseudoCamera (scene/cameras.py:72-93: getWorld2View2, the non-standard getProjectionMatrix, the transposes, the
    full_proj_transform product, camera_center), recorded by tests/golden/make_golden_camera.py."""
    z = np.load(os.path.join(G, "raster_camera_ref.npz"))
    c1, c2 = syn.scene_c1(), syn.scene_c2(P=16)
    cams = [("c1_0", c1["cameras"][0])] + [(f"c2_{i}", c) for i, c in enumerate(c2["cameras"])]
    for tag, cam in cams:
        np.testing.assert_allclose(cam["viewmatrix"], z[f"{tag}_world_view_transform"], rtol=0, atol=1e-6, err_msg=tag)
        np.testing.assert_allclose(cam["projmatrix"], z[f"{tag}_full_proj_transform"], rtol=0, atol=2e-6, err_msg=tag)
        np.testing.assert_allclose(cam["campos"], z[f"{tag}_camera_center"], rtol=0, atol=2e-6, err_msg=tag)
        P = z[f"{tag}_projection_matrix"]          # transposed non-standard projection: P^T[2,2] = P^T[2,3] = 1, no near/far terms
        assert P[2, 2] == 1.0 and P[2, 3] == 1.0 and P[3, 3] == 0.0 and P[3, 2] == 0.0
        assert abs(P[0, 0] - 1.0 / cam["tanfovx"]) < 1e-6 and abs(P[1, 1] - 1.0 / cam["tanfovy"]) < 1e-6


def test_python_sh_and_python_cov_branches_end_to_end(oracle):
    """gaussian_renderer/__init__.py:66-88: with `convert_SHs_python` / `compute_cov3D_python` the reference evaluates SH ->
    colour and the 3-D covariance in Python and hands `colors_precomp` / `cov3D_precomp` to the operator.  Feeding the
    REFERENCE's Python results (golden) through those operator inputs must render what the in-kernel SH / covariance path
    renders -- the two front ends of the rasterizer pinned against each other through the reference's own formulas."""
    z = np.load(os.path.join(G, "raster_camera_ref.npz"))
    sc = syn.scene_c1()
    cam = sc["cameras"][0]
    args = (sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], sc["bg"],
            cam["image_width"], cam["image_height"], cam["tanfovx"], cam["tanfovy"])
    for deg in (0, 3):
        a = oracle.forward(*args, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], sh_degree=deg)
        b = oracle.forward(*args, colors_precomp=np.ascontiguousarray(z[f"c1_colors_precomp_deg{deg}"], dtype=np.float32),
                           cov3D_precomp=np.ascontiguousarray(z["c1_cov3D_precomp"], dtype=np.float32), sh_degree=deg)
        # covariances agree to fp32 rounding, so a radius can differ by one at a ceil() boundary for a rare Gaussian
        assert np.mean(a["radii"] != b["radii"]) < 5e-3 and np.abs(a["radii"] - b["radii"]).max() <= 1
        for k in ("color", "depth", "alpha"):
            np.testing.assert_allclose(b[k], a[k], rtol=0, atol=2e-5, err_msg=f"{k} deg{deg}")
