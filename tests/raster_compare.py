"""Shared helpers for the GPU parity tests: run the HIP path through the drop-in operator, run the
CPU oracle on the same inputs, and compare.  `python tests/raster_compare.py` prints a report
(used during bring-up on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run_oracle(sc, cam, grads=None, colors_precomp=None, cov3D_precomp=None):
    from oracle import raster_oracle as ro
    kw = dict(sh_degree=sc["sh_degree"])
    if colors_precomp is None:
        kw["shs"] = sc["shs"]
    else:
        kw["colors_precomp"] = colors_precomp
    if cov3D_precomp is None:
        kw["scales"], kw["rotations"] = sc["scales"], sc["rotations"]
    else:
        kw["cov3D_precomp"] = cov3D_precomp
    st = ro.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], sc["bg"],
                    cam["image_width"], cam["image_height"], cam["tanfovx"], cam["tanfovy"], **kw)
    g = ro.backward(st, *grads) if grads is not None else None
    return st, g


def run_hip(sc, cam, grads=None, colors_precomp=None, cov3D_precomp=None, device="cuda:0", debug=False,
            alpha_override=None):
    """Calls the native boundary directly (no autograd) and returns numpy state + grads."""
    import torch
    from diff_gaussian_rasterization import _C
    dev = torch.device(device)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    empty = torch.Tensor([])
    W, H = cam["image_width"], cam["image_height"]
    sh = t(sc["shs"]) if colors_precomp is None else empty
    col = empty if colors_precomp is None else t(colors_precomp)
    scales = t(sc["scales"]) if cov3D_precomp is None else empty
    rots = t(sc["rotations"]) if cov3D_precomp is None else empty
    cov = empty if cov3D_precomp is None else t(cov3D_precomp)
    bg, m3, op = t(sc["bg"]), t(sc["means3D"]), t(sc["opacities"])
    vm, pm, cp = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"])
    out = _C.rasterize_gaussians(bg, m3, col, op, scales, rots, 1.0, cov, vm, pm, cam["tanfovx"], cam["tanfovy"],
                                 H, W, sh, sc["sh_degree"], cp, False, debug)
    R, color, depth, alpha, radii, gb, bb, ib = out
    P = m3.shape[0]
    st = dict(R=R, color=color.cpu().numpy(), depth=depth.cpu().numpy(), alpha=alpha.cpu().numpy(),
              radii=radii.cpu().numpy())
    if P > 0:
        v = _C.chunk_views(P, W, H, R, gb, bb, ib)
        for k, x in v.items():
            if x is not None:
                st[k] = x.cpu().numpy()
    g = None
    if grads is not None:
        gC, gD, gA = (t(x) for x in grads)
        if alpha_override is not None:  # isolate the backward kernels from forward rounding differences
            alpha = t(alpha_override).reshape(1, H, W)
        _C.KEEP_BACKWARD_INTERNALS = True
        res = _C.rasterize_gaussians_backward(bg, m3, radii, col, scales, rots, 1.0, cov, vm, pm, cam["tanfovx"],
                                              cam["tanfovy"], gC.reshape(3, H, W), gD.reshape(1, H, W), gA.reshape(1, H, W),
                                              sh, sc["sh_degree"], cp, gb, R, bb, ib, alpha, debug)
        names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
        g = {n: x.cpu().numpy() for n, x in zip(names, res)}
        _C.KEEP_BACKWARD_INTERNALS = False
        g["dL_dconic"] = _C.LAST_BACKWARD_INTERNALS.pop("dL_dconic").reshape(-1, 4).cpu().numpy()
        g["dL_ddepths"] = _C.LAST_BACKWARD_INTERNALS.pop("dL_ddepths").cpu().numpy()
    return st, g


def rel_to_max(a, b):
    """max |a-b| relative to max |b| (the gradient tolerance metric, see tests)."""
    if b.size == 0:
        return 0.0 if a.shape == b.shape else float("inf")
    s = float(np.abs(b).max())
    return float(np.abs(a - b).max()) / (s if s > 0 else 1.0)


def rel_elementwise(a, b, floor_frac=1e-2):
    """Largest |a-b| / |b| over the entries with |b| >= floor_frac * max|b| (element-wise relative error above a magnitude
    floor; entries below the floor are covered by rel_to_max)."""
    if b.size == 0:
        return 0.0
    s = float(np.abs(b).max())
    if s == 0:
        return 0.0
    m = np.abs(b) >= floor_frac * s
    return float((np.abs(a - b)[m] / np.abs(b)[m]).max()) if m.any() else 0.0


def threshold_margin(st_o, py, px, upto):
    """Why can `n_contrib` differ at a pixel between two fp32 implementations that agree to an ulp of exp()?  Only through one
    of the blend loop's three comparisons (forward.cu:343-355: power > 0, alpha < 1/255, T (1 - alpha) < 1e-4) landing within
    rounding of its threshold.  Replays the pixel's list in float64 from the ORACLE's state up to list position `upto` and
    returns the smallest relative distance of any compared quantity to its threshold."""
    W = st_o["W"] if "W" in st_o else st_o["color"].shape[-1]
    gx = (W + 15) // 16
    tile = (py // 16) * gx + (px // 16)
    r0, r1 = (int(v) for v in st_o["ranges"][tile])
    T, margin = 1.0, np.inf
    for pos in range(r0, min(r1, r0 + upto + 1)):
        g = int(st_o["point_list"][pos])
        xy = st_o["means2D"][g].astype(np.float64)
        con = st_o["conic_opacity"][g].astype(np.float64)
        dx, dy = xy[0] - px, xy[1] - py
        power = -0.5 * (con[0] * dx * dx + con[2] * dy * dy) - con[1] * dx * dy
        margin = min(margin, abs(power) / max(1.0, abs(con[0] * dx * dx) + abs(con[2] * dy * dy)))
        if power > 0:
            continue
        alpha = min(0.99, con[3] * np.exp(power))
        margin = min(margin, abs(alpha - 1.0 / 255.0) * 255.0)
        if alpha < 1.0 / 255.0:
            continue
        test_T = T * (1.0 - alpha)
        margin = min(margin, abs(test_T - 1e-4) / 1e-4)
        if test_T < 1e-4:
            break
        T = test_T
    return float(margin)


def compare(st_h, st_o, g_h=None, g_o=None, verbose=True):
    """Returns a dict of parity metrics (exact-match booleans and error magnitudes)."""
    P = st_o["P"]
    vis = st_o["radii"] > 0
    rep = {}
    rep["R_equal"] = int(st_h["R"]) == int(st_o["R"])
    rep["radii_equal"] = bool(np.array_equal(st_h["radii"], st_o["radii"]))
    if P > 0 and "depths" in st_h:
        rep["depth_bits_equal"] = bool(np.array_equal(st_h["depths"][vis].view(np.uint32), st_o["depths"][vis].view(np.uint32)))
        rep["means2D_bits_equal"] = bool(np.array_equal(st_h["means2D"][vis].view(np.uint32), st_o["means2D"][vis].view(np.uint32)))
        rep["tiles_touched_equal"] = bool(np.array_equal(st_h["tiles_touched"].view(np.uint32), st_o["tiles_touched"]))
        rep["point_offsets_equal"] = bool(np.array_equal(st_h["point_offsets"].view(np.uint32), st_o["point_offsets"]))
        rep["ranges_equal"] = bool(np.array_equal(st_h["ranges"].view(np.uint32), st_o["ranges"]))
        co_h, co_o = st_h["conic_opacity"][vis], st_o["conic_opacity"][vis]
        rep["conic_bits_equal"] = bool(np.array_equal(co_h.view(np.uint32), co_o.view(np.uint32)))
        rep["conic_max_rel"] = float((np.abs(co_h - co_o) / np.maximum(np.abs(co_o), 1e-30)).max()) if vis.any() else 0.0
        rgb_h, rgb_o = st_h["rgbd"][vis][:, :3], st_o["features"][vis]
        rep["rgb_bits_equal"] = bool(np.array_equal(rgb_h.view(np.uint32), rgb_o.view(np.uint32)))
        rep["rgb_max_abs"] = float(np.abs(rgb_h - rgb_o).max()) if vis.any() else 0.0
        if st_o["_in"]["cov3D_precomp"] is None:
            rep["cov3D_bits_equal"] = bool(np.array_equal(st_h["cov3D"][vis].view(np.uint32), st_o["cov3D"][vis].view(np.uint32)))
        if st_o["R"] > 0 and rep["R_equal"]:
            rep["keys_equal"] = bool(np.array_equal(st_h["point_list_keys"].view(np.uint64), st_o["keys"]))
            rep["point_list_equal"] = bool(np.array_equal(st_h["point_list"].view(np.uint32), st_o["point_list"]))
        nc_h, nc_o = st_h["n_contrib"].view(np.uint32), st_o["n_contrib"]
        rep["n_contrib_mismatch_frac"] = float((nc_h != nc_o).mean())
        # every mismatch must be a threshold straddle: the float64 replay of that pixel has a comparison within fp32 rounding
        # (the transmittance is a product of up to a few hundred fp32 factors: 2e-4 relative) of its threshold
        ys, xs = np.nonzero(nc_h != nc_o)
        worst = 0.0
        for py, px in list(zip(ys, xs))[:64]:
            worst = max(worst, threshold_margin(st_o, int(py), int(px), int(max(nc_h[py, px], nc_o[py, px])) + 1))
        rep["n_contrib_mismatch_worst_threshold_margin"] = worst
    for k in ("color", "depth", "alpha"):
        a, b = st_h[k], st_o[k]
        tol = 1e-4 * np.maximum(1.0, np.abs(b))
        bad = np.abs(a - b) > tol
        rep[k + "_max_abs"] = float(np.abs(a - b).max())
        rep[k + "_outlier_frac"] = float(bad.mean())
    if g_h is not None:
        for k in g_h:
            rep["grad_" + k + "_relmax"] = rel_to_max(g_h[k], g_o[k])
            rep["gradel_" + k] = rel_elementwise(g_h[k], g_o[k])
    if verbose:
        for k, v in rep.items():
            print(f"  {k:32s} {v}")
    return rep


def _main():
    import time
    import synthetic as syn
    for name, sc in (("C1", syn.scene_c1()), ("C2", syn.scene_c2())):
        for ci, cam in enumerate(sc["cameras"][:2]):
            H, W = cam["image_height"], cam["image_width"]
            rng = np.random.default_rng(7 + ci)
            grads = (rng.normal(size=(3, H, W)) / (H * W), rng.normal(size=(H, W)) / (H * W), rng.normal(size=(H, W)) / (H * W))
            t0 = time.time()
            st_o, g_o = run_oracle(sc, cam, grads)
            t1 = time.time()
            st_h, g_h = run_hip(sc, cam, grads, debug=True)
            print(f"== {name} cam{ci}: P={st_o['P']} R={st_o['R']} oracle {t1 - t0:.2f}s hip {time.time() - t1:.2f}s")
            compare(st_h, st_o, g_h, g_o)
            _, g_h2 = run_hip(sc, cam, grads, alpha_override=st_o["alpha"])
            print("  -- backward fed with the oracle's alpha image:")
            for k in g_h2:
                print(f"  grad_{k:27s} {rel_to_max(g_h2[k], g_o[k])}")


if __name__ == "__main__":
    _main()
