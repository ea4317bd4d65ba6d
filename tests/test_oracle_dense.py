"""Oracle (C, fp32, tiled) vs an independent dense torch-float64 autograd restatement.

Pins the oracle's forward and its hand-derived backward formulas without the reference
binary (which cannot run here: CUDA-only).  Tolerances: forward 2e-5 abs (fp32 vs fp64),
gradients 2e-4 relative to the tensor's max magnitude.
"""
import math

import numpy as np
import pytest
import torch

import synthetic as syn
from dense_ref import dense_render


def _tiny_scene(seed, P=70, W=48, H=32, deg=3, precomp=False):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.0, 1.0, size=(P, 3))
    xyz[:, 2] = xyz[:, 2] * 1.5 + 3.5
    xyz[:3, 2] = -1.0  # a few behind the camera (near-culled)
    scales = np.exp(rng.normal(math.log(0.12), 0.4, size=(P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rng.uniform(0.9, 1.1, size=(P, 1)).astype(np.float32)  # kernel must not normalise (quirk 3)
    opac = (1 / (1 + np.exp(-rng.normal(0.5, 1.2, size=(P, 1))))).astype(np.float32)
    sh = rng.normal(0, 0.25, size=(P, 16, 3)).astype(np.float32)
    sh[:, 0] = syn.rgb2sh(rng.uniform(0, 1, size=(P, 3)))
    cam = syn.make_camera(syn.look_at((0.1, -0.05, 0.0), (0.0, 0.1, 3.0)), math.radians(70), math.radians(50), W, H)
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    return dict(means3D=xyz.astype(np.float32), scales=scales, rotations=q, opacities=opac, shs=sh,
                sh_degree=deg, bg=bg, cam=cam, W=W, H=H)


def _run(oracle, sc, seed):
    cam = sc["cam"]
    W, H = sc["W"], sc["H"]
    st = oracle.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                        sc["bg"], W, H, cam["tanfovx"], cam["tanfovy"], shs=sc["shs"], scales=sc["scales"],
                        rotations=sc["rotations"], sh_degree=sc["sh_degree"])
    rng = np.random.default_rng(seed + 100)
    gC = rng.normal(size=(3, H, W))
    gD = rng.normal(size=(H, W)) * 0.3
    gA = rng.normal(size=(H, W))
    g = oracle.backward(st, gC, gD, gA)

    t = lambda a, rg=True: torch.tensor(np.asarray(a, np.float64), requires_grad=rg)
    m3, sc_, ro, op, sh = t(sc["means3D"]), t(sc["scales"]), t(sc["rotations"]), t(sc["opacities"][:, 0]), t(sc["shs"])
    color, depth, alpha = dense_render(m3, sc_, ro, op, sh, t(cam["viewmatrix"], False), t(cam["projmatrix"], False),
                                       t(cam["campos"], False), t(sc["bg"], False), W, H, cam["tanfovx"], cam["tanfovy"],
                                       sc["sh_degree"], st["rects"], st["radii"])
    loss = (color * torch.tensor(gC)).sum() + (depth * torch.tensor(gD)).sum() + (alpha * torch.tensor(gA)).sum()
    loss.backward()
    return st, g, (color, depth, alpha), dict(means3D=m3.grad, scales=sc_.grad, rotations=ro.grad, opacity=op.grad, sh=sh.grad)


@pytest.mark.parametrize("seed,deg", [(0, 3), (1, 0), (2, 2), (3, 1)])
def test_oracle_matches_dense_autograd(oracle, seed, deg):
    sc = _tiny_scene(seed, deg=deg)
    st, g, (color, depth, alpha), ag = _run(oracle, sc, seed)
    assert (st["radii"] > 0).sum() > 30
    np.testing.assert_allclose(st["color"], color.detach().numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(st["depth"][0], depth.detach().numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(st["alpha"][0], alpha.detach().numpy(), atol=2e-5, rtol=0)

    def close(a, b, name):
        b = b.numpy()
        scale = np.abs(b).max()
        err = np.abs(a - b).max() / scale
        assert err < 2e-4, f"{name}: rel-to-max err {err:.3e} (scale {scale:.3e})"

    close(g["dL_dmeans3D"], ag["means3D"], "means3D")
    close(g["dL_dscales"], ag["scales"], "scales")
    close(g["dL_drotations"], ag["rotations"], "rotations")
    close(g["dL_dopacity"][:, 0], ag["opacity"], "opacity")
    close(g["dL_dsh"], ag["sh"], "sh")
    # entries above the active degree stay exactly zero (quirk 7)
    ncoef = (deg + 1) ** 2
    assert np.all(g["dL_dsh"][:, ncoef:] == 0)
