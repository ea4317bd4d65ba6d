"""CPU restatement of the reference's SSIM (utils/loss_utils.py:36-82) in numpy/scipy, float64 accumulation.

TEST INFRASTRUCTURE ONLY.  PINNED: tests/test_fused_loss.py checks it against tests/golden/ssim_ref.npz, which the
reference's own loss_utils.ssim produced (tests/golden/make_golden_ssim.py).  Follows the reference literally: fp32
1-D gaussian normalised in fp32 (:36-38), 2-D window = outer product in fp32 (:40-44), five zero-padded 11x11
correlations per plane (:64-72), the map (:77), its mean (:81-84).  The gradient w.r.t. img1 is the analytic adjoint of
exactly that graph."""
from math import exp

import numpy as np
from scipy.signal import correlate2d

C1, C2 = 0.01 ** 2, 0.03 ** 2


def window(window_size=11, sigma=1.5):
    g = np.array([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], np.float32)
    g = (g / g.sum(dtype=np.float32)).astype(np.float32)
    return np.outer(g, g).astype(np.float32)


def _conv(p, w):
    return correlate2d(p.astype(np.float64), w.astype(np.float64), mode="same", boundary="fill", fillvalue=0.0)


def ssim_map_and_grad(x, y):
    """x, y [planes, H, W] -> (map [planes,H,W], function g_map -> dL/dx for an upstream gradient of the map)."""
    w = window()
    maps, ctx = [], []
    for a, b in zip(x, y):
        a, b = a.astype(np.float64), b.astype(np.float64)
        mu1, mu2 = _conv(a, w), _conv(b, w)
        ex2, ey2, exy = _conv(a * a, w), _conv(b * b, w), _conv(a * b, w)
        s1, s2, s12 = ex2 - mu1 ** 2, ey2 - mu2 ** 2, exy - mu1 * mu2
        A, B, C, D = 2 * mu1 * mu2 + C1, 2 * s12 + C2, mu1 ** 2 + mu2 ** 2 + C1, s1 + s2 + C2
        m = A * B / (C * D)
        maps.append(m)
        dmu1 = 2 * mu2 * (B - A) / (C * D) - m * 2 * mu1 * (D - C) / (C * D)
        ctx.append((a, b, dmu1, -m / D, 2 * A / (C * D)))

    def grad(g_map):
        out = []
        for (a, b, dmu1, dex2, dexy), gm in zip(ctx, g_map):
            # the window is symmetric: the adjoint of a zero-padded correlation is the same correlation
            out.append(_conv(gm * dmu1, w) + 2 * a * _conv(gm * dex2, w) + b * _conv(gm * dexy, w))
        return np.stack(out)

    return np.stack(maps), grad


def ssim(x, y, size_average=True):
    """x, y [C,H,W] or [N,C,H,W] -> (value, d value / d x); size_average=False gives one mean per batch element and
    the gradient of their SUM."""
    x, y = np.asarray(x), np.asarray(y)
    shp = x.shape
    if x.ndim == 3:
        x, y = x[None], y[None]
    N, C, H, W = x.shape
    m, grad = ssim_map_and_grad(x.reshape(N * C, H, W), y.reshape(N * C, H, W))
    if size_average:
        val = m.mean()
        g = grad(np.full(m.shape, 1.0 / m.size))
    else:
        val = m.reshape(N, -1).mean(1)
        g = grad(np.full(m.shape, 1.0 / (C * H * W)))
    return val, g.reshape(shp)
