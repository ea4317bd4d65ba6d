"""A deterministic stand-in for the NATIVE half of the rasterizer (`_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`,
`_C.mark_visible`), used to pin the PYTHON half -- the autograd operator of diff_gaussian_rasterization/__init__.py -- to the reference's own wrapper
on the CPU: tests/golden/make_golden_raster_wrapper.py drives the REFERENCE's wrapper with it, tests/test_raster_wrapper_golden.py drives this
repository's.  Every output depends on every argument SLOT (slot-weighted sums), so an argument landing in the wrong slot, a gradient returned in the
wrong position, or a gradient scaled (or not) by the confidence changes the numbers.  The backward honours the kernel contract of this repository's
native library when it is handed `confidence=` (every gradient except the screen-space one multiplied by it, include/gvd_raster.h); the reference's
wrapper never passes it and does that multiplication in Python (reference __init__.py:147-157)."""
import torch


def _describe(a):
    if torch.is_tensor(a):
        return ("T", tuple(a.shape), str(a.dtype), round(float(a.double().sum()), 6) if a.numel() else 0.0)
    return ("S", type(a).__name__, a if not isinstance(a, float) else round(a, 9))


def _mix(args):
    """One scalar that depends on which slot every tensor / number sits in."""
    tot = 0.0
    for i, a in enumerate(args):
        if torch.is_tensor(a):
            if a.numel() and a.dtype != torch.uint8:
                tot += (i + 1) * 0.01 * float(a.double().mean())
        elif isinstance(a, (int, float)) and not isinstance(a, bool):
            tot += (i + 1) * 0.001 * float(a)
    return tot


class FakeBackend:
    def __init__(self):
        self.calls = []

    def rasterize_gaussians(self, *args, **kw):
        assert len(args) == 19, len(args)                       # rasterize_points.h:17-38
        self.calls.append(("fwd", [_describe(a) for a in args]))
        means3D, H, W = args[1], int(args[12]), int(args[13])
        P = means3D.shape[0]
        m = _mix(args)
        ramp = torch.arange(H * W, dtype=torch.float32).view(1, H, W) / (H * W)
        color = torch.cat([ramp * (m + c) for c in range(3)], 0)
        depth, alpha = ramp * (m - 0.5), ramp * 0.25 + m
        radii = (torch.arange(P, dtype=torch.int32) % 7)
        buf = lambda n, k: (torch.arange(n, dtype=torch.int64) * k % 251).to(torch.uint8)
        return 3 * P + 1, color, depth, alpha, radii, buf(64, 3), buf(96, 5), buf(80, 7)

    def rasterize_gaussians_backward(self, *args, confidence=None, **kw):
        assert len(args) == 24, len(args)                       # rasterize_points.h:41-65
        self.calls.append(("bwd", [_describe(a) for a in args]))
        means3D, sh = args[1], args[15]
        P = means3D.shape[0]
        M = sh.shape[1] if sh.numel() else 0
        m = _mix(args)
        shapes = [(P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, M, 3), (P, 3), (P, 4)]   # means2D, colors, opacity, means3D, cov3D, sh, scales, rotations
        out = []
        for j, shp in enumerate(shapes):
            n = 1
            for d in shp:
                n *= d
            g = (torch.arange(n, dtype=torch.float32).view(shp) / max(n, 1) + 1.0) * (m + 0.1 * (j + 1))
            if confidence is not None and j != 0:               # the native library's contract with `confidence`
                c = confidence.view(P, *([1] * (len(shp) - 1)))
                g = g * c
            out.append(g)
        return tuple(out)

    def mark_visible(self, *args):
        assert len(args) == 3
        self.calls.append(("vis", [_describe(a) for a in args]))
        return (torch.arange(args[0].shape[0]) % 3 == 0)


def scene(seed=5, P=11, M=16, H=6, W=8):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(P=P, M=M, H=H, W=W, means3D=r(P, 3), means2D=r(P, 3) * 0.0, shs=r(P, M, 3), colors=r(P, 3).abs(), opacities=r(P, 1).sigmoid(),
                scales=r(P, 3).abs(), rotations=torch.nn.functional.normalize(r(P, 4), dim=1), cov3D=r(P, 6), bg=torch.tensor([0.1, 0.2, 0.3]),
                view=r(4, 4), proj=r(4, 4), campos=r(3), confidence=torch.rand(P, 1, generator=g) + 0.25,
                wc=r(3, H, W), wd=r(1, H, W), wa=r(1, H, W))


def drive(module, backend, sc, case):
    """One forward + backward of `module.GaussianRasterizer` (the reference's package or this repository's) on the scene; case 'sh' = SH colours +
    scale / rotation, 'pre' = precomputed colours + covariance.  -> (outputs, gradients by input name)."""
    S = module.GaussianRasterizationSettings(image_height=sc["H"], image_width=sc["W"], tanfovx=0.7, tanfovy=0.6, bg=sc["bg"], scale_modifier=1.25,
                                              viewmatrix=sc["view"], projmatrix=sc["proj"], sh_degree=3 if case == "sh" else 0, campos=sc["campos"],
                                              prefiltered=False, debug=False, confidence=sc["confidence"])
    leaf = lambda k: sc[k].clone().requires_grad_(True)
    inp = dict(means3D=leaf("means3D"), means2D=leaf("means2D"), opacities=leaf("opacities"))
    if case == "sh":
        inp.update(shs=leaf("shs"), scales=leaf("scales"), rotations=leaf("rotations"))
    else:
        inp.update(colors_precomp=leaf("colors"), cov3D_precomp=leaf("cov3D"))
    color, radii, depth, alpha = module.GaussianRasterizer(S)(**inp)
    loss = (color * sc["wc"]).sum() + (depth * sc["wd"]).sum() + (alpha * sc["wa"]).sum()
    loss.backward()
    return ({"color": color.detach(), "radii": radii, "depth": depth.detach(), "alpha": alpha.detach()},
            {k: v.grad for k, v in inp.items()})
