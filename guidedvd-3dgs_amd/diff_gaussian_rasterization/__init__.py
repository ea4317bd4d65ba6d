"""MI355X-native drop-in for the reference package `diff_gaussian_rasterization`
(submodules/diff-gaussian-rasterization-confidence/diff_gaussian_rasterization/__init__.py).

Same public surface, so `gaussian_renderer.render()` (gaussian_renderer/__init__.py:14,42-102) and the
training drivers run unmodified:

    GaussianRasterizationSettings   13-field NamedTuple incl. `confidence`           (ref :161-174)
    GaussianRasterizer              nn.Module with .forward(...) and .markVisible()   (ref :176-225)
    rasterize_gaussians(...)        functional entry                                  (ref :20-42)

Behaviour kept from the reference:
  * returns (color[3,H,W], radii[P] int32, depth[1,H,W], alpha[1,H,W]);
  * exactly one of shs / colors_precomp and one of (scales, rotations) / cov3D_precomp, else Exception;
  * backward multiplies every gradient except the screen-space one (means2D) by the per-Gaussian
    `confidence` [P,1] (ref :147-157) -- the fork's "confidence" feature;
  * with settings.debug the native call's arguments are snapshotted to snapshot_fw.dump /
    snapshot_bw.dump when it raises (ref :83-90, :135-142).

The compute is the HIP library behind `_C` (libgvd_raster.so, C-ABI in include/gvd_raster.h), reached through a compiled torch
autograd operator (lib/_gvd_raster_torch.so, csrc/raster_torch_ext.cpp) or, for debug dumps, the sync-free capacity mode and the
reference's three pybind-level entry points, through ctypes (`_C.py`).  There is no CPU or eager fallback.
"""
import functools
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


_ABSENT = torch.Tensor([])   # one shared "not given" tensor (never written)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    confidence: torch.Tensor


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_native(fn, args, debug, dump_name, what):
    """Runs one native entry point; in debug mode keeps a CPU copy of the inputs and dumps it on failure."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(f"\nAn error occured in {what}. Please forward {dump_name} for debugging.")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        s = raster_settings
        native_args = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                       s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width,
                       sh, s.sh_degree, s.campos, s.prefiltered, s.debug)
        # no input wants a gradient (torch.no_grad() / evaluation renders): autograd will never call backward on these
        # buffers, so the forward need not prepare the backward's partial records (gvd_raster.h: gvd_raster_expect_backward)
        fwd = functools.partial(_C.rasterize_gaussians, expect_backward=any(ctx.needs_input_grad[:8]))
        (num_rendered, color, depth, alpha, radii,
         geom_buf, binning_buf, img_buf) = _call_native(fwd, native_args, s.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = s
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geom_buf, binning_buf, img_buf, alpha)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # untouched outputs arrive as None -> NULL at the C-ABI, no zero fills
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        s = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
         geom_buf, binning_buf, img_buf, alpha) = ctx.saved_tensors
        # outputs the loss did not touch arrive as None (== zero gradient; the native side takes NULL)
        if grad_color is None:
            grad_color = torch.zeros_like(alpha).expand(3, -1, -1).contiguous()
        native_args = (s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                       s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, grad_color, grad_depth, grad_alpha,
                       sh, s.sh_degree, s.campos, geom_buf, ctx.num_rendered, binning_buf, img_buf, alpha, s.debug)
        # The reference multiplies every gradient except the screen-space one by `confidence` [P,1] in
        # Python (ref :147-157, seven elementwise launches); here the gather kernel applies it.
        fused = functools.partial(_C.rasterize_gaussians_backward, confidence=s.confidence)
        (g_means2D, g_colors, g_opacity, g_means3D,
         g_cov3D, g_sh, g_scales, g_rot) = _call_native(fused, native_args, s.debug, "snapshot_bw.dump", "backward")
        return (g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rot, g_cov3D, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    s = raster_settings
    ext = _C.ext()
    if ext is not None and not s.debug and _C._CAPACITY == 0:
        # the same operator as _RasterizeGaussians below, compiled (csrc/raster_torch_ext.cpp): one call in, no Python in the backward
        color, radii, depth, alpha = ext.rasterize(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                                   s.bg, s.scale_modifier, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                                                   s.image_height, s.image_width, s.sh_degree, s.campos, s.prefiltered, s.confidence)
        return color, radii, depth, alpha
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: view-space z > 0.2 (rasterizer_impl.cu:54-66)."""
        with torch.no_grad():
            s = self.raster_settings
            return _C.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        absent = _ABSENT  # empty tensor == "not given" at the native boundary
        pick = lambda t: absent if t is None else t
        return rasterize_gaussians(means3D, means2D, pick(shs), pick(colors_precomp), opacities, pick(scales),
                                   pick(rotations), pick(cov3D_precomp), self.raster_settings)
