"""Per-iteration image losses around the rasterizer, drop-in for `utils/loss_utils.py`:

    from fused_loss import l1_loss, l1_loss_mask, l2_loss, ssim          # train_guidedvd.py:26

`ssim` (loss_utils.py:46-82) runs as one fused HIP forward kernel and one backward kernel (csrc/ssim.hip, C-ABI
include/gvd_loss.h) instead of five depthwise 11x11 convolutions plus ~15 elementwise launches and their autograd
graph.  Same signature and value; the gradient is produced for img1 (the render) only -- the ground-truth image is a
constant in every call site of the reference.  No CPU path: CPU tensors raise.
"""
import ctypes
import os
from math import exp

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GVD_LOSS_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libgvd_loss.so")  # env: A/B and sanitizer builds
_LIB = None


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOOP = _Noop()


def _on(dev):
    """Device guard that is free when `dev` is already the current device (torch.cuda.device() costs ~25 us per use)."""
    idx = dev.index
    return _NOOP if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(dev)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: build the HIP extensions first "
                               f"(python -c 'import __graft_entry__ as g; g.build()')")
        L = ctypes.CDLL(_LIB_PATH)
        L.gvd_loss_last_error.restype = ctypes.c_char_p
        L.gvd_ssim_partial_count.restype = ctypes.c_longlong
        _LIB = L
    return _LIB


def l1_loss(network_output, gt, return_map=False):          # loss_utils.py:18-22
    d = torch.abs(network_output - gt)
    return d if return_map else d.mean()


def l1_loss_mask(network_output, gt, mask=None):            # loss_utils.py:24-28
    if mask is None:
        return l1_loss(network_output, gt)
    return torch.abs((network_output - gt) * mask).sum() / mask.sum()


def l2_loss(network_output, gt, return_map=False):          # loss_utils.py:30-34
    d = (network_output - gt) ** 2
    return d if return_map else d.mean()


def gaussian(window_size, sigma):                           # loss_utils.py:36-38 (fp32 normalisation, like torch.Tensor)
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=torch.float32)
    return g / g.sum()


_GAUSS11 = (ctypes.c_float * 11)(*gaussian(11, 1.5).tolist())


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"gvd_loss error {rc}: {lib().gvd_loss_last_error().decode()}")


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, per_batch):
        x = img1.detach().contiguous().float()
        y = img2.detach().contiguous().float()
        N, C, H, W = x.shape
        planes = N * C
        need_grad = img1.requires_grad
        L = lib()
        partials = torch.empty(L.gvd_ssim_partial_count(planes, H, W), dtype=torch.float32, device=x.device)
        dmaps = torch.empty((3, planes, H, W), dtype=torch.float32, device=x.device) if need_grad else None
        P = ctypes.c_void_p
        with _on(x.device):
            _check(L.gvd_ssim_forward(P(x.data_ptr()), P(y.data_ptr()), _GAUSS11, planes, H, W, P(partials.data_ptr()),
                                      P(dmaps.data_ptr() if need_grad else None), P(None),
                                      P(_stream())))
        per_plane = partials.view(planes, -1).sum(1)          # fixed summation order: reproducible
        ctx.per_batch, ctx.shape = per_batch, (N, C, H, W)
        if need_grad:
            ctx.save_for_backward(x, y, dmaps)
        if per_batch:
            return per_plane.view(N, C).sum(1) / float(C * H * W)
        return per_plane.sum() / float(planes * H * W)

    @staticmethod
    def backward(ctx, g):
        x, y, dmaps = ctx.saved_tensors
        N, C, H, W = ctx.shape
        planes = N * C
        if ctx.per_batch:
            scale = (g.float() / float(C * H * W)).repeat_interleave(C).contiguous()
        else:
            scale = (g.float() / float(planes * H * W)).expand(planes).contiguous()
        d = torch.empty_like(x)
        P = ctypes.c_void_p
        with _on(x.device):
            _check(lib().gvd_ssim_backward(P(x.data_ptr()), P(y.data_ptr()), _GAUSS11, P(dmaps.data_ptr()), P(scale.data_ptr()),
                                           planes, H, W, P(d.data_ptr()), P(_stream())))
        return d, None, None


def ssim(img1, img2, mask=None, window_size=11, size_average=True):
    """loss_utils.py:46-82.  img [C,H,W] or [N,C,H,W]; returns the mean of the SSIM map (size_average) or one mean per
    batch element."""
    if window_size != 11:
        raise NotImplementedError("fused ssim: window_size 11 (every call site of the reference uses the default)")
    if not img1.is_cuda or not img2.is_cuda:
        raise RuntimeError("fused_loss.ssim: tensors must live on a ROCm device (this build has no CPU path)")
    if img2.requires_grad:
        raise NotImplementedError("fused ssim: gradient w.r.t. the second image is not implemented (it is the ground truth)")
    if mask is not None:                                    # loss_utils.py:50-52
        img1 = img1 * mask + (1 - mask)
        img2 = img2 * mask + (1 - mask)
    squeeze = img1.dim() == 3
    if squeeze:
        img1, img2 = img1[None], img2[None]
    out = _SSIM.apply(img1, img2, not size_average)
    return out


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        x = image.detach().contiguous().float()
        y = gt.detach().contiguous().float()
        if x.dim() == 3:
            x, y = x[None], y[None]
        N, C, H, W = x.shape
        planes = N * C
        need_grad = image.requires_grad
        L = lib()
        partials = torch.empty(2 * L.gvd_ssim_partial_count(planes, H, W), dtype=torch.float32, device=x.device)
        dmaps = torch.empty((3, planes, H, W), dtype=torch.float32, device=x.device) if need_grad else None
        out3 = torch.empty(3, dtype=torch.float32, device=x.device)
        P = ctypes.c_void_p
        with _on(x.device):
            _check(L.gvd_photometric_forward(P(x.data_ptr()), P(y.data_ptr()), _GAUSS11, planes, H, W, ctypes.c_float(lambda_dssim),
                                             P(partials.data_ptr()), P(dmaps.data_ptr() if need_grad else None),
                                             P(out3.data_ptr()), P(_stream())))
        ctx.cfg = (planes, H, W, float(lambda_dssim), image.shape)
        if need_grad:
            ctx.save_for_backward(x, y, dmaps)
        ctx.mark_non_differentiable(out3)
        return out3[0], out3

    @staticmethod
    def backward(ctx, g, _g_stats):
        x, y, dmaps = ctx.saved_tensors
        planes, H, W, lam, shape = ctx.cfg
        d = torch.empty_like(x)
        g = g.detach().float().contiguous()
        P = ctypes.c_void_p
        with _on(x.device):
            _check(lib().gvd_photometric_backward(P(x.data_ptr()), P(y.data_ptr()), _GAUSS11, P(dmaps.data_ptr()), P(g.data_ptr()),
                                                  planes, H, W, ctypes.c_float(lam), P(d.data_ptr()),
                                                  P(_stream())))
        return d.view(shape), None, None


def photometric_loss(image, gt_image, lambda_dssim=0.2):
    """The two lines train_guidedvd.py:339-340 / train_baseline.py in one fused pass:

        Ll1  = l1_loss_mask(image, gt_image)
        loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim(image, gt_image))

    Returns (loss, stats) with stats = device tensor [loss, Ll1, ssim] (no gradient) for logging."""
    if not image.is_cuda or not gt_image.is_cuda:
        raise RuntimeError("fused_loss.photometric_loss: tensors must live on a ROCm device (this build has no CPU path)")
    if gt_image.requires_grad:
        raise NotImplementedError("fused photometric loss: the ground-truth image is a constant")
    return _Photometric.apply(image, gt_image, float(lambda_dssim))
