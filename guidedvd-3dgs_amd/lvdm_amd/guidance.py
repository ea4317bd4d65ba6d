"""Scene-grounding guidance loss: masked L2 between one decoded x0 frame and the 3DGS
render of that frame, optionally mixed with a structural term.  Restates LossGuidance of
utils/viewcrafter_wrapper.py:47-165: the `recon` term (:145-147) and the `ssim_guidance` mix
0.8 recon + 0.2 sum(1 - ssim_map) (:150-155; the map through the fused SSIM kernels on the GPU), the `lpips_guidance` add-on
`numel * VggLoss * 0.001` (:157-159, used by scripts/run_scannetpp_guidedvd*.sh; lvdm_amd/vgg_loss.py) and the
`scale_guidance_weight` schedule (:88-94, learning_rate_decay :654-691).
"""
import math
import os

import torch
import torch.nn.functional as F


def log_lerp(t, v0, v1):
    """viewcrafter_wrapper.py:654-660."""
    if v0 <= 0 or v1 <= 0:
        raise ValueError(f"Interpolants {v0} and {v1} must be positive.")
    lv0, lv1 = math.log(v0), math.log(v1)
    return math.exp(min(max(t, 0.0), 1.0) * (lv1 - lv0) + lv0)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    """viewcrafter_wrapper.py:663-691: log-linear interpolation from lr_init (step 0) to lr_final (max_steps)."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    else:
        delay_rate = 1.
    return delay_rate * log_lerp(step / max_steps, lr_init, lr_final)


class LossGuidance:
    def __init__(self, ddim_steps, recur_steps=2, iter_steps=0, recon_loss="l2", w_recon_loss=0.5, save_dir=None,
                 ssim_guidance=False, lpips_guidance=False, device="cuda:0", verbose=False, mean_loss=False,
                 scale_guidance_weight=False):
        assert mean_loss is False, "Important to set it to False. "
        self.ssim_guidance, self.lpips_guidance = bool(ssim_guidance), bool(lpips_guidance)
        if self.lpips_guidance:
            from .vgg_loss import VggLoss
            self.lpips_fn = VggLoss(device)
        self.ddim_steps, self.recur_steps, self.iter_steps = ddim_steps, recur_steps, iter_steps
        self.save_dir = self.root_save_dir = save_dir
        if self.root_save_dir is not None:
            os.makedirs(self.root_save_dir, exist_ok=True)
        self.verbose, self.mean_loss = verbose, mean_loss
        self.w_recon = w_recon_loss
        self.scale_guidance_weight = bool(scale_guidance_weight)
        if self.scale_guidance_weight:   # read by the guided sampler (ddim_guidance.py:249-251)
            self.guidance_weight_fn = lambda step: learning_rate_decay(step, lr_init=0.01, lr_final=1.0, max_steps=2500)
        self.guidance_images = self.guidance_masks = self.guidance_depths = None
        self.current_train_iter = 0

    def set_hw(self, H, W):
        self.H, self.W = H, W

    def set_guidance_images(self, imgs):   # [n,3,H,W] 3DGS renders
        self.guidance_images = F.interpolate(imgs, size=(self.H, self.W), mode="bilinear", align_corners=False).clamp(0, 1)

    def set_guidance_masks(self, masks):   # [n,1,H,W]
        self.guidance_masks = F.interpolate(masks, size=(self.H, self.W), mode="nearest")

    def set_guidance_depths(self, depths):
        self.guidance_depths = F.interpolate(depths, size=(self.H, self.W), mode="nearest")

    def __call__(self, diffused_images, ddim_index, batch_idx_start, batch_idx_end):
        """diffused_images [3,1,H,W] in [-1,1] (one decoded frame) -> ({'recon': sum 0.5 (D-G)^2 mask}, mask.sum())."""
        D = ((diffused_images.permute(1, 0, 2, 3) + 1.) / 2.).clamp(0, 1)
        if self.guidance_masks is None:
            mask = torch.ones_like(D)
        else:
            mask = self.guidance_masks[batch_idx_start:batch_idx_end].expand_as(D)
        G = self.guidance_images[batch_idx_start:batch_idx_end]
        loss = (self.w_recon * torch.square(D - G) * mask).sum()
        numel = mask.sum()
        if self.ssim_guidance:   # viewcrafter_wrapper.py:150-155 with loss_utils.ssim_noavg(:84-118): sum over the map
            loss = 0.8 * loss + 0.2 * _one_minus_ssim_sum(D.float(), G.float(), mask)
        if self.lpips_guidance:  # :157-159
            loss = loss + numel * self.lpips_fn(D.float(), G.float(), mask=mask) * 0.001
        return {"recon": loss}, numel

    def frames_loss(self, diffused_images, batch_idx_start, batch_idx_end):
        """The per-frame calls of __call__ over frames [start, end) as ONE set of tensor ops: diffused_images [3, F, H, W] (what one
        decoder pass of the guided sampler holds) -> (sum over the frames of their `recon` losses, per-frame mask sums [F]).  The
        reference's loop (ddim_guidance.py:296-317) runs ~20 elementwise / reduction launches per frame on one 3 x H x W image each
        -- 25 x that per guided step, plus 25 full-size zero-filled gradient buffers from the slice backward: 4-5 ms of a 240 ms
        step at 320x448.  Plain masked-L2 term only; None with the SSIM / perceptual add-ons (the sampler then loops per frame)."""
        if self.ssim_guidance or self.lpips_guidance:
            return None
        D = ((diffused_images.permute(1, 0, 2, 3) + 1.) / 2.).clamp(0, 1)          # [F, 3, H, W]
        if self.guidance_masks is None:
            mask = torch.ones_like(D)
        else:
            mask = self.guidance_masks[batch_idx_start:batch_idx_end].expand_as(D)
        G = self.guidance_images[batch_idx_start:batch_idx_end]
        per_frame = (self.w_recon * torch.square(D - G) * mask).sum(dim=(1, 2, 3))
        return per_frame.sum(), mask.sum(dim=(1, 2, 3))

    def update_save_dir(self, train_iter):
        """viewcrafter_wrapper.py:167-172: one sub-directory per training iteration that runs the diffusion."""
        if self.root_save_dir is not None:
            self.save_dir = os.path.join(self.root_save_dir, "train_iter" + str(train_iter))
            os.makedirs(self.save_dir, exist_ok=True)
        self.current_train_iter = train_iter

    def save_pred_x0(self, pred_x0, ddim_index):
        """The reference encodes an mp4 of the decoded x0 at EVERY DDIM step when `save_dir` is set (viewcrafter_wrapper.py:174-192)
        -- a host stall on the hot loop that the guided sampler therefore only triggers with a save_dir.  The
        tensor is always stashed; with a save_dir the [T, H, W, 3] uint8 frames are written asynchronously-friendly as a .pt
        file per step (mp4 encoding needs torchvision.io / PyAV, which the caller can run offline on these files)."""
        self.last_pred_x0 = (int(ddim_index), pred_x0)
        if self.save_dir is None:
            return
        frames = (torch.clamp(pred_x0[0].permute(1, 2, 3, 0), -1., 1.) + 1.) / 2.
        torch.save((frames.detach() * 255).to(torch.uint8).cpu(), os.path.join(self.save_dir, f"pred_x0_step{int(ddim_index)}.pt"))


def _one_minus_ssim_sum(x, y, mask):
    """sum(1 - ssim_map(x*m + (1-m), y*m + (1-m))) over [1, C, H, W].  GPU: the fused SSIM kernels (value = mean of
    the map, so the sum is numel * (1 - mean)); CPU tensors (tests under ops.use_reference_math) take the explicit
    five-convolution form of loss_utils._ssim_noavg."""
    xm, ym = x * mask + (1 - mask), y * mask + (1 - mask)
    if x.is_cuda:
        import fused_loss
        return x.numel() * (1.0 - fused_loss.ssim(xm, ym.detach()))
    from . import ops
    if not ops._REFERENCE_MATH:
        raise RuntimeError("LossGuidance(ssim_guidance=True): tensors must live on a ROCm device (this build has no CPU path)")
    from math import exp
    g = torch.tensor([exp(-(i - 5) ** 2 / float(2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float32)
    g = g / g.sum()
    C = x.shape[1]
    w = (g[:, None] @ g[None, :])[None, None].expand(C, 1, 11, 11).contiguous().to(x)
    conv = lambda t: F.conv2d(t, w, padding=5, groups=C)
    mu1, mu2 = conv(xm), conv(ym)
    s1, s2, s12 = conv(xm * xm) - mu1 * mu1, conv(ym * ym) - mu2 * mu2, conv(xm * ym) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return (1.0 - m).sum()
