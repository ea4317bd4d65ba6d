"""Scene-grounding guidance loss: masked L2 between one decoded x0 frame and the 3DGS
render of that frame, optionally mixed with a structural term.  Restates LossGuidance of
utils/viewcrafter_wrapper.py:47-165: the `recon` term (:145-147) and the `ssim_guidance` mix
0.8 recon + 0.2 sum(1 - ssim_map) (:150-155; the map through the fused SSIM kernels on the GPU).  The LPIPS add-on
(:157-159) needs torchvision-VGG weights and stays a 'next' row (N4).
"""
import torch
import torch.nn.functional as F


class LossGuidance:
    def __init__(self, ddim_steps, recur_steps=2, iter_steps=0, recon_loss="l2", w_recon_loss=0.5, save_dir=None,
                 ssim_guidance=False, lpips_guidance=False, device="cuda:0", verbose=False, mean_loss=False,
                 scale_guidance_weight=False):
        assert mean_loss is False, "Important to set it to False. "
        if lpips_guidance:
            raise NotImplementedError("the LPIPS guidance term is not part of this build (SURVEY 8f N4)")
        self.ssim_guidance = bool(ssim_guidance)
        if scale_guidance_weight:
            raise NotImplementedError("scale_guidance_weight needs utils.stepfun.learning_rate_decay (out of scope)")
        self.ddim_steps, self.recur_steps, self.iter_steps = ddim_steps, recur_steps, iter_steps
        self.save_dir = self.root_save_dir = save_dir
        self.verbose, self.mean_loss = verbose, mean_loss
        self.w_recon = w_recon_loss
        self.scale_guidance_weight = False
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
        if self.ssim_guidance:   # viewcrafter_wrapper.py:150-155 with loss_utils.ssim_noavg(:84-118): sum over the map
            loss = 0.8 * loss + 0.2 * _one_minus_ssim_sum(D.float(), G.float(), mask)
        return {"recon": loss}, mask.sum()

    def update_save_dir(self, train_iter):
        self.current_train_iter = train_iter

    def save_pred_x0(self, pred_x0, ddim_index):
        """The reference writes an mp4 of the decoded x0 at EVERY DDIM step (viewcrafter_wrapper.py:174-192) -- a
        pure host stall on the hot loop.  Kept as a hook: tensors are stashed, not encoded."""
        self.last_pred_x0 = (int(ddim_index), pred_x0)


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
