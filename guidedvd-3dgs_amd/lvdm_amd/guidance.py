"""Scene-grounding guidance loss: masked L2 between one decoded x0 frame and the 3DGS
render of that frame.  Restates LossGuidance of utils/viewcrafter_wrapper.py:47-165 (the `recon` term;
SSIM / LPIPS add-ons :150-159 pull in torchvision-VGG and are a 'next' row, N4).
"""
import torch
import torch.nn.functional as F


class LossGuidance:
    def __init__(self, ddim_steps, recur_steps=2, iter_steps=0, recon_loss="l2", w_recon_loss=0.5, save_dir=None,
                 ssim_guidance=False, lpips_guidance=False, device="cuda:0", verbose=False, mean_loss=False,
                 scale_guidance_weight=False):
        assert mean_loss is False, "Important to set it to False. "
        if ssim_guidance or lpips_guidance:
            raise NotImplementedError("SSIM / LPIPS guidance terms are not part of this build (SURVEY 8f N4)")
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
        loss = self.w_recon * torch.square(D - self.guidance_images[batch_idx_start:batch_idx_end]) * mask
        return {"recon": loss.sum()}, mask.sum()

    def update_save_dir(self, train_iter):
        self.current_train_iter = train_iter

    def save_pred_x0(self, pred_x0, ddim_index):
        """The reference writes an mp4 of the decoded x0 at EVERY DDIM step (viewcrafter_wrapper.py:174-192) -- a
        pure host stall on the hot loop.  Kept as a hook: tensors are stashed, not encoded."""
        self.last_pred_x0 = (int(ddim_index), pred_x0)
