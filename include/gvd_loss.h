/*
 * gvd_loss.h -- C-ABI of the fused SSIM loss kernels (the per-iteration loss around the rasterizer).
 *
 * Replaces utils/loss_utils.py:46-82 (`ssim` / `_ssim`: five 11x11 depthwise gaussian convolutions + ~15 elementwise
 * launches, and their autograd backward) with one forward and one backward kernel.
 * Plain C: raw DEVICE pointers (fp32), sizes, hipStream_t as void*.  Returns 0 or a negative code (gvd_loss_last_error()).
 * Images are [N][C][H][W] contiguous (N*C independent planes).  `gauss` = the 11 normalised 1-D window weights on the
 * HOST (the caller builds them exactly as loss_utils.gaussian(11, 1.5) does); padding is zero padding of 5, like
 * F.conv2d(padding=window_size // 2).
 */
#ifndef GVD_LOSS_H_INCLUDED
#define GVD_LOSS_H_INCLUDED
#ifdef __cplusplus
extern "C" {
#endif

/* Number of float partial sums gvd_ssim_forward writes (one per 16x16 tile per plane). */
long long gvd_ssim_partial_count(int planes, int H, int W);

/* ssim_map(img1, img2) summed per tile: partials[plane][tile] (fixed order -> the caller's sum is reproducible).
 * If dmaps != NULL it receives the three per-pixel derivative planes [3][planes][H][W] that gvd_ssim_backward needs
 * (d ssim_map / d mu1, / d E[x^2], / d E[xy]).  ssim_map may be NULL, or [planes][H][W] to receive the map itself. */
int gvd_ssim_forward(const float* img1, const float* img2, const float* gauss, int planes, int H, int W,
                     float* partials, float* dmaps, float* ssim_map, void* stream);

/* dL/dimg1 [planes][H][W] for L = sum_plane plane_scale[plane] * sum_pixels ssim_map  (plane_scale on the DEVICE,
 * e.g. upstream_grad / (C*H*W) for the mean).  img2 is treated as a constant. */
int gvd_ssim_backward(const float* img1, const float* img2, const float* gauss, const float* dmaps, const float* plane_scale,
                      int planes, int H, int W, float* d_img1, void* stream);

/* The training loss of train_baseline.py / train_guidedvd.py:339-340 in one pass,
 *     loss = (1 - lambda_dssim) * mean|img1 - img2| + lambda_dssim * (1 - mean ssim_map),
 * out3 (DEVICE) = {loss, L1 mean, ssim mean}.  partials: 2 * gvd_ssim_partial_count floats of device scratch;
 * dmaps as in gvd_ssim_forward (needed for the backward, may be NULL for evaluation only). */
int gvd_photometric_forward(const float* img1, const float* img2, const float* gauss, int planes, int H, int W,
                            float lambda_dssim, float* partials, float* dmaps, float* out3, void* stream);

/* d loss / d img1 times the DEVICE scalar `upstream`. */
int gvd_photometric_backward(const float* img1, const float* img2, const float* gauss, const float* dmaps,
                             const float* upstream, int planes, int H, int W, float lambda_dssim, float* d_img1,
                             void* stream);

const char* gvd_loss_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
