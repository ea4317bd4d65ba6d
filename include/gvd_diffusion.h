/*
 * gvd_diffusion.h -- C-ABI of the hand-written HIP kernels on the ViewCrafter DDIM hot path.
 *
 * Plain C: raw DEVICE pointers, sizes, hipStream_t as void*.  Every function returns 0 on success
 * or a negative code (gvd_diff_last_error() has the message).
 */
#ifndef GVD_DIFFUSION_H_INCLUDED
#define GVD_DIFFUSION_H_INCLUDED

#ifdef __cplusplus
extern "C" {
#endif

/* Multi-head attention forward, softmax(Q K^T * scale) V, flash style (no N x N matrix in HBM), MFMA
 * 32x32x16 f16/bf16 with fp32 accumulation and fp32 online softmax.
 * Replaces xformers.ops.memory_efficient_attention(q, k, v) at
 *   third_party/ViewCrafter/lvdm/modules/attention.py:175,187 (un-vendored pip dependency `xformers`,
 *   unpinned in requirements.txt:26); its semantics are pinned by the in-tree explicit path :101-135.
 * Layout (no head-split copies): q, out [B, Nq, H*D], k, v [B, Nk, H*D], 16-bit elements, contiguous;
 * head h owns channels [h*D, (h+1)*D).  D must be 64 (every U-Net head).  Any Nq, Nk >= 1. */
int gvd_attention_fwd(const void* q, const void* k, const void* v, void* out,
                      int B, int H, int Nq, int Nk, int D, float scale, int is_bf16, void* stream);

/* Same kernel with explicit addressing: element (batch b, row n, head h, channel d) of q/out lives at
 * b*q_bs + n*q_rs + h*D + d, of k/v at b*kv_bs + n*kv_rs + h*D + d (strides in elements, multiples of 8).
 * Lets temporal attention (sequence = the T frames of one pixel) read a token-major [T, pixels, H*D] tensor in
 * place (bs = H*D, rs = pixels*H*D) instead of materialising the '(b h w) t c' transposes of
 * lvdm/modules/attention.py:370-407. */
int gvd_attention_fwd_strided(const void* q, const void* k, const void* v, void* out,
                              int B, int H, int Nq, int Nk, int D, float scale,
                              long long q_bs, long long q_rs, long long kv_bs, long long kv_rs,
                              float* lse, int is_bf16, void* stream);

/* Attention backward for the guided sampler's autograd pass (ddim_guidance.py:318-345 differentiates pred_x0 w.r.t.
 * x_t through every attention layer; xformers' memory_efficient_attention backward in the reference).
 * Inputs: q, k, v, out (forward result), d_out, and lse = the [B, H, Nq] fp32 log2-domain log-sum-exp the forward
 * wrote (pass a buffer as `lse` above).  Outputs dq [same addressing as q], dk, dv [same addressing as k, v].
 * delta: [B, H, Nq] fp32 device scratch (rowsum(d_out * out)).  Flash style, deterministic (no atomics):
 * one pass over the K/V tiles accumulates dK, dV; a second pass over the Q tiles accumulates dQ. */
int gvd_attention_bwd_strided(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                              const float* lse, float* delta, void* dq, void* dk, void* dv,
                              int B, int H, int Nq, int Nk, int D, float scale,
                              long long q_bs, long long q_rs, long long kv_bs, long long kv_rs,
                              int is_bf16, void* stream);

/* One complete no-grad DDIM update for the v-parameterisation, batch 1, fp32 latents of n elements:
 *   v      = e_uncond + cfg_scale (e_cond - e_uncond)
 *   v      = phi v std(e_cond)/std(v) + (1 - phi) v              (phi = guidance_rescale; skipped if 0)
 *   eps    = sqrt_ac_t v + sqrt_1mac_t x ;   x0 = (sqrt_ac_t x - sqrt_1mac_t v) * x0_rescale
 *   x_prev = sqrt_a_prev x0 + dir_coef eps + sigma_t * temperature * noise
 * Replaces the ~20 elementwise/reduction launches of DDIMSampler.p_sample_ddim
 *   (lvdm/models/samplers/ddim.py:208-280, rescale_noise_cfg utils_diffusion.py:147-158).
 * ws: 4 doubles of device scratch (sum / sum of squares of e_cond and of v), any content on entry. */
int gvd_ddim_step(const float* x, const float* e_cond, const float* e_uncond, const float* noise,
                  float* x_prev, float* x0, double* ws, long long n,
                  float cfg_scale, float guidance_rescale, float sqrt_ac_t, float sqrt_1mac_t,
                  float sqrt_a_prev, float dir_coef, float sigma_t, float x0_rescale, float temperature,
                  void* stream);

/* GroupNorm with fp32 statistics and optional fused SiLU on 16-bit activations, forward.
 *   channels_last == 0 : x, y [N][C][S]   (S = product of the spatial/temporal dims)
 *   channels_last == 1 : x, y [N][S][C]   (token-major; C % 8 == 0)
 * gamma, beta: fp32 [C].  stats: device scratch of 16*N*G + 8*N*C bytes (fp64 group sums, then the per-(n,c) fp32 affine).  Semantics: GroupNormSpecific
 * (lvdm/basics.py:76-86: statistics in fp32, result cast back) followed by nn.SiLU where the reference has
 * `normalization -> SiLU` (openaimodel3d.py:152-156,177-182,259-268,542-546; ae_modules.py nonlinearity). */
int gvd_group_norm(const void* x, void* y, const float* gamma, const float* beta, double* stats,
                   int N, int C, long long S, int G, float eps, int silu, int channels_last, int is_bf16, void* stream);

/* The two phases of gvd_group_norm, for statistics that span more than this device's slice (frame-sharded temporal
 * layers: the 5-D GroupNorm of TemporalConvBlock / TemporalTransformer, openaimodel3d.py:259-268, attention.py:370,
 * normalises over all T frames).  Call _stats on the local slice, all-reduce (sum) the first 2*N*G doubles of `stats`
 * across the shard group, then _apply with S_total = the global element count per (sample, channel). */
int gvd_group_norm_stats(const void* x, double* stats, int N, int C, long long S, int G, int channels_last, int is_bf16,
                         void* stream);
int gvd_group_norm_apply(const void* x, void* y, const float* gamma, const float* beta, double* stats,
                         int N, int C, long long S, long long S_total, int G, float eps, int silu, int channels_last,
                         int is_bf16, void* stream);

/* GroupNorm (+SiLU) backward w.r.t. the input (the guided sampler differentiates pred_x0 w.r.t. x_t only; weights are
 * frozen -- ddim_guidance.py:318-345 requests inputs=x).  x, dy, dx share the forward's layout; fwd_stats is the scratch
 * buffer gvd_group_norm filled for the same x (group sums + per-(n,c) affine); scratch: 16*N*G + 8*N*C bytes. */
int gvd_group_norm_bwd(const void* x, const void* dy, void* dx, const float* gamma, const double* fwd_stats,
                       double* scratch, int N, int C, long long S, int G, float eps, int silu, int channels_last,
                       int is_bf16, void* stream);

/* Two-phase form of gvd_group_norm_bwd (all-reduce the first 2*N*G doubles of `scratch` between the calls). */
int gvd_group_norm_bwd_stats(const void* x, const void* dy, const float* gamma, const double* fwd_stats, double* scratch,
                             int N, int C, long long S, int G, int silu, int channels_last, int is_bf16, void* stream);
int gvd_group_norm_bwd_apply(const void* x, const void* dy, void* dx, const double* fwd_stats, double* scratch,
                             int N, int C, long long S, long long S_total, int G, float eps, int silu, int channels_last,
                             int is_bf16, void* stream);

/* LayerNorm over the last dim of [M, C] 16-bit rows, fp32 statistics, gamma/beta in the same 16-bit type as x.
 * Replaces nn.LayerNorm in BasicTransformerBlock (lvdm/modules/attention.py:283-285).  C % 8 == 0, C <= 2048. */
int gvd_layer_norm(const void* x, void* y, const void* gamma, const void* beta, long long M, int C, float eps,
                   int is_bf16, void* stream);

/* GEGLU gate: y[m, c] = h[m, c] * gelu(h[m, C + c]) (exact erf GELU) for h [M, 2C] -> y [M, C], 16-bit.
 * Replaces `x, gate = self.proj(x).chunk(2, dim=-1); return x * F.gelu(gate)` (lvdm/modules/attention.py:420-423). */
int gvd_geglu(const void* h, void* y, long long M, int C, int is_bf16, void* stream);

/* Input gradients of the two ops above for the guided sampler's autograd pass (weights frozen):
 *   gvd_layer_norm_bwd : dx [M, C] from x, dy and gamma (statistics recomputed per row)
 *   gvd_geglu_bwd      : dh [M, 2C] from h [M, 2C] and dy [M, C] */
int gvd_layer_norm_bwd(const void* x, const void* dy, const void* gamma, void* dx, long long M, int C, float eps,
                       int is_bf16, void* stream);
int gvd_geglu_bwd(const void* h, const void* dy, void* dh, long long M, int C, int is_bf16, void* stream);

const char* gvd_diff_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
