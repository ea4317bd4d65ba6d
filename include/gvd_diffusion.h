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

/* gvd_attention_fwd_strided with (i) separate addressing for out (o_bs, o_rs) -- q, k, v may then be column blocks of one packed
 * [rows, q | k | v] projection (row stride 3 H D) read in place while out is dense; kv_bs = 0 shares one K / V (the frame-invariant
 * text / image context of attention.py:129-142) between all batch entries without expanding it -- and (ii) an accumulate form:
 * accum != NULL (addressed like out; may alias out): out = accum + accum_scale * O, which lands the image-token branch of the
 * cross-attention on the text branch's result (`out + image_cross_attention_scale * out_ip`) without an elementwise pass. */
int gvd_attention_fwd_ex(const void* q, const void* k, const void* v, void* out,
                         int B, int H, int Nq, int Nk, int D, float scale,
                         long long q_bs, long long q_rs, long long kv_bs, long long kv_rs, long long o_bs, long long o_rs,
                         const void* accum, float accum_scale, float* lse, int is_bf16, void* stream);

/* Attention backward for the guided sampler's autograd pass (ddim_guidance.py:318-345 differentiates pred_x0 w.r.t.
 * x_t through every attention layer; xformers' memory_efficient_attention backward in the reference).
 * Inputs: q, k, v, out (forward result), d_out, and lse = the [B, H, Nq] fp32 log2-domain log-sum-exp the forward
 * wrote (pass a buffer as `lse` above).  Outputs dq [same addressing as q], dk, dv [same addressing as k, v].
 * delta: [B, H, Nq] fp32 device scratch (rowsum(d_out * out)).  Flash style, deterministic (no atomics):
 * one pass over the K/V tiles accumulates dK, dV; a second pass over the Q tiles accumulates dQ.  dk == dv == NULL: keys and
 * values carry no gradient (the frame-invariant context of the cross-attention, attention.py:86-99): dQ only. */
int gvd_attention_bwd_strided(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                              const float* lse, float* delta, void* dq, void* dk, void* dv,
                              int B, int H, int Nq, int Nk, int D, float scale,
                              long long q_bs, long long q_rs, long long kv_bs, long long kv_rs,
                              int is_bf16, void* stream);

/* gvd_attention_bwd_strided with separate addressing (o_bs, o_rs) for out / d_out: q, k, v -- and dq, dk, dv, which are addressed
 * like them -- may then be column blocks of one packed [rows, q | k | v] tensor, so the gradient of a fused q|k|v projection is
 * written in place (no slice / concatenate copies under autograd). */
int gvd_attention_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                         const float* lse, float* delta, void* dq, void* dk, void* dv,
                         int B, int H, int Nq, int Nk, int D, float scale,
                         long long q_bs, long long q_rs, long long kv_bs, long long kv_rs, long long o_bs, long long o_rs,
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
/* torch.cat([a, b], channel dim) of two token-major tensors a [N][S][Ca], b [N][S][Cb] (the U-Net decoder's skip concatenation,
 * openaimodel3d.py:592) fused with gvd_group_norm_stats of the result for the GroupNorm that follows (the first norm of the ResBlock the
 * concatenation feeds, openaimodel3d.py:152): writes out [N][S][Ca + Cb] and the first 2*N*G doubles of `stats` in one pass over the
 * sources.  Ca, Cb multiples of 8, (Ca + Cb) % G == 0.  Form the affine with gvd_group_norm_coef(stats, NULL, 1, 1, ...). */
int gvd_cat2_group_norm_stats(const void* xa, int Ca, const void* xb, int Cb, void* out, double* stats, int N, long long S, int G,
                              int is_bf16, void* stream);
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
/* ... with the gradient that reaches the same tensor along a RESIDUAL branch added in (fp32 sum, one rounding):
 * dx = GroupNorm-backward(x, dy) + add.  `add` has dx's layout (nullptr: none).  The reference leaves this sum to autograd's
 * accumulation where a tensor feeds both a norm and a `+ x` (openaimodel3d.py:232-236,270-279; attention.py:296-310): a separate
 * elementwise kernel over two 16-bit gradients per fork. */
int gvd_group_norm_bwd_apply_add(const void* x, const void* dy, const void* add, void* dx, const double* fwd_stats, double* scratch,
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
/* gvd_layer_norm_bwd with the residual branch's gradient added in: dx = LayerNorm-backward(x, dy) + add  (`x + attn(LN(x))`,
 * attention.py:241-244; nullptr: none). */
int gvd_layer_norm_bwd_add(const void* x, const void* dy, const void* gamma, const void* add, void* dx, long long M, int C,
                           float eps, int is_bf16, void* stream);

/* ---- implicit-GEMM convolutions on MFMA (csrc/conv_mfma.hip) -------------------------------------------------------
 * One kernel family for the 3x3 convolutions of the U-Net ResBlock / Upsample (openaimodel3d.py:51-106,210-236), the VAE
 * ResnetBlock / Upsample / conv_out (ae_modules.py:112-128,151-210,575-578) and the (3,1,1) temporal convolution of
 * TemporalConvBlock (openaimodel3d.py:239-279).  These replace torch.nn.Conv2d / Conv3d (MIOpen) + the separate
 * GroupNorm32+SiLU passes in front of them + the elementwise adds behind them.
 *
 *   out = conv(act(x)) + bias + add_nc[n] + residual,   act(v) = silu?(a[n,c] v + b[n,c])  (zero padding after act)
 *
 * mode 0 (spatial 3x3, stride 1, pad 1):  x [N][H_in][W_in][Cin], out [N][H][W][Cout] (16-bit, token-major); H_in, W_in are
 *        derived (pass 0).  upsample = 1: H_in = H/2, W_in = W/2 and the input is nearest-upsampled x2 on the fly (Upsample,
 *        openaimodel3d.py:80-106); upsample = 2: the input is ZERO-STUFFED x2 (odd rows / columns are zero) -- with the
 *        transposed, tap-flipped weights this is the input gradient of a stride-2 convolution.
 * mode 1 (temporal 3 taps, pad 1 in t):   x [H = samples][T = N][W = pixels][Cin], out [samples][T][pixels][Cout] (H = 1: one
 *        video).  The (3,1,1) convolution treats pixels independently, so the samples of a batch (the CFG pair) are extra pixel
 *        tiles of ONE launch; coef (coef_per_n = 1: [samples][Cin]) and the statistics stay per sample.
 * mode 2 (spatial 3x3, stride 2, pad 1: U-Net Downsample, openaimodel3d.py:51-77) and
 * mode 3 (spatial 3x3, stride 2 on an input padded by one zero row / column at the bottom / right only: VAE Downsample,
 *        ae_modules.py:90-109):  x [N][H_in][W_in][Cin] with H = (H_in + pad_lo - 2) / 2 + 1 (pad_lo = 1 / 0), same for W.
 * w_packed: weights in the layout gvd_conv_config describes:  [ceil(Cout/BN)][ceil(Cin/32)][taps][BN][4][8] 16-bit,
 *        element (co_tile, chunk, tap, r, s, j) = W[co_tile*BN + r][chunk*32 + (s ^ ((r >> 2) & 3))*8 + j][tap]
 *        (zero outside Cout x Cin; tap = ky*3 + kx, or kt).  The input gradient of the same convolution is this entry
 *        point with W transposed in (Cout, Cin) and flipped in the taps.
 * coef: fp32 (a, b) pairs [N][Cin] (coef_per_n = 1) or [Cin] (coef_per_n = 0), the per-(sample, channel) affine of the
 *        GroupNorm in front (the second half of the buffer gvd_group_norm_coef / gvd_group_norm fill); NULL = no prologue.
 * bias fp32 [Cout] | NULL;  add_nc 16-bit [N][Cout] | NULL (mode 0 only: ResBlock `h + emb_out[:, :, None, None]`);
 * residual: 16-bit, layout of out | NULL.
 * stats: NULL, or fp64 accumulators [stats_replicas][Nstat][groups][2] (zeroed by the caller) that receive the sum and
 *        sum of squares of the ROUNDED outputs per (sample, group) -- Nstat = N in mode 0, the samples H in mode 1 -- i.e. the first pass
 *        of the next GroupNorm, spread over `stats_replicas` copies to keep the atomics apart (block b adds to copy b % R).
 * Cin % 8 == 0.  16-bit element type: fp16 or bf16 (is_bf16), fp32 accumulation either way.
 * mode 4 (round 4): nearest x2 upsampling + 3x3 evaluated as FOUR 2x2 convolutions of the input map, one per output phase: H, W are
 * the (even) OUTPUT dims, x is [N][H/2][W/2][Cin], w_packed holds the four phase weight sets [4][co tiles][chunks][4 taps][BN][4][8]
 * (taps of the 3x3 kernel that read the same input pixel summed: kernel index sets {0}, {1,2} for phase 0 and {0,1}, {2} for phase 1
 * per dimension; tap k of phase a reads input pixel i + a + k - 1); prologue / epilogue terms as in mode 0, `upsample` = 0.
 * mode 5: the input gradient of mode 4: x = the gradient [N][H][W][Cin] at the UPSAMPLED resolution (H, W even), out
 * [N][H/2][W/2][Cout]; w_packed [co tiles][4 x chunks(Cin)][4 taps][BN][4][8]: the four phase images g[2 j + b] are consecutive runs
 * of input-channel chunks (each padded to 32 channels), tap k of phase b is K_u with u = 2 k - b, K_u = the transposed 3x3 taps summed
 * over the index sets {2}, {1,2}, {0,1}, {0} for u = -1, 0, 1, 2 (per dimension). */
int gvd_conv_mfma(const void* x, const void* w_packed, const float* coef, int coef_per_n, const float* bias, const void* add_nc,
                  const void* residual, void* out, double* stats, int stats_replicas, int groups, int mode, int N, int H, int W,
                  int H_in, int W_in, int Cin, int Cout, int upsample, int silu, int is_bf16, void* stream);

/* The input gradient of a convolution whose input was silu?(GroupNorm(x)) -- the autograd backward of the fused
 * `GroupNorm32 -> SiLU -> conv` pair of ResBlock / TemporalConvBlock / the VAE resnet blocks (openaimodel3d.py:259-268, 152-156,
 * ae_modules.py:90-128) inside ddim_guidance.py:318-345.  d_act = conv(g, w_packed_bwd) exactly as gvd_conv_mfma computes it
 * (mode 0 | 1; Cin = channels of g, Cout = channels of the norm), AND the epilogue accumulates, from the rounded d_act and the
 * norm's input `norm_x` (layout of d_act), the two per-(sample, group) sums gvd_group_norm_bwd_apply needs:
 *   bwd_stats[r][n][g] += ( sum gamma_c dz ,  sum gamma_c dz x ),   dz = d_act * (norm_silu ? silu'(a x + b) : 1)
 * -- the statistics pass of the GroupNorm backward (two full reads of x and d_act) disappears.  norm_coef: the norm's forward
 * affine (a, b) [N][Cout] (norm_coef_per_n = 1) or [Cout] (0), NULL allowed without SiLU; norm_gamma fp32 [Cout].
 * bwd_stats: fp64 [stats_replicas][Nstat][groups][2], zeroed by the caller; merge the replicas with gvd_group_norm_merge into
 * the first 2 N G doubles of the scratch gvd_group_norm_bwd_apply takes.  Cout % 8 == 0. */
int gvd_conv_mfma_norm_bwd(const void* g, const void* w_packed_bwd, void* d_act, double* bwd_stats, int stats_replicas, int groups,
                           int mode, int N, int H, int W, int Cin, int Cout, const void* norm_x, const float* norm_coef,
                           int norm_coef_per_n, const float* norm_gamma, int norm_silu, int is_bf16, void* stream);

/* Frame sheets for maps smaller than a convolution tile (the 5 x 7 / 10 x 14 / 9 x 16 latents of the U-Net's deepest level,
 * openaimodel3d.py:110-220 at 1/8 of the latent): the N maps [N][H][W][C] of one launch are laid out as ONE image
 * [R (H + 1) - 1][Q (W + 1) - 1][C], R = ceil(N / Q), map n at cell (n / Q, n % Q), a zero row / column between neighbours (the
 * zero padding both share), so that gvd_conv_mfma(mode 0, N = 1) on the sheet equals the per-map convolution at the cells.
 * _in: fills the whole sheet (zeros included) and applies act(v) = silu?(a[n,c] v + b[n,c]) when coef != NULL (the fp32 (a, b) pairs
 * of gvd_group_norm_coef -- the prologue the convolution would have run); _out: out[n] = sum over `slices` sheets (1: a plain
 * gvd_conv_mfma result) of cell n + bias + add_nc[n] + residual[n] (fp32 [C] | NULL, 16-bit [N][C] | NULL, 16-bit [N][H][W][C] | NULL;
 * fp32 sum, one rounding).  C % 8 == 0. */
int gvd_conv_sheet_in(const void* x, void* sheet, const float* coef, int silu, int N, int H, int W, int C, int Q, int is_bf16,
                      void* stream);
int gvd_conv_sheet_out(const void* sheet, int slices, void* out, const float* bias, const void* add_nc, const void* residual,
                       double* stats, int groups, int N, int H, int W, int C, int Q, int is_bf16, void* stream);

/* Split-K form of gvd_conv_mfma for launches with fewer workgroups than the chip has slots and a long reduction (the deepest U-Net
 * level: K = 9 x 1280 ... 9 x 2560 over 35-150 pixels per frame; the temporal (3,1,1) form at 35 / 144 pixels): the input channels
 * are cut into `ksplit` runs of 32-channel chunks, slice z (blockIdx.z) convolves its run and writes its 16-bit partial sums to
 * partials + z * (elements of the output); *n_slices receives the number of slices written (<= ksplit).  Stride-1 modes (0, 1), the
 * GroupNorm(+SiLU) prologue as in gvd_conv_mfma, no bias / add / residual / statistics: those belong to the sum,
 *   gvd_conv_sum_slices:  out[r][c] = sum_z partials[z][r][c] + bias[c] + residual[r][c]        (fp32 sum, one rounding)
 *   (stats != NULL: also the sum / sum of squares of the rounded outputs per (sample, group) -- fp64 [n_stat][groups][2], zeroed by
 *    the caller, rows = n_stat x rows per sample: what gvd_conv_mfma's epilogue leaves for the next GroupNorm; same for the sheet form)
 * or, for a frame sheet, gvd_conv_sheet_out with slices = *n_slices (slice stride = the sheet's elements). */
int gvd_conv_mfma_splitk(const void* x, const void* w_packed, const float* coef, int coef_per_n, void* partials, int ksplit,
                         int* n_slices, int mode, int N, int H, int W, int Cin, int Cout, int silu, int is_bf16, void* stream);
int gvd_conv_sum_slices(const void* partials, int slices, void* out, const float* bias, const void* residual, double* stats,
                        int groups, int n_stat, long long rows, int C, int is_bf16, void* stream);

/* stats[n][g][0..1] = sum over replicas r and m < merge of partial[r][n*merge + m][g][0..1]  (N outputs). */
int gvd_group_norm_merge(double* stats, const double* partial, int replicas, int merge, int N, int G, void* stream);

/* Tile configuration gvd_conv_mfma uses for a problem: BN = output channels per workgroup (the packing granule of
 * w_packed), pixels per workgroup tile, tile width (16 | 32; 0 in mode 1). */
int gvd_conv_config(int mode, int N, int H, int W, int Cin, int Cout, int* block_n, int* tile_pixels, int* tile_width);

/* Finish GroupNorm statistics into the norm state the other entry points read: stats = [N][G][2] fp64 sums followed by the
 * fp32 (a, b) affine [N][C].  partial != NULL: first stats[n][g] = sum over replicas r and m < merge of
 * partial[r][n*merge + m][g] (what gvd_conv_mfma accumulated; merge = T turns per-frame sums into the per-video sums of the
 * temporal norms).  partial == NULL: stats already holds the sums (gvd_group_norm_stats).  S_total = elements per channel. */
int gvd_group_norm_coef(double* stats, const double* partial, int replicas, int merge, const float* gamma, const float* beta,
                        int N, int C, long long S_total, int G, float eps, void* stream);

const char* gvd_diff_last_error(void);

/* Profiling aid (no reference counterpart): one empty single-thread launch of `k_profile_marker` on `stream`.  bench.py brackets its
 * timed region with two of them when GVD_BENCH_MARKERS is set, and tests/scripts/prof_summary.py then counts only the launches between
 * the first and the last marker of a rocprofv3 kernel trace (model construction, weight packing and warm-up steps are left out). */
int gvd_profile_marker(int tag, void* stream);

/* NT GEMM on MFMA with the transformer-side epilogues (csrc/gemm_mfma.hip):
 *     Y[b][m][n] = epilogue( alpha * sum_k X[b][m][k] * W[b][n][k] )         X, W, Y 16-bit, fp32 accumulation
 * Replaces every nn.Linear / 1x1 convolution on token rows of the U-Net, the VAE attention block and the once-per-video encoders
 * (F.linear -> hipBLASLt in the reference's stack): lvdm/modules/attention.py:53-57,76,86-99,144,212-246,415-442,
 * lvdm/modules/networks/openaimodel3d.py:109-236,360-387, lvdm/modules/networks/ae_modules.py:26-78.
 * ldx / ldw / ldy: row strides, stride_*: batch strides (elements, multiples of 8; a batch stride may be 0).  K % 8 == 0, N % 8 == 0.
 * Epilogue, in this order (every part optional):
 *   LayerNorm fold   row_stats [batch][M] (mean, rstd) pairs (gvd_row_stats) and col_sum [N]:  v = rstd[m] * (v - mean[m] * col_sum[n])
 *                    -- with W pre-multiplied by the norm's weight this IS  LayerNorm(x) W^T  (attention.py:283-285 + the Linear);
 *   bias [N]         v += bias[n]   (for the fold: sum_k beta[k] W[n][k] + the Linear's bias);
 *   geglu            W rows come in blocks of 32 = [16 value rows | 16 gate rows] of 16 consecutive outputs, each half in natural
 *                    order (a lane of the 16 x 16 x 32 MFMA accumulator layout then owns value and gate of 4 consecutive outputs):
 *                    out[m][j] = value * gelu_erf(gate), Y has N / 2 columns (attention.py:415-423), rounded like the unfused pair.
 *                    N % 32 == 0.  gvd_gemm_geglu_layout() names the row order the library expects (1 = this one; the 32 x 32 x 16
 *                    kernels of rounds 3-5 used 0: row c = 8 rg + 4 hi + e of a half held output 8 hi + 4 rg + e);
 *   residual         Y += R[b][m][n] (row stride ldr, batch stride stride_r), the 16-bit sum of the rounded GEMM result and R. */
int gvd_gemm_nt(const void* x, long long ldx, long long stride_x, const void* w, long long ldw, long long stride_w,
                void* y, long long ldy, long long stride_y, int M, int N, int K, int batch, float alpha, const float* bias,
                const float* row_stats, const float* col_sum, const void* residual, long long ldr, long long stride_r,
                int geglu, int is_bf16, void* stream);

/* Channel-tile width (320 / 256: 8-wave workgroups; 160 / 128: 4-wave workgroups for problems that would under-fill the chip)
 * gvd_gemm_nt uses for an M x N problem with `batch` batches. */
/* The GEGLU feed-forward of the guided sampler's differentiable U-Net evaluation (attention.py:415-450 under autograd with frozen weights)
 * without its two gate row kernels -- the same product and LayerNorm fold / bias as gvd_gemm_nt (no residual), W rows in the gate order above:
 *   mode 2 (forward)   y [.., M, N / 2] = value * gelu(gate), evaluated on the ROUNDED projection exactly as gvd_geglu evaluates it on the tensor
 *                      it reads back, and aux [.., M, N] = that projection (columns stay in the kernel's [16 value | 16 gate] block order: only
 *                      mode 3 reads it);
 *   mode 3 (backward)  the product is d(loss)/d(gated output) [.., M, N] (x = the gradient behind the output projection, w = that projection's
 *                      transposed weight); aux [.., M, 2 N] is the tensor mode 2 saved; y [.., M, 2 N] = d(loss)/d(projection) in the same block
 *                      order (d value = dy gelu(gate), d gate = dy value gelu'(gate), as gvd_geglu_bwd).  N % 16 == 0. */
int gvd_gemm_nt_gate(const void* x, long long ldx, long long stride_x, const void* w, long long ldw, long long stride_w,
                     void* y, long long ldy, long long stride_y, int M, int N, int K, int batch, float alpha, const float* bias,
                     const float* row_stats, const float* col_sum, int mode, void* aux, long long ldaux, long long stride_aux,
                     int is_bf16, void* stream);

int gvd_gemm_tile_n(int M, int N, int batch);
int gvd_gemm_geglu_layout(void);

/* (mean, rstd = 1 / sqrt(var + eps)) of every row of x [M, C] (row stride ldx; C % 8 == 0, C <= 4096) as float pairs:
 * the per-row half of the LayerNorm fold above (nn.LayerNorm's biased variance, attention.py:283-285). */
int gvd_row_stats(const void* x, long long ldx, float* stats, long long M, int C, float eps, int is_bf16, void* stream);

/* Row kernels of the WIDE-HEAD attention -- the VAE's single-head d = 512 block (ae_modules.py:26-78), evaluated as chunked
 * GEMMs on gvd_gemm_nt with the scores of one query (or key) chunk living in the Infinity Cache (lvdm_amd/wide_attention.py).
 * gvd_softmax_rows: in-place softmax over the rows of s [rows, N] (16-bit, row stride ld; N % 8 == 0, N <= 16384), fp32 math;
 *   lse[row] = max + log(sum) (may be NULL).
 * gvd_attn_ds: the backward's element step, in place: s (scaled scores) becomes P = exp(s - lse), dp (= dO V^T) becomes
 *   P (dp - delta); lse / delta indexed by row (by_col = 0) or by (row / rows_per_batch, column) (by_col = 1, the transposed pass).
 * n_valid <= N: columns >= n_valid are padding (token counts are padded to multiples of 8): probability / dS' 0 there. */
int gvd_softmax_rows(void* s, long long ld, long long rows, int N, int n_valid, float* lse, int is_bf16, void* stream);
int gvd_attn_ds(void* s, void* dp, const float* lse, const float* delta, long long rows, int N, int n_valid, long long rows_per_batch,
                int by_col, int is_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif
