// ssim.hip -- fused SSIM forward / backward for gfx950 (C-ABI: include/gvd_loss.h).
//
// Reference: utils/loss_utils.py:46-82.  ssim_map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)) with
// mu = G*x, s1 = G*(x^2) - mu1^2, s12 = G*(xy) - mu1 mu2, G = 11x11 gaussian (sigma 1.5), zero padding 5.
// One workgroup = one 16x16 tile of one plane: the 26x26 halo of both images goes to LDS once, the five windowed
// moments are formed separably (11 + 11 taps instead of 121) from LDS, and the map, its tile sum and the three
// derivative planes come out of the same kernel.  HBM traffic: 2 reads + (3 writes when a gradient is wanted) per
// pixel instead of the ~40 full-image passes of the unfused autograd graph.  Backward: the same separable window over
// the three derivative planes, combined with x and y at the centre pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/gvd_loss.h"

namespace {

thread_local std::string g_err;
int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[320];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

constexpr int TILE = 16, R = 5, HALO = TILE + 2 * R;   // 26
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Gauss { float g[11]; };

__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

__device__ __forceinline__ void load_halo(const float* __restrict__ plane, int H, int W, int x0, int y0, float (*s)[HALO + 1], int tid)
{
    for (int i = tid; i < HALO * HALO; i += 256) {
        const int r = i / HALO, c = i - r * HALO, y = y0 + r - R, x = x0 + c - R;
        s[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? plane[(size_t)y * W + x] : 0.f;
    }
}

__global__ void __launch_bounds__(256) k_ssim_fwd(const float* __restrict__ img1, const float* __restrict__ img2, Gauss gw, int H, int W,
                                                  float* __restrict__ partials, float* __restrict__ dmaps, float* __restrict__ ssim_map,
                                                  long long plane_stride_all, float* __restrict__ l1_partials)
{
    __shared__ float sx[HALO][HALO + 1], sy[HALO][HALO + 1];
    __shared__ float hz[5][HALO][TILE + 1];
    __shared__ float red[4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE, plane = blockIdx.z;
    const size_t poff = (size_t)plane * H * W;
    load_halo(img1 + poff, H, W, x0, y0, sx, tid);
    load_halo(img2 + poff, H, W, x0, y0, sy, tid);
    __syncthreads();
    for (int i = tid; i < HALO * TILE; i += 256) {   // horizontal 11-tap pass of the five moments
        const int r = i / TILE, c = i - r * TILE;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = gw.g[k], x = sx[r][c + k], y = sy[r][c + k];
            a = fmaf(w, x, a); b = fmaf(w, y, b);
            aa = fmaf(w, x * x, aa); bb = fmaf(w, y * y, bb); ab = fmaf(w, x * y, ab);
        }
        hz[0][r][c] = a; hz[1][r][c] = b; hz[2][r][c] = aa; hz[3][r][c] = bb; hz[4][r][c] = ab;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, ex2 = 0.f, ey2 = 0.f, exy = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = gw.g[k];
        mu1 = fmaf(w, hz[0][ty + k][tx], mu1); mu2 = fmaf(w, hz[1][ty + k][tx], mu2);
        ex2 = fmaf(w, hz[2][ty + k][tx], ex2); ey2 = fmaf(w, hz[3][ty + k][tx], ey2); exy = fmaf(w, hz[4][ty + k][tx], exy);
    }
    const int x = x0 + tx, y = y0 + ty;
    const bool in = x < W && y < H;
    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, m12 = mu1 * mu2;
    const float s1 = ex2 - mu1s, s2 = ey2 - mu2s, s12 = exy - m12;
    const float A = 2.f * m12 + C1, B = 2.f * s12 + C2, C = mu1s + mu2s + C1, D = s1 + s2 + C2;
    const float inv = 1.f / (C * D);
    const float val = in ? A * B * inv : 0.f;
    if (in) {
        const size_t p = poff + (size_t)y * W + x;
        if (ssim_map) ssim_map[p] = val;
        if (dmaps) {
            // d/dmu1 at fixed windowed second moments: A' = 2 mu2, B' = -2 mu2, C' = 2 mu1, D' = -2 mu1
            const float dmu1 = (2.f * mu2 * (B - A)) * inv - val * (2.f * mu1 * (D - C)) * inv;
            dmaps[p] = dmu1;
            dmaps[plane_stride_all + p] = -val / D;          // d/dE[x^2] : D' = 1
            dmaps[2 * plane_stride_all + p] = 2.f * A * inv;  // d/dE[xy]  : B' = 2
        }
    }
    float s = wave_sum(val);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const size_t tile = ((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tid == 0) partials[tile] = (red[0] + red[1]) + (red[2] + red[3]);
    if (l1_partials) {   // |x - y| of the same tile from the halo already in LDS (photometric loss: L1 term)
        __syncthreads();
        const float a = in ? fabsf(sx[ty + R][tx + R] - sy[ty + R][tx + R]) : 0.f;
        s = wave_sum(a);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) l1_partials[tile] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// loss = (1 - lambda) mean|x - y| + lambda (1 - mean ssim): fixed-order sums of the per-tile partials by one workgroup
__global__ void __launch_bounds__(256) k_photo_finalize(const float* __restrict__ ssim_part, const float* __restrict__ l1_part, long long n,
                                                        float inv_count, float lambda, float* __restrict__ out3)
{
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) { a += (double)ssim_part[i]; b += (double)l1_part[i]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o; o >>= 1) {
        if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ssim_mean = (float)(sh[0][0] * inv_count), l1_mean = (float)(sh[1][0] * inv_count);
        out3[0] = (1.f - lambda) * l1_mean + lambda * (1.f - ssim_mean);
        out3[1] = l1_mean;
        out3[2] = ssim_mean;
    }
}

__global__ void __launch_bounds__(256) k_ssim_bwd(const float* __restrict__ img1, const float* __restrict__ img2, Gauss gw, int H, int W,
                                                  const float* __restrict__ dmaps, const float* __restrict__ plane_scale,
                                                  float* __restrict__ d_img1, long long plane_stride_all,
                                                  const float* __restrict__ upstream, float w_ssim, float w_l1)
{
    __shared__ float sm[3][HALO][HALO + 1];
    __shared__ float hz[3][HALO][TILE + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE, plane = blockIdx.z;
    const size_t poff = (size_t)plane * H * W;
#pragma unroll
    for (int m = 0; m < 3; m++) load_halo(dmaps + m * plane_stride_all + poff, H, W, x0, y0, sm[m], tid);
    __syncthreads();
    for (int i = tid; i < HALO * TILE; i += 256) {
        const int r = i / TILE, c = i - r * TILE;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = gw.g[k];
            a = fmaf(w, sm[0][r][c + k], a); b = fmaf(w, sm[1][r][c + k], b); d = fmaf(w, sm[2][r][c + k], d);
        }
        hz[0][r][c] = a; hz[1][r][c] = b; hz[2][r][c] = d;
    }
    __syncthreads();
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = gw.g[k];
        a = fmaf(w, hz[0][ty + k][tx], a); b = fmaf(w, hz[1][ty + k][tx], b); d = fmaf(w, hz[2][ty + k][tx], d);
    }
    const int x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
        const size_t p = poff + (size_t)y * W + x;
        const float xv = img1[p], yv = img2[p], gs = a + 2.f * xv * b + yv * d;
        if (upstream) {   // photometric loss: upstream * (w_ssim dssim + w_l1 sign(x - y))
            const float df = xv - yv, sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            d_img1[p] = upstream[0] * (w_ssim * gs + w_l1 * sg);
        } else {
            d_img1[p] = plane_scale[plane] * gs;
        }
    }
}

int check(const char* who, const void* a, const void* b, const float* gauss, int planes, int H, int W)
{
    char msg[128];
    if (!a || !b || !gauss || planes <= 0 || H <= 0 || W <= 0) { snprintf(msg, sizeof msg, "%s: bad arguments", who); return fail(-1, msg); }
    if (planes > 65535) { snprintf(msg, sizeof msg, "%s: more than 65535 planes", who); return fail(-1, msg); }
    return 0;
}

}  // namespace

extern "C" {

const char* gvd_loss_last_error(void) { return g_err.c_str(); }

long long gvd_ssim_partial_count(int planes, int H, int W)
{
    return (long long)planes * ((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE);
}

int gvd_ssim_forward(const float* img1, const float* img2, const float* gauss, int planes, int H, int W, float* partials,
                     float* dmaps, float* ssim_map, void* stream_)
{
    if (int rc = check("gvd_ssim_forward", img1, img2, gauss, planes, H, W)) return rc;
    if (!partials) return fail(-1, "gvd_ssim_forward: partials is NULL");
    Gauss gw;
    for (int i = 0; i < 11; i++) gw.g[i] = gauss[i];
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, planes);
    hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(256), 0, (hipStream_t)stream_, img1, img2, gw, H, W, partials, dmaps, ssim_map,
                       (long long)planes * H * W, (float*)nullptr);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_ssim_fwd", e);
    return 0;
}

int gvd_photometric_forward(const float* img1, const float* img2, const float* gauss, int planes, int H, int W, float lambda_dssim,
                            float* partials, float* dmaps, float* out3, void* stream_)
{
    if (int rc = check("gvd_photometric_forward", img1, img2, gauss, planes, H, W)) return rc;
    if (!partials || !out3) return fail(-1, "gvd_photometric_forward: null pointer");
    Gauss gw;
    for (int i = 0; i < 11; i++) gw.g[i] = gauss[i];
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, planes);
    const long long n = gvd_ssim_partial_count(planes, H, W);
    hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(256), 0, (hipStream_t)stream_, img1, img2, gw, H, W, partials, dmaps, (float*)nullptr,
                       (long long)planes * H * W, partials + n);
    hipLaunchKernelGGL(k_photo_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream_, (const float*)partials, (const float*)(partials + n), n,
                       1.0f / ((float)planes * (float)H * (float)W), lambda_dssim, out3);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_photo_*", e);
    return 0;
}

int gvd_photometric_backward(const float* img1, const float* img2, const float* gauss, const float* dmaps, const float* upstream,
                             int planes, int H, int W, float lambda_dssim, float* d_img1, void* stream_)
{
    if (int rc = check("gvd_photometric_backward", img1, img2, gauss, planes, H, W)) return rc;
    if (!dmaps || !upstream || !d_img1) return fail(-1, "gvd_photometric_backward: null pointer");
    Gauss gw;
    for (int i = 0; i < 11; i++) gw.g[i] = gauss[i];
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, planes);
    const float inv = 1.0f / ((float)planes * (float)H * (float)W);
    hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(256), 0, (hipStream_t)stream_, img1, img2, gw, H, W, dmaps, (const float*)nullptr, d_img1,
                       (long long)planes * H * W, upstream, -lambda_dssim * inv, (1.f - lambda_dssim) * inv);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_ssim_bwd (photometric)", e);
    return 0;
}

int gvd_ssim_backward(const float* img1, const float* img2, const float* gauss, const float* dmaps, const float* plane_scale,
                      int planes, int H, int W, float* d_img1, void* stream_)
{
    if (int rc = check("gvd_ssim_backward", img1, img2, gauss, planes, H, W)) return rc;
    if (!dmaps || !plane_scale || !d_img1) return fail(-1, "gvd_ssim_backward: null pointer");
    Gauss gw;
    for (int i = 0; i < 11; i++) gw.g[i] = gauss[i];
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, planes);
    hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(256), 0, (hipStream_t)stream_, img1, img2, gw, H, W, dmaps, plane_scale, d_img1,
                       (long long)planes * H * W, (const float*)nullptr, 0.f, 0.f);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_ssim_bwd", e);
    return 0;
}

}  // extern "C"
