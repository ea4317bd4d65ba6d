// diffusion_common.h -- shared by the translation units of libgvd_diffusion.so: error slot, vector types and the
// per-element-type MFMA traits (32x32x16, 16-bit operands, fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <type_traits>

#include "../../include/gvd_diffusion.h"

namespace gvdd {

int fail(int code, const char* what, hipError_t e = hipSuccess);   // records the message for gvd_diff_last_error()

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <typename T> struct Tr;
template <> struct Tr<_Float16> {
    typedef h8 vec8;
    static __device__ __forceinline__ f16v mfma(vec8 a, vec8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f4v mfma16(vec8 a, vec8 b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi)
    {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 p = { (_Float16)lo, (_Float16)hi };
        return __builtin_bit_cast(unsigned, p);
    }
    // acc + lo + hi of a packed pair: v_dot2_f32_f16 against (1, 1) -- one plain VALU op for two values, and the sum is over the
    // ROUNDED values the MFMA consumes
    static constexpr bool kHasDot2 = true;
    static __device__ __forceinline__ float add_pair(unsigned packed, float acc)
    {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 ones = { (_Float16)1.0f, (_Float16)1.0f };
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, packed), ones, acc, false);
    }
};
template <> struct Tr<__bf16> {
    typedef b8 vec8;
    static __device__ __forceinline__ f16v mfma(vec8 a, vec8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f4v mfma16(vec8 a, vec8 b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi)
    {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        b2 p = { (__bf16)lo, (__bf16)hi };
        return __builtin_bit_cast(unsigned, p);
    }
    static constexpr bool kHasDot2 = false;   // bf16: the row sum stays in packed fp32 adds
    static __device__ __forceinline__ float add_pair(unsigned, float acc) { return acc; }
};

constexpr int KV_TILE = 64;   // keys per iteration
constexpr int LDS_ROW = 72;   // padded row length (elements): 144 B, 16-byte aligned, spreads bank groups

// B operand of the second product from a 32x32 fp32 C block that holds [k-row][n-col] with n = lane & 31:
// rows 8*k2 .. 8*k2+7 of this lane's 16 registers are packed to 16 bit and exchanged between the wave halves so
// that lower lanes end with k = base+0..7 and upper lanes with k = base+8..15 (the operand's k = 8*hi + j map).
template <typename T>
__device__ __forceinline__ typename Tr<T>::vec8 c_block_to_b_operand(const f16v& s, int k2)
{
    const int r0 = 8 * k2;
    const unsigned u01 = Tr<T>::pack2(s[r0 + 0], s[r0 + 1]), u23 = Tr<T>::pack2(s[r0 + 2], s[r0 + 3]);
    const unsigned u45 = Tr<T>::pack2(s[r0 + 4], s[r0 + 5]), u67 = Tr<T>::pack2(s[r0 + 6], s[r0 + 7]);
    auto sa = __builtin_amdgcn_permlane32_swap(u01, u45, false, false);
    auto sb = __builtin_amdgcn_permlane32_swap(u23, u67, false, false);
    const u4 packed = { sa[0], sb[0], sa[1], sb[1] };
    return __builtin_bit_cast(typename Tr<T>::vec8, packed);
}

// Same, from values already packed in pairs (u[j] = pack(c[2j], c[2j+1])).
template <typename T>
__device__ __forceinline__ typename Tr<T>::vec8 packed_c_to_b_operand(const unsigned (&u)[8], int k2)
{
    auto sa = __builtin_amdgcn_permlane32_swap(u[4 * k2], u[4 * k2 + 2], false, false);
    auto sb = __builtin_amdgcn_permlane32_swap(u[4 * k2 + 1], u[4 * k2 + 3], false, false);
    const u4 packed = { sa[0], sb[0], sa[1], sb[1] };
    return __builtin_bit_cast(typename Tr<T>::vec8, packed);
}

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 exp2_pair(f2 a)
{
    // (a Cody-Waite + cubic VALU exp2 was tried to offload v_exp_f32: 726 -> 539 TFLOP/s; removing the exp altogether
    // only gave 769, so the transcendental rate is not what bounds the loop)
    return f2{ __builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y) };
}

// max of three.  Plain fmaxf so that the compiler sees the instruction: an inline-asm v_max3_f32 reading MFMA results
// hid the XDL-write -> VALU-read hazard from the hazard recognizer (no wait states were inserted, the max was taken
// over partly stale registers, and peaky score rows overflowed -- found by the randomized sweep in
// tests/test_diffusion_gpu.py).  The translation unit is built with -fno-honor-nans, which drops the per-operand
// sNaN-quieting v_max the IEEE-mode lowering would otherwise add and lets the two maxima fuse into v_max3_f32.
__device__ __forceinline__ float max3f(float a, float b, float c)
{
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}

// Phi(g) (the normal CDF) and exp(-g^2 / 2) of the exact-erf GELU's forward / backward row steps: erfc by a Chebyshev-fitted exponent
// polynomial in t = 1 / (1 + z / 2) (|relative error| < 1.2e-7), no cancellation on either side of 0 (diffusion_kernels.hip: k_geglu, k_geglu_bwd).
__device__ __forceinline__ void gelu_cdf_exp(float g, float& cdf, float& e)
{
    const float z = fabsf(g) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.f));
    float c = fmaf(t, 0.17087277f, -0.82215223f);
    c = fmaf(t, c, 1.48851587f);
    c = fmaf(t, c, -1.13520398f);
    c = fmaf(t, c, 0.27886807f);
    c = fmaf(t, c, -0.18628806f);
    c = fmaf(t, c, 0.09678418f);
    c = fmaf(t, c, 0.37409196f);
    c = fmaf(t, c, 1.00002368f);
    c = fmaf(t, c, -1.26551223f);
    const float mz2 = (-0.5f * 1.4426950408889634f) * (g * g);   // -z^2 log2(e)
    e = __builtin_amdgcn_exp2f(mz2);
    const float half_erfc = (0.5f * t) * __builtin_amdgcn_exp2f(fmaf(c, 1.4426950408889634f, mz2));
    cdf = g >= 0.f ? 1.f - half_erfc : half_erfc;
}

// XCD-aware (item, tile) of a workgroup of an (items, tiles) grid whose tiles of one item share operands through L2 (the query
// tiles of one (frame, head) read the same K / V).  Dispatch walks the grid x fastest and workgroup L runs on XCD L % 8 (the GEMM's
// slot map relies on the same observation), so with the plain (blockIdx.x, blockIdx.y) = (item, tile) reading the ~64 workgroups
// resident on an XCD belong to ~64 different items and every one of them pulls its item's K / V through that XCD's 4 MiB L2 alone
// (level-0 self-attention: 3.6 GB fetched per launch against 0.6 GB of tensors, profiles/r04_mfma_pmc.json).  Here XCD k takes a
// contiguous range of the (item-major) work list instead: the tiles of an item run next to each other on ONE XCD.
#ifndef GVD_XCD_ITEMS
#define GVD_XCD_ITEMS 1   // 0 = plain reading (A/B builds)
#endif
__device__ __forceinline__ void xcd_item_tile(int& item, int& tile)
{
#if GVD_XCD_ITEMS
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy;
    const unsigned L = blockIdx.y * gx + blockIdx.x, k = L & 7, i = L >> 3;
    const unsigned a = total >> 3, r = total & 7;
    const unsigned t = k * a + (k < r ? k : r) + i;        // XCD k holds work items [k a + min(k, r), ...): cnt_k = a + (k < r)
    item = (int)(t / gy);
    tile = (int)(t - (unsigned)item * gy);
#else
    item = blockIdx.x;
    tile = blockIdx.y;
#endif
}

// The same for a (pixel tiles, channel tiles, phases | K slices) convolution grid: the channel tiles (and upsampling phases) of one
// pixel tile read the same input patch, so they are made neighbours on one XCD (channel tile fastest, then z, then the pixel tile).
// `weights_first` is the mirror image for launches whose WEIGHTS are the large stream (small maps, 1280 channels, split-K slices):
// the pixel tiles of one (channel tile, K slice) are neighbours on one XCD and share its weight slab; pixel tile fastest.
__device__ __forceinline__ void xcd_conv_ids(int& px, int& co, int& z, bool weights_first)
{
#if GVD_XCD_ITEMS
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
    const unsigned L = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x, k = L & 7, i = L >> 3;
    const unsigned a = total >> 3, r = total & 7;
    const unsigned t = k * a + (k < r ? k : r) + i;
    if (weights_first) {
        const unsigned u = t / gx;
        px = (int)(t - u * gx);
        z = (int)(u / gy);
        co = (int)(u - (unsigned)z * gy);
    } else {
        const unsigned u = t / gy;
        co = (int)(t - u * gy);
        px = (int)(u / gz);
        z = (int)(u - (unsigned)px * gz);
    }
#else
    px = blockIdx.x; co = blockIdx.y; z = blockIdx.z;
#endif
}

}  // namespace gvdd
