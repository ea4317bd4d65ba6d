// diffusion_kernels.hip -- hand-written gfx950 kernels on the ViewCrafter DDIM hot path
// (C-ABI: include/gvd_diffusion.h).
//
//  k_attn_fwd     flash attention forward on MFMA 32x32x16 (f16 / bf16 operands, fp32 accumulate, fp32 online
//                 softmax).  Replaces xformers.memory_efficient_attention (lvdm/modules/attention.py:175,187).
//  k_ddim_stats / k_ddim_apply   the whole no-grad DDIM update (lvdm/models/samplers/ddim.py:208-280).
//
// Attention design (wave64, one wave = 32 query rows):
//   * operands are read straight from the [B, N, H*64] token-major layout the Linear layers produce
//     (row stride H*64) -- no head-split permute/contiguous copies (the reference makes 3 per call);
//   * "swapped" product S^T = K Q^T: MFMA C/D layout puts one QUERY per lane column, so every softmax
//     reduction is over a lane's own 32 registers plus ONE cross-half exchange (lane ^ 32);
//   * P^T goes back into an MFMA B operand without touching LDS: pack to 16 bit, v_permlane32_swap
//     between the wave halves (keys 8s..8s+3 / 8s+4..8s+7 interleave of the 32x32 C layout);
//   * V is written TRANSPOSED into LDS when the tile is staged, so the P V product reads its A operand
//     (V^T) with plain 16-byte ds_read_b128; K and V^T rows are padded to 72 elements (144 B) so the
//     32 lanes of a half-wave hit different bank groups;
//   * O^T = V^T P^T accumulates in 32 fp32 registers, rescaled by the per-lane alpha of the online softmax.
#include <stdlib.h>

#include "diffusion_common.h"

#ifndef GVD_ATTN_QMAJOR
#define GVD_ATTN_QMAJOR 0
#endif
// (Measured, not kept: online softmax + P V per 32-key HALF of a tile, so that a wave's own softmax VALU of half 0 issues under
//  the S MFMAs of half 1 and the P V MFMAs of half 0 under the softmax of half 1: 723 against 734 TFLOP/s at L0 on the same box.
//  Per 64-key tile and wave the loop issues ~240 VALU + 64 v_exp against 32 MFMAs -- ~1650 VALU cycles, 1024 MFMA cycles -- and
//  sits at 57 % / 35 % of the two pipes: dependency stalls with two resident waves per SIMD at 252 VGPRs, not issue order.)
#ifndef GVD_ATTN_OPTIMISTIC
#define GVD_ATTN_OPTIMISTIC 1   // 1 = P formed with the stale running max, the row sum as overflow detector (see the tile body)
#endif
#ifndef GVD_ATTN_HOIST
#define GVD_ATTN_HOIST 0   // 1 = pin the clustered LDS fragment reads with sched_barrier (measured 774 vs 788 TFLOP/s: off)
#endif

namespace gvdd {
thread_local std::string g_err;
int fail(int code, const char* what, hipError_t e)
{
    char buf[384];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}
}  // namespace gvdd

using namespace gvdd;

namespace {

template <typename T, int WAVES, int QB>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2)))
k_attn_fwd(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* out, float* __restrict__ lse,
           int H, int Nq, int Nk, float scale_log2e, long long q_bs, long long q_rs, long long kv_bs, long long kv_rs,
           long long o_bs, long long o_rs, const T* accum, float accum_scale)   // (accum may alias out)
{
    // QB = query blocks (of 32) per wave.  QB = 2 halves the LDS reads and K/V staging per flop (every A operand feeds
    // two MFMAs) and gives the wave two independent softmax chains to interleave with the matrix pipe.
    typedef typename Tr<T>::vec8 vec8;
    constexpr int NT = WAVES * 64;
    __shared__ __attribute__((aligned(16))) T sKb[2][KV_TILE][LDS_ROW];   // double-buffered: one barrier per tile
    __shared__ __attribute__((aligned(16))) T sVtb[2][64][LDS_ROW];

    int bh, qtile;
    xcd_item_tile(bh, qtile);
    const int b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;
    // element (batch b, row n, head h, d) lives at b*bs + n*rs + h*64 + d: covers the plain [B, N, H*64] layout
    // (bs = N*H*64, rs = H*64) and the frame-strided temporal layout [T, pixels, H*64] (bs = H*64, rs = pixels*H*64)
    const size_t rs = (size_t)q_rs, krs = (size_t)kv_rs;
    const T* qb = q + (size_t)b * q_bs + (size_t)h * 64;
    const T* kb = k + (size_t)b * kv_bs + (size_t)h * 64;
    const T* vb = v + (size_t)b * kv_bs + (size_t)h * 64;
    T* ob = out + (size_t)b * o_bs + (size_t)h * 64;                  // out (and accum) have their own strides: q may be a column
    const T* ab = accum ? accum + (size_t)b * o_bs + (size_t)h * 64 : nullptr;   // block of a packed [.., q | k | v] projection

    int query[QB];
    bool valid_q[QB];
    vec8 qf[QB][4];
    f16v o0[QB], o1[QB];
    float m[QB], l[QB];
#pragma unroll
    for (int qi = 0; qi < QB; qi++) {
        query[qi] = qtile * (32 * WAVES * QB) + (wave * QB + qi) * 32 + col;
        valid_q[qi] = query[qi] < Nq;
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            qf[qi][ks] = valid_q[qi] ? *reinterpret_cast<const vec8*>(qb + (size_t)query[qi] * rs + 16 * ks + 8 * hi) : vec8{};
        o0[qi] = f16v{};
        o1[qi] = f16v{};
        m[qi] = -1.0e30f;
        l[qi] = 0.f;
    }

    // Staging maps (per pass of NT threads over the 512 16-byte chunks of a 64-key x 64-channel tile):
    //   K : chunk c -> key row c>>3, channels (c&7)*8..+7 : 8 lanes cover one 128-byte row (coalesced), ds_write_b128.
    //   V : chunk c -> key PAIR kp = c&31 (keys 2kp, 2kp+1), channels (c>>5)*8..+7 : the two keys of each channel are
    //       packed into one dword and written to V^T[channel][2kp] with ds_write_b32 -- 32 consecutive dwords per
    //       half-wave, conflict-free (the per-element b16 scatter of the first version was a 16-way bank conflict).
    constexpr int PASSES = (KV_TILE * 8) / NT;  // 2 for 256 threads, 8 for 64
    vec8 rk[PASSES], rv0[PASSES / 2 > 0 ? PASSES / 2 : 1], rv1[PASSES / 2 > 0 ? PASSES / 2 : 1];
    auto prefetch = [&](int kt) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int c = tid + ps * NT, row = c >> 3, c8 = (c & 7) * 8, key = kt + row;
            rk[ps] = (key < Nk) ? *reinterpret_cast<const vec8*>(kb + (size_t)key * krs + c8) : vec8{};
        }
#pragma unroll
        for (int ps = 0; ps < PASSES / 2; ps++) {
            const int c = tid + ps * NT, kp = c & 31, c8 = (c >> 5) * 8, key = kt + 2 * kp;
            rv0[ps] = (key < Nk) ? *reinterpret_cast<const vec8*>(vb + (size_t)key * krs + c8) : vec8{};
            rv1[ps] = (key + 1 < Nk) ? *reinterpret_cast<const vec8*>(vb + (size_t)(key + 1) * krs + c8) : vec8{};
        }
    };
    auto commit = [&](int buf) {
        T (*sK)[LDS_ROW] = sKb[buf];
        T (*sVt)[LDS_ROW] = sVtb[buf];
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int c = tid + ps * NT, row = c >> 3, c8 = (c & 7) * 8;
            *reinterpret_cast<vec8*>(&sK[row][c8]) = rk[ps];
        }
#pragma unroll
        for (int ps = 0; ps < PASSES / 2; ps++) {
            const int c = tid + ps * NT, kp = c & 31, c8 = (c >> 5) * 8;
            typedef T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const T2 pr = { rv0[ps][i], rv1[ps][i] };
                *reinterpret_cast<T2*>(&sVt[c8 + i][2 * kp]) = pr;
            }
        }
    };

    // One 64-key tile.  TAIL (only the last, partial tile) masks the keys past Nk; the full-tile body is branch-free
    // so that the scheduler can interleave the two query blocks' softmax VALU with the other block's MFMAs.
    // Pipeline: while tile i is computed from LDS buffer i&1, tile i+1 (already in registers) is written to the other
    // buffer and the global loads of tile i+2 are issued; ONE barrier per tile publishes the writes and retires the reads.
    auto tile = [&](int kt, int buf, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const T (*sK)[LDS_ROW] = sKb[buf];
        const T (*sVt)[LDS_ROW] = sVtb[buf];
        if (!TAIL && kt + KV_TILE < Nk) {
            commit(buf ^ 1);
            if (kt + 2 * KV_TILE < Nk) prefetch(kt + 2 * KV_TILE);
        }

        // ---- S^T = K Q^T : two 32-key blocks per query block; each K fragment read feeds QB MFMAs ----
        f16v s0[QB], s1[QB];
        auto scores = [&](auto lean_tag) {
            // lean (the rare recomputation): K fragments loaded per 16-channel step, two registers sets live instead of eight
            constexpr bool LEAN = decltype(lean_tag)::value;
#pragma unroll
            for (int qi = 0; qi < QB; qi++) { s0[qi] = f16v{}; s1[qi] = f16v{}; }
            vec8 ka[4][2];
            if (!LEAN) {
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {   // all eight K fragments in flight before the first MFMA needs one
                    ka[ks][0] = *reinterpret_cast<const vec8*>(&sK[col][16 * ks + 8 * hi]);
                    ka[ks][1] = *reinterpret_cast<const vec8*>(&sK[32 + col][16 * ks + 8 * hi]);
                }
            }
#if GVD_ATTN_HOIST
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                if (LEAN) {
                    ka[ks][0] = *reinterpret_cast<const vec8*>(&sK[col][16 * ks + 8 * hi]);
                    ka[ks][1] = *reinterpret_cast<const vec8*>(&sK[32 + col][16 * ks + 8 * hi]);
                }
#pragma unroll
                for (int qi = 0; qi < QB; qi++) {
                    s0[qi] = Tr<T>::mfma(ka[ks][0], qf[qi][ks], s0[qi]);
                    s1[qi] = Tr<T>::mfma(ka[ks][1], qf[qi][ks], s1[qi]);
                }
                if (LEAN) __builtin_amdgcn_sched_barrier(0);   // keep the next step's loads behind this step's MFMAs
            }
            if (TAIL) {
#pragma unroll
                for (int qi = 0; qi < QB; qi++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int krow = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (kt + krow >= Nk) s0[qi][r] = -3.0e38f;
                        if (kt + 32 + krow >= Nk) s1[qi][r] = -3.0e38f;
                    }
                }
            }
        };
        scores(std::false_type{});
        // the V^T fragments of this tile are requested now and land under the softmax arithmetic
        vec8 va[2][2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; kbk++)
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
                const int kcol = 32 * kbk + 16 * k2 + 8 * hi;
                va[kbk][k2][0] = *reinterpret_cast<const vec8*>(&sVt[col][kcol]);
                va[kbk][k2][1] = *reinterpret_cast<const vec8*>(&sVt[32 + col][kcol]);
            }
#if GVD_ATTN_HOIST
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- online softmax over this lane's 32 scores (+ the other half's 32); the 1/sqrt(d)*log2(e) scale is
        //      folded into the exp2 argument (one fma per score).  The running max m is LAZY: O and l share it, so any m gives
        //      the exact softmax ratio as long as P = exp2(s*c - m) stays inside the 16-bit operand range; raising m costs a
        //      rescale of 32 accumulator registers per query block (a VALU -> MFMA hazard on every one) and is the cold path. ----
        unsigned pk0[QB][8], pk1[QB][8];
        // P = exp2(s*c - m): packed fp32 fma / add (v_pk_*), rounded to 16 bit right away (the MFMA operand type); returns the
        // query's sum of P over the tile's 64 keys
        auto exp_block = [&](int qi) {
            const f16v& t0 = s0[qi];
            const f16v& t1 = s1[qi];
            const f2 c2 = { scale_log2e, scale_log2e }, nm2 = { -m[qi], -m[qi] };
            f2 rs2 = { 0.f, 0.f };
            float rsa = 0.f, rsb = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                f2 a0 = { t0[2 * j], t0[2 * j + 1] }, a1 = { t1[2 * j], t1[2 * j + 1] };
                a0 = __builtin_elementwise_fma(a0, c2, nm2);
                a1 = __builtin_elementwise_fma(a1, c2, nm2);
                const f2 p0 = exp2_pair(a0), p1 = exp2_pair(a1);
                pk0[qi][j] = Tr<T>::pack2(p0.x, p0.y);
                pk1[qi][j] = Tr<T>::pack2(p1.x, p1.y);
                if (Tr<T>::kHasDot2) { rsa = Tr<T>::add_pair(pk0[qi][j], rsa); rsb = Tr<T>::add_pair(pk1[qi][j], rsb); }
                else rs2 += p0 + p1;
            }
            const float rowsum = Tr<T>::kHasDot2 ? rsa + rsb : rs2.x + rs2.y;
            return rowsum + __shfl_xor(rowsum, 32, 64);
        };
#if GVD_ATTN_OPTIMISTIC
        // OPTIMISTIC softmax: P is formed with the running m WITHOUT looking at the tile's maximum (32 v_max3 + the exchange per
        // query block: ~15 % of the loop's VALU work).  The row sum is the overflow detector: sum(P) <= 2^15 bounds every
        // P <= 2^15 (P >= 0), inside the 16-bit operand range and exact in the softmax ratio because O and l share m.  A tile
        // whose sum is larger (always the first one: m starts at -1e30 and P is inf) takes the exact path: maxima, m raised,
        // O and l rescaled, P recomputed.
        float rsum[QB];
        bool redo = false;
#pragma unroll
        for (int qi = 0; qi < QB; qi++) {
            rsum[qi] = exp_block(qi);
            redo |= !(rsum[qi] <= 32768.0f);   // also true for inf / NaN
        }
        if (__any(redo)) {
            scores(std::true_type{});   // the scores were consumed by the optimistic pass (keeping them live costs 64 registers): recompute, K is still in LDS
#pragma unroll
            for (int qi = 0; qi < QB; qi++) {
                const f16v& t0 = s0[qi];
                const f16v& t1 = s1[qi];
                float mt = max3f(t0[0], t1[0], t0[1]);
                mt = max3f(mt, t1[1], t0[2]);
#pragma unroll
                for (int r = 3; r < 16; r += 2) { mt = max3f(mt, t0[r], t1[r - 1]); mt = max3f(mt, t1[r], r + 1 < 16 ? t0[r + 1] : mt); }
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float m_new = fmaxf(m[qi], mt * scale_log2e);
                const float alpha = __builtin_amdgcn_exp2f(m[qi] - m_new);
                l[qi] *= alpha;
                m[qi] = m_new;
#pragma unroll
                for (int r = 0; r < 16; r++) { o0[qi][r] *= alpha; o1[qi][r] *= alpha; }
                rsum[qi] = exp_block(qi);
            }
        }
#pragma unroll
        for (int qi = 0; qi < QB; qi++) l[qi] += rsum[qi];
#else
        float m_cand[QB];
        bool grow = false;
#pragma unroll
        for (int qi = 0; qi < QB; qi++) {
            const f16v& t0 = s0[qi];
            const f16v& t1 = s1[qi];
            float mt = max3f(t0[0], t1[0], t0[1]);
            mt = max3f(mt, t1[1], t0[2]);
#pragma unroll
            for (int r = 3; r < 16; r += 2) { mt = max3f(mt, t0[r], t1[r - 1]); mt = max3f(mt, t1[r], r + 1 < 16 ? t0[r + 1] : mt); }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            m_cand[qi] = fmaxf(m[qi], mt * scale_log2e);
            grow |= (m_cand[qi] - m[qi]) > 8.0f;
        }
        if (__any(grow)) {
#pragma unroll
            for (int qi = 0; qi < QB; qi++) {
                const float alpha = __builtin_amdgcn_exp2f(m[qi] - m_cand[qi]);
                l[qi] *= alpha;
                m[qi] = m_cand[qi];
#pragma unroll
                for (int r = 0; r < 16; r++) { o0[qi][r] *= alpha; o1[qi][r] *= alpha; }
            }
        }
#pragma unroll
        for (int qi = 0; qi < QB; qi++) l[qi] += exp_block(qi);
#endif

        // ---- O^T += V^T P^T : P^T fragments via permlane32_swap of the packed pairs; each V^T read feeds QB MFMAs ----
#pragma unroll
        for (int kbk = 0; kbk < 2; kbk++) {
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
#pragma unroll
                for (int qi = 0; qi < QB; qi++) {
                    const vec8 pf = packed_c_to_b_operand<T>(kbk == 0 ? pk0[qi] : pk1[qi], k2);
                    o0[qi] = Tr<T>::mfma(va[kbk][k2][0], pf, o0[qi]);
                    o1[qi] = Tr<T>::mfma(va[kbk][k2][1], pf, o1[qi]);
                }
            }
        }
        __syncthreads();
    };

    prefetch(0);
    commit(0);
    if (KV_TILE < Nk) prefetch(KV_TILE);
    __syncthreads();
    const int n_full = Nk / KV_TILE * KV_TILE;
    int buf = 0;
    for (int kt = 0; kt < n_full; kt += KV_TILE, buf ^= 1) tile(kt, buf, std::false_type{});
    if (n_full < Nk) tile(n_full, buf, std::true_type{});
    // ---- epilogue: O[query][d] = O^T / l ----
#pragma unroll
    for (int qi = 0; qi < QB; qi++) {
        if (!valid_q[qi]) continue;
        // log2-domain log-sum-exp of the scaled scores, kept for the backward kernels: P = exp2(s*c - lse)
        if (lse && hi == 0) lse[(size_t)bh * Nq + query[qi]] = m[qi] + __builtin_amdgcn_logf(l[qi]);
        const float inv = 1.0f / l[qi];
        T* orow = ob + (size_t)query[qi] * (size_t)o_rs;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int d0 = 8 * rg + 4 * hi;
            uint2 w0, w1;
            w0.x = Tr<T>::pack2(o0[qi][4 * rg] * inv, o0[qi][4 * rg + 1] * inv);
            w0.y = Tr<T>::pack2(o0[qi][4 * rg + 2] * inv, o0[qi][4 * rg + 3] * inv);
            w1.x = Tr<T>::pack2(o1[qi][4 * rg] * inv, o1[qi][4 * rg + 1] * inv);
            w1.y = Tr<T>::pack2(o1[qi][4 * rg + 2] * inv, o1[qi][4 * rg + 3] * inv);
            if (ab) {   // out = accum + accum_scale * O, rounded like the separate ops (O to 16 bit first): the image-token branch
                        // of the cross-attention lands on the text branch's result (attention.py:129-142) without an add kernel
                typedef T T4 __attribute__((ext_vector_type(4)));
                const T* arow = ab + (size_t)query[qi] * (size_t)o_rs;
                const T4 a0 = *reinterpret_cast<const T4*>(arow + d0), a1 = *reinterpret_cast<const T4*>(arow + 32 + d0);
                const T4 n0 = __builtin_bit_cast(T4, w0), n1 = __builtin_bit_cast(T4, w1);
                T4 r0, r1;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    r0[e] = (T)((float)a0[e] + (float)(T)(accum_scale * (float)n0[e]));
                    r1[e] = (T)((float)a1[e] + (float)(T)(accum_scale * (float)n1[e]));
                }
                w0 = __builtin_bit_cast(uint2, r0);
                w1 = __builtin_bit_cast(uint2, r1);
            }
            *reinterpret_cast<uint2*>(orow + d0) = w0;
            *reinterpret_cast<uint2*>(orow + 32 + d0) = w1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Short rows (Nq, Nk <= 32: the TEMPORAL attention, 25 frames of one pixel -- lvdm/modules/attention.py:313-412): the whole problem
// of one (batch entry, head) is 3 x 25 x 128 bytes in, 25 x 128 bytes out and 8 MFMAs, i.e. pure HBM traffic.  The general kernel
// above ran it as one single-wave workgroup per item with 36 KB of (double-buffered 64-key) LDS -- four waves per CU, each waiting
// for its own 9.6 KB before doing anything, always through the optimistic softmax's redo path: 2.6 TB/s.  Here a WAVE walks items
// (grid-stride, ~11 per wave at level 0): Q and K are loaded straight into their MFMA operand registers (no LDS), V goes through a
// wave-private 5 KB transposition, the softmax is the exact one-tile form, and the NEXT item's twelve 16-byte loads per lane are
// issued as soon as the S product has consumed the registers -- they fly under the softmax, P V, and the row-contiguous stores
// (O^T is transposed back through wave-private LDS: full 128-byte rows per 8 lanes).  No barriers anywhere.
template <typename T>
__global__ void __launch_bounds__(256) k_attn_short_fwd(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        T* __restrict__ out, float* __restrict__ lse, int H, int Nq, int Nk, int items,
                                                        float scale_log2e, long long q_bs, long long q_rs, long long kv_bs,
                                                        long long kv_rs, long long o_bs, long long o_rs)
{
    typedef typename Tr<T>::vec8 vec8;
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr int VT_PITCH = 40, O_PITCH = 72;                         // 80 / 144-byte rows: 16-byte aligned fragment reads
    __shared__ __attribute__((aligned(16))) T sVt[4][64][VT_PITCH];    // V^T [channel][key] of the wave's current item
    __shared__ __attribute__((aligned(16))) T sO[4][32][O_PITCH];      // O [query][channel] on its way out
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, r32 = lane & 31;
    const int nw = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    // rows past the end are CLAMPED (a second copy of the last row): their keys are masked before the softmax, their queries are
    // never stored -- no zero fill, no branch around a load
    const size_t qoff = (size_t)(r32 < Nq ? r32 : Nq - 1) * (size_t)q_rs + 8 * hi;
    const size_t koff = (size_t)(r32 < Nk ? r32 : Nk - 1) * (size_t)kv_rs + 8 * hi;
    const int kp = lane & 15, vo = lane >> 4;                           // V staging: key pair (2 kp, 2 kp + 1), channel octets vo, vo + 4
    const size_t voff0 = (size_t)(2 * kp < Nk ? 2 * kp : Nk - 1) * (size_t)kv_rs + 8 * vo;
    const size_t voff1 = (size_t)(2 * kp + 1 < Nk ? 2 * kp + 1 : Nk - 1) * (size_t)kv_rs + 8 * vo;
    vec8 qf[4], kf[4], va[2], vb[2];
    auto load = [&](int it) {
        const int b = it / H, h = it - b * H;
        const T* qb = q + (size_t)b * q_bs + (size_t)h * 64 + qoff;
        const T* kb = k + (size_t)b * kv_bs + (size_t)h * 64;
        const T* vp = v + (size_t)b * kv_bs + (size_t)h * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            qf[ks] = *reinterpret_cast<const vec8*>(qb + 16 * ks);
            kf[ks] = *reinterpret_cast<const vec8*>(kb + koff + 16 * ks);
        }
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            va[ps] = *reinterpret_cast<const vec8*>(vp + voff0 + 32 * ps);
            vb[ps] = *reinterpret_cast<const vec8*>(vp + voff1 + 32 * ps);
        }
    };
    if (gw >= items) return;
    load(gw);
    for (int it = gw; it < items; it += nw) {
        // ---- V^T into the wave's LDS: the two keys of a channel packed into one dword ----
#pragma unroll
        for (int ps = 0; ps < 2; ps++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const T2 pr = { va[ps][i], vb[ps][i] };
                *reinterpret_cast<T2*>(&sVt[wave][(vo + 4 * ps) * 8 + i][2 * kp]) = pr;
            }
        __builtin_amdgcn_wave_barrier();   // (compiler only: V^T is read back by other lanes below)
        // ---- S^T = K Q^T (keys in the rows: a lane owns ONE query's 16 + 16 scores) ----
        f16v s = {};
#pragma unroll
        for (int ks = 0; ks < 4; ks++) s = Tr<T>::mfma(kf[ks], qf[ks], s);
        // ---- the next item's operands (this lane's twelve registers are free again) ----
        const int b = it / H, h = it - b * H;
        {
            const int nx = it + nw < items ? it + nw : items - 1;
            load(nx);
        }
        // ---- exact softmax over the <= 32 keys ----
#pragma unroll
        for (int r = 0; r < 16; r++)
            if ((r & 3) + 8 * (r >> 2) + 4 * hi >= Nk) s[r] = -3.0e38f;
        float mt = max3f(s[0], s[1], s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mt = max3f(mt, s[r], s[r + 1]);
        mt = fmaxf(mt, s[15]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m = mt * scale_log2e;
        unsigned pk[8];
        float rsa = 0.f;
        f2 rs2 = { 0.f, 0.f };
        const f2 c2 = { scale_log2e, scale_log2e }, nm2 = { -m, -m };
#pragma unroll
        for (int j = 0; j < 8; j++) {
            f2 a0 = { s[2 * j], s[2 * j + 1] };
            a0 = __builtin_elementwise_fma(a0, c2, nm2);
            const f2 p0 = exp2_pair(a0);
            pk[j] = Tr<T>::pack2(p0.x, p0.y);
            if (Tr<T>::kHasDot2) rsa = Tr<T>::add_pair(pk[j], rsa);
            else rs2 += p0;
        }
        float l = Tr<T>::kHasDot2 ? rsa : rs2.x + rs2.y;
        l += __shfl_xor(l, 32, 64);
        // ---- O^T = V^T P^T ----
        f16v o0 = {}, o1 = {};
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            const vec8 pf = packed_c_to_b_operand<T>(pk, k2);
            const vec8 a0 = *reinterpret_cast<const vec8*>(&sVt[wave][r32][16 * k2 + 8 * hi]);
            const vec8 a1 = *reinterpret_cast<const vec8*>(&sVt[wave][32 + r32][16 * k2 + 8 * hi]);
            o0 = Tr<T>::mfma(a0, pf, o0);
            o1 = Tr<T>::mfma(a1, pf, o1);
        }
        // ---- O = O^T / l, back to [query][channel] through LDS, stored as whole 128-byte rows ----
        if (lse && hi == 0 && r32 < Nq) lse[(size_t)it * Nq + r32] = m + __builtin_amdgcn_logf(l);
        const float inv = 1.0f / l;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int d0 = 8 * rg + 4 * hi;
            uint2 w0, w1;
            w0.x = Tr<T>::pack2(o0[4 * rg] * inv, o0[4 * rg + 1] * inv);
            w0.y = Tr<T>::pack2(o0[4 * rg + 2] * inv, o0[4 * rg + 3] * inv);
            w1.x = Tr<T>::pack2(o1[4 * rg] * inv, o1[4 * rg + 1] * inv);
            w1.y = Tr<T>::pack2(o1[4 * rg + 2] * inv, o1[4 * rg + 3] * inv);
            *reinterpret_cast<uint2*>(&sO[wave][r32][d0]) = w0;
            *reinterpret_cast<uint2*>(&sO[wave][r32][32 + d0]) = w1;
        }
        __builtin_amdgcn_wave_barrier();   // (compiler only: the read-back below is of OTHER lanes' writes; the LDS itself is in order)
        T* ob = out + (size_t)b * o_bs + (size_t)h * 64 + 8 * (lane & 7);
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const int row = (lane >> 3) + 8 * ps;
            const uint4 w = *reinterpret_cast<const uint4*>(&sO[wave][row][8 * (lane & 7)]);
            if (row < Nq) *reinterpret_cast<uint4*>(ob + (size_t)row * (size_t)o_rs) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ void __launch_bounds__(256) k_ddim_stats(const float* __restrict__ ec, const float* __restrict__ eu, long long n,
                                                    float cfg, double* __restrict__ ws)
{
    double se = 0, se2 = 0, sv = 0, sv2 = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float e = ec[i], u = eu[i];
        const float vv = u + cfg * (e - u);
        se += e; se2 += (double)e * e; sv += vv; sv2 += (double)vv * vv;
    }
    se = wave_sum_d(se); se2 = wave_sum_d(se2); sv = wave_sum_d(sv); sv2 = wave_sum_d(sv2);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ws[0], se); atomicAdd(&ws[1], se2); atomicAdd(&ws[2], sv); atomicAdd(&ws[3], sv2);
    }
}

__global__ void __launch_bounds__(256) k_ddim_apply(const float* __restrict__ x, const float* __restrict__ ec,
                                                    const float* __restrict__ eu, const float* __restrict__ noise,
                                                    float* __restrict__ x_prev, float* __restrict__ x0, const double* __restrict__ ws,
                                                    long long n, float cfg, float phi, float sa, float s1a, float sap, float dirc,
                                                    float sig_temp, float x0r)
{
    float mix = 1.0f;  // v <- v * (phi * std_e / std_v + 1 - phi)
    if (phi > 0.f) {
        const double dn = (double)n;
        const double var_e = (ws[1] - ws[0] * ws[0] / dn) / (dn - 1.0);
        const double var_v = (ws[3] - ws[2] * ws[2] / dn) / (dn - 1.0);
        const float ratio = (float)(sqrt(var_e) / sqrt(var_v));
        mix = phi * ratio + (1.f - phi);
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float xi = x[i], e = ec[i], u = eu[i];
        const float vv = (u + cfg * (e - u)) * mix;
        const float eps = sa * vv + s1a * xi;
        const float p0 = (sa * xi - s1a * vv) * x0r;
        x0[i] = p0;
        x_prev[i] = sap * p0 + dirc * eps + sig_temp * noise[i];
    }
}


// NSC (token-major) kernels: a 256-thread block covers W = min(octets, 256) channel octets x R = 256 / W rows at a time;
// thread t owns octet t % W (+ k W) and rows t / W, t / W + R, ...; the 256 - R W left-over threads idle (< W of them: 6 % for
// C = 320, none for power-of-two channel counts) -- the earlier power-of-two lane map left 37 % (C = 320) to 50 % (C = 128) idle.
__device__ __host__ __forceinline__ int oct_width(int oct) { return oct < 256 ? oct : 256; }

// ------------------------------------------------------------------------------------------------
// GroupNorm (+ optional SiLU), 16-bit activations, fp32 statistics (lvdm/basics.py:76-86 semantics).
// Two layouts:  NCS  x[N][C][S] (channel-first, what the 2-D convolutions produce)
//               NSC  x[N][S][C] (token-major, what the Linear / temporal-GEMM path uses)
// Pass 1 accumulates sum / sum-of-squares per (n, group) into doubles (2 atomics per block), pass 2
// normalises: 2 reads + 1 write of the 16-bit tensor in total, versus ~5 fp32-sized passes through the
// cast -> group_norm -> silu -> cast sequence of the eager path.
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

template <typename T>
__global__ void __launch_bounds__(256) k_gn_stats_ncs(const T* __restrict__ x, double* __restrict__ stats, long long L, int chunks)
{
    typedef typename Tr<T>::vec8 vec8;
    const long long grp = blockIdx.x;            // n * G + g ; its L = cpg * S elements are contiguous
    const T* p = x + grp * L;
    const long long per = ((L + chunks - 1) / chunks + 7) & ~7LL;
    const long long beg = (long long)blockIdx.y * per, end = (beg + per < L) ? beg + per : L;
    float s = 0.f, q = 0.f;
    if ((L & 7) == 0) {
        for (long long i = beg + (long long)threadIdx.x * 8; i < end; i += 256 * 8) {
            const vec8 v = *reinterpret_cast<const vec8*>(p + i);
#pragma unroll
            for (int k = 0; k < 8; k++) { const float f = to_f(v[k]); s += f; q = fmaf(f, f, q); }
        }
    } else {
        for (long long i = beg + threadIdx.x; i < end; i += 256) { const float f = to_f(p[i]); s += f; q = fmaf(f, f, q); }
    }
    double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
    __shared__ double sh[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[2 * w] = ds; sh[2 * w + 1] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[2 * grp], sh[0] + sh[2] + sh[4] + sh[6]);
        atomicAdd(&stats[2 * grp + 1], sh[1] + sh[3] + sh[5] + sh[7]);
    }
}

// Per-(sample, channel) affine of the normalisation, y = a x + b, formed once (N*C threads) so the streaming apply
// kernels do one fma (+ SiLU) per element instead of fp64 divisions.
__global__ void __launch_bounds__(256) k_gn_coef(const double* __restrict__ stats, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float2* __restrict__ coef,
                                                 int N, int C, int G, long long S, float eps)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C, cpg = C / G;
    const double cnt = (double)cpg * (double)S;
    const long long grp = (long long)n * G + c / cpg;
    const double mean = stats[2 * grp] / cnt;
    const double var = stats[2 * grp + 1] / cnt - mean * mean;
    const float rstd = rsqrtf((float)(var > 0 ? var : 0) + eps);
    const float a = rstd * gamma[c];
    coef[i] = make_float2(a, beta[c] - (float)mean * a);
}

__device__ __forceinline__ float silu_f(float f)
{
    return f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * f));
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_apply_ncs(const T* __restrict__ x, T* __restrict__ y, const float2* __restrict__ coef,
                                                      long long S, int silu, long long total)
{
    typedef typename Tr<T>::vec8 vec8;
    const bool vec = (S & 7) == 0;
    const long long step = (long long)gridDim.x * 256 * (vec ? 8 : 1);
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * (vec ? 8 : 1); i < total; i += step) {
        const float2 ab = coef[i / S];           // n * C + c   (8 consecutive elements share it when S % 8 == 0)
        if (vec) {
            vec8 v = *reinterpret_cast<const vec8*>(x + i);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float f = fmaf(to_f(v[k]), ab.x, ab.y);
                if (silu) f = silu_f(f);
                v[k] = (T)f;
            }
            *reinterpret_cast<vec8*>(y + i) = v;
        } else {
            float f = fmaf(to_f(x[i]), ab.x, ab.y);
            if (silu) f = silu_f(f);
            y[i] = (T)f;
        }
    }
}

// NSC: thread -> one octet of 8 consecutive channels, walking down the rows of its block's strip.
template <typename T>
__global__ void __launch_bounds__(256) k_gn_stats_nsc(const T* __restrict__ x, double* __restrict__ stats, int C, int G, long long S,
                                                      int rows_per_block)
{
    typedef typename Tr<T>::vec8 vec8;
    extern __shared__ double sh_g[];  // [G][2]; fp64: a block's strip is up to 8192 rows x cpg channels per group (a thread's
                                      // own fp32 partial spans <= rows / rstep of them)
    const int n = blockIdx.y, cpg = C / G, oct = C / 8;
    for (int i = threadIdx.x; i < 2 * G; i += 256) sh_g[i] = 0.0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < S) ? r0 + rows_per_block : S;
    const T* base = x + (long long)n * S * C;
    const int Wd = oct_width(oct), rstep = 256 / Wd, lane_row = threadIdx.x / Wd;
    for (int o = threadIdx.x % Wd; o < oct && lane_row < rstep; o += Wd) {
        float s[8], q[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { s[k] = 0.f; q[k] = 0.f; }
#pragma unroll 4
        for (long long r = r0 + lane_row; r < r1; r += rstep) {
            const vec8 v = *reinterpret_cast<const vec8*>(base + r * C + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) { const float f = to_f(v[k]); s[k] += f; q[k] = fmaf(f, f, q[k]); }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int g = (o * 8 + k) / cpg;
            atomicAdd(&sh_g[2 * g], (double)s[k]);
            atomicAdd(&sh_g[2 * g + 1], (double)q[k]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * G; i += 256) atomicAdd(&stats[(long long)n * 2 * G + i], sh_g[i]);
}

// Channel concatenation of two token-major tensors (the U-Net decoder's `torch.cat([h, skip], dim=1)`, openaimodel3d.py:592) fused with
// the statistics pass of the GroupNorm that follows it (the first norm of the ResBlock the concatenation feeds): one read of both
// sources, one write of the concatenated rows, the sums on the way -- instead of a copy kernel and a second read of its result.
// Lane map of k_gn_stats_nsc over the octets of the OUTPUT row; an octet lies entirely in one source (Ca, Cb multiples of 8).
template <typename T>
__global__ void __launch_bounds__(256) k_cat2_stats_nsc(const T* __restrict__ xa, const T* __restrict__ xb, T* __restrict__ out,
                                                        double* __restrict__ stats, int Ca, int Cb, int G, long long S, int rows_per_block)
{
    typedef typename Tr<T>::vec8 vec8;
    extern __shared__ double sh_g[];  // [G][2]
    const int C = Ca + Cb, n = blockIdx.y, cpg = C / G, oct = C / 8, oct_a = Ca / 8;
    for (int i = threadIdx.x; i < 2 * G; i += 256) sh_g[i] = 0.0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < S) ? r0 + rows_per_block : S;
    const T* ba = xa + (long long)n * S * Ca;
    const T* bb = xb + (long long)n * S * Cb;
    T* bo = out + (long long)n * S * C;
    const int Wd = oct_width(oct), rstep = 256 / Wd, lane_row = threadIdx.x / Wd;
    for (int o = threadIdx.x % Wd; o < oct && lane_row < rstep; o += Wd) {
        const bool from_a = o < oct_a;
        const T* src = from_a ? ba + o * 8 : bb + (o - oct_a) * 8;
        const long long pitch = from_a ? Ca : Cb;
        float s[8], q[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { s[k] = 0.f; q[k] = 0.f; }
#pragma unroll 4
        for (long long r = r0 + lane_row; r < r1; r += rstep) {
            const vec8 v = *reinterpret_cast<const vec8*>(src + r * pitch);
            *reinterpret_cast<vec8*>(bo + r * C + o * 8) = v;
#pragma unroll
            for (int k = 0; k < 8; k++) { const float f = to_f(v[k]); s[k] += f; q[k] = fmaf(f, f, q[k]); }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int g = (o * 8 + k) / cpg;
            atomicAdd(&sh_g[2 * g], (double)s[k]);
            atomicAdd(&sh_g[2 * g + 1], (double)q[k]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * G; i += 256) atomicAdd(&stats[(long long)n * 2 * G + i], sh_g[i]);
}

// NSC apply: same lane map as the statistics kernel -- a thread keeps the (a, b) pairs of its channel octet in registers and
// streams down its rows (the earlier flat-index version re-read 64 B of coefficients per 16 B of data and paid a 64-bit division
// per vector).
template <typename T>
__global__ void __launch_bounds__(256) k_gn_apply_nsc(const T* __restrict__ x, T* __restrict__ y, const float2* __restrict__ coef,
                                                      int C, long long S, int silu, int rows_per_block)
{
    typedef typename Tr<T>::vec8 vec8;
    const int n = blockIdx.y, oct = C / 8;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < S) ? r0 + rows_per_block : S;
    const T* bx = x + (long long)n * S * C;
    T* by = y + (long long)n * S * C;
    const int Wd = oct_width(oct), rstep = 256 / Wd, lane_row = threadIdx.x / Wd;
    for (int o = threadIdx.x % Wd; o < oct && lane_row < rstep; o += Wd) {
        float a[8], b[8];
        const float4* cf = reinterpret_cast<const float4*>(coef + (long long)n * C + o * 8);
#pragma unroll
        for (int k = 0; k < 4; k++) { const float4 ab = cf[k]; a[2 * k] = ab.x; b[2 * k] = ab.y; a[2 * k + 1] = ab.z; b[2 * k + 1] = ab.w; }
#pragma unroll 4
        for (long long r = r0 + lane_row; r < r1; r += rstep) {
            vec8 v = *reinterpret_cast<const vec8*>(bx + r * C + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float f = fmaf(to_f(v[k]), a[k], b[k]);
                if (silu) f = silu_f(f);
                v[k] = (T)f;
            }
            *reinterpret_cast<vec8*>(by + r * C + o * 8) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward w.r.t. the input only (the guided sampler needs d/dx_t; weights are frozen).
//   z = a_c x + b_c (the forward's per-(n,c) affine),  y = silu(z) or z,  dz = dy silu'(z) or dy
//   dx = a_c dz - rstd/cnt * A - rstd^2/cnt * (x - mean) * Bs,   A = sum_g gamma_c dz,  Bs = rstd * (sum_g gamma_c dz x - mean A)
//      = a_c dz + k0 + k1 x        with per-group k1 = -rstd^2 Bs / cnt,  k0 = -rstd A / cnt - k1 mean
// Pass 1 accumulates (A, sum gamma dz x) per (n, group) in fp64; a tiny kernel forms (k0, k1) per (n, c); pass 2 streams.
__device__ __forceinline__ float silu_grad_f(float z)
{
    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * z));
    return sg * fmaf(z, 1.f - sg, 1.f);
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_bwd_stats_ncs(const T* __restrict__ x, const T* __restrict__ dy, const float2* __restrict__ coef,
                                                          const float* __restrict__ gamma, double* __restrict__ bstats,
                                                          int C, int G, long long S, int silu, int chunks)
{
    typedef typename Tr<T>::vec8 vec8;
    const long long grp = blockIdx.x;            // n * G + g
    const int cpg = C / G, g = (int)(grp % G);
    const long long n = grp / G, L = (long long)cpg * S;
    const T* px = x + grp * L;
    const T* pg = dy + grp * L;
    const long long per = ((L + chunks - 1) / chunks + 7) & ~7LL;
    const long long beg = (long long)blockIdx.y * per, end = (beg + per < L) ? beg + per : L;
    float sa = 0.f, sb = 0.f;
    const bool vec = (S & 7) == 0;
    for (long long i = beg + (long long)threadIdx.x * (vec ? 8 : 1); i < end; i += 256 * (vec ? 8 : 1)) {
        const int c = g * cpg + (int)(i / S);
        const float2 ab = coef[n * C + c];
        const float gm = gamma[c];
        if (vec) {
            const vec8 vx = *reinterpret_cast<const vec8*>(px + i), vg = *reinterpret_cast<const vec8*>(pg + i);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float xf = to_f(vx[k]);
                float dz = to_f(vg[k]);
                if (silu) dz *= silu_grad_f(fmaf(xf, ab.x, ab.y));
                dz *= gm;
                sa += dz;
                sb = fmaf(dz, xf, sb);
            }
        } else {
            const float xf = to_f(px[i]);
            float dz = to_f(pg[i]);
            if (silu) dz *= silu_grad_f(fmaf(xf, ab.x, ab.y));
            dz *= gm;
            sa += dz;
            sb = fmaf(dz, xf, sb);
        }
    }
    double da = wave_sum_d((double)sa), db = wave_sum_d((double)sb);
    __shared__ double sh[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[2 * w] = da; sh[2 * w + 1] = db; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&bstats[2 * grp], sh[0] + sh[2] + sh[4] + sh[6]);
        atomicAdd(&bstats[2 * grp + 1], sh[1] + sh[3] + sh[5] + sh[7]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_bwd_stats_nsc(const T* __restrict__ x, const T* __restrict__ dy, const float2* __restrict__ coef,
                                                          const float* __restrict__ gamma, double* __restrict__ bstats,
                                                          int C, int G, long long S, int silu, int rows_per_block)
{
    typedef typename Tr<T>::vec8 vec8;
    extern __shared__ double sh_g[];  // [G][2]; fp64: a block's strip is up to 8192 rows x cpg channels per group (a thread's
                                      // own fp32 partial spans <= rows / rstep of them)
    const int n = blockIdx.y, cpg = C / G, oct = C / 8;
    for (int i = threadIdx.x; i < 2 * G; i += 256) sh_g[i] = 0.0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < S) ? r0 + rows_per_block : S;
    const T* bx = x + (long long)n * S * C;
    const T* bg = dy + (long long)n * S * C;
    const int Wd = oct_width(oct), rstep = 256 / Wd, lane_row = threadIdx.x / Wd;
    for (int o = threadIdx.x % Wd; o < oct && lane_row < rstep; o += Wd) {
        float a[8], b[8], s1[8], s2[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const float2 ab = coef[(long long)n * C + o * 8 + k]; a[k] = ab.x; b[k] = ab.y; s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll 2
        for (long long r = r0 + lane_row; r < r1; r += rstep) {
            const vec8 vx = *reinterpret_cast<const vec8*>(bx + r * C + o * 8), vg = *reinterpret_cast<const vec8*>(bg + r * C + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float xf = to_f(vx[k]);
                float dz = to_f(vg[k]);
                if (silu) dz *= silu_grad_f(fmaf(xf, a[k], b[k]));
                s1[k] += dz;
                s2[k] = fmaf(dz, xf, s2[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = o * 8 + k;
            const float gm = gamma[c];
            atomicAdd(&sh_g[2 * (c / cpg)], (double)(gm * s1[k]));
            atomicAdd(&sh_g[2 * (c / cpg) + 1], (double)(gm * s2[k]));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * G; i += 256) atomicAdd(&bstats[(long long)n * 2 * G + i], sh_g[i]);
}

__global__ void __launch_bounds__(256) k_gn_bwd_coef(const double* __restrict__ stats, const double* __restrict__ bstats,
                                                     float2* __restrict__ coef2, int N, int C, int G, long long S, float eps)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C, cpg = C / G;
    const double cnt = (double)cpg * (double)S;
    const long long grp = (long long)n * G + c / cpg;
    const double mean = stats[2 * grp] / cnt;
    const double var = stats[2 * grp + 1] / cnt - mean * mean;
    const double rstd = (double)rsqrtf((float)(var > 0 ? var : 0) + eps);
    const double A = bstats[2 * grp], Bs = rstd * (bstats[2 * grp + 1] - mean * A);
    const double k1 = -rstd * rstd * Bs / cnt;
    coef2[i] = make_float2((float)(-rstd * A / cnt - k1 * mean), (float)k1);
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_bwd_apply_ncs(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ add, T* __restrict__ dx,
                                                          const float2* __restrict__ coef, const float2* __restrict__ coef2,
                                                          long long S, int silu, long long total)
{
    typedef typename Tr<T>::vec8 vec8;
    const bool vec = (S & 7) == 0;
    const long long step = (long long)gridDim.x * 256 * (vec ? 8 : 1);
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * (vec ? 8 : 1); i < total; i += step) {
        const float2 ab = coef[i / S], kk = coef2[i / S];
        if (vec) {
            const vec8 vx = *reinterpret_cast<const vec8*>(x + i), vg = *reinterpret_cast<const vec8*>(dy + i);
            vec8 r, va = vec8{};
            if (add) va = *reinterpret_cast<const vec8*>(add + i);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float xf = to_f(vx[k]);
                float dz = to_f(vg[k]);
                if (silu) dz *= silu_grad_f(fmaf(xf, ab.x, ab.y));
                r[k] = (T)(fmaf(ab.x, dz, fmaf(kk.y, xf, kk.x)) + to_f(va[k]));
            }
            *reinterpret_cast<vec8*>(dx + i) = r;
        } else {
            const float xf = to_f(x[i]);
            float dz = to_f(dy[i]);
            if (silu) dz *= silu_grad_f(fmaf(xf, ab.x, ab.y));
            dx[i] = (T)(fmaf(ab.x, dz, fmaf(kk.y, xf, kk.x)) + (add ? to_f(add[i]) : 0.f));
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_bwd_apply_nsc(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ add, T* __restrict__ dx,
                                                          const float2* __restrict__ coef, const float2* __restrict__ coef2,
                                                          int C, long long S, int silu, int rows_per_block)
{
    typedef typename Tr<T>::vec8 vec8;
    const int n = blockIdx.y, oct = C / 8;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < S) ? r0 + rows_per_block : S;
    const long long base = (long long)n * S * C;
    const int Wd = oct_width(oct), rstep = 256 / Wd, lane_row = threadIdx.x / Wd;
    for (int o = threadIdx.x % Wd; o < oct && lane_row < rstep; o += Wd) {
        float a[8], b[8], k0[8], k1[8];   // per-channel constants of this thread's octet, loaded once
        const float4* cf = reinterpret_cast<const float4*>(coef + (long long)n * C + o * 8);
        const float4* kf = reinterpret_cast<const float4*>(coef2 + (long long)n * C + o * 8);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 ab = cf[k], kk = kf[k];
            a[2 * k] = ab.x; b[2 * k] = ab.y; a[2 * k + 1] = ab.z; b[2 * k + 1] = ab.w;
            k0[2 * k] = kk.x; k1[2 * k] = kk.y; k0[2 * k + 1] = kk.z; k1[2 * k + 1] = kk.w;
        }
#pragma unroll 2
        for (long long r = r0 + lane_row; r < r1; r += rstep) {
            const long long i = base + r * C + o * 8;
            const vec8 vx = *reinterpret_cast<const vec8*>(x + i), vg = *reinterpret_cast<const vec8*>(dy + i);
            vec8 rr, va = vec8{};
            if (add) va = *reinterpret_cast<const vec8*>(add + i);   // (uniform) gradient arriving along the residual branch, see k_layer_norm_bwd
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float xf = to_f(vx[k]);
                float dz = to_f(vg[k]);
                if (silu) dz *= silu_grad_f(fmaf(xf, a[k], b[k]));
                rr[k] = (T)(fmaf(a[k], dz, fmaf(k1[k], xf, k0[k])) + to_f(va[k]));
            }
            *reinterpret_cast<vec8*>(dx + i) = rr;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim of [M, C] 16-bit rows (nn.LayerNorm in BasicTransformerBlock, attention.py:283-285).
// One wave per row, the row lives in registers (<= 4 octets per lane), fp32 two-pass statistics.
template <typename T>
__global__ void __launch_bounds__(256) k_layer_norm(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ gamma,
                                                    const T* __restrict__ beta, long long M, int C, float eps)
{
    typedef typename Tr<T>::vec8 vec8;
    const int lane = threadIdx.x & 63, oct = C >> 3;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + row * C;
    vec8 v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int o = lane + 64 * j;
        if (o < oct) {
            v[j] = *reinterpret_cast<const vec8*>(xr + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) s += to_f(v[j][k]);
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (lane + 64 * j < oct) {
#pragma unroll
            for (int k = 0; k < 8; k++) { const float d = to_f(v[j][k]) - mean; q = fmaf(d, d, q); }
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    T* yr = y + row * C;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int o = lane + 64 * j;
        if (o < oct) {
            const vec8 g = *reinterpret_cast<const vec8*>(gamma + o * 8), b = *reinterpret_cast<const vec8*>(beta + o * 8);
            vec8 r;
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = (T)fmaf((to_f(v[j][k]) - mean) * rstd, to_f(g[k]), to_f(b[k]));
            *reinterpret_cast<vec8*>(yr + o * 8) = r;
        }
    }
}

// Phi(g) = 0.5 (1 + erf(g / sqrt 2)) and E = exp(-g^2 / 2) without libdevice's erff (~40 VALU): the complementary error function in
// the Chebyshev form  erfc(z) = t exp(-z^2 + c(t)),  t = 1 / (1 + z / 2)  (Numerical Recipes' erfcc: FRACTIONAL error < 1.2e-7 for all
// z >= 0 -- the negative tail of the gate, where gelu is 1e-4 .. 1e-7, keeps its relative accuracy, which the 1.5e-7 ABSOLUTE bound of
// the shorter Abramowitz-Stegun form in gemm_mfma.hip's fused epilogue does not give; these row kernels sit at the HBM rate either
// way).  Phi = 1 - erfc/2 for g >= 0, erfc/2 for g < 0: no cancellation.  The two row kernels were VALU-bound with erff + expf
// (197 us for the 717 MB of a level-0 gate gradient: 3.6 TB/s); they now move their bytes at 5.5-5.8 TB/s.
// (gelu_cdf_exp: diffusion_common.h -- shared with the GEMM's fused gate epilogues)

// GEGLU gate (attention.py:415-423): y[m, c] = h[m, c] * gelu(h[m, C + c]) for h = proj(x) of width 2C; exact (erf) GELU.
template <typename T>
__global__ void __launch_bounds__(256) k_geglu(const T* __restrict__ h, T* __restrict__ y, long long M, int C)
{
    typedef typename Tr<T>::vec8 vec8;
    const int oct = C >> 3;
    const long long total = M * oct, step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
        const long long m = i / oct;
        const int o = (int)(i - m * oct);
        const T* hr = h + m * 2 * C + o * 8;
        const vec8 a = *reinterpret_cast<const vec8*>(hr), g = *reinterpret_cast<const vec8*>(hr + C);
        vec8 r;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float gf = to_f(g[k]);
            // torch evaluates gelu in fp32 and rounds to the 16-bit type before the product (two separate ops)
            float cdf, e;
            gelu_cdf_exp(gf, cdf, e);
            const T ge = (T)(gf * cdf);
            r[k] = (T)(to_f(a[k]) * to_f(ge));
        }
        *reinterpret_cast<vec8*>(y + m * C + o * 8) = r;
    }
}

// LayerNorm backward w.r.t. the input (weights frozen: guided sampler).  x_hat = (x - mu) rstd, g = dy gamma,
// dx = rstd (g - mean(g) - x_hat mean(g x_hat)) [+ add].  G lanes share a row (G = 8 for C <= 512, 16 for C <= 1024, 32 beyond:
// every lane owns up to 8 16-byte chunks, chunk o = sub + G i), a wave covers 64 / G rows, the statistics are recomputed from the
// row in registers.  (One wave per row, the first form, left 24 of 64 lanes idle at C = 320 and paid three 6-step wave reductions
// per 640-byte row: 98 us for the 56 000 rows of a level-0 activation at 320x448, 1.45 TB/s; `k_row_stats` had the same disease.)
template <typename T, int G>
__global__ void __launch_bounds__(256) k_layer_norm_bwd(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ gamma,
                                                        const T* __restrict__ add, T* __restrict__ dx, long long M, int C, float eps)
{
    typedef typename Tr<T>::vec8 vec8;
    constexpr int RPW = 64 / G, NV = 8;
    const int lane = threadIdx.x & 63, sub = lane & (G - 1), oct = C >> 3;
    const long long row_ = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + (lane / G);
    const bool live = row_ < M;
    const long long row = live ? row_ : M - 1;                // (dead rows recompute the last one and store nothing)
    const T* xr = x + row * C;
    const T* gr = dy + row * C;
    vec8 v[NV], g[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int o = sub + G * j;
        v[j] = vec8{};
        g[j] = vec8{};
        if (o < oct) {
            v[j] = *reinterpret_cast<const vec8*>(xr + o * 8);
            g[j] = *reinterpret_cast<const vec8*>(gr + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) s += to_f(v[j][k]);
        }
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++)
        if (sub + G * j < oct) {
#pragma unroll
            for (int k = 0; k < 8; k++) { const float d = to_f(v[j][k]) - mean; q = fmaf(d, d, q); }
        }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int o = sub + G * j;
        if (o < oct) {
            const vec8 gm = *reinterpret_cast<const vec8*>(gamma + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float gk = to_f(g[j][k]) * to_f(gm[k]);
                sg += gk;
                sgx = fmaf(gk, (to_f(v[j][k]) - mean) * rstd, sgx);
            }
        }
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) { sg += __shfl_xor(sg, d, 64); sgx += __shfl_xor(sgx, d, 64); }
    const float mg = sg / (float)C, mgx = sgx / (float)C;
    T* dr = dx + row * C;
    // `add`: the gradient that reaches the same tensor along the residual branch (x feeds LayerNorm -> ... and `+ x`): summed here in
    // fp32, one rounding, instead of a separate accumulation kernel over the two 16-bit gradients (3 more passes over [M, C])
    const T* ar = add ? add + row * C : nullptr;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int o = sub + G * j;
        if (o < oct && live) {
            vec8 r, av = vec8{};
            if (ar) av = *reinterpret_cast<const vec8*>(ar + o * 8);
            const vec8 gm = *reinterpret_cast<const vec8*>(gamma + o * 8);   // (again: 64 registers of products would not stay resident)
#pragma unroll
            for (int k = 0; k < 8; k++)
                r[k] = (T)(rstd * (to_f(g[j][k]) * to_f(gm[k]) - mg - (to_f(v[j][k]) - mean) * rstd * mgx) + to_f(av[k]));
            *reinterpret_cast<vec8*>(dr + o * 8) = r;
        }
    }
}

// GEGLU backward: y = a gelu(g) for h = [a | g]:  da = dy gelu(g),  dg = dy a gelu'(g),
// gelu'(g) = 0.5 (1 + erf(g / sqrt 2)) + g exp(-g^2 / 2) / sqrt(2 pi).
template <typename T>
__global__ void __launch_bounds__(256) k_geglu_bwd(const T* __restrict__ h, const T* __restrict__ dy, T* __restrict__ dh, long long M, int C)
{
    typedef typename Tr<T>::vec8 vec8;
    const int oct = C >> 3;
    const long long total = M * oct, step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
        const long long m = i / oct;
        const int o = (int)(i - m * oct);
        const T* hr = h + m * 2 * C + o * 8;
        const vec8 a = *reinterpret_cast<const vec8*>(hr), g = *reinterpret_cast<const vec8*>(hr + C);
        const vec8 d = *reinterpret_cast<const vec8*>(dy + m * C + o * 8);
        vec8 ra, rg;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float gf = to_f(g[k]), df = to_f(d[k]);
            float cdf, e;
            gelu_cdf_exp(gf, cdf, e);
            const float pdf = 0.3989422804014327f * e;
            ra[k] = (T)(df * gf * cdf);
            rg[k] = (T)(df * to_f(a[k]) * fmaf(gf, pdf, cdf));
        }
        T* dr = dh + m * 2 * C + o * 8;
        *reinterpret_cast<vec8*>(dr) = ra;
        *reinterpret_cast<vec8*>(dr + C) = rg;
    }
}

__global__ void k_profile_marker(int) {}

}  // namespace

extern "C" {

const char* gvd_diff_last_error(void) { return gvdd::g_err.c_str(); }

int gvd_profile_marker(int tag, void* stream_)
{
    hipLaunchKernelGGL(k_profile_marker, dim3(1), dim3(1), 0, (hipStream_t)stream_, tag);
    return 0;
}

int gvd_attention_fwd(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                      float scale, int is_bf16, void* stream_)
{
    return gvd_attention_fwd_strided(q, k, v, out, B, H, Nq, Nk, D, scale, (long long)Nq * H * D, (long long)H * D,
                                     (long long)Nk * H * D, (long long)H * D, nullptr, is_bf16, stream_);
}

int gvd_attention_fwd_strided(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                              float scale, long long q_bs, long long q_rs, long long kv_bs, long long kv_rs, float* lse,
                              int is_bf16, void* stream_)
{
    return gvd_attention_fwd_ex(q, k, v, out, B, H, Nq, Nk, D, scale, q_bs, q_rs, kv_bs, kv_rs, q_bs, q_rs, nullptr, 1.0f, lse,
                                is_bf16, stream_);
}

int gvd_attention_fwd_ex(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                         float scale, long long q_bs, long long q_rs, long long kv_bs, long long kv_rs, long long o_bs,
                         long long o_rs, const void* accum, float accum_scale, float* lse, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if ((q_bs | q_rs | kv_bs | kv_rs | o_bs | o_rs) & 7) return fail(-1, "gvd_attention_fwd: strides must be multiples of 8 elements");
    if ((uintptr_t)accum & 15) return fail(-1, "gvd_attention_fwd: accum must be 16-byte aligned");
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return fail(-1, "gvd_attention_fwd: bad arguments");
    if (D != 64) return fail(-1, "gvd_attention_fwd: head dim must be 64");
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return fail(-1, "gvd_attention_fwd: pointers must be 16-byte aligned");
    const float sl2 = scale * 1.4426950408889634f;
    const char* no_short = getenv("GVD_ATTN_NO_SHORT");   // (A/B switch: 1 = short rows on the general kernel, as before round 4)
    if (Nq <= 32 && Nk <= 32 && !accum && !(no_short && no_short[0] != '0')) {
        // short rows (temporal attention): a wave per (batch entry, head) item, grid-stride; 4 workgroups of 4 waves per CU
        const long long items = (long long)B * H;
        if (items > 0x7fffffffLL) return fail(-1, "gvd_attention_fwd: too many (batch, head) items");
        // grid = what is RESIDENT at once (the occupancy the compiler reached x 256 CUs): a second round of workgroups would start
        // when the first ends, i.e. with a tail as long as a wave's whole ~11-item walk
        static int s_res[2] = { 0, 0 };
        if (!s_res[is_bf16 ? 1 : 0]) {
            int per_cu = 0, dev = 0, cus = 256;
            hipError_t eo = is_bf16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_attn_short_fwd<__bf16>, 256, 0)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_attn_short_fwd<_Float16>, 256, 0);
            if (eo != hipSuccess || per_cu < 1) per_cu = 2;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            s_res[is_bf16 ? 1 : 0] = per_cu * (cus > 0 ? cus : 256);
        }
        const long long wgs = (items + 3) / 4, res = s_res[is_bf16 ? 1 : 0];
        dim3 sgrid((unsigned)(wgs < res ? wgs : res));
        if (is_bf16)
            hipLaunchKernelGGL((k_attn_short_fwd<__bf16>), sgrid, dim3(256), 0, stream, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v,
                               (__bf16*)out, lse, H, Nq, Nk, (int)items, sl2, q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs);
        else
            hipLaunchKernelGGL((k_attn_short_fwd<_Float16>), sgrid, dim3(256), 0, stream, (const _Float16*)q, (const _Float16*)k,
                               (const _Float16*)v, (_Float16*)out, lse, H, Nq, Nk, (int)items, sl2, q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return fail(-2, "launch k_attn_short_fwd", es);
        return 0;
    }
    // 4 waves x 2 query blocks (256 queries / workgroup) for long sequences, 4 x 1 for medium, one wave for short ones
    // (measured and not kept: 3 query blocks per wave at one wave per SIMD -- 446 registers, no spills -- ran 643 TFLOP/s at L0
    //  against 777 for 2 blocks x 2 waves per SIMD: the second resident wave hides more than the extra operand reuse saves)
    const int mode = Nq >= 512 ? 2 : (Nq > 64 ? 1 : 0);
    const int rows = mode == 2 ? 256 : (mode == 1 ? 128 : 32);
    dim3 grid((unsigned)(B * H), (unsigned)((Nq + rows - 1) / rows));
#define GVD_ATTN_LAUNCH(T, W, Q)                                                                                          \
    hipLaunchKernelGGL((k_attn_fwd<T, W, Q>), grid, dim3(W * 64), 0, stream, (const T*)q, (const T*)k, (const T*)v, (T*)out, lse, \
                       H, Nq, Nk, sl2, q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs, (const T*)accum, accum_scale)
    if (is_bf16) {
        if (mode == 2) GVD_ATTN_LAUNCH(__bf16, 4, 2); else if (mode == 1) GVD_ATTN_LAUNCH(__bf16, 4, 1); else GVD_ATTN_LAUNCH(__bf16, 1, 1);
    } else {
        if (mode == 2) GVD_ATTN_LAUNCH(_Float16, 4, 2); else if (mode == 1) GVD_ATTN_LAUNCH(_Float16, 4, 1); else GVD_ATTN_LAUNCH(_Float16, 1, 1);
    }
#undef GVD_ATTN_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_attn_fwd", e);
    return 0;
}

int gvd_ddim_step(const float* x, const float* e_cond, const float* e_uncond, const float* noise, float* x_prev, float* x0,
                  double* ws, long long n, float cfg_scale, float guidance_rescale, float sqrt_ac_t, float sqrt_1mac_t,
                  float sqrt_a_prev, float dir_coef, float sigma_t, float x0_rescale, float temperature, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !e_cond || !e_uncond || !noise || !x_prev || !x0 || !ws || n <= 1) return fail(-1, "gvd_ddim_step: bad arguments");
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (guidance_rescale > 0.f) {
        hipError_t e = hipMemsetAsync(ws, 0, 4 * sizeof(double), stream);
        if (e != hipSuccess) return fail(-2, "hipMemsetAsync(ws)", e);
        hipLaunchKernelGGL(k_ddim_stats, dim3(blocks), dim3(256), 0, stream, e_cond, e_uncond, n, cfg_scale, ws);
    }
    hipLaunchKernelGGL(k_ddim_apply, dim3(blocks), dim3(256), 0, stream, x, e_cond, e_uncond, noise, x_prev, x0, ws, n,
                       cfg_scale, guidance_rescale, sqrt_ac_t, sqrt_1mac_t, sqrt_a_prev, dir_coef, sigma_t * temperature, x0_rescale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_ddim_*", e);
    return 0;
}


namespace {
int gn_check(const char* who, const void* a, const void* b, int N, int C, long long S, int G, int channels_last)
{
    char msg[160];
    if (!a || !b || N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G) { snprintf(msg, sizeof msg, "%s: bad arguments", who); return fail(-1, msg); }
    if (((uintptr_t)a | (uintptr_t)b) & 15) { snprintf(msg, sizeof msg, "%s: tensors must be 16-byte aligned", who); return fail(-1, msg); }
    if (channels_last && (C % 8)) { snprintf(msg, sizeof msg, "%s: channels-last needs C %% 8 == 0", who); return fail(-1, msg); }
    return 0;
}
int gn_apply_blocks(int N, int C, long long S, int channels_last)
{
    const long long total = (long long)N * C * S;
    const long long vecs = channels_last || (S % 8 == 0) ? total / 8 : total;
    return (int)((vecs + 255) / 256 < 16384 ? (vecs + 255) / 256 : 16384);
}
int gn_rows_per_block(int N, long long S)
{
    // token-major statistics: rows of one block's strip.  ~1024 blocks in flight; short strips for small feature maps
    // (a 128-row strip is a 128-deep serial load chain per thread: 43 us on a 35-token map), 128 rows for large ones
    const long long r = ((long long)N * S + 1023) / 1024;
    return (int)(r < 4 ? 4 : (r > 128 ? 128 : r));
}
int gn_stat_rows(int N, long long S)
{
    // statistics kernels end in 2 G fp64 atomics per block onto N x 2 G addresses: beyond ~2048 blocks those same-address
    // atomics, not the reads, set the time (576x1024x128 map: 4608 blocks 168 us, 1.8 TB/s) -- so long strips for large maps
    const long long r = ((long long)N * S + 2047) / 2048;
    return (int)(r < 4 ? 4 : (r > 8192 ? 8192 : r));
}
int gn_chunks(int C, int G, long long S)
{
    const long long L = (long long)(C / G) * S;
    const int chunks = (int)((L + 32767) / 32768);
    return chunks < 1 ? 1 : (chunks > 256 ? 256 : chunks);
}
}  // namespace

int gvd_group_norm_stats(const void* x, double* stats, int N, int C, long long S, int G, int channels_last, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = gn_check("gvd_group_norm_stats", x, stats, N, C, S, G, channels_last)) return rc;
    hipError_t e = hipMemsetAsync(stats, 0, (size_t)N * G * 2 * sizeof(double), stream);
    if (e != hipSuccess) return fail(-2, "hipMemsetAsync(stats)", e);
    if (!channels_last) {
        const long long L = (long long)(C / G) * S;
        const int chunks = gn_chunks(C, G, S);
        dim3 grid((unsigned)(N * G), (unsigned)chunks);
        if (is_bf16) hipLaunchKernelGGL(k_gn_stats_ncs<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)x, stats, L, chunks);
        else hipLaunchKernelGGL(k_gn_stats_ncs<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)x, stats, L, chunks);
    } else {
        const int rows = gn_stat_rows(N, S);
        dim3 grid((unsigned)((S + rows - 1) / rows), (unsigned)N);
        if (is_bf16) hipLaunchKernelGGL(k_gn_stats_nsc<__bf16>, grid, dim3(256), (size_t)G * 16, stream, (const __bf16*)x, stats, C, G, S, rows);
        else hipLaunchKernelGGL(k_gn_stats_nsc<_Float16>, grid, dim3(256), (size_t)G * 16, stream, (const _Float16*)x, stats, C, G, S, rows);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_gn_stats_*", e);
    return 0;
}

int gvd_cat2_group_norm_stats(const void* xa, int Ca, const void* xb, int Cb, void* out, double* stats, int N, long long S, int G,
                              int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const int C = Ca + Cb;
    if (!xa || !xb || !out || !stats || N <= 0 || S <= 0 || Ca <= 0 || Cb <= 0 || G <= 0 || (Ca & 7) || (Cb & 7) || C % G)
        return fail(-1, "gvd_cat2_group_norm_stats: channel counts must be positive multiples of 8 and Ca + Cb a multiple of G");
    if (((uintptr_t)xa | (uintptr_t)xb | (uintptr_t)out) & 15) return fail(-1, "gvd_cat2_group_norm_stats: tensors must be 16-byte aligned");
    hipError_t e = hipMemsetAsync(stats, 0, (size_t)N * G * 2 * sizeof(double), stream);
    if (e != hipSuccess) return fail(-2, "hipMemsetAsync(stats)", e);
    const int rows = gn_stat_rows(N, S);
    dim3 grid((unsigned)((S + rows - 1) / rows), (unsigned)N);
    if (is_bf16) hipLaunchKernelGGL(k_cat2_stats_nsc<__bf16>, grid, dim3(256), (size_t)G * 16, stream, (const __bf16*)xa, (const __bf16*)xb, (__bf16*)out, stats, Ca, Cb, G, S, rows);
    else hipLaunchKernelGGL(k_cat2_stats_nsc<_Float16>, grid, dim3(256), (size_t)G * 16, stream, (const _Float16*)xa, (const _Float16*)xb, (_Float16*)out, stats, Ca, Cb, G, S, rows);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_cat2_stats_nsc", e);
    return 0;
}

int gvd_group_norm_apply(const void* x, void* y, const float* gamma, const float* beta, double* stats, int N, int C, long long S,
                         long long S_total, int G, float eps, int silu, int channels_last, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = gn_check("gvd_group_norm_apply", x, y, N, C, S, G, channels_last)) return rc;
    if (!gamma || !beta || !stats || S_total < S) return fail(-1, "gvd_group_norm_apply: bad arguments");
    float2* coef = reinterpret_cast<float2*>(stats + (size_t)N * G * 2);
    const long long total = (long long)N * C * S;
    const int ablocks = gn_apply_blocks(N, C, S, channels_last);
    hipLaunchKernelGGL(k_gn_coef, dim3((N * C + 255) / 256), dim3(256), 0, stream, stats, gamma, beta, coef, N, C, G, S_total, eps);
    if (!channels_last) {
        if (is_bf16) hipLaunchKernelGGL(k_gn_apply_ncs<__bf16>, dim3(ablocks), dim3(256), 0, stream, (const __bf16*)x, (__bf16*)y, coef, S, silu, total);
        else hipLaunchKernelGGL(k_gn_apply_ncs<_Float16>, dim3(ablocks), dim3(256), 0, stream, (const _Float16*)x, (_Float16*)y, coef, S, silu, total);
    } else {
        const int rows = gn_rows_per_block(N, S);
        dim3 grid((unsigned)((S + rows - 1) / rows), (unsigned)N);
        if (is_bf16) hipLaunchKernelGGL(k_gn_apply_nsc<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)x, (__bf16*)y, coef, C, S, silu, rows);
        else hipLaunchKernelGGL(k_gn_apply_nsc<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)x, (_Float16*)y, coef, C, S, silu, rows);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_gn_apply_*", e);
    return 0;
}

int gvd_group_norm(const void* x, void* y, const float* gamma, const float* beta, double* stats, int N, int C, long long S,
                   int G, float eps, int silu, int channels_last, int is_bf16, void* stream_)
{
    if (int rc = gvd_group_norm_stats(x, stats, N, C, S, G, channels_last, is_bf16, stream_)) return rc;
    return gvd_group_norm_apply(x, y, gamma, beta, stats, N, C, S, S, G, eps, silu, channels_last, is_bf16, stream_);
}

int gvd_group_norm_bwd_stats(const void* x, const void* dy, const float* gamma, const double* fwd_stats, double* scratch,
                             int N, int C, long long S, int G, int silu, int channels_last, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = gn_check("gvd_group_norm_bwd_stats", x, dy, N, C, S, G, channels_last)) return rc;
    if (!gamma || !fwd_stats || !scratch) return fail(-1, "gvd_group_norm_bwd_stats: bad arguments");
    hipError_t e = hipMemsetAsync(scratch, 0, (size_t)N * G * 2 * sizeof(double), stream);
    if (e != hipSuccess) return fail(-2, "hipMemsetAsync(scratch)", e);
    const float2* coef = reinterpret_cast<const float2*>(fwd_stats + (size_t)N * G * 2);
    if (!channels_last) {
        const int chunks = gn_chunks(C, G, S);
        dim3 grid((unsigned)(N * G), (unsigned)chunks);
        if (is_bf16) hipLaunchKernelGGL(k_gn_bwd_stats_ncs<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)dy, coef, gamma, scratch, C, G, S, silu, chunks);
        else hipLaunchKernelGGL(k_gn_bwd_stats_ncs<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)x, (const _Float16*)dy, coef, gamma, scratch, C, G, S, silu, chunks);
    } else {
        const int rows = gn_stat_rows(N, S);
        dim3 grid((unsigned)((S + rows - 1) / rows), (unsigned)N);
        if (is_bf16) hipLaunchKernelGGL(k_gn_bwd_stats_nsc<__bf16>, grid, dim3(256), (size_t)G * 16, stream, (const __bf16*)x, (const __bf16*)dy, coef, gamma, scratch, C, G, S, silu, rows);
        else hipLaunchKernelGGL(k_gn_bwd_stats_nsc<_Float16>, grid, dim3(256), (size_t)G * 16, stream, (const _Float16*)x, (const _Float16*)dy, coef, gamma, scratch, C, G, S, silu, rows);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_gn_bwd_stats_*", e);
    return 0;
}

int gvd_group_norm_bwd_apply_add(const void* x, const void* dy, const void* add, void* dx, const double* fwd_stats, double* scratch, int N, int C,
                             long long S, long long S_total, int G, float eps, int silu, int channels_last, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = gn_check("gvd_group_norm_bwd_apply", x, dy, N, C, S, G, channels_last)) return rc;
    if (!dx || (((uintptr_t)dx | (uintptr_t)add) & 15) || !fwd_stats || !scratch || S_total < S) return fail(-1, "gvd_group_norm_bwd_apply: bad arguments");
    const float2* coef = reinterpret_cast<const float2*>(fwd_stats + (size_t)N * G * 2);
    float2* coef2 = reinterpret_cast<float2*>(scratch + (size_t)N * G * 2);
    const long long total = (long long)N * C * S;
    const int ablocks = gn_apply_blocks(N, C, S, channels_last);
    hipLaunchKernelGGL(k_gn_bwd_coef, dim3((N * C + 255) / 256), dim3(256), 0, stream, fwd_stats, (const double*)scratch, coef2, N, C, G, S_total, eps);
    if (!channels_last) {
        if (is_bf16) hipLaunchKernelGGL(k_gn_bwd_apply_ncs<__bf16>, dim3(ablocks), dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)dy, (const __bf16*)add, (__bf16*)dx, coef, (const float2*)coef2, S, silu, total);
        else hipLaunchKernelGGL(k_gn_bwd_apply_ncs<_Float16>, dim3(ablocks), dim3(256), 0, stream, (const _Float16*)x, (const _Float16*)dy, (const _Float16*)add, (_Float16*)dx, coef, (const float2*)coef2, S, silu, total);
    } else {
        const int rows = gn_rows_per_block(N, S);
        dim3 grid((unsigned)((S + rows - 1) / rows), (unsigned)N);
        if (is_bf16) hipLaunchKernelGGL(k_gn_bwd_apply_nsc<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)dy, (const __bf16*)add, (__bf16*)dx, coef, (const float2*)coef2, C, S, silu, rows);
        else hipLaunchKernelGGL(k_gn_bwd_apply_nsc<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)x, (const _Float16*)dy, (const _Float16*)add, (_Float16*)dx, coef, (const float2*)coef2, C, S, silu, rows);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_gn_bwd_apply_*", e);
    return 0;
}

int gvd_group_norm_bwd_apply(const void* x, const void* dy, void* dx, const double* fwd_stats, double* scratch, int N, int C,
                             long long S, long long S_total, int G, float eps, int silu, int channels_last, int is_bf16, void* stream_)
{
    return gvd_group_norm_bwd_apply_add(x, dy, nullptr, dx, fwd_stats, scratch, N, C, S, S_total, G, eps, silu, channels_last, is_bf16, stream_);
}

int gvd_group_norm_bwd(const void* x, const void* dy, void* dx, const float* gamma, const double* fwd_stats, double* scratch,
                       int N, int C, long long S, int G, float eps, int silu, int channels_last, int is_bf16, void* stream_)
{
    if (int rc = gvd_group_norm_bwd_stats(x, dy, gamma, fwd_stats, scratch, N, C, S, G, silu, channels_last, is_bf16, stream_)) return rc;
    return gvd_group_norm_bwd_apply(x, dy, dx, fwd_stats, scratch, N, C, S, S, G, eps, silu, channels_last, is_bf16, stream_);
}

int gvd_layer_norm(const void* x, void* y, const void* gamma, const void* beta, long long M, int C, float eps, int is_bf16,
                   void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || !gamma || !beta || M <= 0 || C <= 0 || (C % 8) || C > 2048) return fail(-1, "gvd_layer_norm: bad arguments (C % 8 == 0, C <= 2048)");
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return fail(-1, "gvd_layer_norm: pointers must be 16-byte aligned");
    const dim3 grid((unsigned)((M + 3) / 4));
    if (is_bf16) hipLaunchKernelGGL(k_layer_norm<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)x, (__bf16*)y, (const __bf16*)gamma, (const __bf16*)beta, M, C, eps);
    else hipLaunchKernelGGL(k_layer_norm<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)x, (_Float16*)y, (const _Float16*)gamma, (const _Float16*)beta, M, C, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_layer_norm", e);
    return 0;
}

int gvd_geglu(const void* h, void* y, long long M, int C, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!h || !y || M <= 0 || C <= 0 || (C % 8)) return fail(-1, "gvd_geglu: bad arguments (C % 8 == 0)");
    if (((uintptr_t)h | (uintptr_t)y) & 15) return fail(-1, "gvd_geglu: pointers must be 16-byte aligned");
    const long long vecs = M * (C / 8);
    const int blocks = (int)((vecs + 255) / 256 < 32768 ? (vecs + 255) / 256 : 32768);
    if (is_bf16) hipLaunchKernelGGL(k_geglu<__bf16>, dim3(blocks), dim3(256), 0, stream, (const __bf16*)h, (__bf16*)y, M, C);
    else hipLaunchKernelGGL(k_geglu<_Float16>, dim3(blocks), dim3(256), 0, stream, (const _Float16*)h, (_Float16*)y, M, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_geglu", e);
    return 0;
}

int gvd_layer_norm_bwd_add(const void* x, const void* dy, const void* gamma, const void* add, void* dx, long long M, int C, float eps,
                           int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !dy || !gamma || !dx || M <= 0 || C <= 0 || (C % 8) || C > 2048) return fail(-1, "gvd_layer_norm_bwd: bad arguments (C % 8 == 0, C <= 2048)");
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)add) & 15) return fail(-1, "gvd_layer_norm_bwd: pointers must be 16-byte aligned");
    const int G = C <= 512 ? 8 : (C <= 1024 ? 16 : 32);
    const long long rows_per_block = 4 * (64 / G);
    const dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block));
#define GVD_LNB(T, GG) hipLaunchKernelGGL((k_layer_norm_bwd<T, GG>), grid, dim3(256), 0, stream, (const T*)x, (const T*)dy, (const T*)gamma, (const T*)add, (T*)dx, M, C, eps)
    if (is_bf16) { if (G == 8) GVD_LNB(__bf16, 8); else if (G == 16) GVD_LNB(__bf16, 16); else GVD_LNB(__bf16, 32); }
    else { if (G == 8) GVD_LNB(_Float16, 8); else if (G == 16) GVD_LNB(_Float16, 16); else GVD_LNB(_Float16, 32); }
#undef GVD_LNB
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_layer_norm_bwd", e);
    return 0;
}

int gvd_layer_norm_bwd(const void* x, const void* dy, const void* gamma, void* dx, long long M, int C, float eps, int is_bf16,
                       void* stream_)
{
    return gvd_layer_norm_bwd_add(x, dy, gamma, nullptr, dx, M, C, eps, is_bf16, stream_);
}

int gvd_geglu_bwd(const void* h, const void* dy, void* dh, long long M, int C, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!h || !dy || !dh || M <= 0 || C <= 0 || (C % 8)) return fail(-1, "gvd_geglu_bwd: bad arguments (C % 8 == 0)");
    if (((uintptr_t)h | (uintptr_t)dy | (uintptr_t)dh) & 15) return fail(-1, "gvd_geglu_bwd: pointers must be 16-byte aligned");
    const long long vecs = M * (C / 8);
    const int blocks = (int)((vecs + 255) / 256 < 32768 ? (vecs + 255) / 256 : 32768);
    if (is_bf16) hipLaunchKernelGGL(k_geglu_bwd<__bf16>, dim3(blocks), dim3(256), 0, stream, (const __bf16*)h, (const __bf16*)dy, (__bf16*)dh, M, C);
    else hipLaunchKernelGGL(k_geglu_bwd<_Float16>, dim3(blocks), dim3(256), 0, stream, (const _Float16*)h, (const _Float16*)dy, (_Float16*)dh, M, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_geglu_bwd", e);
    return 0;
}

}  // extern "C"
