// attention_backward.hip -- flash-attention backward on MFMA 32x32x16 for gfx950 (C-ABI: gvd_attention_bwd_strided).
//
// Used by the guided DDIM step only (DDIMSamplerGuidance differentiates pred_x0 w.r.t. x_t through every attention
// layer of both U-Net evaluations; ddim_guidance.py:318-345).  The reference gets this from xformers'
// memory_efficient_attention backward (un-vendored); the math is the explicit softmax-attention gradient:
//     P = softmax(scale Q K^T),  O = P V
//     dV = P^T dO,   dP = dO V^T,   dS = P o (dP - rowsum(dO o O)),   dQ = scale dS K,   dK = scale dS^T Q
//
// Deterministic two-pass design (no atomics):
//   k_attn_delta     delta[b,h,i] = sum_d dO[i,d] O[i,d]                                   (HBM-bound, tiny)
//   k_attn_bwd_dkv   one wave = 32 KEYS (K, V fragments stay in registers as MFMA B operands); the block walks the
//                    Q / dO tiles, staged in LDS both row-major (A operands of S = Q K^T and dP = dO V^T) and
//                    transposed (A operands of dK^T += Q^T dS, dV^T += dO^T P).  S, dP come out of the MFMA with one
//                    KEY per lane column and 16 query rows per lane, which is already the B-operand shape of the
//                    second products after a 16-bit pack + v_permlane32_swap (same trick as the forward kernel).
//   k_attn_bwd_dq    one wave = 32 QUERIES (Q, dO fragments in registers, lse / delta one scalar per lane); walks
//                    the K / V tiles: S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T.
// P is recomputed from the forward's log2-domain log-sum-exp: P = exp2(s * scale*log2(e) - lse).
#include <stdlib.h>

#include "diffusion_common.h"

using namespace gvdd;

namespace {

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

template <typename T>
__global__ void __launch_bounds__(256) k_attn_delta(const T* __restrict__ o, const T* __restrict__ d_o, float* __restrict__ delta,
                                                    int H, int Nq, long long total, long long o_bs, long long o_rs)   // (out / d_out addressing)
{
    typedef typename Tr<T>::vec8 vec8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // (b*H + h) * Nq + n
    if (i >= total) return;
    const long long bh = i / Nq;
    const int n = (int)(i - bh * Nq), b = (int)(bh / H), h = (int)(bh - (long long)b * H);
    const size_t off = (size_t)b * o_bs + (size_t)n * o_rs + (size_t)h * 64;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const vec8 a = *reinterpret_cast<const vec8*>(o + off + 8 * c), g = *reinterpret_cast<const vec8*>(d_o + off + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; j++) s = fmaf(to_f(a[j]), to_f(g[j]), s);
    }
    delta[i] = s;
}

// Staging of a 64-row x 64-channel tile of `src` (rows row0.., zero beyond n_rows) into LDS row-major and/or transposed, split
// into a LOAD half (global -> registers) and a COMMIT half (registers -> LDS) so that the loads of tile i+1 are in flight
// under the MFMAs of tile i.  256 threads.
//   row-major:  chunk c = tid, tid + 256 -> row c>>3, channels (c&7)*8 (coalesced 128-byte rows, ds_write_b128)
//   transposed: thread -> row PAIR tid&31, channels (tid>>5)*8: two rows packed per dword, conflict-free ds_write_b32
// All loads are UNCONDITIONAL (rows past the end read the last valid row and are zeroed at commit time): a predicated
// `ok ? load : 0` makes hipcc wait vmcnt(0) between the loads -- the first version of these kernels spent ~8 dependent L2 round
// trips per tile in its (also un-pipelined) staging and ran the MFMAs 12 % of the time.  (Double-buffering the LDS tiles on top,
// one barrier per tile instead of two, measured within noise: 13.2 vs 12.9 ms at L0 -- not kept.)
template <typename T> struct TileRegs {
    typename Tr<T>::vec8 r[2], t0, t1;
};

template <typename T, bool TRANSPOSED>
__device__ __forceinline__ void tile_load(const T* __restrict__ src, size_t rs, int row0, int n_rows, int tid, TileRegs<T>& R)
{
    typedef typename Tr<T>::vec8 vec8;
    const int last = n_rows - 1;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = tid + i * 256, g = row0 + (c >> 3);
        R.r[i] = *reinterpret_cast<const vec8*>(src + (size_t)(g < last ? g : last) * rs + (c & 7) * 8);
    }
    if (TRANSPOSED) {
        const int g = row0 + 2 * (tid & 31), c8 = (tid >> 5) * 8;
        R.t0 = *reinterpret_cast<const vec8*>(src + (size_t)(g < last ? g : last) * rs + c8);
        R.t1 = *reinterpret_cast<const vec8*>(src + (size_t)(g + 1 < last ? g + 1 : last) * rs + c8);
    }
}

template <typename T, bool TRANSPOSED>
__device__ __forceinline__ void tile_commit(int row0, int n_rows, int tid, const TileRegs<T>& R, T (*sRow)[LDS_ROW], T (*sTr)[LDS_ROW])
{
    typedef typename Tr<T>::vec8 vec8;
    typedef T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = tid + i * 256, row = c >> 3;
        *reinterpret_cast<vec8*>(&sRow[row][(c & 7) * 8]) = (row0 + row < n_rows) ? R.r[i] : vec8{};
    }
    if (TRANSPOSED) {
        const int kp = tid & 31, c8 = (tid >> 5) * 8, g = row0 + 2 * kp;
        const vec8 x0 = (g < n_rows) ? R.t0 : vec8{}, x1 = (g + 1 < n_rows) ? R.t1 : vec8{};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const T2 pr = { x0[i], x1[i] };
            *reinterpret_cast<T2*>(&sTr[c8 + i][2 * kp]) = pr;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
k_attn_bwd_dkv(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ d_o,
               const float* __restrict__ lse, const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv,
               int H, int Nq, int Nk, float scale_log2e, float scale, long long q_bs, long long q_rs, long long kv_bs, long long kv_rs,
               long long o_bs, long long o_rs)
{
    typedef typename Tr<T>::vec8 vec8;
    __shared__ __attribute__((aligned(16))) T sQ[64][LDS_ROW];
    __shared__ __attribute__((aligned(16))) T sdO[64][LDS_ROW];
    __shared__ __attribute__((aligned(16))) T sQt[64][LDS_ROW];
    __shared__ __attribute__((aligned(16))) T sdOt[64][LDS_ROW];
    __shared__ __attribute__((aligned(16))) float sLse[64];
    __shared__ __attribute__((aligned(16))) float sDelta[64];

    int bh, tile_y;
    xcd_item_tile(bh, tile_y);
    const int b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    const size_t rs = (size_t)q_rs, krs = (size_t)kv_rs;
    const T* qb = q + (size_t)b * q_bs + (size_t)h * 64;
    const T* gb = d_o + (size_t)b * o_bs + (size_t)h * 64;   // d_out has its own strides: q may be a column block of a packed projection
    const size_t kvoff = (size_t)b * kv_bs + (size_t)h * 64;
    const float* lse_b = lse + (size_t)bh * Nq;
    const float* delta_b = delta + (size_t)bh * Nq;

    const int key = tile_y * 128 + wave * 32 + col;
    const bool valid_k = key < Nk;
    vec8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const size_t kr = (size_t)(valid_k ? key : Nk - 1) * krs + 16 * ks + 8 * hi;   // unconditional load, zeroed below
        kf[ks] = *reinterpret_cast<const vec8*>(k + kvoff + kr);
        vf[ks] = *reinterpret_cast<const vec8*>(v + kvoff + kr);
        if (!valid_k) { kf[ks] = vec8{}; vf[ks] = vec8{}; }
    }
    f16v dk0 = {}, dk1 = {}, dv0 = {}, dv1 = {};

    TileRegs<T> rq, rg;
    float rl = 0.f, rd = 0.f;
    auto prefetch = [&](int qt) {
        tile_load<T, true>(qb, rs, qt, Nq, tid, rq);
        tile_load<T, true>(gb, (size_t)o_rs, qt, Nq, tid, rg);
        const int r = qt + (tid & 63) < Nq ? qt + (tid & 63) : Nq - 1;
        rl = lse_b[r];
        rd = delta_b[r];
    };
    prefetch(0);
    for (int qt = 0; qt < Nq; qt += 64) {
        __syncthreads();   // every wave is done with the previous tile
        tile_commit<T, true>(qt, Nq, tid, rq, sQ, sQt);
        tile_commit<T, true>(qt, Nq, tid, rg, sdO, sdOt);
        if (tid < 64) {
            const bool ok = qt + tid < Nq;
            sLse[tid] = ok ? rl : 3.0e38f;   // rows past the end: P = exp2(0 - huge) = 0
            sDelta[tid] = ok ? rd : 0.f;
        }
        if (qt + 64 < Nq) prefetch(qt + 64);   // in flight under this tile's MFMAs
        __syncthreads();
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
            // S[q][key] and dP[q][key] for 32 queries x this wave's 32 keys: lane = key column, 16 query rows
            f16v s = {}, dp = {};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const vec8 aq = *reinterpret_cast<const vec8*>(&sQ[32 * sb + col][16 * ks + 8 * hi]);
                const vec8 ag = *reinterpret_cast<const vec8*>(&sdO[32 * sb + col][16 * ks + 8 * hi]);
                s = Tr<T>::mfma(aq, kf[ks], s);
                dp = Tr<T>::mfma(ag, vf[ks], dp);
            }
            unsigned pp[8], pd[8];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {   // rows 8*g4 + 4*hi + {0..3}
                const float4 L = *reinterpret_cast<const float4*>(&sLse[32 * sb + 8 * g4 + 4 * hi]);
                const float4 D = *reinterpret_cast<const float4*>(&sDelta[32 * sb + 8 * g4 + 4 * hi]);
                const float p0 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 0], scale_log2e, -L.x));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 1], scale_log2e, -L.y));
                const float p2 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 2], scale_log2e, -L.z));
                const float p3 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 3], scale_log2e, -L.w));
                pp[2 * g4] = Tr<T>::pack2(p0, p1);
                pp[2 * g4 + 1] = Tr<T>::pack2(p2, p3);
                pd[2 * g4] = Tr<T>::pack2(p0 * (dp[4 * g4 + 0] - D.x), p1 * (dp[4 * g4 + 1] - D.y));
                pd[2 * g4 + 1] = Tr<T>::pack2(p2 * (dp[4 * g4 + 2] - D.z), p3 * (dp[4 * g4 + 3] - D.w));
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
                const vec8 pf = packed_c_to_b_operand<T>(pp, k2), df = packed_c_to_b_operand<T>(pd, k2);
                const int qc = 32 * sb + 16 * k2 + 8 * hi;
                const vec8 g0 = *reinterpret_cast<const vec8*>(&sdOt[col][qc]);
                const vec8 g1 = *reinterpret_cast<const vec8*>(&sdOt[32 + col][qc]);
                const vec8 q0 = *reinterpret_cast<const vec8*>(&sQt[col][qc]);
                const vec8 q1 = *reinterpret_cast<const vec8*>(&sQt[32 + col][qc]);
                dv0 = Tr<T>::mfma(g0, pf, dv0);
                dv1 = Tr<T>::mfma(g1, pf, dv1);
                dk0 = Tr<T>::mfma(q0, df, dk0);
                dk1 = Tr<T>::mfma(q1, df, dk1);
            }
        }
    }
    // keys past Nk hold K = V = 0 fragments and are simply not written
    if (valid_k) {
        T* dkr = dk + kvoff + (size_t)key * krs;
        T* dvr = dv + kvoff + (size_t)key * krs;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int d0 = 8 * rg + 4 * hi;
            uint2 w;
            w.x = Tr<T>::pack2(dk0[4 * rg] * scale, dk0[4 * rg + 1] * scale);
            w.y = Tr<T>::pack2(dk0[4 * rg + 2] * scale, dk0[4 * rg + 3] * scale);
            *reinterpret_cast<uint2*>(dkr + d0) = w;
            w.x = Tr<T>::pack2(dk1[4 * rg] * scale, dk1[4 * rg + 1] * scale);
            w.y = Tr<T>::pack2(dk1[4 * rg + 2] * scale, dk1[4 * rg + 3] * scale);
            *reinterpret_cast<uint2*>(dkr + 32 + d0) = w;
            w.x = Tr<T>::pack2(dv0[4 * rg], dv0[4 * rg + 1]);
            w.y = Tr<T>::pack2(dv0[4 * rg + 2], dv0[4 * rg + 3]);
            *reinterpret_cast<uint2*>(dvr + d0) = w;
            w.x = Tr<T>::pack2(dv1[4 * rg], dv1[4 * rg + 1]);
            w.y = Tr<T>::pack2(dv1[4 * rg + 2], dv1[4 * rg + 3]);
            *reinterpret_cast<uint2*>(dvr + 32 + d0) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
k_attn_bwd_dq(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ d_o,
              const float* __restrict__ lse, const float* __restrict__ delta, T* __restrict__ dq,
              int H, int Nq, int Nk, float scale_log2e, float scale, long long q_bs, long long q_rs, long long kv_bs, long long kv_rs,
              long long o_bs, long long o_rs)
{
    typedef typename Tr<T>::vec8 vec8;
    __shared__ __attribute__((aligned(16))) T sK[64][LDS_ROW];
    __shared__ __attribute__((aligned(16))) T sV[64][LDS_ROW];
    __shared__ __attribute__((aligned(16))) T sKt[64][LDS_ROW];

    int bh, tile_y;
    xcd_item_tile(bh, tile_y);
    const int b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    const size_t rs = (size_t)q_rs, krs = (size_t)kv_rs;
    const size_t qoff = (size_t)b * q_bs + (size_t)h * 64;
    const T* kb = k + (size_t)b * kv_bs + (size_t)h * 64;
    const T* vb = v + (size_t)b * kv_bs + (size_t)h * 64;

    const int query = tile_y * 128 + wave * 32 + col;
    const bool valid_q = query < Nq;
    vec8 qf[4], gf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const size_t qr = (size_t)(valid_q ? query : Nq - 1) * rs + 16 * ks + 8 * hi;   // unconditional load, zeroed below
        qf[ks] = *reinterpret_cast<const vec8*>(q + qoff + qr);
        gf[ks] = *reinterpret_cast<const vec8*>(d_o + (size_t)b * o_bs + (size_t)h * 64 + (size_t)(valid_q ? query : Nq - 1) * (size_t)o_rs + 16 * ks + 8 * hi);
        if (!valid_q) { qf[ks] = vec8{}; gf[ks] = vec8{}; }
    }
    const float L = valid_q ? lse[(size_t)bh * Nq + query] : 3.0e38f;
    const float Dl = valid_q ? delta[(size_t)bh * Nq + query] : 0.f;
    f16v dq0 = {}, dq1 = {};

    TileRegs<T> rk, rv;
    tile_load<T, true>(kb, krs, 0, Nk, tid, rk);
    tile_load<T, false>(vb, krs, 0, Nk, tid, rv);
    for (int kt = 0; kt < Nk; kt += 64) {
        __syncthreads();   // every wave is done with the previous tile
        tile_commit<T, true>(kt, Nk, tid, rk, sK, sKt);
        tile_commit<T, false>(kt, Nk, tid, rv, sV, nullptr);
        if (kt + 64 < Nk) {   // in flight under this tile's MFMAs
            tile_load<T, true>(kb, krs, kt + 64, Nk, tid, rk);
            tile_load<T, false>(vb, krs, kt + 64, Nk, tid, rv);
        }
        __syncthreads();
#pragma unroll
        for (int kbk = 0; kbk < 2; kbk++) {
            // S^T[key][q], dP^T[key][q] for 32 keys x this wave's 32 queries: lane = query column, 16 key rows
            f16v s = {}, dp = {};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const vec8 ak = *reinterpret_cast<const vec8*>(&sK[32 * kbk + col][16 * ks + 8 * hi]);
                const vec8 av = *reinterpret_cast<const vec8*>(&sV[32 * kbk + col][16 * ks + 8 * hi]);
                s = Tr<T>::mfma(ak, qf[ks], s);
                dp = Tr<T>::mfma(av, gf[ks], dp);
            }
            unsigned pd[8];
            const bool tail = kt + 64 > Nk;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float p0 = __builtin_amdgcn_exp2f(fmaf(s[2 * j], scale_log2e, -L));
                float p1 = __builtin_amdgcn_exp2f(fmaf(s[2 * j + 1], scale_log2e, -L));
                if (tail) {   // keys past the end have K = 0 -> s = 0 -> p = exp2(-lse) != 0: mask them
                    const int r0 = 2 * j, r1 = 2 * j + 1;
                    if (kt + 32 * kbk + (r0 & 3) + 8 * (r0 >> 2) + 4 * hi >= Nk) p0 = 0.f;
                    if (kt + 32 * kbk + (r1 & 3) + 8 * (r1 >> 2) + 4 * hi >= Nk) p1 = 0.f;
                }
                pd[j] = Tr<T>::pack2(p0 * (dp[2 * j] - Dl), p1 * (dp[2 * j + 1] - Dl));
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
                const vec8 df = packed_c_to_b_operand<T>(pd, k2);
                const int kc = 32 * kbk + 16 * k2 + 8 * hi;
                const vec8 k0 = *reinterpret_cast<const vec8*>(&sKt[col][kc]);
                const vec8 k1 = *reinterpret_cast<const vec8*>(&sKt[32 + col][kc]);
                dq0 = Tr<T>::mfma(k0, df, dq0);
                dq1 = Tr<T>::mfma(k1, df, dq1);
            }
        }
    }
    if (valid_q) {
        T* dqr = dq + qoff + (size_t)query * rs;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int d0 = 8 * rg + 4 * hi;
            uint2 w;
            w.x = Tr<T>::pack2(dq0[4 * rg] * scale, dq0[4 * rg + 1] * scale);
            w.y = Tr<T>::pack2(dq0[4 * rg + 2] * scale, dq0[4 * rg + 3] * scale);
            *reinterpret_cast<uint2*>(dqr + d0) = w;
            w.x = Tr<T>::pack2(dq1[4 * rg] * scale, dq1[4 * rg + 1] * scale);
            w.y = Tr<T>::pack2(dq1[4 * rg + 2] * scale, dq1[4 * rg + 3] * scale);
            *reinterpret_cast<uint2*>(dqr + 32 + d0) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Short rows (Nq, Nk <= 32: the temporal attention of the guided step).  The three kernels above run such a problem as one
// workgroup per (batch entry, head) with 37-46 KB of 64-row LDS tiles and two barriers -- a few waves per CU, 107 + 66 + 15 us per
// level-0 call at 320x448 for 290 MB of traffic.  Here ONE WAVE owns an item and produces dQ, dK and dV in one pass, like the
// forward (`k_attn_short_fwd`): Q, K, V, dO in their MFMA operand registers (the same registers serve as A operand of one product
// and B operand of its transpose), both orientations of the score block computed (S^T / dP^T with a query per lane for dQ, S / dP
// with a key per lane for dK and dV: 16 small MFMAs instead of a register transposition), delta = rowsum(P o dP) taken from the
// query-per-lane form (== rowsum(dO o O): O is not read at all), Q^T / K^T / dO^T through wave-private LDS, results transposed
// back through LDS into whole 128-byte rows.  No barriers, no atomics, deterministic.
template <typename T>
__global__ void __launch_bounds__(256, 2) k_attn_short_bwd(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        const T* __restrict__ d_o, const float* __restrict__ lse, T* __restrict__ dq,
                                                        T* __restrict__ dk, T* __restrict__ dv, int H, int Nq, int Nk, int items,
                                                        float scale_log2e, float scale, long long q_bs, long long q_rs, long long kv_bs,
                                                        long long kv_rs, long long o_bs, long long o_rs)
{
    typedef typename Tr<T>::vec8 vec8;
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr int TP = 40, OP = 72;
    __shared__ __attribute__((aligned(16))) T sQt[4][64][TP];
    __shared__ __attribute__((aligned(16))) T sKt[4][64][TP];
    __shared__ __attribute__((aligned(16))) T sGt[4][64][TP];
    __shared__ __attribute__((aligned(16))) T sO[4][32][OP];
    __shared__ __attribute__((aligned(16))) float sL[4][32];
    __shared__ __attribute__((aligned(16))) float sD[4][32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, r32 = lane & 31;
    const int nw = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    if (gw >= items) return;
    // operand form: lane (row r32, half hi) holds channels 16 ks + 8 hi .. + 7 of its row (rows past the end: the last row again)
    const int qr = r32 < Nq ? r32 : Nq - 1, kr = r32 < Nk ? r32 : Nk - 1;
    const size_t qo = (size_t)qr * (size_t)q_rs + 8 * hi, go = (size_t)qr * (size_t)o_rs + 8 * hi, ko = (size_t)kr * (size_t)kv_rs + 8 * hi;
    // transposition form: lane -> row pair (2 kp, 2 kp + 1), channel octets vo and vo + 4
    const int kp = lane & 15, vo = lane >> 4;
    const int q0r = 2 * kp < Nq ? 2 * kp : Nq - 1, q1r = 2 * kp + 1 < Nq ? 2 * kp + 1 : Nq - 1;
    const int k0r = 2 * kp < Nk ? 2 * kp : Nk - 1, k1r = 2 * kp + 1 < Nk ? 2 * kp + 1 : Nk - 1;
    const size_t tq0 = (size_t)q0r * (size_t)q_rs + 8 * vo, tq1 = (size_t)q1r * (size_t)q_rs + 8 * vo;
    const size_t tg0 = (size_t)q0r * (size_t)o_rs + 8 * vo, tg1 = (size_t)q1r * (size_t)o_rs + 8 * vo;
    const size_t tk0 = (size_t)k0r * (size_t)kv_rs + 8 * vo, tk1 = (size_t)k1r * (size_t)kv_rs + 8 * vo;

    vec8 qf[4], kf[4], vf[4], gf[4];
    auto load_ops = [&](int it) {
        const int b = it / H, h = it - b * H;
        const T* qb = q + (size_t)b * q_bs + (size_t)h * 64 + qo;
        const T* gb = d_o + (size_t)b * o_bs + (size_t)h * 64 + go;
        const size_t kvb = (size_t)b * kv_bs + (size_t)h * 64 + ko;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            qf[ks] = *reinterpret_cast<const vec8*>(qb + 16 * ks);
            kf[ks] = *reinterpret_cast<const vec8*>(k + kvb + 16 * ks);
            vf[ks] = *reinterpret_cast<const vec8*>(v + kvb + 16 * ks);
            gf[ks] = *reinterpret_cast<const vec8*>(gb + 16 * ks);
        }
    };
    // 32 rows x 64 channels of a C-layout pair (x0: channels 0-31, x1: 32-63; a lane owns one ROW's 16 + 16 values) -> global
    // rows of 128 bytes, through the wave's staging block
    auto store_rows = [&](const f16v& x0, const f16v& x1, float mul, T* base, size_t rstride, int nrows) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int d0 = 8 * rg + 4 * hi;
            uint2 w0, w1;
            w0.x = Tr<T>::pack2(x0[4 * rg] * mul, x0[4 * rg + 1] * mul);
            w0.y = Tr<T>::pack2(x0[4 * rg + 2] * mul, x0[4 * rg + 3] * mul);
            w1.x = Tr<T>::pack2(x1[4 * rg] * mul, x1[4 * rg + 1] * mul);
            w1.y = Tr<T>::pack2(x1[4 * rg + 2] * mul, x1[4 * rg + 3] * mul);
            *reinterpret_cast<uint2*>(&sO[wave][r32][d0]) = w0;
            *reinterpret_cast<uint2*>(&sO[wave][r32][32 + d0]) = w1;
        }
        __builtin_amdgcn_wave_barrier();   // (compiler only: the read-back below is of OTHER lanes' writes; the LDS itself is in order)
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const int row = (lane >> 3) + 8 * ps;
            const uint4 w = *reinterpret_cast<const uint4*>(&sO[wave][row][8 * (lane & 7)]);
            if (row < nrows) *reinterpret_cast<uint4*>(base + (size_t)row * rstride + 8 * (lane & 7)) = w;
        }
    };

    load_ops(gw);
    for (int it = gw; it < items; it += nw) {
        const int b = it / H, h = it - b * H;
        const size_t qbase = (size_t)b * q_bs + (size_t)h * 64, gbase = (size_t)b * o_bs + (size_t)h * 64;
        const size_t kbase = (size_t)b * kv_bs + (size_t)h * 64;
        // ---- the rows once more in transposition form (L2 hits of the lines the operand loads pulled) ----
        vec8 xq[2][2], xk[2][2], xg[2][2];
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            xk[ps][0] = *reinterpret_cast<const vec8*>(k + kbase + tk0 + 32 * ps);
            xk[ps][1] = *reinterpret_cast<const vec8*>(k + kbase + tk1 + 32 * ps);
            xq[ps][0] = *reinterpret_cast<const vec8*>(q + qbase + tq0 + 32 * ps);
            xq[ps][1] = *reinterpret_cast<const vec8*>(q + qbase + tq1 + 32 * ps);
            xg[ps][0] = *reinterpret_cast<const vec8*>(d_o + gbase + tg0 + 32 * ps);
            xg[ps][1] = *reinterpret_cast<const vec8*>(d_o + gbase + tg1 + 32 * ps);
        }
        const float L = lse[(size_t)it * Nq + qr];
        // ---- query per lane: S^T = K Q^T, dP^T = V dO^T; P^T, delta, dS^T ----
        f16v sT = {}, dpT = {};
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            sT = Tr<T>::mfma(kf[ks], qf[ks], sT);
            dpT = Tr<T>::mfma(vf[ks], gf[ks], dpT);
        }
        float p[16], dl = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float pr = __builtin_amdgcn_exp2f(fmaf(sT[r], scale_log2e, -L));
            if ((r & 3) + 8 * (r >> 2) + 4 * hi >= Nk) pr = 0.f;
            p[r] = pr;
            dl = fmaf(pr, dpT[r], dl);
        }
        dl += __shfl_xor(dl, 32, 64);
        unsigned pdT[8];
#pragma unroll
        for (int j = 0; j < 8; j++) pdT[j] = Tr<T>::pack2(p[2 * j] * (dpT[2 * j] - dl), p[2 * j + 1] * (dpT[2 * j + 1] - dl));
        if (hi == 0) {
            sL[wave][r32] = r32 < Nq ? L : 3.0e38f;   // queries past the end: P = exp2(s - huge) = 0 in the key-per-lane form
            sD[wave][r32] = dl;
        }
        // ---- Q^T, K^T, dO^T into the wave's LDS ----
#pragma unroll
        for (int ps = 0; ps < 2; ps++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int ch = (vo + 4 * ps) * 8 + i;
                const T2 pk_ = { xk[ps][0][i], xk[ps][1][i] }, pq_ = { xq[ps][0][i], xq[ps][1][i] }, pg_ = { xg[ps][0][i], xg[ps][1][i] };
                *reinterpret_cast<T2*>(&sKt[wave][ch][2 * kp]) = pk_;
                *reinterpret_cast<T2*>(&sQt[wave][ch][2 * kp]) = pq_;
                *reinterpret_cast<T2*>(&sGt[wave][ch][2 * kp]) = pg_;
            }
        __builtin_amdgcn_wave_barrier();   // (compiler only: what follows reads other lanes' LDS writes)
        // ---- dQ^T = K^T dS^T ----
        f16v a0 = {}, a1 = {};
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            const vec8 df = packed_c_to_b_operand<T>(pdT, k2);
            const vec8 k0 = *reinterpret_cast<const vec8*>(&sKt[wave][r32][16 * k2 + 8 * hi]);
            const vec8 k1 = *reinterpret_cast<const vec8*>(&sKt[wave][32 + r32][16 * k2 + 8 * hi]);
            a0 = Tr<T>::mfma(k0, df, a0);
            a1 = Tr<T>::mfma(k1, df, a1);
        }
        // ---- key per lane: S = Q K^T, dP = dO V^T ----
        f16v s = {}, dp = {};
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            s = Tr<T>::mfma(qf[ks], kf[ks], s);
            dp = Tr<T>::mfma(gf[ks], vf[ks], dp);
        }
        // ---- the next item's operands: in flight under everything below ----
        load_ops(it + nw < items ? it + nw : items - 1);
        store_rows(a0, a1, scale, dq + qbase, (size_t)q_rs, Nq);
        unsigned pp[8], pd[8];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {   // query rows 8 g4 + 4 hi + {0..3}
            const float4 L4 = *reinterpret_cast<const float4*>(&sL[wave][8 * g4 + 4 * hi]);
            const float4 D4 = *reinterpret_cast<const float4*>(&sD[wave][8 * g4 + 4 * hi]);
            const float p0 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 0], scale_log2e, -L4.x));
            const float p1 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 1], scale_log2e, -L4.y));
            const float p2 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 2], scale_log2e, -L4.z));
            const float p3 = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + 3], scale_log2e, -L4.w));
            pp[2 * g4] = Tr<T>::pack2(p0, p1);
            pp[2 * g4 + 1] = Tr<T>::pack2(p2, p3);
            pd[2 * g4] = Tr<T>::pack2(p0 * (dp[4 * g4 + 0] - D4.x), p1 * (dp[4 * g4 + 1] - D4.y));
            pd[2 * g4 + 1] = Tr<T>::pack2(p2 * (dp[4 * g4 + 2] - D4.z), p3 * (dp[4 * g4 + 3] - D4.w));
        }
        // ---- dV^T = dO^T P, dK^T = Q^T dS ----
        f16v dv0 = {}, dv1 = {}, dk0 = {}, dk1 = {};
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            const vec8 pf = packed_c_to_b_operand<T>(pp, k2), df = packed_c_to_b_operand<T>(pd, k2);
            const int qc = 16 * k2 + 8 * hi;
            const vec8 g0 = *reinterpret_cast<const vec8*>(&sGt[wave][r32][qc]);
            const vec8 g1 = *reinterpret_cast<const vec8*>(&sGt[wave][32 + r32][qc]);
            const vec8 q0 = *reinterpret_cast<const vec8*>(&sQt[wave][r32][qc]);
            const vec8 q1 = *reinterpret_cast<const vec8*>(&sQt[wave][32 + r32][qc]);
            dv0 = Tr<T>::mfma(g0, pf, dv0);
            dv1 = Tr<T>::mfma(g1, pf, dv1);
            dk0 = Tr<T>::mfma(q0, df, dk0);
            dk1 = Tr<T>::mfma(q1, df, dk1);
        }
        store_rows(dv0, dv1, 1.0f, dv + kbase, (size_t)kv_rs, Nk);
        store_rows(dk0, dk1, scale, dk + kbase, (size_t)kv_rs, Nk);
    }
}

template <typename T>
int launch_bwd(const void* q, const void* k, const void* v, const void* out, const void* d_out, const float* lse, float* delta,
               void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, float scale, long long q_bs, long long q_rs,
               long long kv_bs, long long kv_rs, long long o_bs, long long o_rs, hipStream_t stream)
{
    const float sl2 = scale * 1.4426950408889634f;
    const char* no_short = getenv("GVD_ATTN_NO_SHORT");   // (A/B switch: 1 = short rows on the three general kernels)
    if (Nq <= 32 && Nk <= 32 && dk && dv && !(no_short && no_short[0] != '0')) {
        // short rows (temporal attention): one wave per (batch entry, head) item, dQ / dK / dV in one pass; `out` and `delta` unused
        const long long items = (long long)B * H;
        if (items > 0x7fffffffLL) return fail(-1, "gvd_attention_bwd: too many (batch, head) items");
        static int s_res = 0;   // resident workgroups (per element type the same kernel shape: take the f16 figure)
        if (!s_res) {
            int per_cu = 0, dev = 0, cus = 256;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_attn_short_bwd<T>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            s_res = per_cu * (cus > 0 ? cus : 256);
        }
        const long long wgs = (items + 3) / 4;
        hipLaunchKernelGGL(k_attn_short_bwd<T>, dim3((unsigned)(wgs < s_res ? wgs : s_res)), dim3(256), 0, stream, (const T*)q, (const T*)k,
                           (const T*)v, (const T*)d_out, lse, (T*)dq, (T*)dk, (T*)dv, H, Nq, Nk, (int)items, sl2, scale, q_bs, q_rs, kv_bs,
                           kv_rs, o_bs, o_rs);
        const hipError_t es = hipGetLastError();
        if (es != hipSuccess) return fail(-2, "launch k_attn_short_bwd", es);
        return 0;
    }
    const long long total = (long long)B * H * Nq;
    hipLaunchKernelGGL(k_attn_delta<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const T*)out, (const T*)d_out,
                       delta, H, Nq, total, o_bs, o_rs);
    if (dk && dv)   // (both null: the keys / values carry no gradient -- the frame-invariant context of the cross-attention)
        hipLaunchKernelGGL(k_attn_bwd_dkv<T>, dim3((unsigned)(B * H), (unsigned)((Nk + 127) / 128)), dim3(256), 0, stream, (const T*)q,
                           (const T*)k, (const T*)v, (const T*)d_out, lse, (const float*)delta, (T*)dk, (T*)dv, H, Nq, Nk, sl2, scale,
                           q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs);
    hipLaunchKernelGGL(k_attn_bwd_dq<T>, dim3((unsigned)(B * H), (unsigned)((Nq + 127) / 128)), dim3(256), 0, stream, (const T*)q,
                       (const T*)k, (const T*)v, (const T*)d_out, lse, (const float*)delta, (T*)dq, H, Nq, Nk, sl2, scale,
                       q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_attn_bwd_*", e);
    return 0;
}

}  // namespace

extern "C" int gvd_attention_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                                    const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq,
                                    int Nk, int D, float scale, long long q_bs, long long q_rs, long long kv_bs,
                                    long long kv_rs, long long o_bs, long long o_rs, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !k || !v || !out || !d_out || !lse || !delta || !dq || (!dk != !dv) || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0)
        return fail(-1, "gvd_attention_bwd: bad arguments (dk and dv: both or neither)");
    if (D != 64) return fail(-1, "gvd_attention_bwd: head dim must be 64");
    if ((q_bs | q_rs | kv_bs | kv_rs | o_bs | o_rs) & 7) return fail(-1, "gvd_attention_bwd: strides must be multiples of 8 elements");
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)d_out | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15)   // (null dk / dv pass)
        return fail(-1, "gvd_attention_bwd: pointers must be 16-byte aligned");
    if (is_bf16) return launch_bwd<__bf16>(q, k, v, out, d_out, lse, delta, dq, dk, dv, B, H, Nq, Nk, scale, q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs, stream);
    return launch_bwd<_Float16>(q, k, v, out, d_out, lse, delta, dq, dk, dv, B, H, Nq, Nk, scale, q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs, stream);
}

extern "C" int gvd_attention_bwd_strided(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                                         const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq,
                                         int Nk, int D, float scale, long long q_bs, long long q_rs, long long kv_bs,
                                         long long kv_rs, int is_bf16, void* stream_)
{
    return gvd_attention_bwd_ex(q, k, v, out, d_out, lse, delta, dq, dk, dv, B, H, Nq, Nk, D, scale, q_bs, q_rs, kv_bs, kv_rs, q_bs,
                                q_rs, is_bf16, stream_);
}
