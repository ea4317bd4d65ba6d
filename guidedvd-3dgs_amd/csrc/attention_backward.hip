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

    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    const size_t rs = (size_t)q_rs, krs = (size_t)kv_rs;
    const T* qb = q + (size_t)b * q_bs + (size_t)h * 64;
    const T* gb = d_o + (size_t)b * o_bs + (size_t)h * 64;   // d_out has its own strides: q may be a column block of a packed projection
    const size_t kvoff = (size_t)b * kv_bs + (size_t)h * 64;
    const float* lse_b = lse + (size_t)bh * Nq;
    const float* delta_b = delta + (size_t)bh * Nq;

    const int key = blockIdx.y * 128 + wave * 32 + col;
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

    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    const size_t rs = (size_t)q_rs, krs = (size_t)kv_rs;
    const size_t qoff = (size_t)b * q_bs + (size_t)h * 64;
    const T* kb = k + (size_t)b * kv_bs + (size_t)h * 64;
    const T* vb = v + (size_t)b * kv_bs + (size_t)h * 64;

    const int query = blockIdx.y * 128 + wave * 32 + col;
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

template <typename T>
int launch_bwd(const void* q, const void* k, const void* v, const void* out, const void* d_out, const float* lse, float* delta,
               void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, float scale, long long q_bs, long long q_rs,
               long long kv_bs, long long kv_rs, long long o_bs, long long o_rs, hipStream_t stream)
{
    const float sl2 = scale * 1.4426950408889634f;
    const long long total = (long long)B * H * Nq;
    hipLaunchKernelGGL(k_attn_delta<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const T*)out, (const T*)d_out,
                       delta, H, Nq, total, o_bs, o_rs);
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
    if (!q || !k || !v || !out || !d_out || !lse || !delta || !dq || !dk || !dv || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0)
        return fail(-1, "gvd_attention_bwd: bad arguments");
    if (D != 64) return fail(-1, "gvd_attention_bwd: head dim must be 64");
    if ((q_bs | q_rs | kv_bs | kv_rs | o_bs | o_rs) & 7) return fail(-1, "gvd_attention_bwd: strides must be multiples of 8 elements");
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)d_out | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15)
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
