// knn.hip -- exact 3-nearest-neighbour mean squared distance for gfx950 (C-ABI: include/gvd_knn.h).
//
// What the reference computes (simple_knn.cu:149-186): for every point, the three smallest (squared distance, position
// in Morton order) pairs over all other points.  How it is computed here (HBM / LDS-bound integer + fp32 work, no MFMA):
//   k_bounds_*     two-level min/max of the cloud, initial value 0 like the reference's reduce (bounds stay on device)
//   k_morton       30-bit Morton codes, same fp32 expression as the reference -> same codes
//   k_radix_*      stable LSD radix sort of (code, id), four 8-bit passes (histogram per 2048-key tile -> one exclusive scan over
//                  (digit, tile) -> stable scatter: a key's slot is the scanned base of its (digit, tile) + the keys of the same
//                  digit in earlier 64-key chunks of the tile + its rank among its wave's lanes with that digit, taken from eight
//                  ballots).  The reference sorts with cub::DeviceRadixSort (simple_knn.cu:208-216); a stable sort by the whole key
//                  has one result, so the order is the same.  No library call is left in this package.
//   k_gather_box   sorted positions -> contiguous float4 array (kills the points[indices[i]] indirection of the
//                  reference's inner loop) + per-1024-point box bounds
//   k_knn          one workgroup = 256 consecutive sorted points (spatially coherent).  For every box that ANY of its
//                  points cannot prune, the box's 1024 points are staged once in LDS (16 KB) and each interested lane
//                  scans them with broadcast ds_reads; candidates arrive in sorted order with strict-'>' insertion, so
//                  ties resolve exactly as in the reference.  Pruning supersets are harmless: a pruned box only holds
//                  candidates strictly farther than the current third-best.
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/gvd_knn.h"

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

constexpr int BOX = 1024;      // simple_knn.cu:12
constexpr int QB = 256;        // queries per workgroup
constexpr int RED_BLOCKS = 256;

struct Bounds { float mn[3], mx[3]; };
struct BoxMM { float mn[3], mx[3]; };

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

__global__ void __launch_bounds__(256) k_bounds_partial(const float* __restrict__ pts, int P, Bounds* __restrict__ part)
{
    float mn[3] = { 0.f, 0.f, 0.f }, mx[3] = { 0.f, 0.f, 0.f };   // init {0,0,0}: simple_knn.cu:194
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
        for (int a = 0; a < 3; a++) { const float c = pts[3 * (size_t)i + a]; mn[a] = fminf(mn[a], c); mx[a] = fmaxf(mx[a], c); }
    __shared__ float sh[4][6];
    for (int a = 0; a < 3; a++) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) { sh[threadIdx.x >> 6][a] = mn[a]; sh[threadIdx.x >> 6][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        part[blockIdx.x].mn[a] = fminf(fminf(sh[0][a], sh[1][a]), fminf(sh[2][a], sh[3][a]));
        part[blockIdx.x].mx[a] = fmaxf(fmaxf(sh[0][3 + a], sh[1][3 + a]), fmaxf(sh[2][3 + a], sh[3][3 + a]));
    }
}

__global__ void __launch_bounds__(64) k_bounds_final(const Bounds* __restrict__ part, int n, Bounds* __restrict__ out)
{
    float mn[3] = { 0.f, 0.f, 0.f }, mx[3] = { 0.f, 0.f, 0.f };
    for (int i = threadIdx.x; i < n; i += 64)
        for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], part[i].mn[a]); mx[a] = fmaxf(mx[a], part[i].mx[a]); }
    for (int a = 0; a < 3; a++) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if (threadIdx.x == 0)
        for (int a = 0; a < 3; a++) { out->mn[a] = mn[a]; out->mx[a] = mx[a]; }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)   // simple_knn.cu:46-53
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256) k_morton(const float* __restrict__ pts, int P, const Bounds* __restrict__ bd,
                                                uint32_t* __restrict__ codes, uint32_t* __restrict__ ids)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t c = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float mn = bd->mn[a], mx = bd->mx[a];
        const float t = __fmul_rn(__fdiv_rn(__fsub_rn(pts[3 * (size_t)i + a], mn), __fsub_rn(mx, mn)), 1023.0f);
        c |= prep_morton((uint32_t)t) << a;   // float -> uint32 truncation, simple_knn.cu:57-59
    }
    codes[i] = c;
    ids[i] = (uint32_t)i;
}

// sorted position -> float4 (x, y, z, original id bits) and the bounds of each 1024-point box (simple_knn.cu:71-117)
__global__ void __launch_bounds__(BOX) k_gather_box(const float* __restrict__ pts, int P, const uint32_t* __restrict__ order,
                                                    float4* __restrict__ sp, BoxMM* __restrict__ boxes)
{
    const int i = blockIdx.x * BOX + threadIdx.x;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    if (i < P) {
        const uint32_t id = order[i];
        const float x = pts[3 * (size_t)id], y = pts[3 * (size_t)id + 1], z = pts[3 * (size_t)id + 2];
        sp[i] = make_float4(x, y, z, __uint_as_float(id));
        mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
    }
    __shared__ float sh[BOX / 64][6];
    for (int a = 0; a < 3; a++) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) { sh[threadIdx.x >> 6][a] = mn[a]; sh[threadIdx.x >> 6][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = sh[0][a], hi = sh[0][3 + a];
        for (int w = 1; w < BOX / 64; w++) { lo = fminf(lo, sh[w][a]); hi = fmaxf(hi, sh[w][3 + a]); }
        boxes[blockIdx.x].mn[a] = lo;
        boxes[blockIdx.x].mx[a] = hi;
    }
}

__device__ __forceinline__ float dist_box_point(const BoxMM& b, float x, float y, float z)   // simple_knn.cu:119-129
{
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (x < b.mn[0] || x > b.mx[0]) dx = fminf(fabsf(x - b.mn[0]), fabsf(x - b.mx[0]));
    if (y < b.mn[1] || y > b.mx[1]) dy = fminf(fabsf(y - b.mn[1]), fabsf(y - b.mx[1]));
    if (z < b.mn[2] || z > b.mx[2]) dz = fminf(fabsf(z - b.mn[2]), fabsf(z - b.mx[2]));
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

__device__ __forceinline__ void update3(float px, float py, float pz, const float4 q, float* best, int* bi, int qpos)
{
    const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
    float dist = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));   // nvcc's contraction of dx*dx+dy*dy+dz*dz
    int pi = qpos;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            const int ti = bi[j];
            best[j] = dist; bi[j] = pi;
            dist = t; pi = ti;
        }
    }
}

__global__ void __launch_bounds__(QB) k_knn(int P, const float4* __restrict__ sp, const BoxMM* __restrict__ boxes, int n_boxes,
                                            float* __restrict__ mean_dists, int* __restrict__ nearest)
{
    __shared__ float4 s_pts[BOX];
    const int idx = blockIdx.x * QB + threadIdx.x;
    const bool live = idx < P;
    float4 me = live ? sp[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    int bi[3] = { -1, -1, -1 };
    if (live) {   // the +-3 window of the sorted order gives the rejection radius (simple_knn.cu:159-170)
        const int lo = idx - 3 > 0 ? idx - 3 : 0, hi = idx + 3 < P - 1 ? idx + 3 : P - 1;
        for (int i = lo; i <= hi; i++)
            if (i != idx) update3(me.x, me.y, me.z, sp[i], best, bi, i);
    }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    bi[0] = bi[1] = bi[2] = -1;

    for (int b = 0; b < n_boxes; b++) {
        bool want = false;
        if (live) {
            const float d = dist_box_point(boxes[b], me.x, me.y, me.z);
            want = !(d > reject || d > best[2]);
        }
        if (!__syncthreads_or(want)) continue;   // nobody in this workgroup can be improved by box b
        const int base = b * BOX, n = (P - base) < BOX ? (P - base) : BOX;
        for (int i = threadIdx.x; i < n; i += QB) s_pts[i] = sp[base + i];
        __syncthreads();
        if (want) {
            for (int i = 0; i < n; i++)
                if (base + i != idx) update3(me.x, me.y, me.z, s_pts[i], best, bi, base + i);
        }
        __syncthreads();
    }
    if (live) {
        const uint32_t id = __float_as_uint(me.w);
        mean_dists[id] = __fdiv_rn(__fadd_rn(__fadd_rn(best[0], best[1]), best[2]), 3.0f);
#pragma unroll
        for (int j = 0; j < 3; j++) nearest[3 * (size_t)id + j] = bi[j] < 0 ? 0 : (int)__float_as_uint(sp[bi[j]].w);
    }
}

// ---- stable LSD radix sort of (key, value) pairs, 8 bits per pass ----
constexpr int RS_TILE = 2048, RS_THREADS = 256, RS_CHUNKS = RS_TILE / 64;   // a tile's order: chunk c = keys [64 c, 64 c + 64), lane = key

// digit counts of every tile: hist[d * n_tiles + tile]
__global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const uint32_t* __restrict__ keys, int n, int shift, int n_tiles,
                                                           uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
    for (int i = threadIdx.x; i < RS_TILE; i += RS_THREADS)
        if (base + i < n) atomicAdd(&s_h[(keys[base + i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = s_h[threadIdx.x];
}

// exclusive scan of hist in (digit, tile) order, in place, by one workgroup: thread t owns a contiguous run
__global__ void __launch_bounds__(1024) k_radix_scan(uint32_t* __restrict__ hist, int count)
{
    __shared__ uint32_t s_w[16];
    const int per = (count + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < count ? lo + per : count;
    uint32_t sum = 0;
    for (int i = lo; i < hi; i++) sum += hist[i];
    uint32_t inc = sum;                                        // inclusive scan over the 1024 thread sums
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((threadIdx.x & 63) >= o) inc += t; }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += s_w[w];
    for (int i = lo; i < hi; i++) { const uint32_t v = hist[i]; hist[i] = run; run += v; }
}

__global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int n,
                                                              int shift, int n_tiles, const uint32_t* __restrict__ bases,
                                                              uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t s_cnt[RS_CHUNKS][256];                 // keys of digit d in chunk c, then: ... in the chunks before c
    for (int i = threadIdx.x; i < RS_CHUNKS * 256; i += RS_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t key[RS_TILE / RS_THREADS], val[RS_TILE / RS_THREADS], rank[RS_TILE / RS_THREADS];
#pragma unroll
    for (int r = 0; r < RS_TILE / RS_THREADS; r++) {
        const int chunk = r * (RS_THREADS / 64) + wave, i = base + chunk * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keys[i] : 0u;
        val[r] = valid ? vals[i] : 0u;
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);            // lanes of this chunk holding the same digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        rank[r] = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank[r] == 0) s_cnt[chunk][d] = (uint32_t)__popcll(peers);
    }
    __syncthreads();
    {   // thread d: exclusive prefix of its digit over the tile's chunks, started at the digit's scanned base for this tile
        uint32_t run = bases[(size_t)threadIdx.x * n_tiles + blockIdx.x];
        for (int c = 0; c < RS_CHUNKS; c++) { const uint32_t v = s_cnt[c][threadIdx.x]; s_cnt[c][threadIdx.x] = run; run += v; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_TILE / RS_THREADS; r++) {
        const int chunk = r * (RS_THREADS / 64) + wave, i = base + chunk * 64 + lane;
        if (i < n) {
            const uint32_t dst = s_cnt[chunk][(key[r] >> shift) & 255u] + rank[r];
            keys_out[dst] = key[r];
            vals_out[dst] = val[r];
        }
    }
}

struct Layout {
    size_t part, bounds, codes, ids, codes_s, order, sp, boxes, sort_tmp, sort_bytes, total;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

Layout make_layout(int P)
{
    Layout L{};
    const size_t n = (size_t)(P > 0 ? P : 1), nb = (n + BOX - 1) / BOX;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
    L.part = take(sizeof(Bounds) * RED_BLOCKS);
    L.bounds = take(sizeof(Bounds));
    L.codes = take(4 * n);
    L.ids = take(4 * n);
    L.codes_s = take(4 * n);
    L.order = take(4 * n);
    L.sp = take(16 * n);
    L.boxes = take(sizeof(BoxMM) * nb);
    L.sort_bytes = 4 * 256 * ((n + RS_TILE - 1) / RS_TILE);      // the (digit, tile) histogram of a radix pass
    L.sort_tmp = take(L.sort_bytes);
    L.total = off;
    return L;
}

}  // namespace

extern "C" {

const char* gvd_knn_last_error(void) { return g_err.c_str(); }

size_t gvd_knn_workspace_bytes(int P) { return make_layout(P).total; }

int gvd_knn_mean_dist(const float* points, int P, float* mean_dists, int* nearest_idx, void* workspace, size_t workspace_bytes,
                      void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0) return fail(-1, "gvd_knn_mean_dist: negative point count");
    if (P == 0) return 0;
    if (!points || !mean_dists || !nearest_idx || !workspace) return fail(-1, "gvd_knn_mean_dist: null pointer");
    const Layout L = make_layout(P);
    if (workspace_bytes < L.total) return fail(-1, "gvd_knn_mean_dist: workspace too small (see gvd_knn_workspace_bytes)");
    if ((uintptr_t)workspace & 255) return fail(-1, "gvd_knn_mean_dist: workspace must be 256-byte aligned");
    char* ws = (char*)workspace;
    Bounds* part = (Bounds*)(ws + L.part);
    Bounds* bounds = (Bounds*)(ws + L.bounds);
    uint32_t *codes = (uint32_t*)(ws + L.codes), *ids = (uint32_t*)(ws + L.ids), *codes_s = (uint32_t*)(ws + L.codes_s),
             *order = (uint32_t*)(ws + L.order);
    float4* sp = (float4*)(ws + L.sp);
    BoxMM* boxes = (BoxMM*)(ws + L.boxes);
    const int n_boxes = (P + BOX - 1) / BOX;
    const int rb = (P + 255) / 256 < RED_BLOCKS ? (P + 255) / 256 : RED_BLOCKS;

    hipLaunchKernelGGL(k_bounds_partial, dim3(rb), dim3(256), 0, stream, points, P, part);
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(64), 0, stream, (const Bounds*)part, rb, bounds);
    hipLaunchKernelGGL(k_morton, dim3((P + 255) / 256), dim3(256), 0, stream, points, P, (const Bounds*)bounds, codes, ids);
    {   // four stable 8-bit passes, ping-pong (codes, ids) <-> (codes_s, order): the sorted pairs end up in (codes, ids)
        uint32_t* hist = (uint32_t*)(ws + L.sort_tmp);
        const int n_tiles = (P + RS_TILE - 1) / RS_TILE;
        uint32_t *k_in = codes, *v_in = ids, *k_out = codes_s, *v_out = order;
        for (int shift = 0; shift < 32; shift += 8) {
            hipLaunchKernelGGL(k_radix_hist, dim3(n_tiles), dim3(RS_THREADS), 0, stream, (const uint32_t*)k_in, P, shift, n_tiles, hist);
            hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, stream, hist, 256 * n_tiles);
            hipLaunchKernelGGL(k_radix_scatter, dim3(n_tiles), dim3(RS_THREADS), 0, stream, (const uint32_t*)k_in, (const uint32_t*)v_in, P,
                               shift, n_tiles, (const uint32_t*)hist, k_out, v_out);
            uint32_t* t = k_in; k_in = k_out; k_out = t;
            t = v_in; v_in = v_out; v_out = t;
        }
        order = v_in;   // == ids after an even number of passes
    }
    hipError_t e;
    hipLaunchKernelGGL(k_gather_box, dim3(n_boxes), dim3(BOX), 0, stream, points, P, (const uint32_t*)order, sp, boxes);
    hipLaunchKernelGGL(k_knn, dim3((P + QB - 1) / QB), dim3(QB), 0, stream, P, (const float4*)sp, (const BoxMM*)boxes, n_boxes,
                       mean_dists, nearest_idx);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_knn pipeline", e);
    return 0;
}

}  // extern "C"
