// raster_forward.hip -- forward pass of the MI355X-native Gaussian rasterizer (gfx950, wave64).
//
// Pipeline (one view):
//   k_preprocess   per Gaussian: project, cov3D/cov2D, conic, radius, tile rect, SH->RGB
//                  (forward.cu:155-256) + per-block LDS histogram of touched tiles
//   k_colscan      per tile: exclusive prefix of the block histograms -> per-(block,tile) bases
//   k_tilescan     exclusive scan over tiles -> ranges[tile] (rasterizer_impl.cu:116-138 result),
//                  num_rendered, per-block instance bases (the InclusiveSum of rasterizer_impl.cu:278)
//   k_scatter      per Gaussian: emit (depth bits<<32 | id) into its tile bucket (counting sort by
//                  tile, LDS cursors) + point_offsets           (duplicateWithKeys, rasterizer_impl.cu:70-111)
//   k_sort_tiles   one workgroup per tile: bitonic sort of the bucket in LDS by (depth bits, id)
//                  == the stable radix sort on (tile|depth) of rasterizer_impl.cu:304-309
//   k_render_fwd   one workgroup per 16x16 tile: LDS-staged, culled, compacted tile list;
//                  front-to-back alpha blend (forward.cu:261-381)
//
// Why not a global radix sort: the key's high word is the tile id, so a counting sort by tile
// followed by an LDS-resident per-tile sort moves ~20 B/instance through HBM instead of
// ~6 passes x 24 B/instance.  The sorted order is identical: (tile, depth bits, Gaussian id)
// is a total order and equals the stable-sort order of the reference's emission sequence.
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "raster_kernels.h"
#include "raster_layout.h"
#include "raster_math.h"

namespace gvd {

// experiments (tests/scripts/r5_fwd_trace.py): per-workgroup stamps of k_render_fwd.  Compiled only with -DGVD_RFWD_TRACE.
#ifdef GVD_RFWD_TRACE
__device__ unsigned long long g_ftrace[4096 * 8];
#define GVD_FT(i, v) do { if (blockIdx.x < 4096) g_ftrace[blockIdx.x * 8 + (i)] = (v); } while (0)
#else
#define GVD_FT(i, v) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// small wave/block helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// k_preprocess
// ------------------------------------------------------------------------------------------------
template <bool LDS_HIST>
__global__ void __launch_bounds__(256) k_preprocess(PreprocessArgs a)
{
    extern __shared__ uint32_t s_hist[];  // T counters (LDS_HIST)
    __shared__ uint32_t s_total;
    const int tid = threadIdx.x;
    if (LDS_HIST) {
        for (int t = tid; t < a.T; t += 256) s_hist[t] = 0;
    }
    if (tid == 0) s_total = 0;
    __syncthreads();

    const float3 campos = make_float3(a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]);
    uint32_t my_total = 0;
    const int base = blockIdx.x * a.items_per_block;
    for (int it = 0; it < a.items_per_block; it += 256) {
        const int idx = base + it + tid;
        if (idx >= a.P) break;
        int radius_out = 0;
        uint32_t touched = 0;
        do {
            const float3 p = make_float3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
            // in_frustum (auxiliary.h:139-164)
            const float4 ph = xform4x4(p, a.projmatrix);
            const float p_w = 1.0f / (ph.w + 0.0000001f);
            const float projx = ph.x * p_w, projy = ph.y * p_w;
            const float3 pv = xform4x3(p, a.viewmatrix);
            if (pv.z <= 0.2f) {
                if (a.prefiltered) {  // auxiliary.h:156-160
                    printf("Point is filtered although prefiltered is set. This shouldn't happen!");
                    __builtin_trap();
                }
                break;
            }
            float c3[6];
            if (a.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) c3[k] = a.cov3D_precomp[6 * idx + k];
            } else {
                const float3 sc = make_float3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
                const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
                cov3d_from_scale_rot(sc, a.scale_modifier, q, c3);
#pragma unroll
                for (int k = 0; k < 6; k++) a.cov3D[6 * idx + k] = c3[k];
            }
            const float3 cov = cov2d(pv, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, a.viewmatrix);
            const float det = fmaf(cov.x, cov.z, -(cov.y * cov.y));
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float conx = cov.z * det_inv, cony = -cov.y * det_inv, conz = cov.x * det_inv;
            const float mid = 0.5f * (cov.x + cov.z);
            const float disc = sqrtf(fmaxf(0.1f, fmaf(mid, mid, -det)));
            const float lambda1 = mid + disc;
            const float lambda2 = mid - disc;
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float pix_x = ndc2pix(projx, a.W), pix_y = ndc2pix(projy, a.H);
            const int ri = f2i_rz(my_radius);
            const int4 r = get_rect(pix_x, pix_y, ri, a.gx, a.gy);
            const int area = (r.z - r.x) * (r.w - r.y);
            if (area == 0) break;

            float3 rgb;
            uint32_t clamp_bits = 0;
            if (a.colors_precomp == nullptr) {
                rgb = sh_to_rgb(a.D, p, campos, a.shs + (size_t)idx * a.M * 3, &clamp_bits);
            } else {
                rgb = make_float3(a.colors_precomp[3 * idx], a.colors_precomp[3 * idx + 1], a.colors_precomp[3 * idx + 2]);
            }
            a.clamped[idx] = clamp_bits;
            a.depths[idx] = pv.z;
            reinterpret_cast<float2*>(a.means2D)[idx] = make_float2(pix_x, pix_y);
            reinterpret_cast<float4*>(a.conic_opacity)[idx] = make_float4(conx, cony, conz, a.opacities[idx]);
            reinterpret_cast<float4*>(a.rgbd)[idx] = make_float4(rgb.x, rgb.y, rgb.z, pv.z);
            radius_out = ri;
            touched = (uint32_t)area;
            // tile histogram
            for (int y = r.y; y < r.w; y++)
                for (int x = r.x; x < r.z; x++) {
                    if (LDS_HIST) atomicAdd(&s_hist[y * a.gx + x], 1u);
                    else atomicAdd(&a.hist[y * a.gx + x], 1u);
                }
        } while (0);
        a.radii[idx] = radius_out;
        a.tiles_touched[idx] = touched;
        my_total += touched;
    }
    my_total = wave_sum_u32(my_total);
    if (lane_id() == 0 && my_total) atomicAdd(&s_total, my_total);
    __syncthreads();
    if (LDS_HIST) {
        uint32_t* row = a.hist + (size_t)blockIdx.x * a.T;
        for (int t = tid; t < a.T; t += 256) row[t] = s_hist[t];
    }
    if (tid == 0) a.block_total[blockIdx.x] = s_total;
}

// ------------------------------------------------------------------------------------------------
// k_colscan: hist[b][t] -> exclusive prefix over b (in place); tile_count[t] = column total.
// Block = 1024 threads = kColSegs b-segments x kColTiles tiles; grid = ceil(T / kColTiles).  The kernel is a chain of
// dependent global round trips (sum pass, then rewrite pass), so the column is cut into many short segments: 64
// segments of ceil(B/64) rows need 2 + 2 trips of 8 loads in flight for B = 782 where 16 segments needed 7 + 7.
// ------------------------------------------------------------------------------------------------
constexpr int kColTiles = 16, kColSegs = 1024 / kColTiles;
__device__ __forceinline__ void colscan_body(uint32_t* __restrict__ hist, uint32_t* __restrict__ tile_count, int B, int T)
{
    __shared__ uint32_t s_seg[kColSegs][kColTiles];
    const int tl = threadIdx.x % kColTiles, seg = threadIdx.x / kColTiles;
    const int t = blockIdx.x * kColTiles + tl;
    const int per = (B + kColSegs - 1) / kColSegs;
    const int b0 = min(B, seg * per), b1 = min(B, b0 + per);
    constexpr int U = 8;  // loads in flight per lane (the loop is latency-, not bandwidth-bound)
    uint32_t sum = 0;
    if (t < T) {
        for (int b = b0; b < b1; b += U) {
            uint32_t v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = (b + u < b1) ? hist[(size_t)(b + u) * T + t] : 0u;
#pragma unroll
            for (int u = 0; u < U; u++) sum += v[u];
        }
    }
    s_seg[seg][tl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int s = 0; s < seg; s++) run += s_seg[s][tl];
    if (t < T) {
        for (int b = b0; b < b1; b += U) {
            uint32_t v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = (b + u < b1) ? hist[(size_t)(b + u) * T + t] : 0u;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (b + u < b1) hist[(size_t)(b + u) * T + t] = run;
                run += v[u];
            }
        }
        if (seg == kColSegs - 1) tile_count[t] = run;  // the last segment's running total is the column total
    }
}

__global__ void __launch_bounds__(1024) k_colscan(uint32_t* __restrict__ hist, uint32_t* __restrict__ tile_count, int B, int T)
{
    colscan_body(hist, tile_count, B, T);
}

// ------------------------------------------------------------------------------------------------
// k_tilescan: single block (1024 threads).  ranges, num_rendered, max list length, chunk_base.
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w /*NT / 64*/, uint32_t* total)
{
    const uint32_t incl = wave_incl_scan_u32(v);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const uint32_t x = s_w[i];
        if (i < w) wbase += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return wbase + incl - v;
}

// 256 buckets, descending in list length, ~6 % relative resolution (4-bit mantissa).
__device__ __forceinline__ uint32_t cost_bucket(uint32_t c)
{
    const uint32_t v = c + 1u;
    const int e = 31 - __clz((int)v);                     // 0..31
    const uint32_t m = e >= 4 ? ((v >> (e - 4)) & 15u) : ((v << (4 - e)) & 15u);
    const uint32_t k = min(255u, (uint32_t)e * 16u + m);  // ascending in c
    return 255u - k;
}

// Image region (0..7) of tile t: 2x2-tile blocks (32x32 pixels) dealt to the regions in a skewed round-robin, so that
// every region is a uniform sample of the image (similar total list length: the blend kernels last as long as their
// slowest XCD) while the ~5 tiles a Gaussian touches still fall into few regions.
__device__ __forceinline__ int xcd_region(int t, int gx)
{
    const int ty = t / gx, tx = t - ty * gx;
    return ((tx >> 1) + 3 * (ty >> 1)) & 7;
}

// ROLE bit 0: ranges, cursors, num_rendered / longest list / overflow (scalars, status, host mirror), per-block instance bases;
// ROLE bit 1: tile_order.  k_tilescan does both in one workgroup; inside the k_scatter launch two workgroups share the job, so that the
// single-workgroup chain of dependent steps (~24 us at 256 threads) is two shorter ones side by side -- and the host's ticket is out early.
template <int NT, int ROLE>
__device__ __forceinline__ void tilescan_body(const TileScanArgs& a)
{
    __shared__ uint32_t s_w[NT / 64];
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_bins[8][256];   // per XCD region: tiles per cost bucket, then running output ranks
    __shared__ uint32_t s_rcount[8];
    const int tid = threadIdx.x;
    if (tid == 0) s_max = 0;
    if (ROLE & 2) for (int i = tid; i < 8 * 256; i += NT) (&s_bins[0][0])[i] = 0;
    __syncthreads();
    // ---- tiles ----
    // Every global value this block needs is fetched once, up front and together (one round trip; the kernel is a single
    // workgroup, so each dependent trip to memory is ~1 us of GPU idle time); up to kKeep tile counts per thread stay in
    // registers, larger images re-read them (L2 hits).
    constexpr int kKeep = NT >= 1024 ? 4 : 8;
    const int per = (a.T + NT - 1) / NT;
    const int t0 = tid * per, t1 = min(a.T, t0 + per);
    const int perb = (a.B + NT - 1) / NT;
    const int b0 = tid * perb, b1 = min(a.B, b0 + perb);
    uint32_t kept[kKeep];
#pragma unroll
    for (int i = 0; i < kKeep; i++) kept[i] = (t0 + i < t1) ? a.tile_count[t0 + i] : 0u;
    uint32_t lb = 0, first_bt = 0;
    if (ROLE & 1) {
        for (int b = b0; b < b1; b++) {
            const uint32_t x = a.block_total[b];
            if (b == b0) first_bt = x;
            lb += x;
        }
    }
#define GVD_TILE_COUNT(T_, I_) ((I_) < kKeep ? kept[(I_) < kKeep ? (I_) : 0] : a.tile_count[T_])
    uint32_t local = 0, lmax = 0;
#pragma unroll 4
    for (int t = t0, i = 0; t < t1; t++, i++) {
        const uint32_t c = GVD_TILE_COUNT(t, i);
        local += c;
        lmax = max(lmax, c);
    }
    uint32_t total = 0;
    if (ROLE & 1) {
        uint32_t run = block_excl_scan<NT>(local, s_w, &total);
#pragma unroll 4
        for (int t = t0, i = 0; t < t1; t++, i++) {
            const uint32_t c = GVD_TILE_COUNT(t, i);
            // untouched tiles stay (0,0) like the reference's memset (rasterizer_impl.cu:311)
            a.ranges[2 * t] = c ? run : 0u;
            a.ranges[2 * t + 1] = c ? run + c : 0u;
            if (a.cursor) a.cursor[t] = run;
            run += c;
        }
        if (lmax) atomicMax(&s_max, lmax);
    }
    if (ROLE & 2) {
#pragma unroll 4
        for (int t = t0, i = 0; t < t1; t++, i++) atomicAdd(&s_bins[xcd_region(t, a.gx)][cost_bucket(GVD_TILE_COUNT(t, i))], 1u);
    }
    // ---- tile_order: the blend kernels' workgroup b works on tile_order[b].  Two goals:
    //   * longest lists first (LPT), so that the tail of the launch is short;
    //   * XCD locality: block b is observed to run on XCD b % 8 and every XCD has its own L2, so position b gets a
    //     tile of image region b % 8 (xcd_region): an XCD then pulls mostly its own regions' Gaussians (means2D /
    //     conic / colour records) through its L2 instead of every XCD pulling all of them.
    //   Within a region tiles are ranked by a 256-bucket counting sort on the list length; rank r of region x goes to
    //   position 8 r + x.  The ranks beyond the smallest region's size fill the end of the order. ----
    __syncthreads();
    if (ROLE & 2) {
        const int w = tid >> 6, lane = tid & 63;
        for (int rg = w; rg < 8; rg += NT / 64) {   // a wave per region: exclusive scan of the region's 256 buckets (4 per lane)
            uint32_t v[4], sum = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { v[i] = s_bins[rg][4 * lane + i]; sum += v[i]; }
            const uint32_t incl = wave_incl_scan_u32(sum);
            uint32_t run = incl - sum;
#pragma unroll
            for (int i = 0; i < 4; i++) { s_bins[rg][4 * lane + i] = run; run += v[i]; }
            if (lane == 63) s_rcount[rg] = incl;
        }
        __syncthreads();
        uint32_t cmin = 0xffffffffu;
#pragma unroll
        for (int x = 0; x < 8; x++) cmin = min(cmin, s_rcount[x]);
#pragma unroll 4
        for (int t = t0, i = 0; t < t1; t++, i++) {
            const int x = xcd_region(t, a.gx);
            const uint32_t r = atomicAdd(&s_bins[x][cost_bucket(GVD_TILE_COUNT(t, i))], 1u);
            uint32_t pos;
            if (r < cmin) {
                pos = 8u * r + (uint32_t)x;
            } else {
                pos = 8u * cmin + (r - cmin);
                for (int y = 0; y < x; y++) pos += s_rcount[y] - cmin;
            }
            a.tile_order[pos] = (uint32_t)t;
        }
    }
#undef GVD_TILE_COUNT
    if (!(ROLE & 1)) return;
    // ---- per-block instance bases (exclusive scan of block_total) ----
    uint32_t totb;
    uint32_t runb = block_excl_scan<NT>(lb, s_w, &totb);
    for (int b = b0; b < b1; b++) {
        a.chunk_base[b] = runb;
        runb += (b == b0) ? first_bt : a.block_total[b];
    }
    __syncthreads();
    if (tid == 0) {
        a.scalars[0] = total;  // num_rendered
        a.scalars[1] = s_max;  // longest tile list
        a.scalars[2] = (total > a.capacity) ? 1u : 0u;
        if (a.d_status) *a.d_status = (total > a.capacity) ? -4 : 0;
        if (a.host_mirror) {
            // the host spins on word 2 (capi.hip): data first, then the ticket, each a system-scope store to pinned memory
            a.host_mirror[0] = total; a.host_mirror[1] = s_max;
            __threadfence_system();
            a.host_mirror[2] = a.ticket;
        }
    }
}

__global__ void __launch_bounds__(1024) k_tilescan(TileScanArgs a) { tilescan_body<1024, 3>(a); }

// (A merged k_colscan + k_tilescan launch -- the last column-scan workgroup to arrive, by an agent-scope fence and a
// counter, runs the tile scan -- was measured at 61 us against 8.6 + 12.3 us for the two launches: on a multi-XCD part the
// release/acquire pair is a write-back / invalidate of the XCDs' private L2s.  A kernel boundary is the cheaper hand-off.)

// ------------------------------------------------------------------------------------------------
// k_scatter
// ------------------------------------------------------------------------------------------------
// FUSED (round 5; LDS_HIST only): the launch carries the tile scan.  Workgroups 0 and 1 run tilescan_body's two roles (ranges,
// num_rendered, the host mirror | tile_order) for the kernels that follow; every scatter workgroup derives what IT needs from the same two
// small arrays itself -- the exclusive scan of the T tile totals (its LDS cursors) and the sum of the block totals in front of it
// (its instance base) -- 6 KB of L2 reads and one extra barrier per workgroup instead of a single-workgroup k_tilescan launch
// (12 us of dependent round trips on one CU + a launch boundary) between k_colscan and k_scatter.
template <bool LDS_HIST, bool FUSED>
__global__ void __launch_bounds__(256) k_scatter(ScatterArgs a, TileScanArgs ts)
{
    extern __shared__ uint32_t s_cur[];  // T cursors (LDS_HIST)
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_bw[4];
    const int tid = threadIdx.x;
    if (FUSED && blockIdx.x < 2) {   // (wave-uniform) the tile scan's two roles, side by side
        if (blockIdx.x == 0) tilescan_body<256, 1>(ts);
        else tilescan_body<256, 2>(ts);
        return;
    }
    const int blk = FUSED ? (int)blockIdx.x - 2 : (int)blockIdx.x;
    uint32_t my_base = 0;
    if (LDS_HIST && FUSED) {
        const uint32_t* row = a.hist + (size_t)blk * a.T;
        const int per = (a.T + 255) / 256;
        const int t0 = tid * per, t1 = min(a.T, t0 + per);
        uint32_t local = 0;
        for (int t = t0; t < t1; t++) local += ts.tile_count[t];
        uint32_t lb = 0;
        for (int b = tid; b < blk; b += 256) lb += ts.block_total[b];
        const uint32_t incl = wave_incl_scan_u32(local);
        lb = wave_sum_u32(lb);
        const int w = tid >> 6;
        if (lane_id() == 63) s_w[w] = incl;
        if (lane_id() == 0) s_bw[w] = lb;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < w) wbase += s_w[i];
            my_base += s_bw[i];
        }
        uint32_t run = wbase + incl - local;
        for (int t = t0; t < t1; t++) {
            s_cur[t] = run + row[t];
            run += ts.tile_count[t];
        }
    } else if (LDS_HIST) {
        const uint32_t* row = a.hist + (size_t)blk * a.T;
        for (int t = tid; t < a.T; t += 256) s_cur[t] = a.ranges[2 * t] + row[t];
    }
    __syncthreads();
    const uint32_t first = FUSED ? my_base : a.chunk_base[blk];
    uint32_t carry = first;
    const int base = blk * a.items_per_block;
    for (int it = 0; it < a.items_per_block; it += 256) {
        if (base + it >= a.P) break;  // uniform
        const int idx = base + it + tid;
        const uint32_t touched = (idx < a.P) ? a.tiles_touched[idx] : 0u;
        // block inclusive scan (4 waves)
        const uint32_t incl = wave_incl_scan_u32(touched);
        const int w = tid >> 6;
        if (lane_id() == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t x = s_w[i];
            if (i < w) wbase += x;
            tot += x;
        }
        if (idx < a.P) a.point_offsets[idx] = carry + wbase + incl;
        carry += tot;
        if (touched) {
            const float2 xy = reinterpret_cast<const float2*>(a.means2D)[idx];
            const int4 r = get_rect(xy.x, xy.y, a.radii[idx], a.gx, a.gy);
            const uint64_t hi = ((uint64_t)__float_as_uint(a.depths[idx])) << 32;
            const uint64_t entry = hi | (uint32_t)idx;
            for (int y = r.y; y < r.w; y++)
                for (int x = r.x; x < r.z; x++) {
                    const int t = y * a.gx + x;
                    const uint32_t pos = LDS_HIST ? atomicAdd(&s_cur[t], 1u) : atomicAdd(&a.cursor[t], 1u);
                    if (pos < a.capacity) a.bucket[pos] = entry;
                }
        }
        __syncthreads();
    }
    // The block's instances are the contiguous slots [chunk_base, carry) of the backward's partial-record arrays (Gaussian
    // order).  k_render_bwd's quadrant waves write the sub-records they reach and flag them; only the 4-byte flag words are
    // cleared here (coalesced, under this kernel's atomic latency) -- no memset launch at the head of every backward, and since
    // round 5 no 48-byte-per-instance zeroing either.
    if (a.pflags) {
        const uint32_t beg = min(first, a.capacity), end = min(carry, a.capacity);
        for (uint32_t i = beg + tid; i < end; i += 256) a.pflags[i] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// k_sort_tiles: ascending sort of bucket[range) as u64 = (depth bits << 32 | gaussian id).
// Sorting network: bitonic "flip + disperse" with every comparator ascending, so a list of any
// length n is sorted by treating indices >= n as +inf (comparators touching them are no-ops).
// CLASS 0: n <= 2048, LDS, 256 threads.  CLASS 1: n <= 16384, LDS (128 KiB), 1024 threads.
// CLASS 2: n > 16384, in global memory, 1024 threads (rare; correctness path).
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void bitonic_sort_u64(uint64_t* d, uint32_t n)
{
    uint32_t npad = 1;
    while (npad < n) npad <<= 1;
    const uint32_t half = npad >> 1;
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        // flip: i-th comparator of each k-block pairs l with k-1-l
        {
            const uint32_t hk = k >> 1;
            for (uint32_t i = threadIdx.x; i < half; i += NT) {
                const uint32_t blk = i / hk, l = i - blk * hk;
                const uint32_t lo = blk * k + l, hi = blk * k + (k - 1 - l);
                if (hi < n) {
                    const uint64_t x = d[lo], y = d[hi];
                    if (x > y) { d[lo] = y; d[hi] = x; }
                }
            }
            __syncthreads();
        }
        for (uint32_t j = k >> 2; j >= 1; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < half; i += NT) {
                const uint32_t lo = 2 * j * (i / j) + (i % j), hi = lo + j;
                if (hi < n) {
                    const uint64_t x = d[lo], y = d[hi];
                    if (x > y) { d[lo] = y; d[hi] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// Merge sort by ranking, for lists that fit two LDS buffers (<= kFusedSortMax): every thread first sorts 8 consecutive
// entries in registers (19-comparator network), then log2(n/8) passes merge neighbouring runs of width w: an entry's
// output slot is its index in its own run plus the number of smaller entries in the partner run, found by a binary
// search (keys are unique, so no tie rule is needed).  A pass is one barrier and <= log2(w)+1 dependent LDS reads per
// entry, against log2(w)+1 barrier-separated compare-exchange stages of the bitonic network: 8 barriers instead of 66
// for 2048 entries, and the chain of dependent LDS round trips is ~5x shorter -- this chain, on the longest list, is
// what the sort costs (most workgroups have long finished).  Returns the buffer holding the sorted list; ends with a
// barrier.  Entries at index >= n do not exist (runs are simply shorter at the end).
template <int NT>
__device__ __forceinline__ uint64_t* merge_sort_u64(uint64_t* buf0, uint64_t* buf1, uint32_t n)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t base = 8 * tid; base < n; base += 8 * NT) {
        uint64_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (base + i < n) ? buf0[base + i] : ~0ull;
#define GVD_CE(A, B) { const uint64_t x_ = v[A], y_ = v[B]; v[A] = x_ < y_ ? x_ : y_; v[B] = x_ < y_ ? y_ : x_; }
        GVD_CE(0, 1) GVD_CE(2, 3) GVD_CE(4, 5) GVD_CE(6, 7)
        GVD_CE(0, 2) GVD_CE(1, 3) GVD_CE(4, 6) GVD_CE(5, 7)
        GVD_CE(1, 2) GVD_CE(5, 6) GVD_CE(0, 4) GVD_CE(3, 7)
        GVD_CE(1, 5) GVD_CE(2, 6)
        GVD_CE(1, 4) GVD_CE(3, 6)
        GVD_CE(2, 4) GVD_CE(3, 5)
        GVD_CE(3, 4)
#undef GVD_CE
#pragma unroll
        for (int i = 0; i < 8; i++) if (base + i < n) buf0[base + i] = v[i];
    }
    __syncthreads();
    uint64_t* src = buf0;
    uint64_t* dst = buf1;
    for (uint32_t w = 8; w < n; w <<= 1) {
        // (Advancing a thread's searches in lockstep -- K independent LDS reads per halving step -- was measured slower
        //  than these early-exit loops: 31 us against 23 us for the separate sort kernel on the C2 scene.)
        for (uint32_t i = tid; i < n; i += NT) {
            const uint32_t run = i / w, p = i - run * w;
            const uint32_t other = (run ^ 1u) * w;                       // first entry of the partner run
            const uint32_t len = other < n ? min(w, n - other) : 0u;      // its length (0: no partner, the run is copied)
            const uint64_t key = src[i];
            uint32_t lo = 0, hi = len;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (src[other + mid] < key) lo = mid + 1; else hi = mid;
            }
            dst[(run & ~1u) * w + p + lo] = key;
        }
        __syncthreads();
        uint64_t* t = src; src = dst; dst = t;
    }
    return src;
}

template <int CLASS>
__global__ void __launch_bounds__(CLASS == 0 ? 256 : 1024) k_sort_tiles(SortArgs a)
{
    constexpr int NT = (CLASS == 0) ? 256 : 1024;
    extern __shared__ uint64_t s_keys[];
    const uint32_t tile = blockIdx.x;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    if (r1 > a.capacity || r1 < r0) return;  // overflowed forward: leave untouched
    const uint32_t n = r1 - r0;
    if (CLASS == 0) { if (n > 2048) return; }
    if (CLASS == 1) { if (n <= 2048 || n > 16384) return; }
    if (CLASS == 2) { if (n <= 16384) return; }
    if (n == 0) return;
    uint64_t* g = a.bucket + r0;
    uint64_t* d;
    if (CLASS == 2) {
        d = g;
    } else {
        d = s_keys;
        for (uint32_t i = threadIdx.x; i < n; i += NT) d[i] = g[i];
        __syncthreads();
    }
    if (CLASS == 0) d = merge_sort_u64<NT>(d, d + kFusedSortMax, n);
    else bitonic_sort_u64<NT>(d, n);
    const uint64_t thi = ((uint64_t)tile) << 32;
    for (uint32_t i = threadIdx.x; i < n; i += NT) {
        const uint64_t e = d[i];
        if (CLASS != 2) g[i] = e;
        a.point_list[r0 + i] = (uint32_t)e;
        a.keys[r0 + i] = thi | (e >> 32);
    }
}

// ------------------------------------------------------------------------------------------------
// k_render_fwd: one workgroup (4 waves) per 16x16 tile; wave w owns the 8x8 pixel quadrant (w & 1, w >> 1).
// The workgroup first sorts the tile's list together (merge_sort_u64; lists over kFusedSortMax arrive sorted) -- the only
// cooperative phase.  After it THE FOUR WAVES RUN INDEPENDENTLY (round 5; the round-4 form staged 256 entries per batch together
// and met at two barriers per batch, so every batch lasted as long as its slowest quadrant -- the k_render_bwd trace of round 5
// priced that wait at ~a quarter of a workgroup's life): per trip of 64 list entries each lane fetches one entry
// (id -> xy, conic/opacity, rgb+depth; the next trip's gathers are in flight during the blend), runs the conservative test
// against the wave's OWN quadrant, and the survivors are compacted into the wave's private LDS list with one ballot + prefix
// (order preserved, original list position kept for n_contrib).  Every pixel then walks that list from LDS (broadcast reads)
// with the reference's exact per-pixel sequence (forward.cu:329-368).  A wave stops as soon as ITS 64 pixels are finished.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_render_fwd(RenderArgs a)
{
    __shared__ float2 s_xy4[4][64];   // one compacted list per quadrant (= per wave)
    __shared__ float4 s_co4[4][64];
    __shared__ float4 s_cd4[4][64];
    __shared__ uint32_t s_pos4[4][64];

    const int tile = (int)a.tile_order[blockIdx.x];
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int qx = tx * 16 + (w & 1) * 8, qy = ty * 16 + (w >> 1) * 8;   // wave w owns the 8x8 quadrant (w & 1, w >> 1)
    const int px = qx + (lane & 7);
    const int py = qy + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const float qx0 = (float)qx, qy0 = (float)qy;

    const uint32_t r0 = a.ranges[2 * tile];
    uint32_t r1 = a.ranges[2 * tile + 1];
    if (r1 > a.capacity) r1 = r0;  // overflowed forward: render background, status already flagged
    if (r1 - r0 > a.sorted_limit) r1 = r0;   // mis-guessed sort class (speculative forward): this list was never sorted, its ids are garbage
    const uint32_t n = r1 - r0;
#ifdef GVD_RFWD_TRACE
    const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { GVD_FT(0, __builtin_amdgcn_s_memrealtime()); GVD_FT(3, (unsigned long long)n); }
#endif

    // ---- fused per-tile sort (the k_sort_tiles<0> work, done by the workgroup that consumes the list) ----
    // A separate sort launch lasts as long as its longest list (a chain of barrier-separated LDS stages) while most
    // of the GPU idles; here that chain overlaps with the other tiles' blending.
    extern __shared__ uint64_t s_sort_lds[];   // 2 x kFusedSortMax entries (+ the occupancy pad)
    uint64_t* s_sorted = s_sort_lds;
    const bool sorted_here = a.fused_sort && n <= kFusedSortMax;
    if (sorted_here && n) {
        uint64_t* g = a.bucket + r0;
        for (uint32_t i = tid; i < n; i += 256) s_sorted[i] = g[i];
        __syncthreads();
        s_sorted = merge_sort_u64<256>(s_sorted, s_sorted + kFusedSortMax, n);
        const uint64_t thi = ((uint64_t)(uint32_t)tile) << 32;
        for (uint32_t i = tid; i < n; i += 256) {
            const uint64_t e = s_sorted[i];
            g[i] = e;
            a.point_list[r0 + i] = (uint32_t)e;
            a.keys[r0 + i] = thi | (e >> 32);
        }
    }
    // ---- from here on no workgroup barrier: each wave blends its quadrant on its own ----
#ifdef GVD_RFWD_TRACE
    const unsigned long long tf1 = __builtin_amdgcn_s_memtime();
#endif

    float T = inside ? 1.0f : 0.0f, T_keep = 1.0f;   // live transmittance (0 = pixel finished) / value kept for the background
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, weight = 0.f, D = 0.f;
    uint32_t last_contributor = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    float2* const s_xy = s_xy4[w];
    float4* const s_co = s_co4[w];
    float4* const s_cd = s_cd4[w];
    uint32_t* const s_pos = s_pos4[w];

    // The records of trip b + 64 are fetched while trip b is blended (the gather id -> means2D / conic / colour is
    // two dependent trips to L2 that would otherwise sit between two blend loops).
    float2 nxy = make_float2(0.f, 0.f);
    float4 nco = make_float4(0.f, 0.f, 0.f, 0.f), ncd = make_float4(0.f, 0.f, 0.f, 0.f);
#define GVD_FETCH(E)                                                                              \
        if ((E) < n) {                                                                            \
            const uint32_t id = sorted_here ? (uint32_t)s_sorted[E] : a.point_list[r0 + (E)];     \
            nxy = reinterpret_cast<const float2*>(a.means2D)[id];                                 \
            nco = reinterpret_cast<const float4*>(a.conic_opacity)[id];                           \
            ncd = reinterpret_cast<const float4*>(a.rgbd)[id];                                    \
        }
    GVD_FETCH((uint32_t)lane)
    for (uint32_t b = 0; b < n; b += 64) {
        if (__builtin_amdgcn_ballot_w64(T != 0.0f) == 0ull) break;  // wave-uniform: this quadrant is finished
        // ---- stage + cull against this quadrant + compact ----
        const uint32_t e = b + (uint32_t)lane;
        const float2 xy = nxy;
        const float4 co = nco, cd = ncd;
        const bool keep = e < n && rect_may_contribute(xy.x, xy.y, co.x, co.y, co.z, co.w, qx0, qy0, 7.0f, 7.0f);
        GVD_FETCH(e + 64u)
        // the cull bit travels to the backward (one byte per (entry, quadrant), 64 contiguous bytes per trip): its quadrant wave
        // then gathers the records of the kept entries only, instead of every entry's to repeat this test
        if (e < n) a.qmask[(size_t)w * a.capacity + r0 + e] = keep ? 1 : 0;
        const unsigned long long m = __ballot(keep);
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (keep) {
            const uint32_t slot = (uint32_t)__popcll(m & below);
            s_xy[slot] = xy; s_co[slot] = co; s_cd[slot] = cd;
            s_pos[slot] = e + 1;  // value of `contributor` when this entry is visited
        }
        // ---- blend ----
        // Branch-free, 4 entries per trip: the per-entry geometry (power, exp, alpha) of the 4 entries
        // is independent work the scheduler can overlap with the LDS latency, and only the short
        // T-recurrence is serial.  Per-pixel semantics are exactly forward.cu:329-368: an entry
        // contributes iff power<=0, alpha>=1/255 and the pixel is not done; the first entry with
        // T*(1-alpha) < 1e-4 marks the pixel done and is not blended.
// `done` is folded into the running transmittance: T is the LIVE transmittance (0 once the pixel has stopped, so
// every later product is exactly 0) and T_keep holds the value the reference keeps for the background term.  An
// entry the reference skips (power > 0 or alpha < 1/255) runs with alpha = 0: test_T == T exactly and every
// accumulator gets +0.  The per-entry critical path is then mul -> v_cmp -> v_cndmask on the VALU only (the old form
// went VALU -> SALU mask logic -> VALU through `done` on every entry).
#define GVD_BLEND_ONE(XY, CO, CD, POS)                                                            \
        {                                                                                         \
            const float dx = (XY).x - pixfx, dy = (XY).y - pixfy;                                 \
            const float power = gauss_power((CO).x, (CO).y, (CO).z, dx, dy);                      \
            const float alpha_raw = fminf(0.99f, (CO).w * __expf(power));                         \
            const bool hit = !(power > 0.0f) && !(alpha_raw < 1.0f / 255.0f);                     \
            const float alpha = hit ? alpha_raw : 0.0f;                                           \
            const float test_T = T * (1.f - alpha);                                               \
            const bool go = !(test_T < 0.0001f);        /* false when stopping now or already stopped (T == 0) */ \
            const float Tc = go ? T : 0.0f;                                                       \
            const float wgt = alpha * Tc;               /* blend weight formed once: c * (alpha * T), not (c * alpha) * T */ \
            C0 = fmaf((CD).x, wgt, C0);                                                           \
            C1 = fmaf((CD).y, wgt, C1);                                                           \
            C2 = fmaf((CD).z, wgt, C2);                                                           \
            weight = fmaf(alpha, Tc, weight);                                                     \
            D = fmaf((CD).w, wgt, D);                                                             \
            T_keep = go ? test_T : T_keep;                                                        \
            T = go ? test_T : 0.0f;                                                               \
            last_contributor = (go && hit) ? (POS) : last_contributor;                            \
        }
        uint32_t j = 0;
        for (; j + 4 <= cnt; j += 4) {
            if (__builtin_amdgcn_ballot_w64(T != 0.0f) == 0ull) break;  // wave-uniform: this quadrant is finished
            const float2 xy0 = s_xy[j], xy1 = s_xy[j + 1], xy2 = s_xy[j + 2], xy3 = s_xy[j + 3];
            const float4 co0 = s_co[j], co1 = s_co[j + 1], co2 = s_co[j + 2], co3 = s_co[j + 3];
            const float4 cd0 = s_cd[j], cd1 = s_cd[j + 1], cd2 = s_cd[j + 2], cd3 = s_cd[j + 3];
            const uint32_t p0 = s_pos[j], p1 = s_pos[j + 1], p2 = s_pos[j + 2], p3 = s_pos[j + 3];
            GVD_BLEND_ONE(xy0, co0, cd0, p0)
            GVD_BLEND_ONE(xy1, co1, cd1, p1)
            GVD_BLEND_ONE(xy2, co2, cd2, p2)
            GVD_BLEND_ONE(xy3, co3, cd3, p3)
        }
        for (; j < cnt; j++) {
            const float2 xy0 = s_xy[j];
            const float4 co0 = s_co[j];
            const float4 cd0 = s_cd[j];
            const uint32_t p0 = s_pos[j];
            GVD_BLEND_ONE(xy0, co0, cd0, p0)
        }
#undef GVD_BLEND_ONE
    }
#undef GVD_FETCH
    if (inside) {
        const size_t pid = (size_t)py * a.W + px;
        const size_t HW = (size_t)a.H * a.W;
        a.n_contrib[pid] = last_contributor;
        a.out_color[pid] = fmaf(T_keep, a.bg[0], C0);
        a.out_color[HW + pid] = fmaf(T_keep, a.bg[1], C1);
        a.out_color[2 * HW + pid] = fmaf(T_keep, a.bg[2], C2);
        a.out_alpha[pid] = weight;
        a.out_depth[pid] = D;
    }
#ifdef GVD_RFWD_TRACE
    if (lane == 0) {
        GVD_FT(4 + w, __builtin_amdgcn_s_memtime() - tf1);                 // this wave's blend phase
        if (w == 0) { GVD_FT(2, tf1 - tf0); }                               // the sort phase
        unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicMax(&g_ftrace[blockIdx.x * 8 + 1], t1);                       // last wave out
    }
#endif
}

// rasterizer_impl.cu:54-66 (checkFrustum)
__global__ void __launch_bounds__(256) k_mark_visible(int P, const float* __restrict__ means3D,
                                                      const float* __restrict__ viewmatrix, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    const float3 pv = xform4x3(p, viewmatrix);
    present[idx] = pv.z > 0.2f;
}

// ------------------------------------------------------------------------------------------------
// launchers (called from capi.hip)
// ------------------------------------------------------------------------------------------------
void launch_preprocess(const PreprocessArgs& a, int blocks, bool lds_hist, hipStream_t s)
{
    if (lds_hist) hipLaunchKernelGGL(k_preprocess<true>, dim3(blocks), dim3(256), (size_t)a.T * 4, s, a);
    else hipLaunchKernelGGL(k_preprocess<false>, dim3(blocks), dim3(256), 0, s, a);
}
void launch_colscan(uint32_t* hist, uint32_t* tile_count, int B, int T, hipStream_t s)
{
    hipLaunchKernelGGL(k_colscan, dim3((T + kColTiles - 1) / kColTiles), dim3(1024), 0, s, hist, tile_count, B, T);
}
void launch_tilescan(const TileScanArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_tilescan, dim3(1), dim3(1024), 0, s, a);
}
void launch_scatter(const ScatterArgs& a, const TileScanArgs* fused_scan, int blocks, bool lds_hist, hipStream_t s)
{
    TileScanArgs ts{};
    if (fused_scan) ts = *fused_scan;
    if (lds_hist && fused_scan) hipLaunchKernelGGL((k_scatter<true, true>), dim3(blocks + 2), dim3(256), (size_t)a.T * 4, s, a, ts);
    else if (lds_hist) hipLaunchKernelGGL((k_scatter<true, false>), dim3(blocks), dim3(256), (size_t)a.T * 4, s, a, ts);
    else hipLaunchKernelGGL((k_scatter<false, false>), dim3(blocks), dim3(256), 0, s, a, ts);
}
void launch_sort_tiles(const SortArgs& a, int T, int max_class, bool short_lists_too, hipStream_t s)
{
    if (max_class >= 1) {   // the 128 KiB dynamic-LDS attribute is per DEVICE: set it once on each device that launches it
        static std::atomic<unsigned long long> attr_mask{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        const unsigned long long bit = 1ull << (dev & 63);
        if (dev >= 64 || !(attr_mask.load(std::memory_order_relaxed) & bit)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_tiles<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            attr_mask.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    // lists of <= kFusedSortMax entries are normally sorted inside k_render_fwd
    if (short_lists_too) hipLaunchKernelGGL(k_sort_tiles<0>, dim3(T), dim3(256), (size_t)kFusedSortMax * 16, s, a);
    if (max_class >= 1) hipLaunchKernelGGL(k_sort_tiles<1>, dim3(T), dim3(1024), 16384 * 8, s, a);
    if (max_class >= 2) hipLaunchKernelGGL(k_sort_tiles<2>, dim3(T), dim3(1024), 0, s, a);
}
static size_t env_bytes(const char* name, size_t dflt)
{
    const char* e = getenv(name);
    return e ? (size_t)atol(e) : dflt;
}
void launch_render_fwd(const RenderArgs& a, int T, hipStream_t s)
{
    // Extra (unused) dynamic LDS caps the resident workgroups per CU so that the hardware dispatcher
    // hands out the LPT-ordered tiles dynamically instead of placing every tile at t=0.
    static const size_t pad = env_bytes("GVD_FWD_LDS_PAD", 0);
    hipLaunchKernelGGL(k_render_fwd, dim3(T), dim3(256), (a.fused_sort ? (size_t)kFusedSortMax * 16 : 0) + pad, s, a);
}
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s)
{
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}

}  // namespace gvd

#ifdef GVD_RFWD_TRACE
extern "C" int gvd_debug_ftrace_read(unsigned long long* dst, size_t n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gvd::g_ftrace), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
extern "C" int gvd_debug_ftrace_clear(void)
{
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(gvd::g_ftrace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 4096 * 8);
}
#endif
