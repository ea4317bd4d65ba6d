// raster_backward.hip -- backward pass of the MI355X-native Gaussian rasterizer (gfx950, wave64).
//
//   k_render_bwd   one WAVE per (16x16 tile, 8x8 quadrant), back-to-front replay (backward.cu:415-601).
//                  The reference issues 10 global float atomics per (pixel, Gaussian) hit; on
//                  MI355X same-address device-scope atomics serialise at ~11 ns each, so instead
//                  every wave reduces its 64 pixels in registers and writes ONE 48-byte partial
//                  sub-record per (Gaussian, tile, quadrant) it walked, at the instance's slot in
//                  Gaussian order (slot = point_offsets[id-1] + row-major index of the tile inside
//                  the rect; sub-slot = quadrant), flagged in a byte of pflags[instance].
//   k_gather_bwd   per Gaussian: sums its contiguous run of flagged sub-records in a fixed order
//                  (=> deterministic gradients, no atomics), then the reference's
//                  computeCov2DCUDA (backward.cu:144-274), preprocessCUDA-bwd (:346-412),
//                  SH bwd (:20-139) and cov3D bwd (:278-341) in one pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "raster_kernels.h"
#include "raster_layout.h"
#include "raster_math.h"

namespace gvd {

constexpr int kNV = 10;  // reduced values per (Gaussian, tile)

#ifndef GVD_BWD_MERGE
#define GVD_BWD_MERGE 0   // 1: two list entries in one serial step when no pixel is active in both.  Built, bit-identical, and SLOWER: k_render_bwd 103.3 -> 110.9 us
                          // (profiles/r06_bwd_merge_ab.txt: the wave-uniform test and the per-lane selects cost more than the 9.5 % of steps they remove)
#endif

// experiments (tests/scripts/r5_bwd_trace.py): per-workgroup stamps of k_render_bwd -- s_memrealtime at entry / exit, XCC + HW ids,
// the tile's walk length, and each wave's s_memtime cycles inside the walk.  Compiled only with -DGVD_RBWD_TRACE.
#ifdef GVD_RBWD_TRACE
__device__ unsigned long long g_rtrace[32768 * 8];
#define GVD_RT(i, v) do { if (blockIdx.x < 32768) g_rtrace[blockIdx.x * 8 + (i)] = (v); } while (0)
#else
#define GVD_RT(i, v) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// k_render_bwd: ONE WAVE PER (TILE, 8x8 QUADRANT), no workgroup barriers (round 5).
//
// The round-4 form ran a 256-thread workgroup per tile: cooperative staging of 256 list entries, four quadrant waves walking their
// compacted lists, and the four waves' sums meeting in LDS -- three barriers per batch.  A per-workgroup trace of that kernel
// (tests/scripts/r5_bwd_trace.py, profiles/r05_bwd_trace_*.txt) showed where its time went: wave 0 spends 63 % of the workgroup's
// life in the walk, 9 % in staging and 23 % parked at the batch-end barrier waiting for the slowest quadrant (the quadrant lists of
// a batch differ by ~25 %, and the wait repeats every batch), and the launch lasts as long as its longest tile (1100 entries,
// ~0.17 us each under a three-way shared SIMD).  Here a wave owns its quadrant from the first list entry to the last:
//   * it stages 64 entries per trip itself (lane = entry: the forward's cull bit for this quadrant, then id -> records for the kept
//     entries, ballot compaction into its private LDS strip; the next trip's gathers are in flight during the walk), and only the
//     list prefix in front of ITS last contributor (max n_contrib of its 64 pixels), not the tile's;
//   * its per-entry sums go to a private sub-record: partials[(4 * instance + quadrant)] (48 bytes), flagged in pflags[instance]
//     byte `quadrant`; k_gather_bwd adds the flagged sub-records in the fixed order instance-major, quadrant-minor, so the
//     gradients stay bitwise deterministic without any cross-wave meeting point.  The forward's k_scatter zeroes the 4-byte flag
//     words instead of the 48-byte records.
// 4 T independent units instead of T workgroups: the dispatcher balances quadrants, nothing waits for a neighbour, and the LDS
// footprint (4 KiB per wave + 8 KiB for the turn-around strip) no longer limits occupancy.
//
// The per-Gaussian sums are formed by an LDS TURN-AROUND, not by a cross-lane reduction per entry.  The walk keeps only what is serial
// per pixel and leaves, per (entry, pixel),
//     q = G * dL_dalpha   (backward.cu:577-598: every conic / mean / opacity term is q times a polynomial in (dx, dy))
//     w = alpha * T       (backward.cu:520-537: dL_dcolor[c] = w * dL_dpixel[c])
// in a row of the wave's LDS strip; every kRows entries the wave turns around -- lane = (entry e, pixel group g) -- and each lane
// accumulates the moments sum q, q dx, q dy, q dx^2, q dx dy, q dy^2 and sum w dL_dpix[c] over its group's pixels in its own
// registers (13 VALU per pixel for kRows entries at once instead of 17 term + 25 DPP-reduction instructions per entry: the DPP form,
// round 2's wave_reduce20, measured 141 us in this kernel against 113 us; 8 rows 134 us), the groups meet with log2(64 / kRows)
// shuffles, and the lanes of group 0 write the entry's sub-record and flag straight to global memory.
// DA: the caller supplied a gradient for the depth and / or the alpha image (else those recurrences are compiled out).
// ------------------------------------------------------------------------------------------------
template <bool DA, int kRows>
__global__ void __launch_bounds__(64) k_render_bwd(RenderBwdArgs a)
{
    static_assert(kRows == 8 || kRows == 16, "rows per turn-around");
    constexpr int kStride = 65;               // float2 per strip row: 64 pixels + the row's slot id; 130 words = 2 (mod 32)
    __shared__ float4 s_rec[2 * 64];          // per staged entry {x, y, list position (bits), record slot (bits), conic a b c, opacity}
    __shared__ float4 s_cd[64];
    __shared__ float2 s_qw[kRows][kStride];
    __shared__ float4 s_dl[64];               // the wave's pixels' (dL_dpix rgb, dL_ddepth)

        // unit -> (tile, quadrant, segment), such that unit u and tile_order position p agree modulo 8 -- the XCD a workgroup lands on and
    // the image region k_tilescan dealt to that position (its L2).
    // The grid is [extra units | main units].  Main unit m: segment 0 of (position (m >> 5) * 8 + (m & 7), quadrant (m >> 3) & 3).
    // Extra unit e (the first a.extra_units of the grid, so that the extra segments of the LONG lists -- tile_order puts those first --
    // start with the launch): groups of 96 = 8 positions x 4 quadrants x segments 1..3.  Only the first a.split_positions positions of
    // tile_order can be cut; both counts are multiples of 8, so every unit u still agrees with its position modulo 8.
    const uint32_t unit = blockIdx.x;
    uint32_t pos, seg;
    int quad;
    if (unit < a.extra_units) {
        const uint32_t grp = unit / 96u, in = unit - grp * 96u;
        pos = grp * 8u + (in & 7u);
        quad = (int)((in >> 3) & 3u);
        seg = 1u + (in >> 5);
    } else {
        const uint32_t m = unit - a.extra_units;
        pos = (m >> 5) * 8u + (m & 7u);
        quad = (int)((m >> 3) & 3u);
        seg = 0u;
    }
    if (pos >= (uint32_t)(a.gx * a.gy)) return;
    const int tile = (int)a.tile_order[pos];
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int qx = tx * 16 + (quad & 1) * 8, qy = ty * 16 + (quad >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const float qx0 = (float)qx, qy0 = (float)qy;
    const size_t pid = (size_t)py * a.W + px;
    const size_t HW = (size_t)a.H * a.W;

#ifdef GVD_RBWD_TRACE
    unsigned long long walk_cycles = 0, stage_cycles = 0, tail_cycles = 0;
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        GVD_RT(0, __builtin_amdgcn_s_memrealtime());
        GVD_RT(2, ((unsigned long long)xcc << 32) | hwid);
        GVD_RT(1, 0ull); GVD_RT(3, 0ull); GVD_RT(4, 0ull); GVD_RT(5, 0ull); GVD_RT(6, 0ull); GVD_RT(7, 0ull);
    }
#endif
    // A capped forward that overflowed left truncated lists and point_offsets that index past the partial buffer:
    // do nothing (k_gather_bwd then writes zero gradients); the overflow itself is reported through d_status.
    if (a.scalars[2]) return;
    const uint32_t r0 = a.ranges[2 * tile];
    uint32_t r1 = a.ranges[2 * tile + 1];
    if (r1 > a.capacity) r1 = r0;
    if (seg && r1 - r0 <= a.split_len) return;   // a short list has one segment: the others leave before touching a pixel

    const float T_final = inside ? (1.f - a.alphas[pid]) : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? a.n_contrib[pid] : 0u;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f, dLa = 0.f;
    if (inside) {
        dLp0 = a.dL_dpix[pid];
        dLp1 = a.dL_dpix[HW + pid];
        dLp2 = a.dL_dpix[2 * HW + pid];
        if (DA && a.dL_dpix_depth) dLd = a.dL_dpix_depth[pid];  // NULL == all-zero gradient
        if (DA && a.dL_dalphas) dLa = a.dL_dalphas[pid];
    }
    s_dl[lane] = make_float4(dLp0, dLp1, dLp2, dLd);
    float bg_dot = 0.f;  // backward.cu:575-577 accumulation order
    bg_dot += a.bg[0] * dLp0;
    bg_dot += a.bg[1] * dLp1;
    bg_dot += a.bg[2] * dLp2;
    uint32_t n_walk;     // entries at or behind this list position contribute to no pixel of the quadrant
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        n_walk = min((uint32_t)__builtin_amdgcn_readfirstlane((int)m), r1 - r0);
    }
#ifdef GVD_RBWD_TRACE
    if (lane == 0) { GVD_RT(3, (unsigned long long)n_walk | ((unsigned long long)(r1 - r0) << 32)); GVD_RT(4, __builtin_amdgcn_s_memtime() - tk0); }
    if (n_walk == 0 && lane == 0) GVD_RT(1, __builtin_amdgcn_s_memrealtime());
#endif
    if (n_walk == 0) return;
    // LONG WALKS ARE CUT INTO SEGMENTS, ONE UNIT EACH.  The launch lasts at least as long as its longest unit -- a serial per-pixel chain
    // of ~0.1 us per list entry, 1100 entries on the C2 view against a mean of 360 (profiles/r05_bwd_trace_*) -- and the machine drains
    // behind it.  A walk of more than split_len entries is cut into nseg = min(4, ceil(n_walk / split_len)) parts at multiples of 64.
    // The unit of part s first REPLAYS everything behind its part, [end_s, n_walk), in a light pass that keeps only the per-pixel
    // recurrences (T, accum_rec, last_*: the same instructions in the same order, so the state it arrives with is bit-identical to
    // the sequential walk's -- no checkpoint from the forward, no change in any result), then walks its own part in full.  A light
    // step is ~32 of the ~74 VALU instructions of a full one.
    constexpr uint32_t kMaxSeg = 4;
    uint32_t nseg = 1;
    if (pos < a.split_positions && n_walk > a.split_len) nseg = min(kMaxSeg, (n_walk + a.split_len - 1u) / a.split_len);
    if (seg >= nseg) return;
    const uint32_t per = (((n_walk + nseg - 1u) / nseg) + 63u) & ~63u;        // wave-uniform; parts are multiples of 64 entries
    const uint32_t part_lo = min(seg * per, n_walk), part_hi = min(part_lo + per, n_walk);
    if (part_lo >= part_hi) return;

    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc_d = 0.f, acc_a = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;
    const float nddelx_dx = -0.5f * a.W, nddely_dy = -0.5f * a.H;   // -(d delta / d mean2D): backward.cu:490-491
    const float nTf = -T_final;
    const bool has_bg = (a.bg[0] != 0.f) || (a.bg[1] != 0.f) || (a.bg[2] != 0.f);  // wave-uniform (kernel argument)
    const unsigned long long below = (1ull << lane) - 1ull;

    // turn-around roles: entry e of the strip, pixel group g (pixels g * kRows .. + kRows - 1 of the quadrant, lane order)
    const int te = lane & (kRows - 1), tg = lane / kRows;
    const float gy0 = qy0 + (float)(tg * (kRows / 8));   // first pixel row of this lane's group (kRows / 8 rows per group)

    // ---- the records of trip b + 64 are fetched while trip b is walked: the list position's id and the forward's cull bit for this
    // quadrant (both unit-stride), then the record gathers -- and the two gathers of the instance's partial-record slot -- for the kept
    // entries only ----
    const uint8_t* const my_mask = a.qmask + (size_t)quad * a.capacity + r0;
    uint32_t n_id = 0, n_ord = 0, n_poff = 0;
    int n_rad = 0;
    bool n_keep = false;
    float2 n_xy = make_float2(0.f, 0.f);
    float4 n_co = make_float4(0.f, 0.f, 0.f, 0.f), n_cd = make_float4(0.f, 0.f, 0.f, 0.f);
#define GVD_BWD_FETCH(BASE)                                                                       \
    n_keep = false;                                                                               \
    if ((BASE) + (uint32_t)lane < hi - lo) {                                                      \
        n_ord = hi - 1u - (BASE) - (uint32_t)lane;  /* value of `contributor` after its decrement */ \
        n_id = a.point_list[r0 + n_ord];                                                          \
        n_keep = my_mask[n_ord] != 0;                                                             \
        if (n_keep) {                                                                             \
            n_xy = reinterpret_cast<const float2*>(a.means2D)[n_id];                              \
            n_co = reinterpret_cast<const float4*>(a.conic_opacity)[n_id];                        \
            n_cd = reinterpret_cast<const float4*>(a.rgbd)[n_id];                                 \
            if (!LIGHT) {                                                                         \
                n_rad = a.radii[n_id];                                                            \
                n_poff = n_id ? a.point_offsets[n_id - 1] : 0u;                                   \
            }                                                                                     \
        }                                                                                         \
    }
  // entries [lo, hi) of the list, back to front; LIGHT: recurrences only (no gradient terms, no sub-records)
  auto run = [&](const uint32_t lo, const uint32_t hi, auto light_tag) {
    constexpr bool LIGHT = decltype(light_tag)::value;
    GVD_BWD_FETCH(0u)

    for (uint32_t base = 0; base < hi - lo; base += 64) {
#ifdef GVD_RBWD_TRACE
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- stage (descending list order) + compact the entries that passed the forward's test against this quadrant ----
        const bool keep = n_keep;
        const unsigned long long m = __ballot(keep);
        const uint32_t n = (uint32_t)__popcll(m);
        if (keep) {
            const uint32_t slot = (uint32_t)__popcll(m & below);
            // the instance's sub-record: slot in Gaussian order (point_offsets[id - 1] + row-major index of the tile in the rect), x 4 + quadrant
            uint32_t g = 0;
            if (!LIGHT) {
                const int4 r = get_rect(n_xy.x, n_xy.y, n_rad, a.gx, a.gy);
                g = n_poff + (uint32_t)((ty - r.y) * (r.z - r.x) + (tx - r.x));
            }
            s_rec[2 * slot] = make_float4(n_xy.x, n_xy.y, __uint_as_float(n_ord), __uint_as_float(g * 4u + (uint32_t)quad));
            s_rec[2 * slot + 1] = n_co;
            s_cd[slot] = n_cd;
        }
        GVD_BWD_FETCH(base + 64u)
#ifdef GVD_RBWD_TRACE
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
        stage_cycles += tw0 - ts0;
#endif

// 1 / (1 - alpha) for T / (1 - alpha) (backward.cu:510) and for the background term -T_final / (1 - alpha) * (bg . dL_dpix)
// (backward.cu:575-577): v_rcp_f32 + one Newton step, shared by both uses.  Measured against the oracle on the C2 scene the
// gradient errors equal those of the IEEE division; the raw v_rcp_f32 (1 ulp) alone doubled the dL_dscales error and was rejected.
#define GVD_BWD_RCP(D) ([&] { const float r0_ = __builtin_amdgcn_rcpf(D); return fmaf(fmaf(-(D), r0_, 1.0f), r0_, r0_); }())
#define GVD_BWD_GEOM(J, SL)                                                                       \
        const float4 gxy##J = s_rec[2 * (SL)];                                                    \
        const float4 con##J = s_rec[2 * (SL) + 1];                                                \
        const uint32_t ord##J = __float_as_uint(gxy##J.z);                                        \
        const float dx##J = gxy##J.x - pixfx, dy##J = gxy##J.y - pixfy;                           \
        const float pw##J = gauss_power(con##J.x, con##J.y, con##J.z, dx##J, dy##J);              \
        const float G##J = __expf(pw##J);                                                         \
        const float alpha##J = fminf(0.99f, con##J.w * G##J);                                     \
        const bool act##J = (ord##J < last_contributor) && !(pw##J > 0.0f) && !(alpha##J < 1.0f / 255.0f);  \
        /* the wave's active lanes: the AND of the three compares' lane masks */                  \
        const unsigned long long msk##J = __builtin_amdgcn_ballot_w64(ord##J < last_contributor) &  \
                             __builtin_amdgcn_ballot_w64(!(pw##J > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha##J < 1.0f / 255.0f)); \
        const bool any##J = msk##J != 0ull;
// The serial part and the entry's row of the strip.  Inactive lanes run with alpha = G = 0: then Tn == T, every accum_rec' equals the
// value the next active entry would have formed (fmaf(1, acc, 0 * c) == acc bit-exactly) and q = w = 0.
#define GVD_BWD_SERIAL(J, SL)                                                                     \
        {                                                                                         \
            const float4 c = s_cd[SL];                                                            \
            const float am = act##J ? alpha##J : 0.f;                                             \
            const float gm = act##J ? G##J : 0.f;                                                 \
            const float one_m_a = 1.f - am;                                                       \
            const float rinv = GVD_BWD_RCP(one_m_a);                                              \
            T = T * rinv;                                                                         \
            const float wgt = am * T; (void)wgt; (void)gm;                                        \
            const float oml = 1.f - last_alpha;                                                   \
            acc0 = fmaf(oml, acc0, last_alpha * lc0);                                             \
            acc1 = fmaf(oml, acc1, last_alpha * lc1);                                             \
            acc2 = fmaf(oml, acc2, last_alpha * lc2);                                             \
            if (DA) {                                                                             \
                acc_d = fmaf(oml, acc_d, last_alpha * last_depth);                                \
                acc_a = fmaf(oml, acc_a, last_alpha);                                             \
            }                                                                                     \
            if (!LIGHT) {                                                                         \
                float d_ = (c.x - acc0) * dLp0;                                                   \
                d_ = fmaf(c.y - acc1, dLp1, d_);                                                  \
                d_ = fmaf(c.z - acc2, dLp2, d_);                                                  \
                if (DA) {                                                                         \
                    d_ = fmaf(c.w - acc_d, dLd, d_);                                              \
                    d_ = fmaf(1.f - acc_a, dLa, d_);                                              \
                }                                                                                 \
                d_ *= T;                                                                          \
                if (BG) { /* (-T_final / (1 - alpha)) * (bg . dL_dpix): quotient refined by one residual step */ \
                    float qb = nTf * rinv;                                                        \
                    qb = fmaf(fmaf(-one_m_a, qb, nTf), rinv, qb);                                 \
                    d_ = fmaf(qb, bg_dot, d_);                                                    \
                }                                                                                 \
                float2* row = &s_qw[rows][0];                                                     \
                row[lane] = make_float2(gm * d_, wgt);                                            \
                if (lane == 0) row[64] = make_float2(__uint_as_float(SL), 0.f);                   \
                rows++;                                                                           \
            }                                                                                     \
            lc0 = c.x; lc1 = c.y; lc2 = c.z; last_depth = c.w; last_alpha = am;                   \
        }
// TWO ENTRIES IN ONE SERIAL STEP (round 6; GVD_BWD_MERGE).  When no pixel of the quadrant is active in BOTH entries of a pair, every lane
// has at most one of them to walk: it picks its own (alpha, G, colour) and the pair costs one pass through the serial block instead of two.
// Bit-identical to the two steps: an inactive step multiplies T by exactly 1 and leaves, through (last_alpha, last colour), the same
// pending update of the accum_rec recurrences that the next walked entry applies -- here it simply stays pending one entry longer
// (fmaf(1 - a, acc, a c) is evaluated once, on the same operands).  The lane writes its (q, w) into its entry's row of the strip and
// zeros into the other one.  Measured on the C2 views: tests/scripts/lane_stats.py counts 9.5 % of the walked entries as removable this
// way (windows of 2; 16 % for windows of 4, 22 % of 8 -- at the price of per-lane entry queues).
#define GVD_BWD_MERGED(SL)                                                                        \
        {                                                                                         \
            const float4 c = s_cd[act0 ? (SL) : (SL) + 1u];                                       \
            const float am = act0 ? alpha0 : (act1 ? alpha1 : 0.f);                               \
            const float gm = act0 ? G0 : (act1 ? G1 : 0.f);                                       \
            const float one_m_a = 1.f - am;                                                       \
            const float rinv = GVD_BWD_RCP(one_m_a);                                              \
            T = T * rinv;                                                                         \
            const float wgt = am * T; (void)wgt; (void)gm;                                        \
            const float oml = 1.f - last_alpha;                                                   \
            acc0 = fmaf(oml, acc0, last_alpha * lc0);                                             \
            acc1 = fmaf(oml, acc1, last_alpha * lc1);                                             \
            acc2 = fmaf(oml, acc2, last_alpha * lc2);                                             \
            if (DA) {                                                                             \
                acc_d = fmaf(oml, acc_d, last_alpha * last_depth);                                \
                acc_a = fmaf(oml, acc_a, last_alpha);                                             \
            }                                                                                     \
            if (!LIGHT) {                                                                         \
                float d_ = (c.x - acc0) * dLp0;                                                   \
                d_ = fmaf(c.y - acc1, dLp1, d_);                                                  \
                d_ = fmaf(c.z - acc2, dLp2, d_);                                                  \
                if (DA) {                                                                         \
                    d_ = fmaf(c.w - acc_d, dLd, d_);                                              \
                    d_ = fmaf(1.f - acc_a, dLa, d_);                                              \
                }                                                                                 \
                d_ *= T;                                                                          \
                if (BG) {                                                                         \
                    float qb = nTf * rinv;                                                        \
                    qb = fmaf(fmaf(-one_m_a, qb, nTf), rinv, qb);                                 \
                    d_ = fmaf(qb, bg_dot, d_);                                                    \
                }                                                                                 \
                const uint32_t mine = act0 ? rows : rows + 1u, other = act0 ? rows + 1u : rows;   \
                s_qw[mine][lane] = make_float2(gm * d_, wgt);                                     \
                s_qw[other][lane] = make_float2(0.f, 0.f);                                        \
                if (lane == 0) {                                                                  \
                    s_qw[rows][64] = make_float2(__uint_as_float(SL), 0.f);                       \
                    s_qw[rows + 1u][64] = make_float2(__uint_as_float((SL) + 1u), 0.f);           \
                }                                                                                 \
                rows += 2;                                                                        \
            }                                                                                     \
            lc0 = c.x; lc1 = c.y; lc2 = c.z; last_depth = c.w; last_alpha = am;                   \
        }
        // ---- the turn-around: rows [0, nrows) of the strip -> the entries' sub-records ----
        auto flush = [&](const uint32_t nrows) {
#pragma clang fp contract(fast)
            if ((uint32_t)te < nrows) {
                const float2* row = &s_qw[te][0];
                const uint32_t sl = __float_as_uint(row[64].x);
                const float4 g0 = s_rec[2 * sl];
                const float gxr = g0.x - qx0, gyr = g0.y - gy0;
                const float2* qp = row + tg * kRows;
                const float4* dp = &s_dl[tg * kRows];
                float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f, S5 = 0.f, S6 = 0.f, S7 = 0.f, S8 = 0.f, S9 = 0.f;
#pragma unroll
                for (int t = 0; t < kRows; t++) {
                    const float2 qw = qp[t];
                    const float4 dl = dp[t];
                    const float dx = gxr - (float)(t & 7), dy = gyr - (float)(t >> 3);
                    const float qdx = qw.x * dx, qdy = qw.x * dy;
                    S0 += qw.x; S1 += qdx; S2 += qdy;
                    S3 = fmaf(qdx, dx, S3); S4 = fmaf(qdx, dy, S4); S5 = fmaf(qdy, dy, S5);
                    S6 = fmaf(qw.y, dl.x, S6); S7 = fmaf(qw.y, dl.y, S7); S8 = fmaf(qw.y, dl.z, S8);
                    if (DA) S9 = fmaf(qw.y, dl.w, S9);
                }
#pragma unroll
                for (int msk = kRows; msk < 64; msk <<= 1) {
                    S0 += __shfl_xor(S0, msk, 64); S1 += __shfl_xor(S1, msk, 64); S2 += __shfl_xor(S2, msk, 64);
                    S3 += __shfl_xor(S3, msk, 64); S4 += __shfl_xor(S4, msk, 64); S5 += __shfl_xor(S5, msk, 64);
                    S6 += __shfl_xor(S6, msk, 64); S7 += __shfl_xor(S7, msk, 64); S8 += __shfl_xor(S8, msk, 64);
                    if (DA) S9 += __shfl_xor(S9, msk, 64);
                }
                if (tg == 0) {
                    const float4 con = s_rec[2 * sl + 1];
                    const float ox = con.w * nddelx_dx, oy = con.w * nddely_dy, oh = -0.5f * con.w;
                    const size_t sub = (size_t)__float_as_uint(g0.w);   // 4 * instance slot + quadrant
                    float4* dst = reinterpret_cast<float4*>(a.partials + sub * kPartialStride);
                    dst[0] = make_float4(ox * fmaf(con.y, S2, con.x * S1),      // sum (o q) (-0.5 W) (a dx + b dy)
                                         oy * fmaf(con.y, S1, con.z * S2), oh * S3, oh * S4);
                    dst[1] = make_float4(oh * S5, S0, S6, S7);
                    dst[2] = make_float4(S8, DA ? S9 : 0.f, 0.f, 0.f);
                    reinterpret_cast<uint8_t*>(a.pflags)[sub] = 1;
                }
            }
        };
        // The walk is instantiated twice, with and without the background term (6 VALU instructions per entry that are
        // exactly zero for the black background of train_guidedvd.py:301): one wave-uniform branch per trip picks.
        auto walk = [&](auto bg_tag) {
#pragma clang fp contract(fast)  // gradient terms are tolerance-checked (1e-4), not bit-pinned
            constexpr bool BG = decltype(bg_tag)::value;
            uint32_t rows = 0;   // wave-uniform: filled rows of the strip
            uint32_t j = 0;
            for (; j + 2 <= n; j += 2) {
                GVD_BWD_GEOM(0, j)
                GVD_BWD_GEOM(1, j + 1)
#if GVD_BWD_MERGE
                if (any0 && any1 && !(msk0 & msk1)) GVD_BWD_MERGED(j)
                else
#endif
                {
                    if (any0) GVD_BWD_SERIAL(0, j)
                    if (any1) GVD_BWD_SERIAL(1, j + 1)
                }
                if (rows >= (uint32_t)kRows - 1u) { flush(rows); rows = 0; }
            }
            if (j < n) {
                GVD_BWD_GEOM(0, j)
                if (any0) GVD_BWD_SERIAL(0, j)
            }
            if (rows) flush(rows);
        };
        if (has_bg) walk(std::true_type{});
        else walk(std::false_type{});
#undef GVD_BWD_SERIAL
#undef GVD_BWD_MERGED
#undef GVD_BWD_GEOM
#undef GVD_BWD_RCP
#ifdef GVD_RBWD_TRACE
        const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
        walk_cycles += tw1 - tw0;
#endif
    }
  };
    if (part_hi < n_walk) run(part_hi, n_walk, std::true_type{});
    run(part_lo, part_hi, std::false_type{});
#undef GVD_BWD_FETCH
#ifdef GVD_RBWD_TRACE
    if (lane == 0) { GVD_RT(5, stage_cycles); GVD_RT(6, walk_cycles); GVD_RT(7, tail_cycles); GVD_RT(1, __builtin_amdgcn_s_memrealtime()); }
#endif
}

// ------------------------------------------------------------------------------------------------
// Per-Gaussian part of the backward.  The 3*M SH gradients are an outer product, dsh[3k+c] = (w_k * dRGB[c]) * conf.
// STAGED == false: `dsh` is the Gaussian's global row and is written directly.  STAGED == true (M == 16): `dsh` is the
// Gaussian's row of the block's LDS tile and receives only the FACTORS (kShW + k: w_k, kShRGB + c: dRGB[c], kShConf,
// kShCount: number of basis functions of the active degree); k_gather_bwd expands them when it streams the tile out.
constexpr int kShW = 0, kShRGB = 16, kShConf = 19, kShCount = 20, kShRow = 21;
// Sums of one Gaussian's flagged sub-records of quadrant column q (fixed instance order).  Called by the four lanes of a quad
// together: per trip the quad fetches the flag words of four instances (lane j loads instance g + j's word, the quad exchanges
// them), then every lane has its column's up to four sub-records in flight at once -- one round trip for the flags and one for
// the records per four instances.
__device__ __forceinline__ void gather_column(const GatherBwdArgs& a, const uint32_t beg, const uint32_t end, const int q, uint32_t mine,
                                              float* s)
{
    // `mine`: this lane's flag word of the first trip (instance beg + q), fetched by the caller for all of its Gaussians at once; the
    // word of the NEXT trip is requested together with this trip's records, so a trip is one memory round trip, not two
    for (uint32_t g = beg; g < end; g += 4) {
        const uint32_t next = (g + 4u + (uint32_t)q < end) ? a.pflags[g + 4u + q] : 0u;
        uint32_t f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) f[j] = (uint32_t)__shfl((int)mine, j, 4);
        float4 r[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((f[j] >> (8 * q)) & 0xffu) {
                const float4* rec = reinterpret_cast<const float4*>(a.partials + ((size_t)(g + j) * 4 + q) * kPartialStride);
                r[j][0] = rec[0]; r[j][1] = rec[1]; r[j][2] = rec[2];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((f[j] >> (8 * q)) & 0xffu) {
                s[0] += r[j][0].x; s[1] += r[j][0].y; s[2] += r[j][0].z; s[3] += r[j][0].w;
                s[4] += r[j][1].x; s[5] += r[j][1].y; s[6] += r[j][1].z; s[7] += r[j][1].w;
                s[8] += r[j][2].x; s[9] += r[j][2].y;
            }
        }
        mine = next;
    }
}

// `s`: the Gaussian's ten per-pixel-sum totals (k_gather_bwd's first phase).
template <bool STAGED>
__device__ __forceinline__ void gather_body(const GatherBwdArgs& a, const int idx, const float* __restrict__ s, float* dsh)
{
    const bool visible = a.radii[idx] > 0 && a.scalars[2] == 0;  // overflowed forward: all-zero gradients
    // Optional per-Gaussian confidence (the fork's Python-side scaling, ref __init__.py:147-157, folded
    // in): every returned gradient except the screen-space one is multiplied by conf AFTER it has been
    // formed exactly as without confidence (x * 1.0f == x, so conf == NULL and conf == 1 agree bitwise).
    const float conf = a.confidence ? a.confidence[idx] : 1.0f;
    a.dL_dmean2D[3 * idx] = s[0];
    a.dL_dmean2D[3 * idx + 1] = s[1];
    a.dL_dmean2D[3 * idx + 2] = 0.f;
    reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(s[2], s[3], 0.f, s[4]);
    a.dL_dopacity[idx] = s[5] * conf;
    a.dL_dcolor[3 * idx] = s[6] * conf;
    a.dL_dcolor[3 * idx + 1] = s[7] * conf;
    a.dL_dcolor[3 * idx + 2] = s[8] * conf;
    a.dL_ddepth[idx] = s[9];

    float* dm = a.dL_dmean3D + 3 * idx;
    float* dcov = a.dL_dcov3D + 6 * idx;
    float* ds = a.dL_dscale + 3 * idx;
    float* dq = a.dL_drot + 4 * idx;
    if (!visible) {
        dm[0] = dm[1] = dm[2] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = 0.f;
        if (STAGED) dsh[kShCount] = 0.f;   // zero basis functions: the whole row streams out as +0
        else for (int i = 0; i < 3 * a.M; i++) dsh[i] = 0.f;
        ds[0] = ds[1] = ds[2] = 0.f;
        dq[0] = dq[1] = dq[2] = dq[3] = 0.f;
        return;
    }
    const float* view = a.viewmatrix;
    const float* proj = a.projmatrix;
    // ---------------- computeCov2DCUDA (backward.cu:144-274) ----------------
    const float* cov3D = a.cov3D + 6 * idx;
    const float m0 = a.means3D[3 * idx], m1 = a.means3D[3 * idx + 1], m2 = a.means3D[3 * idx + 2];
    const float dcx = s[2], dcy = s[3], dcz = s[4];
    float t0 = view[0] * m0 + view[4] * m1 + view[8] * m2 + view[12];
    float t1 = view[1] * m0 + view[5] * m1 + view[9] * m2 + view[13];
    const float t2 = view[2] * m0 + view[6] * m1 + view[10] * m2 + view[14];
    const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
    const float txtz = t0 / t2, tytz = t1 / t2;
    t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
    t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float fx = a.focal_x, fy = a.focal_y;
    // GLM column-major X[c][r]
    const float J[3][3] = { { fx / t2, 0.0f, -(fx * t0) / (t2 * t2) }, { 0.0f, fy / t2, -(fy * t1) / (t2 * t2) }, { 0.f, 0.f, 0.f } };
    const float Wm[3][3] = { { view[0], view[4], view[8] }, { view[1], view[5], view[9] }, { view[2], view[6], view[10] } };
    const float Vrk[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
    float T[3][3], A[3][3], c2[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) T[j][i] = Wm[0][i] * J[j][0] + Wm[1][i] * J[j][1] + Wm[2][i] * J[j][2];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) A[j][i] = T[i][0] * Vrk[0][j] + T[i][1] * Vrk[1][j] + T[i][2] * Vrk[2][j];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) c2[j][i] = A[0][i] * T[j][0] + A[1][i] * T[j][1] + A[2][i] * T[j][2];
    const float ca = c2[0][0] + 0.3f, cb = c2[0][1], cc = c2[1][1] + 0.3f;
    const float denom = ca * cc - cb * cb;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
        dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
        dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
        dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
        dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
        dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
        dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
        dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
        dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
    const float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                          (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
    const float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                          (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
    const float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                          (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
    const float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                          (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
    const float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                          (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
    const float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                          (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
    const float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
    const float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
    const float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
    const float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
    const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -fx * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -fy * tz2 * dL_dJ12;
    const float dL_dtz = -fx * tz2 * dL_dJ00 - fy * tz2 * dL_dJ11 + (2 * fx * t0) * tz3 * dL_dJ02 + (2 * fy * t1) * tz3 * dL_dJ12;
    float g0 = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;  // assigned, backward.cu:273
    float g1 = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
    float g2 = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

    // ---------------- preprocessCUDA backward (backward.cu:346-412) ----------------
    const float m_hom_w = proj[3] * m0 + proj[7] * m1 + proj[11] * m2 + proj[15];
    const float m_w = 1.0f / (m_hom_w + 0.0000001f);
    const float mul1 = (proj[0] * m0 + proj[4] * m1 + proj[8] * m2 + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m0 + proj[5] * m1 + proj[9] * m2 + proj[13]) * m_w * m_w;
    const float g2x = s[0], g2y = s[1];
    g0 += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    g1 += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    g2 += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    const float mul3 = view[2] * m0 + view[6] * m1 + view[10] * m2 + view[14];
    const float gd = s[9];
    g0 += (view[2] - view[3] * mul3) * gd;
    g1 += (view[6] - view[7] * mul3) * gd;
    g2 += (view[10] - view[11] * mul3) * gd;

    // ---------------- SH backward (backward.cu:20-139) ----------------
    if (a.has_sh) {
        const float dox = m0 - a.campos[0], doy = m1 - a.campos[1], doz = m2 - a.campos[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float x = dox / len, y = doy / len, z = doz / len;
        const float* sh = a.shs + (size_t)idx * a.M * 3;
        const uint32_t cl = a.clamped[idx];
        float dRGB[3] = { s[6] * ((cl & 1u) ? 0.f : 1.f), s[7] * ((cl & 2u) ? 0.f : 1.f), s[8] * ((cl & 4u) ? 0.f : 1.f) };
        float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
        const int D = a.D;
#define SHV(k, ch) sh[3 * (k) + (ch)]
#define DSH(k, wgt) { const float w_ = (wgt); if (STAGED) dsh[kShW + (k)] = w_; else { dsh[3 * (k)] = (w_ * dRGB[0]) * conf; dsh[3 * (k) + 1] = (w_ * dRGB[1]) * conf; dsh[3 * (k) + 2] = (w_ * dRGB[2]) * conf; } }
        DSH(0, SH_C0);
        int written = 1;
        if (D > 0) {
            DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
            written = 4;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] = -SH_C1 * SHV(3, ch);
                dRGBdy[ch] = -SH_C1 * SHV(1, ch);
                dRGBdz[ch] = SH_C1 * SHV(2, ch);
            }
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                DSH(4, SH_C2_0 * xy); DSH(5, SH_C2_1 * yz); DSH(6, SH_C2_2 * (2.f * zz - xx - yy));
                DSH(7, SH_C2_3 * xz); DSH(8, SH_C2_4 * (xx - yy));
                written = 9;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += SH_C2_0 * y * SHV(4, ch) + SH_C2_2 * 2.f * -x * SHV(6, ch) + SH_C2_3 * z * SHV(7, ch) + SH_C2_4 * 2.f * x * SHV(8, ch);
                    dRGBdy[ch] += SH_C2_0 * x * SHV(4, ch) + SH_C2_1 * z * SHV(5, ch) + SH_C2_2 * 2.f * -y * SHV(6, ch) + SH_C2_4 * 2.f * -y * SHV(8, ch);
                    dRGBdz[ch] += SH_C2_1 * y * SHV(5, ch) + SH_C2_2 * 2.f * 2.f * z * SHV(6, ch) + SH_C2_3 * x * SHV(7, ch);
                }
                if (D > 2) {
                    DSH(9, SH_C3_0 * y * (3.f * xx - yy));
                    DSH(10, SH_C3_1 * xy * z);
                    DSH(11, SH_C3_2 * y * (4.f * zz - xx - yy));
                    DSH(12, SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy));
                    DSH(13, SH_C3_4 * x * (4.f * zz - xx - yy));
                    DSH(14, SH_C3_5 * z * (xx - yy));
                    DSH(15, SH_C3_6 * x * (xx - 3.f * yy));
                    written = 16;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        dRGBdx[ch] += (SH_C3_0 * SHV(9, ch) * 3.f * 2.f * xy +
                                       SH_C3_1 * SHV(10, ch) * yz +
                                       SH_C3_2 * SHV(11, ch) * -2.f * xy +
                                       SH_C3_3 * SHV(12, ch) * -3.f * 2.f * xz +
                                       SH_C3_4 * SHV(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                                       SH_C3_5 * SHV(14, ch) * 2.f * xz +
                                       SH_C3_6 * SHV(15, ch) * 3.f * (xx - yy));
                        dRGBdy[ch] += (SH_C3_0 * SHV(9, ch) * 3.f * (xx - yy) +
                                       SH_C3_1 * SHV(10, ch) * xz +
                                       SH_C3_2 * SHV(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                                       SH_C3_3 * SHV(12, ch) * -3.f * 2.f * yz +
                                       SH_C3_4 * SHV(13, ch) * -2.f * xy +
                                       SH_C3_5 * SHV(14, ch) * -2.f * yz +
                                       SH_C3_6 * SHV(15, ch) * -3.f * 2.f * xy);
                        dRGBdz[ch] += (SH_C3_1 * SHV(10, ch) * xy +
                                       SH_C3_2 * SHV(11, ch) * 4.f * 2.f * yz +
                                       SH_C3_3 * SHV(12, ch) * 3.f * (2.f * zz - xx - yy) +
                                       SH_C3_4 * SHV(13, ch) * 4.f * 2.f * xz +
                                       SH_C3_5 * SHV(14, ch) * (xx - yy));
                    }
                }
            }
        }
#undef SHV
#undef DSH
        if (STAGED) {
            dsh[kShRGB] = dRGB[0]; dsh[kShRGB + 1] = dRGB[1]; dsh[kShRGB + 2] = dRGB[2];
            dsh[kShConf] = conf;
            dsh[kShCount] = (float)written;
        } else {
            for (int k = written; k < a.M; k++) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
        }
        const float ddx = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
        const float ddy = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
        const float ddz = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
        const float sum2 = dox * dox + doy * doy + doz * doz;  // dnormvdv, auxiliary.h:107-117
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        g0 += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
        g1 += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
        g2 += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
    } else {
        if (STAGED) dsh[kShCount] = 0.f;
        else for (int i = 0; i < 3 * a.M; i++) dsh[i] = 0.f;
    }
    dm[0] = g0 * conf; dm[1] = g1 * conf; dm[2] = g2 * conf;

    // ---------------- cov3D backward (backward.cu:278-341) ----------------
    if (a.has_scales) {
        const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        const float R[3][3] = { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
                                { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
                                { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } };
        const float sv[3] = { a.scale_modifier * a.scales[3 * idx], a.scale_modifier * a.scales[3 * idx + 1], a.scale_modifier * a.scales[3 * idx + 2] };
        float Mm[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) Mm[j][i] = sv[i] * R[j][i];
        const float dS[3][3] = { { dcov[0], 0.5f * dcov[1], 0.5f * dcov[2] }, { 0.5f * dcov[1], dcov[3], 0.5f * dcov[4] }, { 0.5f * dcov[2], 0.5f * dcov[4], dcov[5] } };
        float dM[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++)
                dM[j][i] = (2.0f * Mm[0][i]) * dS[j][0] + (2.0f * Mm[1][i]) * dS[j][1] + (2.0f * Mm[2][i]) * dS[j][2];
        float Rt[3][3], dMt[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) { Rt[j][i] = R[i][j]; dMt[j][i] = dM[i][j]; }
#pragma unroll
        for (int k = 0; k < 3; k++) ds[k] = Rt[k][0] * dMt[k][0] + Rt[k][1] * dMt[k][1] + Rt[k][2] * dMt[k][2];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 3; i++) dMt[k][i] *= sv[k];
        dq[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
        dq[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
        dq[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
        dq[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
        if (a.confidence) {
#pragma unroll
            for (int k = 0; k < 3; k++) ds[k] *= conf;
#pragma unroll
            for (int k = 0; k < 4; k++) dq[k] *= conf;
        }
    } else {
        ds[0] = ds[1] = ds[2] = 0.f;
        dq[0] = dq[1] = dq[2] = dq[3] = 0.f;
    }
}

// A workgroup handles kGatherG = 256 consecutive Gaussians in two phases.
//   Phase 1, four lanes per Gaussian (64 Gaussians per pass, 4 passes): lane q adds the Gaussian's flagged sub-records of quadrant
//   column q over its contiguous run of instances (a quad reads one instance's 192 contiguous bytes together), then the four
//   columns meet as (q0 + q1) + (q2 + q3) -- a fixed order, so the totals are bit-identical from run to run; they go to LDS.
//   Phase 2, one lane per Gaussian: computeCov2D / preprocess / SH / cov3D backward on the totals.
// dL_dsh is 48 floats (192 B) per Gaussian: written per thread it is a 192-byte-stride scatter (64 cache lines per store
// instruction).  Instead each Gaussian drops the 21 factors of its row into an LDS tile (odd row stride: conflict-free) and the
// block expands and streams the tile out as contiguous float4 (1 KiB per wave store).
#ifndef GVD_GATHER_G
#define GVD_GATHER_G 256
#endif
constexpr int kGatherG = GVD_GATHER_G;   // Gaussians == threads per workgroup (A/B builds: -DGVD_GATHER_G=128 / 512)
__global__ void __launch_bounds__(kGatherG) k_gather_bwd(GatherBwdArgs a)
{
    extern __shared__ float s_sh[];  // [kGatherG][kShRow] when M == 16 (launch passes the size), else unused
    __shared__ float s_sum[kGatherG][kNV + 1];
    const int tid = threadIdx.x;
    {
        const int q = tid & 3;
        const bool live = a.scalars[2] == 0;   // overflowed forward: all-zero gradients
        // the run bounds and the first flag word of all four passes' Gaussians up front: two memory round trips for the workgroup
        // instead of three dependent ones per pass
        constexpr int NPASS = 4, GPP = kGatherG / 4;   // four lanes per Gaussian: GPP Gaussians per pass
        uint32_t beg[NPASS], end[NPASS], first[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; pass++) {
            const int idx = blockIdx.x * kGatherG + pass * GPP + (tid >> 2);
            beg[pass] = end[pass] = 0u;
            if (live && idx < a.P) {
                const int rad = a.radii[idx];
                const uint32_t b = idx ? a.point_offsets[idx - 1] : 0u, e = a.point_offsets[idx];
                if (rad > 0) { beg[pass] = b; end[pass] = e; }
            }
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; pass++) first[pass] = (beg[pass] + (uint32_t)q < end[pass]) ? a.pflags[beg[pass] + q] : 0u;
#pragma unroll
        for (int pass = 0; pass < NPASS; pass++) {
            const int gi = pass * GPP + (tid >> 2);
            float s[kNV];
#pragma unroll
            for (int v = 0; v < kNV; v++) s[v] = 0.f;
            gather_column(a, beg[pass], end[pass], q, first[pass], s);
#pragma unroll
            for (int v = 0; v < kNV; v++) {
                float t = s[v];
                t += __shfl_xor(t, 1, 64);   // (q0 + q1) | (q2 + q3)
                t += __shfl_xor(t, 2, 64);   // the same two numbers added in either order: one value in all four lanes
                s[v] = t;
            }
            if (q == 0) {
#pragma unroll
                for (int v = 0; v < kNV; v++) s_sum[gi][v] = s[v];
            }
        }
    }
    __syncthreads();
    const bool stage_sh = (a.M == 16);
    {
        const int idx = blockIdx.x * kGatherG + tid;
        if (idx < a.P) {
            if (stage_sh) gather_body<true>(a, idx, &s_sum[tid][0], s_sh + tid * kShRow);
            else gather_body<false>(a, idx, &s_sum[tid][0], a.dL_dsh + (size_t)idx * a.M * 3);
        }
    }
    if (stage_sh) {
        __syncthreads();
        const size_t block_base = (size_t)blockIdx.x * kGatherG * 48;
        const size_t total = (size_t)a.P * 48;
#pragma unroll
        for (int k = 0; k < 48 / 4; k++) {
            const int i = (k * kGatherG + tid) * 4;  // float index inside the block's kGatherG x 48 tile
            if (block_base + i < total) {
                const int g = i / 48, c = i - g * 48;
                const float* r = s_sh + g * kShRow;
                const int count = (int)r[kShCount];
                const float conf = r[kShConf];
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int e = c + j, kk = e / 3, ch = e - kk * 3;
                    o[j] = kk < count ? (r[kShW + kk] * r[kShRGB + ch]) * conf : 0.f;   // same product order as the direct form
                }
                *reinterpret_cast<float4*>(a.dL_dsh + block_base + i) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// dL_dcov3D is an input of the scale/rotation backward above, so it is scaled last, in place.
__global__ void __launch_bounds__(256) k_scale_cov(int n, float* __restrict__ dL_dcov3D, const float* __restrict__ conf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dL_dcov3D[i] *= conf[i / 6];
}

void launch_render_bwd(const RenderBwdArgs& a, int T, hipStream_t s)
{
    static const int rows8 = getenv("GVD_BWD_ROWS8") ? atoi(getenv("GVD_BWD_ROWS8")) : 0;   // A/B switch (round 5): 8-row strip
    const bool da = a.dL_dpix_depth || a.dL_dalphas;
    const int units = (int)a.extra_units + ((T + 7) / 8) * 32;   // extra segments of the first split_positions tiles, then one unit per (tile, quadrant)
#define GVD_LAUNCH(K) hipLaunchKernelGGL(K, dim3(units), dim3(64), 0, s, a)
    if (rows8) { if (da) GVD_LAUNCH((k_render_bwd<true, 8>)); else GVD_LAUNCH((k_render_bwd<false, 8>)); }
    else { if (da) GVD_LAUNCH((k_render_bwd<true, 16>)); else GVD_LAUNCH((k_render_bwd<false, 16>)); }
#undef GVD_LAUNCH
}
void launch_gather_bwd(const GatherBwdArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_bwd, dim3((a.P + kGatherG - 1) / kGatherG), dim3(kGatherG), a.M == 16 ? (size_t)kGatherG * kShRow * 4 : 0, s, a);
    // only needed when the caller consumes dL_dcov3D (precomputed-covariance path)
    if (a.confidence && !a.has_scales)
        hipLaunchKernelGGL(k_scale_cov, dim3((a.P * 6 + 255) / 256), dim3(256), 0, s, a.P * 6, a.dL_dcov3D, a.confidence);
}

}  // namespace gvd

#ifdef GVD_RBWD_TRACE
extern "C" int gvd_debug_rtrace_read(unsigned long long* dst, size_t n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gvd::g_rtrace), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
