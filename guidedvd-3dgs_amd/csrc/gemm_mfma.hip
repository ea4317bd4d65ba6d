// gemm_mfma.hip -- hand-written gfx950 NT GEMM  Y[M, N] = X[M, K] . W[N, K]^T  on MFMA 16x16x32 with the transformer-side
// epilogues of the ViewCrafter U-Net fused in (C-ABI: include/gvd_diffusion.h, gvd_gemm_nt / gvd_row_stats).
//
// Replaces (reference lines, all nn.Linear / 1x1 convolutions on token rows; hipBLASLt through F.linear until round 2):
//   CrossAttention  to_q / to_k / to_v / to_out (+ to_k_ip / to_v_ip)          lvdm/modules/attention.py:53-57,76,86-99,144
//   BasicTransformerBlock  x + attn(LayerNorm(x)), x + ff(LayerNorm(x))         attention.py:212-246  (LayerNorm folded, residual fused)
//   GEGLU / FeedForward                                                         attention.py:415-442  (gate fused)
//   SpatialTransformer / TemporalTransformer proj_in / proj_out (+ x_in)        attention.py:249-412
//   ResBlock emb_layers, skip_connection (1x1), time / fps embedding MLPs       lvdm/modules/networks/openaimodel3d.py:109-236,360-387
//   VAE AttnBlock q / k / v / proj_out (1x1 convolutions) and its d = 512 attention as chunked GEMMs   ae_modules.py:26-78
//
// Layout: X rows = tokens (row stride ldx), W rows = output channels (row stride ldw), both K-contiguous 16-bit; Y rows = tokens.
//
// Design as of round 6 (one workgroup = 512 threads = 8 waves, one workgroup per CU; a 4-wave form with two workgroups per CU for narrow N):
//   * Tile: BN = 320 (or 256) output channels x 256 tokens.  Waves 2 (channels) x 4 (tokens); a wave owns 160 (128) channels x 64 tokens as
//     MFMA 16x16x32 blocks (round 6: more flops per watt than 32x32x16 on the power-capped board, tests/scripts/r6_mfma_power.hip; the
//     32x32x16 form of rounds 3-5 is gone from k_gemm_nt; k_gemm_skinny keeps it).  MFMA roles as in conv_mfma.hip: A = W rows (channels), B = X rows (tokens): a lane ends with 4
//     consecutive channels of ONE token per register quad, so the epilogue writes channel-contiguous rows.
//   * Staging is LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), filled linearly (wave base + lane x 16 B); the
//     bank-conflict-free layout for the ds_read_b128 operand reads is an XOR of the 16-byte slot with row bits, applied to the per-lane GLOBAL
//     address (inside the row's own line, so coalescing is untouched) and to the read address.
//   * K loop (8-wave form): a RING of four 32-channel slots, the DMA of half-tile h + 3 issued by hand-written asm between the MFMA steps of
//     h, counted vmcnt waits, the fragments that open h + 1 read across the barrier -- see the comment above k_gemm_nt.  (Rounds 3-5: two
//     64-channel stages, one barrier per K-tile, ~500 of 2750 cycles per K-tile without an MFMA in flight.)  Operand fragments are read two
//     MFMA steps ahead of their use, pinned with sched_group_barrier (GVD_GEMM_RDAHEAD).
//   * PERSISTENT workgroups (one per CU) walk their tiles; the three half-tiles that open the NEXT tile are issued before the epilogue of the
//     current one, so their HBM / L2 latency hides under the epilogue's stores (with K = 320 a tile is only ten half-tiles long).
//   * Epilogue, in 16-token passes: scale, the LAYERNORM FOLD, bias in fp32 registers; the GEGLU gate (modes of GemmArgs::geglu: inference
//     gate; gate + saved pre-activation for autograd; the gate's BACKWARD on an incoming gradient -- round 6, gvd_gemm_nt_gate); 16-bit
//     rounding; a wave-private transposition through LDS (no barriers); residual; 16-byte non-temporal stores, row-contiguous.
//   * LayerNorm fold: LN(x) W^T = rstd (x W'^T - mean s) + c with W' = W gamma (per input channel), s[n] = sum_k W'[n, k],
//     c[n] = sum_k beta[k] W[n, k] + bias[n]: the GEMM runs on the raw tokens and the normalisation is two per-row scalars
//     (gvd_row_stats) and two per-column vectors in the epilogue -- no normalised tensor is ever written or read.
//   * Slot -> tile map is XCD-aware: workgroup b runs on XCD b % 8 (observed) and keeps to slots = b mod 8; the column tiles
//     of one token tile are consecutive slots of ONE XCD, so the 256-token panel is pulled through one L2 once.
#include <stdlib.h>

#include "diffusion_common.h"

using namespace gvdd;

namespace {

struct GemmArgs {
    const void* x; const void* w; void* y;
    long long ldx, ldw, ldy, sx, sw, sy;   // row / batch strides (elements)
    int M, N, K, batch;
    int tiles_m, tiles_n, mgroups;         // mgroups = ceil(tiles_m / 8)
    int ngroup;                            // channel tiles per L2-resident group (slot order, see decode)
    float alpha;                           // scale on the accumulator (1 / sqrt(d) for attention scores)
    const float* bias;                     // [N] or nullptr (added after the fold)
    const float2* row_stats;               // [batch][M] (mean, rstd) or nullptr
    const float* col_sum;                  // [N] s[n] (with row_stats)
    const void* res; long long ldr, sr;    // residual rows (layout of y) or nullptr
    int geglu;                             // gate mode (gvd_diffusion.h): 0 none, 1 gate (y has N / 2 columns), 2 gate + the pre-activation to `aux`,
                                           // 3 gate backward (the product is d/d(gated output); `aux` = the saved pre-activation, y = its gradient, 2 N columns)
    const void* aux; long long ldaux, saux;   // [.., M, N] (mode 2, written) / [.., M, 2 N] (mode 3, read), columns in the kernel's [16 value | 16 gate] order
};

#ifndef GVD_GEMM_RDAHEAD
#define GVD_GEMM_RDAHEAD 2   // A fragments read ahead of their MFMAs (steps of 4 MFMAs)
#endif
#ifndef GVD_GEMM_NT_STORE
#define GVD_GEMM_NT_STORE 1   // the output rows leave as non-temporal stores (0: plain, for A/B builds): a 150-590 MB output of the level-0 shapes only
                              // passes through the caches on its way out -- 230 400 x 960 x 320 0.259 -> 0.234 ms, x 2560 0.588 -> 0.551, DDIM step
                              // 238.2 -> 234.7 ms (profiles/r04_nt_stores.txt); neutral on the small shapes
#endif
constexpr int BM = 256, WN = 4;   // tokens per tile: 4 wave columns x 4 blocks of 16
__device__ const uint4 g_zero16 = { 0u, 0u, 0u, 0u };   // source of K-tail slots
#ifdef GVD_GEMM_TRACE
__device__ unsigned long long g_trace[1024];   // experiments: s_memtime stamps of workgroup 0, every wave (128 slots = 16 tiles each): tests/scripts/r4_gemm_trace.py
#define GVD_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && trace_n + (i) < 128) g_trace[(threadIdx.x >> 6) * 128 + trace_n + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GVD_STAMP(i) do { } while (0)
#endif

// exact-form GELU 0.5 g (1 + erf(g / sqrt 2)) written as  relu(g) - 0.5 |g| erfc(|g| / sqrt 2)  with erfc by Abramowitz-Stegun
// 7.1.26 (erfc(x) = P(t) exp(-x^2), t = 1 / (1 + p x), |error| <= 1.5e-7: far below the 16-bit rounding of the result).  This form
// needs no sign restore and no 1 +- erf (the negative tail is the product itself, not a cancellation), and the -0.5 rides in the
// polynomial's coefficients: 11 plain VALU + v_rcp + v_exp per gate against ~40 for libdevice erff.  The gate runs once per output
// element and, the two pipes of a SIMD being all but serial on this part, its instruction count is kernel time: the PMC pass of the
// level-0 feed-forward GEMM showed VALU-active 0.44 against MFMA-busy 0.27 with the previous 14-instruction form plus three 16-bit
// round trips per element (profiles/r03_mfma_pmc.json).
__device__ __forceinline__ float gelu_erf(float g)
{
    const float ag = fabsf(g);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ag, 1.f));
    float q = fmaf(t, -0.5f * 1.061405429f, 0.5f * 1.453152027f);
    q = fmaf(t, q, -0.5f * 1.421413741f);
    q = fmaf(t, q, 0.5f * 0.284496736f);
    q = fmaf(t, q, -0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * (g * g));   // exp(-x^2), x = |g| / sqrt 2
    return fmaf(ag, (t * q) * e, fmaxf(g, 0.f));
}

// pack four fp32 values to 16 bit: (lo dword, hi dword)
template <typename T>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(Tr<T>::pack2(a, b), Tr<T>::pack2(c, d)); }

// WM: wave rows (channel halves) -- 2: 8 waves, one workgroup per CU, a RING of four 32-channel K slots (below); 1: 4 waves, two
// workgroups per CU (one's epilogue stores drain under the other's K loop), two K stages of 32 channels.
//
// The 8-wave K loop (RING).  A trace of the two-stage form (tests/scripts/r4_gemm_trace.py on a -DGVD_GEMM_TRACE build; the round-6 printout stayed in a session's scratch directory) showed what a K-tile costs beyond
// its MFMAs: the two waves of a SIMD run their MFMA streams one after the other (the older wave wins the pipe), the older one then waits at
// the K-tile's barrier, and after the barrier BOTH wait for their first operand fragments -- ~500 of 2750 cycles per 64-channel K-tile with
// no MFMA in flight on the SIMD.  Here K is cut in 32-channel half-tiles h = 0, 1, ... living in slot h & 3 of a ring of four; during half-tile
// h the workgroup's DMA for h + 3 goes out (into the slot of h - 1, free since the barrier that ended h - 1), every wave waits at the END of h
// for its own pieces of h + 2 (one full half-tile after their issue: `s_waitcnt vmcnt(pieces of h + 3)`), and the barrier that follows makes
// h + 2 visible to everyone -- so the fragments that OPEN half-tile h + 1 can be read during h, ahead of the barrier, and the MFMA stream of a
// wave continues across it without a gap.  The DMA instructions are hand-written (hipcc counts a builtin LDS-DMA as a pending LDS write and
// drains it -- vmcnt(0) -- in front of the next LDS read it cannot tell apart) and issued between the MFMA steps; the three half-tiles that
// open the NEXT tile go out before this tile's epilogue, whose staging lives in slot 3 and the space behind it.
template <typename T, int MI, int WM, int BK, int GM>
__global__ void __launch_bounds__(WM * WN * 64, 2) k_gemm_nt(const GemmArgs a)
{
    constexpr bool GEGLU = GM == 1;   // the inference gate: fp32 through the gate, y only.  Modes 2 / 3 stage the plain (rounded) tile and gate in the read-back
    typedef typename Tr<T>::vec8 vec8;
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr bool RING = WM == 2;
    static_assert(BK == 32, "32-channel K slots: 64-byte LDS rows");
    constexpr int NT = WM * WN * 64, ROWB = BK * 2, SPR = ROWB / 16;
    constexpr int MB = MI * 2, NB = 4;              // 16-row blocks of a wave: MB along the channels (A operand), NB along the tokens (B operand)
    constexpr int BN = WM * MI * 32;
    constexpr int STAGE = (BN + BM) * ROWB;
    constexpr int NQ = STAGE / 16;                  // 16-byte DMA pieces per stage
    constexpr int NP = (NQ + NT - 1) / NT;          // ... per thread (the last sweep may be partial)
    constexpr int EP_PITCH = MI * 64 + 16;          // epilogue staging: one token row of a wave (MI * 32 channels, 16 bit) + pad
    constexpr int EP_WAVE = 16 * EP_PITCH;          // ... 16 tokens per pass
    constexpr int EP_BASE = RING ? 3 * STAGE : STAGE;                                  // staging: behind the slots that receive the next tile
    constexpr int VEC_BASE = RING ? EP_BASE + WM * WN * EP_WAVE : EP_BASE;             // per-column vectors of the accumulator initialisation
    static_assert(NQ % 64 == 0 && (BN * SPR) % 64 == 0, "a wave's DMA instruction is one kind of row");
    // XOR of a row's 16-byte slots that makes the ds_read_b128 operand reads conflict free.  A 16 x 16 x 32 operand read has lane
    // (r16 = lane & 15, g4 = lane >> 4) fetch slot g4 of row r16 of its block; the LDS serves the 16-lane groups {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} (+ 32), i.e. rows {0-3, 12-15 | 4-11} at slots {g | g + 1}.  With 64-byte rows (four rows per 64-bank line)
    // slot ^ perm[(row >> 2) & 3], perm = (0, 2, 3, 1), gives 4 distinct slots x 4 rows-mod-4 in every group (the plain (row >> 2) & 3
    // of the 32 x 32 x 16 reads of rounds 3-5 would put rows 0-3 and 4-7 of a group on the same slot).  Depends on row mod 16 only.
    auto swz = [](int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; };

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // Per-lane geometry.  Re-derived from the thread id at the top of every tile (behind an opaque asm): kept live across the
    // whole persistent loop these values were spilled to scratch by the 5-block kernels (160 accumulators + fragments at the
    // 256-register cap) and reloaded with exposed latency (~10k cycles per tile); recomputing them is ~30 VALU instructions.
    int lane, wave, g4, r16, wm, wn, lrow, lslot;
    int a_off, b_off[NB];             // MFMA operand addresses within a slot (16-row block offsets do not change swz(row))
    auto geometry = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        lane = t & 63; wave = t >> 6;
        wave = __builtin_amdgcn_readfirstlane(wave);
        g4 = lane >> 4; r16 = lane & 15;
        wm = wave / WN; wn = wave % WN;
        lrow = lane / SPR;
        lslot = (lane % SPR) ^ swz(lane / SPR + wave * (64 / SPR));
        const int sl = (g4 ^ swz(r16)) << 4;
        a_off = (wm * MI * 32 + r16) * ROWB + sl;
#pragma unroll
        for (int ni = 0; ni < NB; ni++) b_off[ni] = (BN + (wn * NB + ni) * 16 + r16) * ROWB + sl;
    };
    geometry();
    const int nk = (a.K + BK - 1) / BK, nk_full = a.K / BK;

    // ---- persistent workgroup: slots s = blockIdx.x, + gridDim.x, ... (gridDim.x % 8 == 0, so a workgroup stays on "its" XCD's
    //      slots); slot -> (batch, token tile, channel tile): the channel tiles of a token tile are consecutive slots of ONE XCD,
    //      i.e. they run back to back there and the 256-token panel is pulled through one L2 once ----
    //      Within an XCD the order is: for each GROUP of `ngroup` channel tiles (their W rows, <= ~1.5 MB, stay in the XCD's
    //      4 MiB L2) / for each token tile of the XCD / for each channel tile of the group -- W is fetched from beyond L2 once per
    //      group instead of once per token tile, and a token panel is re-read `ngroup` times back to back (L2 hits).
    const int per_batch = a.mgroups * 8 * a.tiles_n, total = per_batch * a.batch;
    auto decode = [&](int slot, int& b, int& m0, int& n0) {
        b = slot / per_batch;
        const int r = slot - b * per_batch, xcd = r & 7, j = r >> 3;
        const int span = a.ngroup * a.mgroups;                       // slots of one full group
        int g = j / span;
        const int full = a.tiles_n / a.ngroup;
        if (g > full) g = full;
        const int gsz = g < full ? a.ngroup : a.tiles_n - full * a.ngroup, jj = j - g * span;
        const int tm = (jj / gsz) * 8 + xcd;
        m0 = tm * BM;
        n0 = (g * a.ngroup + jj % gsz) * BN;
        return tm < a.tiles_m;
    };
    auto next_valid = [&](int slot) {
        int b, m0, n0;
        while (slot < total && !decode(slot, b, m0, n0)) slot += gridDim.x;
        return slot;
    };

    // ---- DMA source map: piece q = p * NT + tid of a stage fills LDS bytes [16 q, 16 q + 16): row q / SPR, physical slot q % SPR,
    //      which holds the row's LOGICAL slot (q % SPR) ^ swz(row) -- the permutation stays inside the row's own 64-byte
    //      line, so the global reads stay coalesced ----
    // A wave's DMA instruction of sweep p covers 64 / SPR consecutive rows: row = R0(p, wave) + lane / SPR, physical slot
    // lane % SPR; the row's swizzle depends on (wave, lane) only (R0's contribution to the swizzled bits is 0), so the per-lane
    // state is TWO registers (row-in-instruction, logical slot) and everything else about a piece is wave-uniform (scalar):
    // no per-piece offset array lives across the K loop.  Rows past the matrix edge are clamped (loaded twice, never stored).
    constexpr int WQ = BN * SPR;                             // pieces [0, WQ) are W rows
    const char* wbase = nullptr;
    const char* xbase = nullptr;
    int wlim = 0, xlim = 0;
    auto aim = [&](int b, int m0, int n0) {
        wbase = reinterpret_cast<const char*>((const T*)a.w + (size_t)b * a.sw + (size_t)n0 * a.ldw);
        xbase = reinterpret_cast<const char*>((const T*)a.x + (size_t)b * a.sx + (size_t)m0 * a.ldx);
        wlim = a.N - 1 - n0;
        xlim = a.M - 1 - m0;
    };
    // -- the 4-wave form: builtin LDS-DMA, a K-tile's pieces as one block in front of its MFMAs --
    auto piece_src = [&](int p, int kt, int lrow) {          // (lrow: the caller's opaque copy -- keeps these ~10 instructions per piece inside the
        const int q0 = p * NT + wave * 64;                   //  K loop instead of NP hoisted 64-bit addresses in registers the loop does not have)
        const bool isw = q0 < WQ;
        int r = q0 / SPR - (isw ? 0 : BN) + lrow;
        const int lim = isw ? wlim : xlim;
        r = r < lim ? r : lim;
        const unsigned off = (unsigned)r * (unsigned)(isw ? a.ldw : a.ldx) * 2u + (unsigned)lslot * 16u + (unsigned)kt * ROWB;
        return (isw ? wbase : xbase) + off;
    };
    auto issue = [&](int kt, int stage) {                    // a full K-tile
        unsigned char* dst = lds + stage * STAGE + wave * 1024;
        int lr = lrow;
        asm volatile("" : "+v"(lr));
#pragma unroll
        for (int p = 0; p < NP; p++) {
            if ((p + 1) * NT > NQ && p * NT + wave * 64 >= NQ) continue;     // (last, partial sweep: wave-uniform)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)piece_src(p, kt, lr),
                                             (__attribute__((address_space(3))) void*)(dst + p * NT * 16), 16, 0, 0);
        }
    };
    auto issue_tail = [&](int kt, int stage) {               // the last, partial K-tile: slots past K read zeros
        unsigned char* dst = lds + stage * STAGE + wave * 1024;
        int lr = lrow;
        asm volatile("" : "+v"(lr));
#pragma unroll
        for (int p = 0; p < NP; p++) {
            if ((p + 1) * NT > NQ && p * NT + wave * 64 >= NQ) continue;
            const char* g = piece_src(p, kt, lr);
            if (kt * BK + lslot * 8 >= a.K) g = reinterpret_cast<const char*>(&g_zero16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(dst + p * NT * 16), 16, 0, 0);
        }
    };
    // -- the ring form: one piece of half-tile `h` as a hand-written instruction group: the base (tile origin + K position) is scalar, the
    //    lane adds (clamped row) x (row stride) + its slot -- three vector instructions.  `on` (wave-uniform; 0 = off) switches the piece off
    //    without a branch (EXEC = 0), so that ONE copy of the MFMA stream serves every half-tile (two copies joined by a branch cost the
    //    accumulators their fixed registers: 200-380 spilled VGPRs).  hipcc does not count these loads: the waits below do.  In the last,
    //    partial sweep (5-block tiles: 4.5 pieces per wave) waves 4-7 repeat the piece of waves 0-3 -- the same bytes to the same place --
    //    so that every wave has the same number of loads in flight for the counted waits. --
    auto ring_piece = [&](int p, int h, unsigned on) {
        int q0 = p * NT + wave * 64;
        if ((p + 1) * NT > NQ && q0 >= NQ) q0 -= NT / 2;
        const bool isw = q0 < WQ;
        const int R0 = q0 / SPR - (isw ? 0 : BN), lim = isw ? wlim : xlim;
        int r = R0 + lrow;
        r = r < lim ? r : lim;
        const unsigned voff = (unsigned)r * ((unsigned)(isw ? a.ldw : a.ldx) * 2u) + (unsigned)lslot * 16u;
        const char* sb = (isw ? wbase : xbase) + (size_t)h * ROWB;
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds + (h & 3) * STAGE + q0 * 16);
        unsigned keep_m0;
        unsigned long long keep_exec;
        // (five scalar instructions between any producer of the operands and the load: covers the M0 and VALU-written-SGPR wait states)
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %3\n\t"
                     "s_mov_b64 %1, exec\n\t"
                     "s_cmp_lg_u32 %4, 0\n\t"
                     "s_cselect_b64 exec, exec, 0\n\t"
                     "global_load_lds_dwordx4 %2, %5\n\t"
                     "s_mov_b64 exec, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0), "=&s"(keep_exec) : "v"(voff), "s"(dst), "s"(on), "s"(sb) : "memory", "scc");
    };
    auto pos = [](int d) { return (unsigned)(d & ~(d >> 31)); };   // d > 0 ? d : 0 in integer form (a compare would reach the asm operand as a vector register)
    auto ring_open = [&]() {                                  // the three half-tiles that open a tile
#pragma unroll
        for (int h = 0; h < 3; h++)
#pragma unroll
            for (int p = 0; p < NP; p++) ring_piece(p, h, pos(nk - h));
    };
    auto issue_first = [&]() {
        if constexpr (RING) ring_open();
        else { if (nk_full > 0) issue(0, 0); else issue_tail(0, 0); }
    };

    int slot = next_valid(blockIdx.x);
    if (slot >= total) return;
    int cb, cm0, cn0;
    decode(slot, cb, cm0, cn0);
    aim(cb, cm0, cn0);
    issue_first();

    int trace_n = 0;
    (void)trace_n;
    bool more = true;
    const bool scaled = a.row_stats != nullptr || a.alpha != 1.0f;       // (uniform) the epilogue multiplies by a per-row / global scale
    while (true) {
        geometry();
        // ---- accumulators start at the epilogue's additive terms, so that the epilogue itself is one multiply (or nothing), the
        //      16-bit pack and the stores: with  y = scale_m (sum_k x w' + init),
        //        LayerNorm fold:  scale_m = rstd_m,  init = c_n / rstd_m - mean_m s_n     (== rstd (acc - mean s) + c)
        //        otherwise:       scale   = alpha,   init = bias_n / alpha
        //      The per-column vectors travel through wave-private LDS: two loads per lane instead of forty. ----
        GVD_STAMP(0);
        f4v acc[MB][NB];
        float rscale[NB];
        {
            // (4-wave form: the wave's OWN epilogue staging area, free once its previous epilogue is done -- other waves may still be inside
            //  theirs, so no other part of stage 1 may be touched here; ring form: its own area behind the staging)
            float* const vec = reinterpret_cast<float*>(lds + VEC_BASE + wave * (RING ? MI * 64 * 4 : EP_WAVE));
            float rinv[NB], rmean[NB];
            const bool vecs = a.bias != nullptr || a.row_stats != nullptr;      // (uniform)
            if (vecs) {
                const int c = cn0 + wm * MI * 32 + lane * 4;
                const int cs = (lane < MI * 8 && c < a.N) ? c : 0;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = bv;
                if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + cs);
                if (a.row_stats) sv = *reinterpret_cast<const float4*>(a.col_sum + cs);
                float2 st[NB];
#pragma unroll
                for (int ni = 0; ni < NB; ni++) {
                    const int m = cm0 + (wn * NB + ni) * 16 + r16;
                    st[ni] = a.row_stats ? (a.row_stats + (size_t)cb * a.M)[m < a.M ? m : a.M - 1] : make_float2(0.f, a.alpha);
                }
                if (lane < MI * 8) {
                    if (c >= a.N) { bv = make_float4(0.f, 0.f, 0.f, 0.f); sv = bv; }
                    *reinterpret_cast<float4*>(vec + lane * 4) = bv;
                    *reinterpret_cast<float4*>(vec + MI * 32 + lane * 4) = sv;
                }
#pragma unroll
                for (int ni = 0; ni < NB; ni++) { rscale[ni] = st[ni].y; rinv[ni] = __builtin_amdgcn_rcpf(st[ni].y); rmean[ni] = st[ni].x; }
            } else {
#pragma unroll
                for (int ni = 0; ni < NB; ni++) { rscale[ni] = a.alpha; rinv[ni] = 0.f; rmean[ni] = 0.f; }
            }
#pragma unroll
            for (int mi = 0; mi < MB; mi++) {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = bv;
                if (vecs) {
                    bv = *reinterpret_cast<const float4*>(vec + mi * 16 + 4 * g4);
                    sv = *reinterpret_cast<const float4*>(vec + MI * 32 + mi * 16 + 4 * g4);
                }
                const float bq[4] = { bv.x, bv.y, bv.z, bv.w }, sq[4] = { sv.x, sv.y, sv.z, sv.w };
#pragma unroll
                for (int ni = 0; ni < NB; ni++)
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[mi][ni][e] = fmaf(bq[e], rinv[ni], -rmean[ni] * sq[e]);
            }
        }
        GVD_STAMP(1);

        static_assert(NB == 4 && GVD_GEMM_RDAHEAD >= 2 && GVD_GEMM_RDAHEAD <= 4, "the scheduling groups below are literals");
        if constexpr (RING) {
            // ---- half-tiles 0, 1 of this tile landed (issued before the previous epilogue; the wait also drains that epilogue's stores) ----
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#if !(defined(GVD_GEMM_TRACE) && GVD_GEMM_TRACE == 2)
            GVD_STAMP(2);
#endif
            // fragments that open half-tile 0
            vec8 bfc[NB], afc[GVD_GEMM_RDAHEAD];
#pragma unroll
            for (int ni = 0; ni < NB; ni++) bfc[ni] = *reinterpret_cast<const vec8*>(lds + b_off[ni]);
#pragma unroll
            for (int d = 0; d < GVD_GEMM_RDAHEAD; d++) afc[d] = *reinterpret_cast<const vec8*>(lds + a_off + d * 16 * ROWB);
            // two half-tiles per trip (K % 64 == 0: the launcher's condition for this form): the fragments read across a barrier change
            // hands between two register sets instead of being copied at the top of every half-tile
            vec8 bfd[NB], afd[GVD_GEMM_RDAHEAD];
            auto half_tile = [&](int h, vec8 (&bin)[NB], vec8 (&ain)[GVD_GEMM_RDAHEAD], vec8 (&bout)[NB], vec8 (&aout)[GVD_GEMM_RDAHEAD]) {
                const unsigned char* cur = lds + (h & 3) * STAGE;
                const unsigned char* nxt = lds + ((h + 1) & 3) * STAGE;
                const unsigned dma_on = pos(nk - 3 - h);                     // half-tile h + 3 exists
                vec8 af[MB];
#pragma unroll
                for (int d = 0; d < GVD_GEMM_RDAHEAD; d++) af[d] = ain[d];
#pragma unroll
                for (int mi = 0; mi < MB; mi++) {
                    // operand fragments are read GVD_GEMM_RDAHEAD steps (of 4 MFMAs) ahead of their use; the last steps read what opens h + 1
                    if (mi + GVD_GEMM_RDAHEAD < MB) af[mi + GVD_GEMM_RDAHEAD] = *reinterpret_cast<const vec8*>(cur + a_off + (mi + GVD_GEMM_RDAHEAD) * 16 * ROWB);
                    else {
                        const int d = mi + GVD_GEMM_RDAHEAD - MB;      // 0 .. RDAHEAD - 1
                        if (d == 0) { bout[0] = *reinterpret_cast<const vec8*>(nxt + b_off[0]); bout[1] = *reinterpret_cast<const vec8*>(nxt + b_off[1]); }
                        if (d == GVD_GEMM_RDAHEAD - 1) { bout[2] = *reinterpret_cast<const vec8*>(nxt + b_off[2]); bout[3] = *reinterpret_cast<const vec8*>(nxt + b_off[3]); }
                        aout[d] = *reinterpret_cast<const vec8*>(nxt + a_off + d * 16 * ROWB);
                    }
                    if ((mi & 1) && mi / 2 < NP) ring_piece(mi / 2, h + 3, dma_on);
#pragma unroll
                    for (int ni = 0; ni < NB; ni++) acc[mi][ni] = Tr<T>::mfma16(af[mi], bin[ni], acc[mi][ni]);
                }
                // the same order for the machine scheduler (which otherwise sinks every read to just in front of its first use)
#pragma unroll
                for (int mi = 0; mi < MB; mi++) {
                    if (mi + GVD_GEMM_RDAHEAD < MB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else {
                        const int d = mi + GVD_GEMM_RDAHEAD - MB;
                        if (d == 0 || d == GVD_GEMM_RDAHEAD - 1) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
#if defined(GVD_GEMM_TRACE) && GVD_GEMM_TRACE == 2   // half-tile 5: 2 before the DMA wait, 3 after it, 4 after the barrier; 5: the same point one half-tile later
                if (h == 5) GVD_STAMP(2);
#endif
                // my pieces of h + 2 have landed when at most the NP pieces of h + 3 are still in flight
                if (dma_on) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(GVD_GEMM_TRACE) && GVD_GEMM_TRACE == 2
                if (h == 5) GVD_STAMP(3);
#endif
                // (my reads of slot h have been consumed by MFMAs above: complete; the reads that opened h + 1 may cross the barrier --
                //  their slot is refilled only after the NEXT one)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#if defined(GVD_GEMM_TRACE) && GVD_GEMM_TRACE == 2
                if (h == 5) GVD_STAMP(4);
                if (h == 6) GVD_STAMP(5);
#else
                if (h == 0) GVD_STAMP(3);
#endif
            };
            for (int h = 0; h < nk; h += 2) {
                half_tile(h, bfc, afc, bfd, afd);
                half_tile(h + 1, bfd, afd, bfc, afc);
            }
#if !(defined(GVD_GEMM_TRACE) && GVD_GEMM_TRACE == 2)
            GVD_STAMP(4);
            GVD_STAMP(5);
#endif
        } else {
            for (int kt = 0; kt < nk; kt++) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                 // tile kt landed for everyone; stage (kt + 1) & 1 is free
                if (kt == 0) GVD_STAMP(2);
                if (kt == 1) GVD_STAMP(3);
                if (kt + 1 < nk_full) issue(kt + 1, (kt + 1) & 1);
                else if (kt + 1 < nk) issue_tail(kt + 1, (kt + 1) & 1);
                const unsigned char* st = lds + (kt & 1) * STAGE;
                vec8 af[MB], bf[NB];
#pragma unroll
                for (int ni = 0; ni < NB; ni++) bf[ni] = *reinterpret_cast<const vec8*>(st + b_off[ni]);
#pragma unroll
                for (int d = 0; d < GVD_GEMM_RDAHEAD; d++) af[d] = *reinterpret_cast<const vec8*>(st + a_off + d * 16 * ROWB);
#pragma unroll
                for (int mi = 0; mi < MB; mi++) {
                    if (mi + GVD_GEMM_RDAHEAD < MB) af[mi + GVD_GEMM_RDAHEAD] = *reinterpret_cast<const vec8*>(st + a_off + (mi + GVD_GEMM_RDAHEAD) * 16 * ROWB);
#pragma unroll
                    for (int ni = 0; ni < NB; ni++) acc[mi][ni] = Tr<T>::mfma16(af[mi], bf[ni], acc[mi][ni]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + GVD_GEMM_RDAHEAD, 0);
#pragma unroll
                for (int mi = 0; mi < MB; mi++) {
                    if (mi + GVD_GEMM_RDAHEAD < MB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
            GVD_STAMP(4);
            __syncthreads();   // all operand reads retired: both stages are free
            GVD_STAMP(5);
        }

        // ---- the next tile's opening K-tile(s) go out now: their latency hides under this tile's epilogue ----
        const int tb = cb, tm0 = cm0, tn0 = cn0;
        slot = next_valid(slot + gridDim.x);
        more = slot < total;
        if (more) {
            decode(slot, cb, cm0, cn0);
            aim(cb, cm0, cn0);
            issue_first();
        }

        // ---- epilogue.  A lane holds, per 16 x 16 block (mi, ni), the 4 consecutive channels 16 mi + 4 g4 + e of ONE token (16 ni + r16):
        //      scale, LayerNorm fold, bias (and the GEGLU gate) in fp32 registers, 16-bit rounding, then a WAVE-PRIVATE transposition
        //      through LDS (8-byte writes of a lane's four channels, 16 tokens per pass, no barriers) so that the global stores are
        //      row-contiguous: 20 (16) consecutive lanes cover the wave's 320 (256) bytes of one token row.  (Row-per-lane stores straight
        //      from the registers ran at a fraction of the write bandwidth: the store path handles a row segment per cycle.) ----
        GVD_STAMP(6);
        T* __restrict__ yb = (T*)a.y + (size_t)tb * a.sy;
        const T* __restrict__ rb = a.res ? (const T*)a.res + (size_t)tb * a.sr : nullptr;
        unsigned char* const ep = lds + EP_BASE + wave * EP_WAVE;
        constexpr int NOCT = GEGLU ? MI * 2 : MI * 4;                        // 16-byte chunks per token row of this wave
        constexpr int NIT = (NOCT * 16 + 63) / 64;                           // read-back sweeps of a 16-token block (the last may be partial)
        const int wcol = GEGLU ? (tn0 >> 1) + wm * MI * 16 : tn0 + wm * MI * 32, ncols = GEGLU ? (a.N >> 1) : a.N;
#pragma unroll
        for (int ni = 0; ni < NB; ni++) {
            const int mrow = tm0 + (wn * NB + ni) * 16;
            unsigned char* const row = ep + r16 * EP_PITCH + g4 * 8;         // this lane's token row, its 4-channel column
            if (GEGLU) {
                // W rows come as [16 values | 16 gates] of 16 consecutive outputs: value block 2 j, gate block 2 j + 1 -- this lane
                // holds value and gate of the 4 outputs 16 j + 4 g4 + e
#pragma unroll
                for (int j = 0; j < MI; j++) {
                    f4v v = acc[2 * j][ni], gt = acc[2 * j + 1][ni];
                    if (scaled) { v *= rscale[ni]; gt *= rscale[ni]; }
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = v[e] * gelu_erf(gt[e]);   // fp32 through the gate: one rounding, at the store
                    *reinterpret_cast<uint2*>(row + j * 32) = pack4<T>(o[0], o[1], o[2], o[3]);
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < MB; mi++) {
                    f4v v = acc[mi][ni];
                    if (scaled) v *= rscale[ni];
                    *reinterpret_cast<uint2*>(row + mi * 32) = pack4<T>(v[0], v[1], v[2], v[3]);
                }
            }
            // read back row-contiguous: chunk c of the wave's [16 tokens][NOCT chunks] block -> token c / NOCT, chunk c % NOCT.
            // The residual pieces of a block are fetched together first (unconditional, clamped addresses): their latency is exposed once.
            uint4 rr[NIT], rg[GM == 3 ? NIT : 1];
            if (GM == 3) {
                // gate backward: the tile is d/d(gated output) for outputs col .. col + 7; value and gate of those outputs sit in the saved
                // pre-activation's [16 value | 16 gate] blocks at (col >> 4) * 32 + (col & 8) and + 16
                const T* __restrict__ hb = (const T*)a.aux + (size_t)tb * a.saux;
#pragma unroll
                for (int i = 0; i < NIT; i++) {
                    const int c = i * 64 + lane, tok = c / NOCT, oc = c - tok * NOCT;
                    int m = mrow + (tok < 16 ? tok : 15), col = wcol + oc * 8;
                    m = m < a.M ? m : a.M - 1;
                    col = col < ncols ? col : 0;
                    const T* hp = hb + (size_t)m * a.ldaux + ((col >> 4) * 32 + (col & 8));
                    rr[i] = *reinterpret_cast<const uint4*>(hp);
                    rg[i] = *reinterpret_cast<const uint4*>(hp + 16);
                }
            } else if (rb) {
#pragma unroll
                for (int i = 0; i < NIT; i++) {
                    const int c = i * 64 + lane, tok = c / NOCT, oc = c - tok * NOCT;
                    int m = mrow + (tok < 16 ? tok : 15), col = wcol + oc * 8;
                    m = m < a.M ? m : a.M - 1;
                    col = col < ncols ? col : 0;
                    rr[i] = *reinterpret_cast<const uint4*>(rb + (size_t)m * a.ldr + col);
                }
            }
#pragma unroll
            for (int i = 0; i < NIT; i++) {
                const int c = i * 64 + lane, tok = c / NOCT, oc = c - tok * NOCT;
                const bool live = (NOCT * 16) % 64 == 0 || tok < 16;
                uint4 w = *reinterpret_cast<const uint4*>(ep + (live ? tok : 0) * EP_PITCH + oc * 16);
                const int m = mrow + tok, col = wcol + oc * 8;
                if (GM == 3) {
                    // attention.py:415-423 backward, as k_geglu_bwd evaluates it on the ROUNDED operands: d value = dy gelu(g), d gate = dy value gelu'(g)
                    const vec8 dyv = __builtin_bit_cast(vec8, w), av = __builtin_bit_cast(vec8, rr[i]), gv = __builtin_bit_cast(vec8, rg[i]);
                    vec8 ra, rgt;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float gf = (float)gv[k], df = (float)dyv[k];
                        float cdf, e;
                        gelu_cdf_exp(gf, cdf, e);
                        const float pdf = 0.3989422804014327f * e;
                        ra[k] = (T)(df * gf * cdf);
                        rgt[k] = (T)(df * (float)av[k] * fmaf(gf, pdf, cdf));
                    }
                    if (live && m < a.M && col < ncols) {
                        T* dp = yb + (size_t)m * a.ldy + ((col >> 4) * 32 + (col & 8));
                        *reinterpret_cast<vec8*>(dp) = ra;
                        *reinterpret_cast<vec8*>(dp + 16) = rgt;
                    }
                    continue;
                }
                if (rb) {
                    const uint4 r = rr[i];
                    const unsigned wi[4] = { w.x, w.y, w.z, w.w }, ri[4] = { r.x, r.y, r.z, r.w };
                    unsigned oo[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {   // the 16-bit sum of the ROUNDED product and the residual, as the separate ops
                        const T2 x2 = __builtin_bit_cast(T2, wi[q]), r2 = __builtin_bit_cast(T2, ri[q]);
                        oo[q] = Tr<T>::pack2((float)x2[0] + (float)r2[0], (float)x2[1] + (float)r2[1]);
                    }
                    w = make_uint4(oo[0], oo[1], oo[2], oo[3]);
                }
                if (live && m < a.M && col < ncols) {
                    T* dst = GM == 2 ? (T*)a.aux + (size_t)tb * a.saux + (size_t)m * a.ldaux + col : yb + (size_t)m * a.ldy + col;   // (mode 2: the pre-activation)
#if GVD_GEMM_NT_STORE
                    typedef unsigned u4v __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(u4v{ w.x, w.y, w.z, w.w }, reinterpret_cast<u4v*>(dst));
#else
                    *reinterpret_cast<uint4*>(dst) = w;
#endif
                }
            }
            if (GM == 2) {
                // the gate on the staged, ROUNDED pre-activation (what k_geglu reads back from memory on the unfused path: bit-identical):
                // output chunk oc2 = 8 outputs 16 j + 8 half .. of this wave: values at bytes 64 j + 16 half of the token row, gates 32 further
                constexpr int NOCT2 = MI * 2, NIT2 = (NOCT2 * 16 + 63) / 64;
                const int wcol2 = (tn0 >> 1) + wm * MI * 16, ncols2 = a.N >> 1;
#pragma unroll
                for (int i = 0; i < NIT2; i++) {
                    const int c = i * 64 + lane, tok = c / NOCT2, oc = c - tok * NOCT2;
                    const bool live = (NOCT2 * 16) % 64 == 0 || tok < 16;
                    const unsigned char* sp = ep + (live ? tok : 0) * EP_PITCH + (oc >> 1) * 64 + (oc & 1) * 16;
                    const vec8 av = *reinterpret_cast<const vec8*>(sp), gv = *reinterpret_cast<const vec8*>(sp + 32);
                    vec8 r;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float gf = (float)gv[k];
                        float cdf, e;
                        gelu_cdf_exp(gf, cdf, e);
                        const T ge = (T)(gf * cdf);               // torch: gelu in fp32, rounded, then the product (two separate ops)
                        r[k] = (T)((float)av[k] * (float)ge);
                    }
                    const int m = mrow + tok, col = wcol2 + oc * 8;
                    if (live && m < a.M && col < ncols2) *reinterpret_cast<vec8*>(yb + (size_t)m * a.ldy + col) = r;
                }
            }
        }
        GVD_STAMP(7);
#ifdef GVD_GEMM_TRACE
        trace_n += 8;
#endif
        if (!more) break;
    }
}

// ---- skinny problems: M <= 256 rows (the cross-attention context projections -- 77 text / 256 image tokens x 1024 -> 640 .. 2560 --
// the per-frame embedding Linears, the time-embedding MLP: ~115 launches per DDIM step) ----
// One persistent tile of the kernel above walks K in 32-channel steps behind a barrier each: 21-25 us for a 77 x 2560 x 1024 product
// whose 5 MB of weights are 1.3 us of HBM time.  Here the N x K weight matrix is cut into (32 channels) x (a quarter of K) pieces, one
// WAVE each: the wave loads its W piece and the matching X columns straight into MFMA operand registers (no LDS, no barrier in the
// loop -- every load of the piece is in flight at once), multiplies, and the four K-quarters of a channel block meet in LDS.
template <typename T, int TB>   // TB: 32-row blocks of X per workgroup (1, 2, 4)
__global__ void __launch_bounds__(256) k_gemm_skinny(const GemmArgs a)
{
    typedef typename Tr<T>::vec8 vec8;
    __shared__ float red[4][TB][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, r32 = lane & 31;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * (TB * 32);   // (more than 128 rows: two workgroups of 4 row blocks each)
    const int ksl = ((a.K + 63) / 64) * 16;                    // K-slice of a wave: a multiple of 16
    const int k0 = wave * ksl, k1 = k0 + ksl < a.K ? k0 + ksl : a.K;
    const int ch = n0 + r32 < a.N ? n0 + r32 : a.N - 1;
    const T* wr = (const T*)a.w + (size_t)ch * a.ldw + 8 * hi;
    const T* xr[TB];
#pragma unroll
    for (int tb = 0; tb < TB; tb++) {
        const int m = m0 + tb * 32 + r32;
        xr[tb] = (const T*)a.x + (size_t)(m < a.M ? m : a.M - 1) * a.ldx + 8 * hi;
    }
    f16v acc[TB];
#pragma unroll
    for (int tb = 0; tb < TB; tb++) acc[tb] = f16v{};
    constexpr int UN = TB <= 2 ? 16 : 8;                       // K-steps whose loads are in flight together (registers: UN x (1 + TB) fragments)
#pragma unroll UN
    for (int k = k0; k < k1; k += 16) {
        const bool ok = k + 8 * hi < a.K;                      // (K % 8 == 0: a lane's 8 channels are inside or outside together)
        const int kk = ok ? k : -8 * hi;                       // (dummy read: the row's first 8 channels, always inside the matrix)
        vec8 af = *reinterpret_cast<const vec8*>(wr + kk);
        if (!ok) af = vec8{};
#pragma unroll
        for (int tb = 0; tb < TB; tb++) {
            vec8 bf = *reinterpret_cast<const vec8*>(xr[tb] + kk);
            if (!ok) bf = vec8{};                               // (BOTH operands: the dummy read may hold inf / NaN bits, and 0 x NaN = NaN --
            acc[tb] = Tr<T>::mfma(af, bf, acc[tb]);            //  with K = 8 the upper half-wave reads past the row: found by the guided schedule test)
        }
    }
#pragma unroll
    for (int tb = 0; tb < TB; tb++)
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave][tb][r][lane] = acc[tb][r];
    __syncthreads();
    // thread -> (token block, channel quad g, lane): channels n0 + 8 g + 4 hi + {0..3} of token 32 tb + r32
    const int g = wave;
#pragma unroll
    for (int tb = 0; tb < TB; tb++) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int r = 4 * g + e;
            v[e] = (red[0][tb][r][lane] + red[1][tb][r][lane]) + (red[2][tb][r][lane] + red[3][tb][r][lane]);
        }
        const int c0 = n0 + 8 * g + 4 * hi, m = m0 + tb * 32 + r32;
        if (m < a.M && c0 < a.N) {                             // (N % 8 == 0 and c0 % 4 == 0: the quad is inside or outside)
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = v[e] * a.alpha + (a.bias ? a.bias[c0 + e] : 0.f);
            *reinterpret_cast<uint2*>((T*)a.y + (size_t)m * a.ldy + c0) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
    }
}

// (mean, rstd) of every row of x [M, C] (16-bit), LayerNorm's biased variance, the row held in registers (mean first, then the
// centred sum of squares).  G lanes share a row (G = 8 for C <= 512 ... 64 for C <= 4096; every lane owns up to 8 16-byte chunks,
// chunk o = sub + G i), so a wave covers 64 / G rows: at C = 320 one wave per row left 24 of 64 lanes idle and paid two 6-step
// reductions for 640 bytes (64 us for the 230 400 rows of a level-0 activation; this form moves the same bytes in ~35).
template <typename T, int G>
__global__ void __launch_bounds__(256) k_row_stats(const T* __restrict__ x, float2* __restrict__ out, long long M, int C, long long ldx, float eps)
{
    typedef typename Tr<T>::vec8 vec8;
    constexpr int RPW = 64 / G;                               // rows per wave
    const int lane = threadIdx.x & 63, sub = lane & (G - 1), oct = C >> 3;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + (lane / G);
    const bool live = row < M;
    const T* xr = x + (live ? row : 0) * ldx;
    constexpr int MAXV = 8;                                  // C <= G * 8 * 8
    vec8 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        const int o = sub + G * i;
        v[i] = vec8{};
        if (o < oct) {
            v[i] = *reinterpret_cast<const vec8*>(xr + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) s += (float)v[i][k];
        }
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        if (sub + G * i < oct) {
#pragma unroll
            for (int k = 0; k < 8; k++) { const float d = (float)v[i][k] - mean; q = fmaf(d, d, q); }
        }
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) q += __shfl_xor(q, d, 64);
    if (live && sub == 0) out[row] = make_float2(mean, rsqrtf(q / (float)C + eps));
}

// ---- row kernels of the wide-head attention (ae_modules.py:26-78 as chunked GEMMs, lvdm_amd/wide_attention.py) ----
// In-place softmax over the rows of s [rows, N] (16-bit, row stride ld), fp32 math, one wave per row with the row in registers;
// lse[row] = max + log(sum) (natural log) for the backward.
template <typename T, int MAXV>
__global__ void __launch_bounds__(256) k_softmax_rows(T* __restrict__ s, long long ld, long long rows, int N, int n_valid, float* __restrict__ lse)
{
    typedef typename Tr<T>::vec8 vec8;
    const int lane = threadIdx.x & 63, oct = N >> 3;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    T* sr = s + row * ld;
    vec8 v[MAXV];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        const int o = lane + 64 * i;
        if (o < oct) {
            v[i] = *reinterpret_cast<const vec8*>(sr + o * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) if (o * 8 + k < n_valid) mx = fmaxf(mx, (float)v[i][k]);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float sum = 0.f;
    float e[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        if (lane + 64 * i < oct) {
#pragma unroll
            for (int k = 0; k < 8; k++) {   // (columns >= n_valid are padding keys: probability 0)
                e[i][k] = (lane + 64 * i) * 8 + k < n_valid ? __builtin_amdgcn_exp2f(((float)v[i][k] - mx) * 1.4426950408889634f) : 0.f;
                sum += e[i][k];
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        const int o = lane + 64 * i;
        if (o < oct) {
            vec8 p;
#pragma unroll
            for (int k = 0; k < 8; k++) p[k] = (T)(e[i][k] * inv);
            *reinterpret_cast<vec8*>(sr + o * 8) = p;
        }
    }
    if (lse && lane == 0) lse[row] = mx + __logf(sum);
}

// Backward element step, in place: s holds the scaled scores and becomes P = exp(s - lse); dp holds dO V^T and becomes
// dS' = P (dp - delta).  lse / delta are per ROW (by_col = 0: index row) or per COLUMN (by_col = 1: the transposed pass; index
// (row / rows_per_batch) * N + column).
template <typename T>
__global__ void __launch_bounds__(256) k_attn_ds(T* __restrict__ s, T* __restrict__ dp, const float* __restrict__ lse,
                                                 const float* __restrict__ delta, long long rows, int N, int n_valid, long long rows_per_batch, int by_col)
{
    typedef typename Tr<T>::vec8 vec8;
    const int oct = N >> 3;
    const long long total = rows * oct, step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
        const long long r = i / oct;
        const int o = (int)(i - r * oct);
        vec8 sv = *reinterpret_cast<const vec8*>(s + r * N + o * 8), dv = *reinterpret_cast<const vec8*>(dp + r * N + o * 8);
        float l[8], d[8];
        if (by_col) {
            const long long base = (r / rows_per_batch) * N + o * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) { l[k] = lse[base + k]; d[k] = delta[base + k]; }
        } else {
            const float lr = lse[r], dr = delta[r];
#pragma unroll
            for (int k = 0; k < 8; k++) { l[k] = lr; d[k] = dr; }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float p = o * 8 + k < n_valid ? __builtin_amdgcn_exp2f(((float)sv[k] - l[k]) * 1.4426950408889634f) : 0.f;
            const T pt = (T)p;
            sv[k] = pt;
            dv[k] = (T)((float)pt * ((float)dv[k] - d[k]));
        }
        *reinterpret_cast<vec8*>(s + r * N + o * 8) = sv;
        *reinterpret_cast<vec8*>(dp + r * N + o * 8) = dv;
    }
}

template <typename T, int MI, int WM, int BK, int GM>
hipError_t launch_gemm(const GemmArgs& a, hipStream_t stream)
{
    constexpr int BN = WM * MI * 32;
    constexpr int stage = (BN + BM) * BK * 2, epw = 16 * (MI * 64 + 16);
    // 8-wave (ring) form: four K slots, the epilogue staging from slot 3 on, the initialisation vectors behind it; 4-wave form: two K
    // stages, the staging (and the vectors) overlaying the second
    constexpr int smem = WM == 2 ? 3 * stage + WM * WN * epw + WM * WN * MI * 64 * 4 : stage + (stage > WM * WN * epw ? stage : WM * WN * epw);
    static_assert(WM != 2 || 3 * stage + WM * WN * epw >= 4 * stage, "the staging area covers slot 3");
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = k_gemm_nt<T, MI, WM, BK, GM>;
    static bool attr_done[64] = {};   // (per instantiation)
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 64 && !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        attr_done[dev] = true;
    }
    const long long slots = (long long)a.mgroups * 8 * a.tiles_n * a.batch;          // (a multiple of 8)
    const int resident = (dev < 64 && cus[dev] ? cus[dev] : 256) / 8 * 8 * (WM == 1 ? 2 : 1);   // persistent: every CU full
    const unsigned grid = (unsigned)(slots < resident ? slots : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, stream, a);
    return hipGetLastError();
}

// Workgroup form: 1 = 8 waves, 320 / 256-channel tiles, K-step 64, one per CU (the faster main loop); 0 = 4 waves, 160 / 128-channel
// tiles, K-step 32, two per CU -- taken when the 8-wave form would leave CUs without a tile (small M x N: the 9 x 16 level, the
// context projections, the wide attention's P V product).  GVD_GEMM_VARIANT = 0 / 1 forces one form (A/B runs).
int gemm_variant(long long M, int N, int batch, int K = 32)
{
    if (K & 63) return 0;   // (the ring form reads whole 32-channel half-tiles, two per trip of its loop)
    static const int forced = [] { const char* e = getenv("GVD_GEMM_VARIANT"); return e ? atoi(e) : -1; }();
    if (forced == 0 || forced == 1) return forced;
    const int w5 = (N + 319) / 320 * 320, w4 = (N + 255) / 256 * 256;
    const long long tiles = (M + BM - 1) / BM * ((w5 <= w4 ? w5 / 320 : w4 / 256)) * batch;
    return tiles >= 192 ? 1 : 0;
}

}  // namespace

extern "C" {

static int tile_n(long long M, int N, int batch, int K = 32)
{
    // 5-block (160 / 320-channel) or 4-block (128 / 256) tiles: the one whose launch is shorter under the simple model
    // rounds-of-resident-workgroups x tile width (padding of N and a mostly empty last round both cost); ties go to the wider tile
    static const int forced = [] { const char* e = getenv("GVD_GEMM_BLOCKS"); return e ? atoi(e) : 0; }();
    const int v = gemm_variant(M, N, batch, K), big = v == 1 ? 2 : 1, resident = v == 1 ? 256 : 512;
    const int w5 = 160 * big, w4 = 128 * big;
    if (forced == 4 || forced == 5) return forced == 5 ? w5 : w4;
    const long long mt = (M + BM - 1) / BM * batch;
    auto cost = [&](int w) { const long long tiles = mt * ((N + w - 1) / w); return (tiles + resident - 1) / resident * w; };
    return cost(w5) <= cost(w4) ? w5 : w4;
}

int gvd_gemm_tile_n(int M, int N, int batch) { return tile_n(M, N, batch); }

int gvd_gemm_geglu_layout(void) { return 1; }   // 1: [16 values | 16 gates] in natural order (16 x 16 x 32 MFMA blocks); see gvd_diffusion.h

static int gemm_nt_impl(const void* x, long long ldx, long long stride_x, const void* w, long long ldw, long long stride_w,
                        void* y, long long ldy, long long stride_y, int M, int N, int K, int batch, float alpha, const float* bias,
                        const float* row_stats, const float* col_sum, const void* residual, long long ldr, long long stride_r,
                        int geglu, const void* aux, long long ldaux, long long stride_aux, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (geglu < 0 || geglu > 3) return fail(-1, "gvd_gemm_nt: gate mode is 0 .. 3");
    if (geglu >= 2 && (!aux || (ldaux & 7) || (stride_aux & 7) || ((uintptr_t)aux & 15) || residual))
        return fail(-1, "gvd_gemm_nt_gate: modes 2 / 3 need the pre-activation tensor (aligned, strides multiples of 8) and take no residual");
    if (geglu == 3 && (N & 15)) return fail(-1, "gvd_gemm_nt_gate: mode 3 needs N % 16 == 0");
    if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return fail(-1, "gvd_gemm_nt: bad arguments");
    if ((K & 7) || (ldx & 7) || (ldw & 7) || (ldy & 7) || (stride_x & 7) || (stride_w & 7) || (stride_y & 7) ||
        (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)residual) & 15))
        return fail(-1, "gvd_gemm_nt: K, row and batch strides must be multiples of 8 elements and tensors 16-byte aligned");
    if ((geglu == 1 || geglu == 2) ? (N & 31) : (N & 7)) return fail(-1, "gvd_gemm_nt: N must be a multiple of 8 (32 with the GEGLU epilogue)");
    if ((row_stats == nullptr) != (col_sum == nullptr)) return fail(-1, "gvd_gemm_nt: the LayerNorm fold needs row_stats AND col_sum");
    if (residual && ((ldr & 7) || (stride_r & 7))) return fail(-1, "gvd_gemm_nt: residual strides must be multiples of 8");
    GemmArgs a{};
    a.x = x; a.w = w; a.y = y; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.sx = stride_x; a.sw = stride_w; a.sy = stride_y;
    a.M = M; a.N = N; a.K = K; a.batch = batch; a.alpha = alpha; a.bias = bias;
    a.row_stats = reinterpret_cast<const float2*>(row_stats); a.col_sum = col_sum;
    a.res = residual; a.ldr = ldr; a.sr = stride_r; a.geglu = geglu;
    a.aux = aux; a.ldaux = ldaux; a.saux = stride_aux;
    static const bool no_skinny = [] { const char* e = getenv("GVD_GEMM_NO_SKINNY"); return e && e[0] != '0'; }();   // (A/B switch)
    if (M <= 256 && batch == 1 && !row_stats && !residual && !geglu && !no_skinny) {   // (geglu: any gate mode)
        // skinny: a wave per (32 channels, quarter of K); y = alpha x w^T + bias
        const int tb = (M + 31) / 32;
        const dim3 grid((unsigned)((N + 31) / 32), tb > 4 ? 2u : 1u);
#define GVD_SK(T, TB_) hipLaunchKernelGGL((k_gemm_skinny<T, TB_>), grid, dim3(256), 0, stream, a)
#define GVD_SK_T(T) do { if (tb <= 1) GVD_SK(T, 1); else if (tb <= 2) GVD_SK(T, 2); else GVD_SK(T, 4); } while (0)
        if (is_bf16) GVD_SK_T(__bf16); else GVD_SK_T(_Float16);
#undef GVD_SK_T
#undef GVD_SK
        const hipError_t es = hipGetLastError();
        if (es != hipSuccess) return fail(-2, "launch k_gemm_skinny", es);
        return 0;
    }
    const int bn = tile_n(M, N, batch, K);
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + bn - 1) / bn; a.mgroups = (a.tiles_m + 7) / 8;
    {   // channel tiles whose W rows (bn x K x 2 bytes each) share an XCD's L2 with the token panels in flight
        static const long long budget = [] { const char* e = getenv("GVD_GEMM_L2_BYTES"); return e ? atoll(e) : (3LL << 19); }();
        long long g = budget / ((long long)bn * K * 2);
        // (a lone last tile as its own group re-reads every token panel for it: 7 + 1 of the level-0 feed-forward's 8 tiles.  Up to a
        //  third over the budget, one group: 230 400 x 2560 x 320 0.614 -> 0.588 ms, profiles/r04_gemm_l2_budget.txt)
        if ((long long)a.tiles_n * bn * K * 2 * 3 <= budget * 4) g = a.tiles_n;
        a.ngroup = (int)(g < 1 ? 1 : (g > a.tiles_n ? a.tiles_n : g));
    }
    if ((long long)a.mgroups * 8 * a.tiles_n * batch >= (1LL << 31)) return fail(-1, "gvd_gemm_nt: grid too large");
    hipError_t e;
#define GVD_GEMM_GM(T_, MI_, WM_, BK_)                                                                                              \
    (geglu == 0 ? launch_gemm<T_, MI_, WM_, BK_, 0>(a, stream) : geglu == 1 ? launch_gemm<T_, MI_, WM_, BK_, 1>(a, stream)              \
     : geglu == 2 ? launch_gemm<T_, MI_, WM_, BK_, 2>(a, stream) : launch_gemm<T_, MI_, WM_, BK_, 3>(a, stream))
#define GVD_GEMM_GO(MI_, WM_, BK_) (is_bf16 ? GVD_GEMM_GM(__bf16, MI_, WM_, BK_) : GVD_GEMM_GM(_Float16, MI_, WM_, BK_))
    switch (bn) {
    case 320: e = GVD_GEMM_GO(5, 2, 32); break;
    case 256: e = GVD_GEMM_GO(4, 2, 32); break;
    case 160: e = GVD_GEMM_GO(5, 1, 32); break;
    default: e = GVD_GEMM_GO(4, 1, 32); break;
    }
#undef GVD_GEMM_GO
#undef GVD_GEMM_GM
    if (e != hipSuccess) return fail(-2, "launch k_gemm_nt", e);
    return 0;
}

int gvd_gemm_nt(const void* x, long long ldx, long long stride_x, const void* w, long long ldw, long long stride_w,
                void* y, long long ldy, long long stride_y, int M, int N, int K, int batch, float alpha, const float* bias,
                const float* row_stats, const float* col_sum, const void* residual, long long ldr, long long stride_r,
                int geglu, int is_bf16, void* stream)
{
    return gemm_nt_impl(x, ldx, stride_x, w, ldw, stride_w, y, ldy, stride_y, M, N, K, batch, alpha, bias, row_stats, col_sum, residual, ldr, stride_r,
                        geglu ? 1 : 0, nullptr, 0, 0, is_bf16, stream);
}

int gvd_gemm_nt_gate(const void* x, long long ldx, long long stride_x, const void* w, long long ldw, long long stride_w,
                     void* y, long long ldy, long long stride_y, int M, int N, int K, int batch, float alpha, const float* bias,
                     const float* row_stats, const float* col_sum, int mode, void* aux, long long ldaux, long long stride_aux,
                     int is_bf16, void* stream)
{
    if (mode != 2 && mode != 3) return fail(-1, "gvd_gemm_nt_gate: mode is 2 (gate + pre-activation) or 3 (gate backward)");
    return gemm_nt_impl(x, ldx, stride_x, w, ldw, stride_w, y, ldy, stride_y, M, N, K, batch, alpha, bias, row_stats, col_sum, nullptr, 0, 0,
                        mode, aux, ldaux, stride_aux, is_bf16, stream);
}

int gvd_row_stats(const void* x, long long ldx, float* stats, long long M, int C, float eps, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !stats || M <= 0 || C <= 0 || (C & 7) || C > 4096 || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)stats & 7))
        return fail(-1, "gvd_row_stats: C must be a multiple of 8 and <= 4096, pointers aligned");
    const int G = C <= 512 ? 8 : C <= 1024 ? 16 : C <= 2048 ? 32 : 64;
    const long long rows_per_block = 4 * (64 / G);
    const unsigned blocks = (unsigned)((M + rows_per_block - 1) / rows_per_block);
#define GVD_RS(T, GG) hipLaunchKernelGGL((k_row_stats<T, GG>), dim3(blocks), dim3(256), 0, stream, (const T*)x, reinterpret_cast<float2*>(stats), M, C, ldx, eps)
#define GVD_RS_T(T) do { if (G == 8) GVD_RS(T, 8); else if (G == 16) GVD_RS(T, 16); else if (G == 32) GVD_RS(T, 32); else GVD_RS(T, 64); } while (0)
    if (is_bf16) GVD_RS_T(__bf16); else GVD_RS_T(_Float16);
#undef GVD_RS_T
#undef GVD_RS
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_row_stats", e);
    return 0;
}

int gvd_softmax_rows(void* s, long long ld, long long rows, int N, int n_valid, float* lse, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!s || rows <= 0 || N <= 0 || (N & 7) || (ld & 7) || N > 16384 || n_valid <= 0 || n_valid > N || ((uintptr_t)s & 15))
        return fail(-1, "gvd_softmax_rows: N must be a multiple of 8 and <= 16384, pointer aligned");
    const unsigned blocks = (unsigned)((rows + 3) / 4);
#define GVD_SM(T, V) hipLaunchKernelGGL((k_softmax_rows<T, V>), dim3(blocks), dim3(256), 0, stream, (T*)s, ld, rows, N, n_valid, lse)
    if (is_bf16) { if (N <= 4096) GVD_SM(__bf16, 8); else GVD_SM(__bf16, 32); }
    else { if (N <= 4096) GVD_SM(_Float16, 8); else GVD_SM(_Float16, 32); }
#undef GVD_SM
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_softmax_rows", e);
    return 0;
}

int gvd_attn_ds(void* s, void* dp, const float* lse, const float* delta, long long rows, int N, int n_valid, long long rows_per_batch,
                int by_col, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!s || !dp || !lse || !delta || rows <= 0 || N <= 0 || (N & 7) || rows_per_batch <= 0 || (((uintptr_t)s | (uintptr_t)dp) & 15))
        return fail(-1, "gvd_attn_ds: bad arguments");
    const long long vecs = rows * (N / 8);
    const unsigned blocks = (unsigned)((vecs + 255) / 256 < 16384 ? (vecs + 255) / 256 : 16384);
    if (is_bf16) hipLaunchKernelGGL(k_attn_ds<__bf16>, dim3(blocks), dim3(256), 0, stream, (__bf16*)s, (__bf16*)dp, lse, delta, rows, N, n_valid, rows_per_batch, by_col);
    else hipLaunchKernelGGL(k_attn_ds<_Float16>, dim3(blocks), dim3(256), 0, stream, (_Float16*)s, (_Float16*)dp, lse, delta, rows, N, n_valid, rows_per_batch, by_col);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_attn_ds", e);
    return 0;
}

#ifdef GVD_GEMM_TRACE
int gvd_gemm_trace_read(unsigned long long* host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * (n < 1024 ? n : 1024)) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
