// conv_mfma.hip -- hand-written gfx950 implicit-GEMM convolutions of the ViewCrafter U-Net / VAE on MFMA 32x32x16
// (C-ABI: include/gvd_diffusion.h, gvd_conv_mfma / gvd_conv_config / gvd_group_norm_coef).
//
// Replaces (reference lines):
//   ResBlock         GroupNorm32 -> SiLU -> Conv2d 3x3 (+ emb add, + skip)   lvdm/modules/networks/openaimodel3d.py:152-156,210-236
//   Upsample         nearest x2 -> Conv2d 3x3                                 openaimodel3d.py:51-83
//   TemporalConvBlock 4 x [GroupNorm32 -> SiLU -> Conv3d (3,1,1)] + identity  openaimodel3d.py:239-279
//   VAE ResnetBlock / Upsample / conv_out                                     lvdm/modules/networks/ae_modules.py:151-210,112-128,575-578
//
// Layout: activations token-major  x[n][y][x][Cin]  (16-bit), output  out[n][y][x][Cout];  the temporal (3,1,1) form works
// on  x[t][pixel][C]  (b = 1).  fp32 accumulation, fp32 GroupNorm affine + SiLU.
//
// Design (one workgroup = 256 threads = 4 waves; wave64):
//   * A workgroup owns a PATCH of output pixels (TH x TW of one image, or PB pixels x all T frames) and BN output
//     channels.  K runs over 32-channel chunks of Cin; for every chunk the input patch INCLUDING ITS HALO is staged
//     ONCE into LDS and all taps (9 spatial / 3 temporal) read it through a tap-dependent LDS address shift -- an input
//     element crosses L2->LDS once per chunk instead of once per tap, and the fused GroupNorm(+SiLU) prologue
//     (y = silu(a x + b), fp32, zero padding applied AFTER the activation like the reference's conv padding) costs one
//     evaluation per staged element instead of nine.
//   * Weights are pre-packed on the host into the exact LDS image of a (cout tile, chunk, tap) slab: rows of 32 input
//     channels (64 B) whose four 16-byte slots are XOR-swizzled by ((row >> 2) & 3), so the MFMA operand reads
//     (ds_read_b128, lanes = 32 consecutive rows at one k-slot) are bank-conflict free without padding and staging is a
//     linear 16-byte copy.
//   * MFMA roles: A = weights (rows = output channels), B = activated pixels (columns = pixels).  The C/D layout then gives
//     every lane 4 consecutive output channels of ONE pixel per register quad.
//   * Patch pixel rows are 80 bytes (64 + 16 pad) so that 32 consecutive pixels at one k-slot hit distinct bank groups;
//     TW = 16 tiles pad the patch row pitch to a multiple of 256 B for the same reason.
//   * All staging loads are unconditional (dummy address + select): a predicated load makes hipcc serialise the stage's loads
//     behind vmcnt(0) waits; removing the predicates took the L0-L2 shapes from 700-820 to 790-985 TFLOP/s.
//   * Pipeline: one barrier per (chunk, tap).  Weights of step i+1 and (late in a chunk) the next chunk's patch are
//     fetched into registers before the MFMAs of step i and written to the other LDS buffer after them.  Two workgroups
//     per CU (<= 80 KiB LDS, <= 256 VGPRs) de-synchronise and cover each other's staging.
//   * Epilogue: accumulators go through LDS so that global stores are 16-byte, row-contiguous; bias, the per-frame
//     embedding add (ResBlock `h + emb_out`), the residual and the 16-bit rounding are fused, and the GroupNorm statistics
//     the NEXT norm needs (sum / sum of squares per (sample, group) of the ROUNDED outputs) are reduced in-block and added
//     to fp64 accumulators -- the separate statistics pass over the activation disappears.
#include <stdlib.h>

#include "diffusion_common.h"

#ifndef GVD_CONV_WHOLE
#define GVD_CONV_WHOLE 1    // temporal form: fetch the whole next patch at tap 0, write it at tap 2 (0 = the two-halves schedule, for A/B builds)
#endif
#ifndef GVD_CONV_DBG
#define GVD_CONV_DBG 0   // experiments only (round 3, profiles/r03_conv_lds_hunt.txt; the in-loop bits 1 / 2 / 16 / 32 went with the round-4 loop):
#endif                   // 4 = no epilogue staging writes, 8 = no epilogue staging reads

using namespace gvdd;

namespace {

#ifdef GVD_CONV_TRACE
// experiments (tests/scripts/r4_conv_trace.py): s_memtime stamps of the first 2048 workgroups' wave 0 -- start, first barrier (staging of
// chunk 0 done), end of the K loop, end of the epilogue -- plus the XCC / CU the workgroup ran on
__device__ unsigned long long g_ctrace[2048 * 6];
#define GVD_CSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048) g_ctrace[blockIdx.x * 6 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GVD_CSTAMP(i) do { } while (0)
#endif


#ifndef GVD_CONV_RDAHEAD
#define GVD_CONV_RDAHEAD 2   // A fragments read ahead of their MFMAs, in steps of NI MFMAs (0: the compiler's own order, for A/B builds)
#endif
// sched_group_barrier takes literal sizes: one instantiation per (reads, MFMAs) group
template <int N> __device__ __forceinline__ void sgb_read()
{
    if constexpr (N == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    else if constexpr (N == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    else if constexpr (N == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    else if constexpr (N == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    else if constexpr (N == 5) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
    else if constexpr (N == 6) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
    else if constexpr (N == 7) __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
    else if constexpr (N == 8) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    else static_assert(N == 0, "add a case");
}
template <int N> __device__ __forceinline__ void sgb_mfma()
{
    if constexpr (N == 1) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    else if constexpr (N == 2) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    else if constexpr (N == 4) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    else static_assert(N == 0, "add a case");
}
// the order of the read-ahead block of the K loop below: NI + RD reads, then per (ks, mi) [its read(s)] [NI MFMAs]
template <int MI, int NI, int RD, int KS = 0, int M = 0> __device__ __forceinline__ void conv_sched()
{
    if constexpr (KS == 0 && M == 0) sgb_read<NI + RD>();
    if constexpr (KS < 2) {
        if constexpr (M + RD < MI) sgb_read<1>();
        else if constexpr (KS == 0) sgb_read<(M + RD - MI == 0 ? NI : 0) + 1>();
        sgb_mfma<NI>();
        if constexpr (M + 1 < MI) conv_sched<MI, NI, RD, KS, M + 1>();
        else conv_sched<MI, NI, RD, KS + 1, 0>();
    }
}

struct ConvArgs {
    const void* x;        // input activations
    const void* w;        // packed weights
    const float2* coef;   // per-(n, cin) affine of the fused GroupNorm prologue (nullptr: plain convolution)
    const float* bias;    // [Cout] or nullptr
    const void* add_nc;   // [N][Cout] 16-bit per-sample channel add (nullptr if none; spatial mode only)
    const void* res;      // residual, layout of out (nullptr if none)
    void* out;
    double* stats;        // [R][Nstat][G][2] accumulators (nullptr if not wanted)
    int N, H, W, Cin, Cout;   // output geometry (temporal: N = T frames, H = 1, W = pixels per frame)
    int Hin, Win;             // input geometry of the spatial modes
    int ups, silu;            // ups: 0 none, 1 nearest x2 on the fly, 2 zero-stuffed x2 (input gradient of a stride-2 convolution)
    int pad_lo;               // stride-2 mode: zero rows / columns in front of the image (1: U-Net Downsample, 0: VAE Downsample)
    int tiles_x, tiles_y;
    int nchunks;
    int xcd_map;          // 1: read the grid through xcd_conv_ids (the channel tiles / phases of a pixel tile next to each other on one XCD)
    int cps;              // split-K: input-channel chunks per blockIdx.z slice (== nchunks: no split); slice z writes its partial sums to
    long long split_stride;   // out + z * split_stride (elements); no bias / add / residual / statistics on such launches
    int cpg, G, R;
    int PB;               // temporal: pixels per tile
    int NS;               // temporal: samples in this launch (x / out / res / bx are [NS][N frames][W pixels][C]; coef, statistics and
                          // bcoef are indexed by the sample when coef_per_n / bcoef_per_n are set); tiles never straddle samples
    int coef_per_n;       // 1: coef is [N][Cin], 0: one [Cin] vector for all n (temporal, batch 1)
    // Input-gradient launches whose OUTPUT is the gradient w.r.t. silu(GroupNorm(bx)): `stats` then receives the two sums the
    // GroupNorm backward needs per (sample, group) instead of sum / sum of squares (see the epilogue).
    const void* bx;        // the norm's input, layout of out (nullptr: forward statistics)
    const float2* bcoef;   // the norm's forward affine (a, b) per (n, channel) ([Cout] for all n when !bcoef_per_n)
    const float* bgamma;   // the norm's weight [Cout]
    int bsilu, bcoef_per_n;
};

constexpr int BK = 32;          // input channels per chunk
constexpr int PIX_BYTES = 80;   // one patch pixel: 32 channels (64 B) + 16 B pad
constexpr int PB_MAX = 32;      // temporal: max pixels per tile

template <int MODE, int PIX> struct Geo;
template <int PIX> struct Geo<0, PIX> {   // spatial, 16-pixel-wide tiles
    static constexpr int TW = 16, TH = PIX / 16, PW = TW + 2, PH = TH + 2, PITCH = 1536, NTAPS = 9;
    static constexpr int PATCH_BYTES = PH * PITCH, NPP_MAX = PH * PW * 4;
};
template <int PIX> struct Geo<1, PIX> {   // spatial, 32-pixel-wide tiles
    static constexpr int TW = 32, TH = PIX / 32, PW = TW + 2, PH = TH + 2, PITCH = PW * PIX_BYTES, NTAPS = 9;
    static constexpr int PATCH_BYTES = PH * PITCH, NPP_MAX = PH * PW * 4;
};
template <int PIX> struct Geo<3, PIX> {   // spatial stride 2, 32-pixel-wide output tiles: the patch holds (2 TH + 1) x (2 TW + 1) input pixels
    static constexpr int TW = 32, TH = PIX / 32, PW = 2 * TW + 1, PH = 2 * TH + 1, PITCH = PW * PIX_BYTES, NTAPS = 9;
    static constexpr int PATCH_BYTES = PH * PITCH, NPP_MAX = PH * PW * 4;
};
template <int PIX> struct Geo<4, PIX> {   // nearest x2 upsampling + 3x3 as FOUR 2x2 convolutions of the low-resolution map (one per output
    // phase): 32-pixel-wide tiles of INPUT pixels, the 3x3 halo patch of Geo<1>, 4 taps; blockIdx.z = phase (ay, ax)
    static constexpr int TW = 32, TH = PIX / 32, PW = TW + 2, PH = TH + 2, PITCH = PW * PIX_BYTES, NTAPS = 4;
    static constexpr int PATCH_BYTES = PH * PITCH, NPP_MAX = PH * PW * 4;
};
template <int PIX> struct Geo<5, PIX> {   // input gradient of mode 4: the four PHASE IMAGES g[2 j + b] of the output-resolution gradient are the
    // "input channels" of one 2x2 convolution per phase on the low-resolution grid (chunks run over phase x channel chunk)
    static constexpr int TW = 32, TH = PIX / 32, PW = TW + 2, PH = TH + 2, PITCH = PW * PIX_BYTES, NTAPS = 4;
    static constexpr int PATCH_BYTES = PH * PITCH, NPP_MAX = PH * PW * 4;
};
template <int PIX> struct Geo<2, PIX> {   // temporal: rows = (t, p), one zero halo frame on both sides
    static constexpr int NTAPS = 3, TW = 1, TH = 1, PW = 1, PITCH = 0;   // (spatial members: unused placeholders)
    static constexpr int PATCH_BYTES = (PIX + 2 * PB_MAX) * PIX_BYTES, NPP_MAX = (PIX + 2 * PB_MAX) * 4;
};

__device__ __forceinline__ float silu32(float f) { return f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * f)); }
// d silu(z) / dz = s (1 + z (1 - s)), s = sigmoid(z)  (the form of k_gn_bwd_*: diffusion_kernels.hip silu_grad_f)
__device__ __forceinline__ float silu_grad32(float z)
{
    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * z));
    return sg * fmaf(z, 1.f - sg, 1.f);
}

// dynamic LDS of one workgroup: weights + patch double buffers during the main loop, fp32 output staging + statistics scratch
// in the epilogue (the two phases alias)
template <int MI, int NI, int WM, int WN, int MODE>
constexpr int lds_bytes()
{
    constexpr int BN = WM * MI * 32, PIX = WN * NI * 32;
    constexpr int main_loop = 2 * BN * 64 + 2 * ((Geo<MODE, PIX>::PATCH_BYTES + 15) & ~15) + 16;   // (+ the dummy slot)
    constexpr int ep_pix = (BN > 160) ? 32 : (BN > 32 ? 64 : PIX);
    // epilogue: MI >= 2 -- four wave-private fp32 staging areas of SP pixel rows (pitch MI * 128 + 16), overlaid after a barrier by the
    // statistics scratch [2][WN * pixels per sweep][BN]; MI == 1 -- the shared pass-by-pass staging + scratch of rounds 2-3
    constexpr int w_stage = WM * WN * (MI >= 5 ? 16 : 32) * (MI * 128 + 16), w_red = 2 * (WN * (64 / (MI * 4))) * BN * 4;
    constexpr int epilogue = MI >= 2 ? (w_stage > w_red ? w_stage : w_red) : ep_pix * (BN * 4 + 16) + 2 * (256 / (BN / 8)) * BN * 4;
    return main_loop > epilogue ? main_loop : epilogue;
}

// PRO: the fused prologue is a template argument, not a branch in the loop (0: plain, 1: GroupNorm affine, SiLU by a select)
template <typename T, int MI, int NI, int WM, int WN, int MODE, int PRO>
__global__ void __launch_bounds__(256, MODE == 3 ? 1 : 2) k_conv_mfma(const ConvArgs a)   // (stride 2: one workgroup per CU, its patch fills the LDS)
{
    typedef typename Tr<T>::vec8 vec8;
    constexpr int BN = WM * MI * 32, PIX = WN * NI * 32;
    typedef Geo<MODE, PIX> G_;
    constexpr int NTAPS = G_::NTAPS;
    constexpr int WBYTES = BN * 64;
    constexpr int PBYTES = (G_::PATCH_BYTES + 15) & ~15;
    constexpr int NWP = BN * 4, WPT = (NWP + 255) / 256;
    constexpr int PPT = (G_::NPP_MAX + 255) / 256;
    // epilogue staging: EP_PIX pixels x BN channels in fp32 (row pitch BN*4 + 16), then the statistics scratch
    constexpr int EP_PITCH = BN * 4 + 16;
    constexpr int EP_PIX = (BN > 160) ? 32 : (BN > 32 ? 64 : PIX);
    constexpr int NOCT = BN / 8, EP_ROWS = 256 / NOCT, EP_ACTIVE = EP_ROWS * NOCT;
    static_assert(PIX % EP_PIX == 0 && EP_PIX % 32 == 0, "epilogue passes cover whole 32-pixel blocks");
    // (the launcher sizes the dynamic LDS as the larger of the main-loop buffers and this epilogue staging: lds_bytes())

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    GVD_CSTAMP(0);
#ifdef GVD_CONV_TRACE
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_ctrace[blockIdx.x * 6 + 4] = hwid;
        g_ctrace[blockIdx.x * 6 + 5] = xcc;
    }
#endif
    unsigned char* const wbuf = lds;                    // [2][WBYTES]
    unsigned char* const pbuf = lds + 2 * WBYTES;       // [2][PBYTES]

    const T* __restrict__ x = (const T*)a.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, r32 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware reading of the grid (diffusion_common.h) where the input map is the large stream: the channel tiles / phases of a
    // pixel tile then share its patch through one L2 instead of fetching it gridDim.y (x 4 phases) times from beyond (xcd_rule())
    int pxi = blockIdx.x, co_tile = blockIdx.y, bz = blockIdx.z;
    if (a.xcd_map) xcd_conv_ids(pxi, co_tile, bz, a.xcd_map == 2);
    const int Cin = a.Cin, Cout = a.Cout;
    // MODE 4 (upsampling as four phase convolutions): y[2 i + a] = sum_t W_t U(x)[2 i + a + t] touches x[i + a + k - 1], k = 0, 1, per
    // dimension -- a 2 x 2 convolution of the LOW-resolution map per output phase (a_y, a_x), with the taps that fall on the same input
    // pixel summed on the host (conv.py: packed(..., "up2")): 16 tap evaluations per input pixel instead of 36, the GroupNorm + SiLU
    // prologue once per input pixel instead of once per upsampled patch pixel.  blockIdx.z is the phase there, not a split-K slice.
    // MODE 5 (its input gradient): gx[i] = sum_a gU[2 i + a] = sum_u K_u g[2 i + u], u = -1 .. 2 per dimension, written over the phase
    // images g_b[j] = g[2 j + b] of the output-resolution gradient: u = 2 m + b, so phase b contributes taps m in {0, 1} (b = 0) or
    // {-1, 0} (b = 1) on the LOW-resolution grid -- the same 2 x 2 structure, with (phase, channel chunk) as the reduction dimension
    // (a.nchunks = 4 x chunks per phase; K_u summed on the host: conv.py packed(..., "up2_bwd")).  No 2 x 2 sum pass afterwards.
    constexpr bool UP2 = MODE == 4, DN2 = MODE == 5;
    const int phase = UP2 ? bz : 0, ay = phase >> 1, ax = phase & 1;
    const int chunk0 = (UP2 || DN2) ? 0 : bz * a.cps;                                  // split-K slice: chunks [chunk0, chunk0 + nloc)
    const int nloc = (UP2 || DN2) ? a.nchunks : (a.nchunks - chunk0 < a.cps ? a.nchunks - chunk0 : a.cps);
    const int cpp = DN2 ? a.nchunks >> 2 : 1;                                                  // MODE 5: channel chunks per phase image
    auto phase_of = [&](int vchunk) { return (vchunk >= cpp) + (vchunk >= 2 * cpp) + (vchunk >= 3 * cpp); };   // (uniform)

    // ---- tile origin ----
    constexpr bool SPATIAL = MODE != 2;
    int n = 0, ty0 = 0, tx0 = 0, p0 = 0;
    if (SPATIAL) {
        const int per_img = a.tiles_x * a.tiles_y;
        n = pxi / per_img;
        const int rem = pxi - n * per_img;
        ty0 = (rem / a.tiles_x) * G_::TH;
        tx0 = (rem % a.tiles_x) * G_::TW;
    } else {
        n = pxi / a.tiles_x;                              // sample: the (3,1,1) convolution treats pixels independently, so the samples of
        p0 = (pxi - n * a.tiles_x) * a.PB;         // a batch are just more pixel tiles of ONE launch (per-sample norms via coef_per_n)
    }
    const int PB = a.PB;
    const size_t sample_in = SPATIAL ? 0 : (size_t)n * a.N * a.W * a.Cin, sample_out = SPATIAL ? 0 : (size_t)n * a.N * a.W * a.Cout;

    // ---- patch staging map (piece = 16 bytes = 8 channels of one patch pixel) ----
    const int k8 = tid & 3;   // 256 % 4 == 0: a thread always stages the same channel octet of the chunk
    int goff[PPT], loff[PPT];
    const int Hin = a.Hin, Win = a.Win;
    int npp;
    if (SPATIAL) npp = G_::NPP_MAX; else npp = (a.N + 2) * PB * 4;
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const int q = tid + i * 256;
        int pixel = q >> 2;
        // ds_write_b128 is serviced in groups of 8 contiguous lanes against 32 banks (MI355X_MICROARCH.md): a group writes the 4 slots
        // of TWO patch pixels, and at the 80-byte pixel pitch neighbours p, p + 1 overlap by one slot mod 128 bytes (2-way: 86 % of the
        // kernel's SQ_LDS_BANK_CONFLICT cycles, profiles/r03_conv_lds_hunt.txt).  Pixels p and p + 4 are 320 = 64 (mod 128) bytes
        // apart -- disjoint halves of the bank window -- so within every full block of 16 patch pixels the lanes take them in the
        // order 0 4 8 12 1 5 ...; the global reads stay 64 contiguous bytes per lane quad, the operand reads do not change.
        if (((pixel >> 4) + 1) * 64 <= npp) pixel = (pixel & ~15) + ((pixel & 3) << 2) + ((pixel >> 2) & 3);
        goff[i] = -1;
        loff[i] = 0;
        if (q < npp) {
            if (MODE == 3) {        // stride 2: patch origin = input pixel (2 ty0 - pad, 2 tx0 - pad)
                const int py = pixel / G_::PW, px = pixel - py * G_::PW;
                const int gy = 2 * ty0 + py - a.pad_lo, gx = 2 * tx0 + px - a.pad_lo;
                loff[i] = py * G_::PITCH + px * PIX_BYTES + k8 * 16;
                if (gy >= 0 && gy < Hin && gx >= 0 && gx < Win) goff[i] = ((n * Hin + gy) * Win + gx) * Cin;
            } else if (DN2) {       // pixel (gy, gx) of the low-resolution grid <-> pixel (2 gy, 2 gx) of the gradient image (phase offset added per chunk)
                const int py = pixel / G_::PW, px = pixel - py * G_::PW;
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                loff[i] = py * G_::PITCH + px * PIX_BYTES + k8 * 16;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) goff[i] = ((n * (2 * a.H) + 2 * gy) * (2 * a.W) + 2 * gx) * Cin;
            } else if (SPATIAL) {
                const int py = pixel / G_::PW, px = pixel - py * G_::PW;
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                loff[i] = py * G_::PITCH + px * PIX_BYTES + k8 * 16;
                const int sh = a.ups ? 1 : 0;
                bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                if (a.ups == 2) ok = ok && !((gy | gx) & 1);   // zero-stuffed upsampling: odd rows / columns are zeros
                if (ok) goff[i] = ((n * Hin + (gy >> sh)) * Win + (gx >> sh)) * Cin;
            } else {
                const int tt = pixel / PB - 1, pp = pixel - (tt + 1) * PB, gp = p0 + pp;
                loff[i] = pixel * PIX_BYTES + k8 * 16;
                if (tt >= 0 && tt < a.N && gp < a.W) goff[i] = (int)sample_in + (tt * a.W + gp) * Cin;
            }
        } else {
            loff[i] = -1;
        }
    }
    const float2* __restrict__ coef = a.coef ? a.coef + (a.coef_per_n ? (size_t)n * Cin : 0) : nullptr;

    // ---- staging + main loop --------------------------------------------------------------------------------------------------
    // STRAIGHT-LINE by construction (round 4).  The loop over a chunk's taps is unrolled, so "which tap fetches / writes the next
    // patch" is resolved at compile time; every global load and LDS store of the loop is UNCONDITIONAL (the last step re-fetches the
    // last slab / patch, pieces past the end of a stage go to a dummy LDS slot); the GroupNorm(+SiLU) prologue is a template argument
    // of the loop, not a branch in it.  Why: hipcc's s_waitcnt insertion counts outstanding loads exactly only along straight-line
    // code.  With `if (more_w) load`, `if (tap == T_L0) load`, `if (coef)`, `if (loff < 0) continue` in the body (rounds 2-3) the ISA
    // had `s_waitcnt vmcnt(0)` in front of every weight fetch and behind the coefficient fetch of every chunk -- each loop step waited
    // out a whole memory round trip (~1300 clocks per step against 256-640 clocks of matrix work: tests/scripts/r4_conv_trace.py),
    // hidden only as far as the CU's second workgroup could fill in.
    constexpr int PH0 = (PPT + 1) / 2;     // the next patch travels in two halves (pieces [0, PH0), [PH0, PPT)): half the staging registers
    constexpr bool WHOLE = (MODE == 2) && GVD_CONV_WHOLE;   // temporal (3 taps): whole patch fetched at tap 0, written at tap 2
    constexpr int NPREG = WHOLE ? PPT : PH0;
    vec8 preg[NPREG];
    float4 cfr[4];                         // (a, b) pairs of this thread's 8 channels of the next chunk, as loaded (no repacking: a
                                           // register move right behind the load would wait for it)
    bool chan_ok = false;
    unsigned char* const dummy = lds + 2 * WBYTES + 2 * PBYTES;   // 16-byte slot for the stores of pieces past the end of a stage
#pragma unroll
    for (int i = 0; i < PPT; i++) loff[i] = loff[i] < 0 ? 2 * PBYTES : loff[i];      // (relative to pbuf: pbuf + 2 PBYTES == dummy)
    auto load_cf = [&](int chunk) {
        const int c0 = (DN2 ? chunk - phase_of(chunk) * cpp : chunk0 + chunk) * BK + k8 * 8;
        const float4* cp = reinterpret_cast<const float4*>(coef + (c0 < Cin ? c0 : 0));
#pragma unroll
        for (int j = 0; j < 4; j++) cfr[j] = cp[j];
    };
    auto load_p = [&](int chunk, auto half_tag) {
        constexpr int HALF = decltype(half_tag)::value;
        const int pb_ = DN2 ? phase_of(chunk) : 0;                                   // MODE 5: which phase image this chunk reads
        const int c0 = (DN2 ? chunk - pb_ * cpp : chunk0 + chunk) * BK + k8 * 8;
        const int c0s = (c0 < Cin ? c0 : 0) + (DN2 ? ((pb_ >> 1) * (2 * a.W) + (pb_ & 1)) * Cin : 0);
        if (HALF == 0) chan_ok = c0 < Cin;       // (both halves of a patch are written before the next patch's first half is fetched)
#pragma unroll
        for (int i = HALF * PH0; i < (HALF ? PPT : PH0); i++)
            preg[WHOLE ? i : i - HALF * PH0] = *reinterpret_cast<const vec8*>(x + (size_t)(goff[i] >= 0 ? goff[i] : 0) + c0s);
    };
    const bool do_silu = a.silu != 0;
    auto store_p = [&](int buf, auto half_tag) {
        constexpr int HALF = decltype(half_tag)::value;
        unsigned char* pb = pbuf + buf * PBYTES;
#pragma unroll
        for (int i = HALF * PH0; i < (HALF ? PPT : PH0); i++) {
            if (PRO && MI >= 5) __builtin_amdgcn_sched_barrier(0);   // (one piece at a time: interleaved, their fp32 temporaries spill the 5-block tiles)
            vec8 v = preg[WHOLE ? i : i - HALF * PH0];
            const bool ok = goff[i] >= 0 && chan_ok;
            if (PRO) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float4 ab = cfr[j >> 1];
                    const float f = (j & 1) ? fmaf((float)v[j], ab.z, ab.w) : fmaf((float)v[j], ab.x, ab.y);
                    v[j] = (T)(do_silu ? silu32(f) : f);      // (a select, not a branch: every norm on this path carries SiLU)
                }
            }
            if (!ok) v = vec8{};   // conv zero padding applies to the ACTIVATED tensor (and dummy reads are dropped here)
            *reinterpret_cast<vec8*>((loff[i] == 2 * PBYTES ? pbuf : pb) + loff[i]) = v;
        }
    };
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;

    // ---- weight staging (linear copy of the pre-swizzled slab), one step ahead ----
    const T* __restrict__ wt = (const T*)a.w + (((size_t)phase * gridDim.y + co_tile) * a.nchunks + chunk0) * NTAPS * (BN * BK);   // (phase = 0 outside MODE 4)
    const int total = nloc * NTAPS;
    vec8 wreg[WPT];
    auto load_w = [&](int it) {
        const T* src = wt + (size_t)(it < total ? it : total - 1) * (BN * BK);
#pragma unroll
        for (int i = 0; i < WPT; i++) {
            const int q = tid + i * 256;
            wreg[i] = *reinterpret_cast<const vec8*>(src + (q < NWP ? q : NWP - 1) * 8);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; i++) {
            const int q = tid + i * 256;
            *reinterpret_cast<vec8*>(q < NWP ? wbuf + buf * WBYTES + q * 16 : dummy) = wreg[i];
        }
    };
    // ---- MFMA operand addresses ----
    int a_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) a_off[ks] = (wm * MI * 32 + r32) * 64 + ((((2 * ks + hi) ^ ((r32 >> 2) & 3))) << 4);
    int b_off[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        const int m = (wn * NI + ni) * 32 + r32;
        if (MODE == 0) b_off[ni] = (m >> 4) * G_::PITCH + (m & 15) * PIX_BYTES + hi * 16;
        else if (MODE == 1 || MODE == 4 || MODE == 5) b_off[ni] = (m >> 5) * G_::PITCH + (m & 31) * PIX_BYTES + hi * 16;
        else if (MODE == 3) b_off[ni] = 2 * (m >> 5) * G_::PITCH + 2 * (m & 31) * PIX_BYTES + hi * 16;
        else b_off[ni] = m * PIX_BYTES + hi * 16;
    }

    f16v acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) acc[mi][ni] = f16v{};

    // taps at which the next chunk's patch halves are fetched / written (two-halves schedule)
    constexpr int T_L0 = NTAPS >= 9 ? NTAPS - 4 : 0, T_S0 = NTAPS >= 9 ? NTAPS - 3 : 1, T_S1 = NTAPS - 1;
    {
        load_w(0);
        if (PRO) load_cf(0);
        load_p(0, H0{});
        if (WHOLE) load_p(0, H1{});
        store_w(0);
        store_p(0, H0{});
        if (!WHOLE) load_p(0, H1{});
        store_p(0, H1{});
        GVD_CSTAMP(1);
        const int last = nloc - 1;
        for (int chunk = 0; chunk < nloc; chunk++) {
            const int nxt = chunk < last ? chunk + 1 : last;      // (the last chunk re-fetches itself: never read)
            const int pcur = chunk & 1;
#pragma unroll
            for (int tap = 0; tap < NTAPS; tap++) {
                const int it = chunk * NTAPS + tap, wcur = it & 1;
                __builtin_amdgcn_sched_barrier(0);   // (nothing moves between taps: the unrolled body must not be scheduled as one block --
                __syncthreads();                     //  hoisting a later tap's loads over this one's MFMAs spills the big tiles)
                __builtin_amdgcn_sched_barrier(0);
                load_w(it + 1);
                if (tap == T_L0) {
                    if (PRO) load_cf(nxt);
                    load_p(nxt, H0{});
                    if (WHOLE) load_p(nxt, H1{});
                }
                // (the fetches above stay above, the LDS writes below stay below: left alone, the machine scheduler sinks every
                //  global load to right in front of the ds_write that consumes it -- shortest live range, whole latency exposed)
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* wb = wbuf + wcur * WBYTES;
                int shift;
                if (UP2) shift = (ay + (tap >> 1)) * G_::PITCH + (ax + (tap & 1)) * PIX_BYTES;
                else if (DN2) { const int pc = phase_of(chunk); shift = ((tap >> 1) + 1 - (pc >> 1)) * G_::PITCH + ((tap & 1) + 1 - (pc & 1)) * PIX_BYTES; }
                else if (SPATIAL) { const int dy = tap / 3, dx = tap - 3 * dy; shift = dy * G_::PITCH + dx * PIX_BYTES; }
                else shift = tap * PB * PIX_BYTES;
                const unsigned char* pb = pbuf + pcur * PBYTES + shift;
                if constexpr (GVD_CONV_RDAHEAD > 0 && MI <= 4) {   // (the 5-block tiles sit at the register cap: any pinned read-ahead spills them)
                    // operand fragments read AHEAD of the MFMAs that use them, in an order pinned for the machine scheduler (left alone it sinks
                    // every read to right in front of its first use: the LDS latency in front of every group of MFMAs -- gemm_mfma.hip)
                    vec8 af[2][MI], bf[2][NI];
                    auto rd_a = [&](int ks, int mi) { af[ks][mi] = *reinterpret_cast<const vec8*>(wb + a_off[ks] + mi * 2048); };
                    auto rd_b = [&](int ks, int ni) { bf[ks][ni] = *reinterpret_cast<const vec8*>(pb + b_off[ni] + ks * 32); };
                    constexpr int RD = GVD_CONV_RDAHEAD < MI ? GVD_CONV_RDAHEAD : (MI > 1 ? MI - 1 : 1);
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) rd_b(0, ni);
#pragma unroll
                    for (int d = 0; d < RD; d++) rd_a(0, d);
#pragma unroll
                    for (int ks = 0; ks < 2; ks++)
#pragma unroll
                        for (int mi = 0; mi < MI; mi++) {
                            if (mi + RD < MI) rd_a(ks, mi + RD);
                            else if (ks == 0) {
                                const int d = mi + RD - MI;
                                if (d == 0) {
#pragma unroll
                                    for (int ni = 0; ni < NI; ni++) rd_b(1, ni);
                                }
                                rd_a(1, d);
                            }
#pragma unroll
                            for (int ni = 0; ni < NI; ni++) acc[mi][ni] = Tr<T>::mfma(af[ks][mi], bf[ks][ni], acc[mi][ni]);
                        }
                    conv_sched<MI, NI, RD>();
                } else {
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    vec8 af[MI], bf[NI];
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) af[mi] = *reinterpret_cast<const vec8*>(wb + a_off[ks] + mi * 2048);
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) bf[ni] = *reinterpret_cast<const vec8*>(pb + b_off[ni] + ks * 32);
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int ni = 0; ni < NI; ni++) acc[mi][ni] = Tr<T>::mfma(af[mi], bf[ni], acc[mi][ni]);
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                store_w(wcur ^ 1);
                if (WHOLE) {
                    if (tap == T_S1) { store_p(pcur ^ 1, H0{}); store_p(pcur ^ 1, H1{}); }
                } else {
                    if (tap == T_S0) { store_p(pcur ^ 1, H0{}); load_p(nxt, H1{}); }
                    if (tap == T_S1) store_p(pcur ^ 1, H1{});
                }
            }
        }
    }
    __syncthreads();   // all operand reads retired: the LDS is reused by the epilogue
    GVD_CSTAMP(2);

    // ---- epilogue ----
    T* __restrict__ out = (T*)a.out + (UP2 ? (size_t)0 : (size_t)bz * (size_t)a.split_stride);
    const T* __restrict__ res = (const T*)a.res;
    const T* __restrict__ bx = (const T*)a.bx;
    float ssum[8], ssq[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { ssum[j] = 0.f; ssq[j] = 0.f; }
    constexpr bool WEPI = MI >= 2;
    // statistics scratch of the final reduction: [2][RROWS][BN] fp32 at the start of the LDS (written after a barrier)
    constexpr int W_NOCT = MI * 4;                          // 16-byte output chunks (8 channels) per pixel row of a wave
    constexpr int W_PPS = 64 / W_NOCT;                      // pixels a wave finishes per sweep (4 / 8; 3 for the 5-block tiles: 60 lanes)
    constexpr int RROWS = WEPI ? WN * W_PPS : (256 / (BN / 8));
    float* const red = reinterpret_cast<float*>(WEPI ? lds : lds + ((BN > 160) ? 32 : (BN > 32 ? 64 : PIX)) * (BN * 4 + 16));
    int red_row = 0, red_col = 0;
    bool red_on = false;
    if constexpr (WEPI) {
        // WAVE-PRIVATE (round 4).  A wave owns MI * 32 channels x NI * 32 pixels of the tile; it transposes them through its OWN fp32
        // staging area -- pixel rows of MI * 128 + 16 bytes, SP pixels at a time -- and finishes them itself: lane = (pixel in sweep,
        // channel octet), 32 bytes read back, bias / add_nc / residual added in fp32, ONE rounding, a 16-byte store in which the
        // lanes of a pixel cover MI * 64 contiguous bytes, statistics of the rounded values.  No workgroup barrier: the four waves
        // run their epilogues in parallel.  (Rounds 2-3 went pass by pass through one shared staging area: per pass ONE wave wrote,
        // a barrier, all threads read, a barrier -- 8 barriers per tile, three waves idle during every write; the timeline of
        // tests/scripts/r4_conv_trace.py had 25-35 % of a workgroup's life in there on the short-K tiles: the VAE's 128-channel
        // stage, the temporal form.)  Same arithmetic as before, to the bit.
        constexpr int SP = MI >= 5 ? 16 : 32;               // pixels staged at a time (the 5-block rows are 656 bytes: 4 x 16 rows = 42 KB)
        constexpr int WP = MI * 128 + 16;                   // staging pitch (bytes): = 16 mod 128, conflict-free for the 8-lane write groups
        constexpr int LPS = W_PPS * W_NOCT;                 // lanes that take part in a sweep
        constexpr int NSW = (SP + W_PPS - 1) / W_PPS;       // sweeps per staged block
        unsigned char* const wep = lds + wave * (SP * WP);
        const int oc = lane % W_NOCT, pofs = lane / W_NOCT;
        const bool lane_on = lane < LPS;
        const int cout0 = co_tile * BN + wm * MI * 32 + oc * 8;
        const bool full_oct = (cout0 + 8 <= Cout) && ((Cout & 7) == 0);
        float badd[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            badd[j] = 0.f;
            if (lane_on && cout0 + j < Cout) {
                if (a.bias) badd[j] = a.bias[cout0 + j];
                if (SPATIAL && a.add_nc) badd[j] += (float)((const T*)a.add_nc)[(size_t)n * Cout + cout0 + j];
            }
        }
        const float4* __restrict__ bcp = (bx && a.bsilu && full_oct)
            ? reinterpret_cast<const float4*>(a.bcoef + (a.bcoef_per_n ? (size_t)n * Cout : 0) + cout0) : nullptr;
        const T* __restrict__ psrc = res ? res : bx;        // (a launch has a residual OR a norm input, never both)
        auto pixel = [&](int m, bool& valid, size_t& off) {
            if (MODE == 2) {
                const int tt = m / PB, pp = m - tt * PB;
                valid = tt < a.N && p0 + pp < a.W;
                off = sample_out + ((size_t)tt * a.W + p0 + pp) * Cout;
            } else {
                const int ty = MODE == 0 ? (m >> 4) : (m >> 5), tx = MODE == 0 ? (m & 15) : (m & 31);
                valid = ty0 + ty < a.H && tx0 + tx < a.W;
                if (UP2) off = (((size_t)n * (2 * a.H) + 2 * (ty0 + ty) + ay) * (2 * a.W) + 2 * (tx0 + tx) + ax) * Cout;   // (a.H, a.W: the input map)
                else off = (((size_t)n * a.H + ty0 + ty) * a.W + tx0 + tx) * Cout;
            }
        };
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
#pragma unroll
            for (int h = 0; h < 32 / SP; h++) {
                const int m0 = (wn * NI + ni) * 32 + h * SP;       // first tile pixel of this staged block
                // residual / norm-input octets of the block's sweeps: fetched before the staging, consumed after it
                constexpr bool PREF = MI <= 4;             // (no registers for it next to the 160 accumulators of the 5-block tiles)
                vec8 pre[PREF ? NSW : 1];
                if (PREF && full_oct && psrc && lane_on) {
#pragma unroll
                    for (int sw = 0; sw < NSW; sw++) {
                        bool valid;
                        size_t off;
                        const int pl = sw * W_PPS + pofs;
                        pixel(m0 + pl, valid, off);
                        pre[sw] = *reinterpret_cast<const vec8*>(psrc + ((valid && pl < SP) ? off : 0) + cout0);
                    }
                }
                if ((r32 / SP) == h || SP == 32) {
                    const int pl = r32 - h * SP;
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int rg = 0; rg < 4; rg++)
                            *reinterpret_cast<float4*>(wep + pl * WP + (mi * 32 + 8 * rg + 4 * hi) * 4) =
                                make_float4(acc[mi][ni][4 * rg], acc[mi][ni][4 * rg + 1], acc[mi][ni][4 * rg + 2], acc[mi][ni][4 * rg + 3]);
                }
                if (lane_on && cout0 < Cout) {
#pragma unroll(PREF ? NSW : 1)
                    for (int sw = 0; sw < NSW; sw++) {
                        const int pl = sw * W_PPS + pofs;
                        bool valid;
                        size_t off;
                        pixel(m0 + pl, valid, off);
                        if (!valid || pl >= SP) continue;
                        const float4 v0 = *reinterpret_cast<const float4*>(wep + pl * WP + oc * 32);
                        const float4 v1 = *reinterpret_cast<const float4*>(wep + pl * WP + oc * 32 + 16);
                        const float v[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
                        vec8 o;
                        if (full_oct) {
                            vec8 rv = vec8{};
                            if (res) rv = PREF ? pre[PREF ? sw : 0] : *reinterpret_cast<const vec8*>(res + off + cout0);
                            if (bx) {   // (uniform) sums of dz and dz * x, dz = d_out * silu'(a x + b); gamma applied per column below
                                const vec8 xv = PREF ? pre[PREF ? sw : 0] : *reinterpret_cast<const vec8*>(bx + off + cout0);
                                float sg[8];
#pragma unroll
                                for (int j = 0; j < 8; j++) sg[j] = 1.f;
                                if (bcp) {
#pragma unroll
                                    for (int j = 0; j < 4; j++) {
                                        const float4 ab = bcp[j];
                                        sg[2 * j] = silu_grad32(fmaf((float)xv[2 * j], ab.x, ab.y));
                                        sg[2 * j + 1] = silu_grad32(fmaf((float)xv[2 * j + 1], ab.z, ab.w));
                                    }
                                }
#pragma unroll
                                for (int j = 0; j < 8; j++) {
                                    o[j] = (T)(v[j] + badd[j] + (float)rv[j]);
                                    const float dz = (float)o[j] * sg[j];
                                    ssum[j] += dz;
                                    ssq[j] = fmaf(dz, (float)xv[j], ssq[j]);
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < 8; j++) {
                                    o[j] = (T)(v[j] + badd[j] + (float)rv[j]);
                                    const float f = (float)o[j];
                                    ssum[j] += f;
                                    ssq[j] = fmaf(f, f, ssq[j]);
                                }
                            }
                            *reinterpret_cast<vec8*>(out + off + cout0) = o;
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                if (cout0 + j >= Cout) continue;
                                float f = v[j] + badd[j];
                                if (res) f += (float)res[off + cout0 + j];
                                const T hh = (T)f;
                                out[off + cout0 + j] = hh;
                                f = (float)hh;
                                ssum[j] += f;
                                ssq[j] = fmaf(f, f, ssq[j]);
                            }
                        }
                    }
                }
            }
        }
        red_row = wn * W_PPS + pofs;
        red_col = wm * MI * 32 + oc * 8;
        red_on = lane_on;
        __syncthreads();   // every wave is out of its staging area: the statistics scratch below may overlay it
    } else {
        unsigned char* const ep = lds;                                         // [EP_PIX][EP_PITCH] fp32
        const int oct = tid % NOCT, prow = tid / NOCT;
        const bool ep_thread = tid < EP_ACTIVE;
        const int cout0 = co_tile * BN + oct * 8;
        const bool full_oct = (cout0 + 8 <= Cout) && ((Cout & 7) == 0);
        float badd[8];
    #pragma unroll
        for (int j = 0; j < 8; j++) {
            badd[j] = 0.f;
            if (ep_thread && cout0 + j < Cout) {
                if (a.bias) badd[j] = a.bias[cout0 + j];
                if (SPATIAL && a.add_nc) badd[j] += (float)((const T*)a.add_nc)[(size_t)n * Cout + cout0 + j];
            }
        }
        // GroupNorm-backward statistics mode: the forward affine of this thread's 8 channels (needed through silu' only) is
        // re-read per pixel row from L1 -- held in registers it would cost 16 VGPRs in every mode of the 256-register tiles
        const float4* __restrict__ bcp = (bx && a.bsilu && full_oct)
            ? reinterpret_cast<const float4*>(a.bcoef + (a.bcoef_per_n ? (size_t)n * Cout : 0) + cout0) : nullptr;

        constexpr int NPASS = PIX / EP_PIX;
        // Rows of a pass this thread finishes: pl = prow + k EP_ROWS.  Their residual / norm-input octets are fetched BEFORE the pass's
        // accumulators go through LDS (unconditional loads, clamped addresses), so the HBM / L2 round trip sits under the staging and its
        // barrier instead of in front of every row: a tile of the VAE's 128-channel stage made 16 such dependent trips (~0.7 us each
        // against ~10 us of matrix work; the residual form cost +10 %, the norm-backward form +37 % -- tests/scripts/r4_conv_ablate.py).
        constexpr int NROW = (EP_PIX + EP_ROWS - 1) / EP_ROWS;
        constexpr bool PREFETCH = MI <= 4;   // (the 5-block tiles have no registers to spare next to their 160 accumulators)
        auto row_geometry = [&](int pass, int pl, bool& valid, size_t& off) {
            const int m = pass * EP_PIX + pl;
            if (MODE == 2) {
                const int tt = m / PB, pp = m - tt * PB;
                valid = tt < a.N && p0 + pp < a.W;
                off = sample_out + ((size_t)tt * a.W + p0 + pp) * Cout;
            } else {
                const int ty = MODE == 0 ? (m >> 4) : (m >> 5), tx = MODE == 0 ? (m & 15) : (m & 31);
                valid = ty0 + ty < a.H && tx0 + tx < a.W;
                if (UP2) off = (((size_t)n * (2 * a.H) + 2 * (ty0 + ty) + ay) * (2 * a.W) + 2 * (tx0 + tx) + ax) * Cout;   // (a.H, a.W: the input map)
                else off = (((size_t)n * a.H + ty0 + ty) * a.W + tx0 + tx) * Cout;
            }
            valid = valid && pl < EP_PIX;
        };
        for (int pass = 0; pass < NPASS; pass++) {
            vec8 pre[PREFETCH ? NROW : 1];   // (a launch has a residual OR a norm input, never both: gvd_conv_mfma / gvd_conv_mfma_norm_bwd)
            if constexpr (PREFETCH) {
                const T* __restrict__ psrc = res ? res : bx;
                if (ep_thread && full_oct && psrc) {
    #pragma unroll
                    for (int k = 0; k < NROW; k++) {
                        bool valid;
                        size_t off;
                        row_geometry(pass, prow + k * EP_ROWS, valid, off);
                        pre[k] = *reinterpret_cast<const vec8*>(psrc + (valid ? off : 0) + cout0);
                    }
                }
            }
            // accumulators of this pass's pixel blocks -> LDS [pixel][channel] fp32 (a lane owns 4 consecutive channels per quad)
    #pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int pblk = wn * NI + ni;
                if ((pblk * 32) / EP_PIX != pass) continue;
                const int pl = pblk * 32 - pass * EP_PIX + r32;
    #pragma unroll
                for (int mi = 0; mi < MI; mi++) {
    #pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int cl = (wm * MI + mi) * 32 + 8 * rg + 4 * hi;
                        const float4 v = make_float4(acc[mi][ni][4 * rg], acc[mi][ni][4 * rg + 1], acc[mi][ni][4 * rg + 2], acc[mi][ni][4 * rg + 3]);
    #if GVD_CONV_DBG & 4
                        if (v.x == 123.456f)
    #endif
                        *reinterpret_cast<float4*>(ep + pl * EP_PITCH + cl * 4) = v;
                    }
                }
            }
            __syncthreads();
            if (ep_thread) {
    #pragma unroll(PREFETCH ? NROW : 1)
                for (int k = 0; k < NROW; k++) {
                    const int pl = prow + k * EP_ROWS;
                    if (pl >= EP_PIX) break;
                    bool valid;
                    size_t off;
                    row_geometry(pass, pl, valid, off);
                    if (!valid || cout0 >= Cout) continue;
                    // a thread reads 32 contiguous bytes as two ds_read_b128; threads oct and oct + 8 of a 16-lane read group are 256
                    // bytes apart (the same banks), so the upper eight read their halves in the opposite order: the 2-way conflict on
                    // these reads was the 16-33 % SQ_LDS_BANK_CONFLICT of the round-2 counters (the operand reads are conflict free)
                    const int sw = (oct >> 3) & 1;
    #if GVD_CONV_DBG & 8
                    const float4 va = make_float4((float)pl, 0.f, 1.f, 2.f), vb = va;
    #else
                    const float4 va = *reinterpret_cast<const float4*>(ep + pl * EP_PITCH + oct * 32 + (sw ? 16 : 0));
                    const float4 vb = *reinterpret_cast<const float4*>(ep + pl * EP_PITCH + oct * 32 + (sw ? 0 : 16));
    #endif
                    const float4 v0 = sw ? vb : va, v1 = sw ? va : vb;
                    float v[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
                    vec8 o;
                    if (full_oct) {
                        vec8 rv = vec8{};
                        if (res) rv = PREFETCH ? pre[PREFETCH ? k : 0] : *reinterpret_cast<const vec8*>(res + off + cout0);
                        if (bx) {   // wave-uniform (kernel argument): sums of dz and dz * x, dz = d_out * silu'(a x + b); gamma applied per column below
                            const vec8 xv = PREFETCH ? pre[PREFETCH ? k : 0] : *reinterpret_cast<const vec8*>(bx + off + cout0);
                            float sg[8];
    #pragma unroll
                            for (int j = 0; j < 8; j++) sg[j] = 1.f;
                            if (bcp) {
    #pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const float4 ab = bcp[j];
                                    sg[2 * j] = silu_grad32(fmaf((float)xv[2 * j], ab.x, ab.y));
                                    sg[2 * j + 1] = silu_grad32(fmaf((float)xv[2 * j + 1], ab.z, ab.w));
                                }
                            }
    #pragma unroll
                            for (int j = 0; j < 8; j++) {
                                o[j] = (T)(v[j] + badd[j] + (float)rv[j]);
                                const float dz = (float)o[j] * sg[j];
                                ssum[j] += dz;
                                ssq[j] = fmaf(dz, (float)xv[j], ssq[j]);
                            }
                        } else {
    #pragma unroll
                            for (int j = 0; j < 8; j++) {
                                o[j] = (T)(v[j] + badd[j] + (float)rv[j]);
                                const float f = (float)o[j];
                                ssum[j] += f;
                                ssq[j] = fmaf(f, f, ssq[j]);
                            }
                        }
                        *reinterpret_cast<vec8*>(out + off + cout0) = o;
                    } else {
    #pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if (cout0 + j >= Cout) continue;
                            float f = v[j] + badd[j];
                            if (res) f += (float)res[off + cout0 + j];
                            const T h = (T)f;
                            out[off + cout0 + j] = h;
                            f = (float)h;
                            ssum[j] += f;
                            ssq[j] = fmaf(f, f, ssq[j]);
                        }
                    }
                }
            }
            __syncthreads();
        }
        red_row = prow;
        red_col = oct * 8;
        red_on = ep_thread;
    }
    GVD_CSTAMP(3);
    // ---- GroupNorm statistics of the (rounded) outputs for the next norm ----
    if (a.stats) {
        if (red_on) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                red[red_row * BN + red_col + j] = ssum[j];
                red[(RROWS + red_row) * BN + red_col + j] = ssq[j];
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {   // per-channel totals into row 0 (column-private: no race)
            float s = 0.f, q = 0.f;
            for (int rr = 0; rr < RROWS; rr++) { s += red[rr * BN + c]; q += red[(RROWS + rr) * BN + c]; }
            if (bx) {   // backward statistics carry the norm's weight: sum gamma dz, sum gamma dz x
                const int cg = co_tile * BN + c;
                const float gm = cg < Cout ? a.bgamma[cg] : 0.f;
                s *= gm;
                q *= gm;
            }
            red[c] = s;
            red[RROWS * BN + c] = q;
        }
        __syncthreads();
        const int cb = co_tile * BN, ce = (cb + BN < Cout) ? cb + BN : Cout;
        if (cb < ce) {
            const int g_first = cb / a.cpg, g_last = (ce - 1) / a.cpg;
            const int g = g_first + tid;
            if (g <= g_last) {
                const int c_lo = (g * a.cpg > cb) ? g * a.cpg : cb, c_hi = ((g + 1) * a.cpg < ce) ? (g + 1) * a.cpg : ce;
                double s = 0.0, q = 0.0;
                for (int c = c_lo; c < c_hi; c++) { s += (double)red[c - cb]; q += (double)red[RROWS * BN + c - cb]; }
                const int rep = blockIdx.x % a.R;
                const int nstat = n, Nstat = SPATIAL ? a.N : a.NS;
                double* dst = a.stats + (((size_t)rep * Nstat + nstat) * a.G + g) * 2;
                atomicAdd(dst, s);
                atomicAdd(dst + 1, q);
            }
        }
    }
}

// stats[nout][g][0..1] = sum over replicas r and merged samples m of partial[r][nout*merge + m][g][0..1]; one wave per output
__global__ void __launch_bounds__(256) k_gn_merge(const double* __restrict__ partial, double* __restrict__ stats, int R, int Nin,
                                                  int merge, int G, int total)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= total) return;
    const int k = i & 1, g = (i >> 1) % G, nout = (i >> 1) / G;
    double s = 0.0;
    for (int j = lane; j < R * merge; j += 64) {
        const int r = j / merge, m = j - r * merge;
        s += partial[(((size_t)r * Nin + (size_t)nout * merge + m) * G + g) * 2 + k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) stats[i] = s;
}

__global__ void __launch_bounds__(256) k_gn_coef2(const double* __restrict__ stats, const float* __restrict__ gamma,
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

// k_gn_merge + k_gn_coef2 in one launch (the unsharded case): one wave per (sample, group) sums the replicas / merged samples in the
// order k_gn_merge does (lane-strided, then the same butterfly), writes the two sums, and lanes 0 .. cpg-1 (strided) form the affine
// of the group's channels.  A U-Net forward has ~110 norms fed this way: one ~4 us launch each instead of two.
__global__ void __launch_bounds__(64) k_gn_merge_coef(const double* __restrict__ partial, double* __restrict__ stats, int R, int Nin, int merge,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float2* __restrict__ coef,
                                                      int C, int G, long long S, float eps)
{
    const int lane = threadIdx.x, g = blockIdx.x % G, nout = blockIdx.x / G, cpg = C / G;
    double s0 = 0.0, s1 = 0.0;
    for (int j = lane; j < R * merge; j += 64) {
        const int r = j / merge, m = j - r * merge;
        const double* p = partial + (((size_t)r * Nin + (size_t)nout * merge + m) * G + g) * 2;
        s0 += p[0];
        s1 += p[1];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); }
    if (lane == 0) {
        stats[2 * ((size_t)nout * G + g)] = s0;
        stats[2 * ((size_t)nout * G + g) + 1] = s1;
    }
    const double cnt = (double)cpg * (double)S;
    const double mean = s0 / cnt;
    const double var = s1 / cnt - mean * mean;
    const float rstd = rsqrtf((float)(var > 0 ? var : 0) + eps);
    for (int k = lane; k < cpg; k += 64) {
        const int c = g * cpg + k;
        const float a = rstd * gamma[c];
        coef[(size_t)nout * C + c] = make_float2(a, beta[c] - (float)mean * a);
    }
}

// (Measured alternative, not kept: staging the weight slabs with LDS-DMA (global_load_lds) instead of through registers was
//  within +-3 % on every U-Net / VAE shape -- two workgroups per CU already cover the ds_write pass.)

// Which XCD-aware grid reading a launch takes (measured: profiles/r04_conv_xcd_map.txt).  With the plain reading the pixel tiles of one
// (channel tile, slice) are dealt round-robin to the 8 XCDs: every XCD pulls every weight slab AND the gy channel tiles of a pixel tile
// fetch its patch gy times.  Reading 1 (inputs first) makes the channel tiles / upsampling phases of a pixel tile neighbours on one XCD:
// the patch is fetched once, every XCD streams all weights once per round of resident workgroups.  Reading 2 (weights first) makes the
// pixel tiles of one (channel tile, K slice) neighbours: a slab lives in 8 / (gy gz) L2s (>= 1), the patches are fetched gy times.
// Bytes from beyond L2, per launch:  reading 1: in + 8 w rounds;  reading 2: min(gy, 8) in + max(1, 8 / (gy gz)) w.  Split-K slices share
// no input, so they only take reading 2.
static int xcd_rule(long long in_bytes, long long w_bytes, unsigned gx, unsigned gy, unsigned gz, bool split)
{
    static const int forced = [] { const char* e = getenv("GVD_CONV_XCD_MAP"); return e ? atoi(e) : -1; }();   // (A/B switch: 0 plain, 1, 2)
    if (forced >= 0 && forced <= 2) return (forced == 1 && split) ? 0 : forced;
    if (gx * gy * gz < 16) return 0;
    const double rounds = (double)((gx * gy * gz + 511) / 512);
    const unsigned share = gy * gz;
    const double c1 = split ? 1e30 : (double)in_bytes + 8.0 * (double)w_bytes * rounds;
    const double c2 = (double)(gy < 8 ? gy : 8) * (double)in_bytes + (share >= 8 ? 1.0 : 8.0 / share) * (double)w_bytes;
    if (gy * gz == 1) return 1;      // one workgroup per pixel tile: a contiguous run of pixel tiles per XCD, so neighbouring tiles find
                                     // their shared halo rows (27 % of a 16 x 16 tile's patch) in that L2 (guided 320x448 -0.5 %)
    return c1 <= c2 ? 1 : 2;
}

template <typename T, int MI, int NI, int WM, int WN, int MODE, int PRO>
hipError_t launch_pro(const ConvArgs& a, dim3 grid, hipStream_t stream)
{
    constexpr int smem = lds_bytes<MI, NI, WM, WN, MODE>();
    auto kern = k_conv_mfma<T, MI, NI, WM, WN, MODE, PRO>;
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 64 && !attr_done[dev]) {   // per-device attribute (dynamic LDS above 64 KiB)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
    return hipGetLastError();
}

template <typename T, int MI, int NI, int WM, int WN, int MODE>
hipError_t launch_one(const ConvArgs& a, dim3 grid, hipStream_t stream)
{
    return a.coef ? launch_pro<T, MI, NI, WM, WN, MODE, 1>(a, grid, stream) : launch_pro<T, MI, NI, WM, WN, MODE, 0>(a, grid, stream);
}

template <typename T, int MODE>
hipError_t launch_cfg(int cfg, const ConvArgs& a, dim3 grid, hipStream_t stream)
{
    switch (cfg) {
    case 0: return launch_one<T, 5, 2, 1, 4, MODE>(a, grid, stream);   // 160 channels x 256 pixels
    case 1: return launch_one<T, 5, 2, 2, 2, MODE>(a, grid, stream);   // 320 x 128
    case 2: return launch_one<T, 4, 2, 1, 4, MODE>(a, grid, stream);   // 128 x 256
    case 3: return launch_one<T, 1, 2, 1, 4, MODE>(a, grid, stream);   //  32 x 256
    case 4: return launch_one<T, 2, 2, 2, 2, MODE>(a, grid, stream);   // 128 x 128
    default: return hipErrorInvalidValue;
    }
}

template <typename T>
hipError_t launch_stride2(int cfg, const ConvArgs& a, dim3 grid, hipStream_t stream)
{   // the stride-2 patch is (2 TH + 1) x 65 pixels: only the 128-pixel tiles fit the LDS double buffer
    switch (cfg) {
    case 1: return launch_one<T, 5, 2, 2, 2, 3>(a, grid, stream);
    case 4: return launch_one<T, 2, 2, 2, 2, 3>(a, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

const int CFG_BN[5] = { 160, 320, 128, 32, 128 };
const int CFG_PIX[5] = { 256, 128, 256, 256, 128 };

// tile configuration for a problem: cfg index, tile width class (0: 16, 1: 32) -- shared by gvd_conv_config and the launcher
void choose(int mode, int N, int H, int W, int Cout, int* cfg, int* tw32)
{
    const long long pixels = (long long)N * H * W;
    auto padded = [&](int pix, int tw) { const int th = pix / tw; return (long long)((H + th - 1) / th) * th * ((W + tw - 1) / tw) * tw; };
    auto width32 = [&](int pix) { return padded(pix, 32) <= padded(pix, 16) ? 1 : 0; };
    // workgroups a configuration launches (spatial: padded tiles per image; temporal: pixel tiles of PIX / N pixels)
    auto groups = [&](int c) {
        const int pix = CFG_PIX[c], bn = CFG_BN[c];
        long long tiles;
        if (mode == 1) { int pb = pix / N; pb = pb < 1 ? 1 : (pb > PB_MAX ? PB_MAX : pb); tiles = (long long)((W + pb - 1) / pb) * H; }   // (temporal: H = samples)
        else tiles = (mode >= 2 ? padded(pix, 32) : padded(pix, width32(pix) ? 32 : 16)) / pix * N;
        return tiles * ((Cout + bn - 1) / bn);
    };
    int c;
    if (mode >= 2) c = (Cout % 160 == 0) ? 1 : 4;          // stride 2
    else if (Cout <= 32) c = 3;
    else if (Cout % 160 == 0) c = (mode == 0 && pixels >= 40000) ? 0 : 1;    // (temporal: 320 x 128 measured 2-7 % ahead of 160 x 256 at every level)
    else c = 2;
    // small problems (the 9x16 level, the 72x128 VAE stage, the temporal form at 35 / 144 pixels): the big tiles launch fewer
    // workgroups than the chip has CU slots (2 x 256) -- take the 128 x 128 tile when it at least fills one slot per CU better.
    // Temporal: only when the big tile leaves most CUs empty (tests/scripts/r4_tile_sweep.py: 224 workgroups of 320 x 128 beat 560 of
    // 128 x 128 -- two rounds -- by 20 %; at 56 / 116 workgroups the small tile wins by 25-35 %).
    if (mode < 2 && Cout >= 128 && c != 4 && groups(c) < (mode == 1 ? 160 : 384) && groups(4) > groups(c)) c = 4;
    // images that fit ONE 16 x 16 tile but not an 8 x 16 one (the 10 x 14 latents of a 320 x 448 video): 128 x 256 -- a single round
    // of workgroups with the shortest loop step that covers the image (391 against 446-467 us for the other three)
    if (mode == 0 && H <= 16 && W <= 16 && H * W > 128 && Cout % 128 == 0 && groups(2) <= 512 && groups(2) > 256) c = 2;
    static const int forced = [] { const char* e = getenv("GVD_CONV_FORCE_CFG"); return e ? atoi(e) : -1; }();   // experiments: one tile configuration for every stride-1 launch with Cout > 32
    if (forced >= 0 && forced <= 4 && forced != 3 && mode < 2 && Cout > 32) c = forced;
    *cfg = c;
    *tw32 = 1;
    if (mode == 0) *tw32 = width32(CFG_PIX[c]);
}

// ---- frame sheets for maps smaller than a tile ---------------------------------------------------------------------------------
// A 5 x 7, 10 x 14 or 9 x 16 latent fills 27-56 % of the smallest spatial tile (16 x 8 pixels), and tiles do not span frames: the
// level-3 convolutions of the U-Net ran at 180-430 TFLOP/s.  The frames of such a launch are laid out as ONE sheet -- Q maps side by
// side, R down, one ZERO row / column between neighbours (the zero padding both share) -- the ordinary 3 x 3 kernel runs on the
// sheet as a single image (72-88 % of the tile slots real), and the maps are cut out again.  k_sheet_in also applies the
// GroupNorm + SiLU affine that the convolution's prologue would have (per FRAME coefficients: the sheet is one image); k_sheet_out
// adds the per-frame embedding term and the residual (fp32 sum, one rounding).  Both move a few MB.
template <typename T>
__global__ void __launch_bounds__(256) k_sheet_in(const T* __restrict__ x, T* __restrict__ v, const float2* __restrict__ coef, int silu,
                                                  int N, int H, int W, int C, int Q, int Hv, int Wv)
{
    typedef typename Tr<T>::vec8 vec8;
    const int oct = C >> 3;
    const long long total = (long long)Hv * Wv * oct;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / oct;
        const int o = (int)(i - pix * oct), vr = (int)(pix / Wv), vc = (int)(pix - (long long)vr * Wv);
        const int r = vr / (H + 1), y = vr - r * (H + 1), q = vc / (W + 1), xx = vc - q * (W + 1), n = r * Q + q;
        vec8 val = vec8{};
        if (y < H && xx < W && n < N) {
            val = *reinterpret_cast<const vec8*>(x + (((size_t)n * H + y) * W + xx) * C + o * 8);
            if (coef) {
                const float4* cf = reinterpret_cast<const float4*>(coef + (size_t)n * C + o * 8);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float4 ab = cf[k];
                    float f0 = fmaf((float)val[2 * k], ab.x, ab.y), f1 = fmaf((float)val[2 * k + 1], ab.z, ab.w);
                    if (silu) { f0 = silu32(f0); f1 = silu32(f1); }
                    val[2 * k] = (T)f0;
                    val[2 * k + 1] = (T)f1;
                }
            }
        }
        *reinterpret_cast<vec8*>(v + (size_t)i * 8) = val;
    }
}

// out[n][r][c] = sum over the split-K slices of the convolution's partial sums + bias[c] + add_nc[n][c] + res[n][r][c] (fp32 sum, one
// rounding), read either from a plain [n][r][c] result (SHEET = false) or from the cells of a frame sheet; with `stats`, also the sum and
// sum of squares of the ROUNDED outputs per (n, group) for the next GroupNorm (what a convolution epilogue leaves; fp64 atomics onto
// [n][G][2], zeroed by the caller).  Lane map of the NSC norm kernels: a thread owns a channel octet and streams down a strip of rows.
template <typename T, bool SHEET>
__global__ void __launch_bounds__(256) k_slices_out(const T* __restrict__ part, T* __restrict__ out, const float* __restrict__ bias,
                                                    const T* __restrict__ add_nc, const T* __restrict__ res, double* __restrict__ stats,
                                                    int C, int G, long long S, int rows_per_block, int slices, long long slice_stride,
                                                    int H, int W, int Q, int Wv)
{
    typedef typename Tr<T>::vec8 vec8;
    extern __shared__ double sh_g[];   // [G][2] when stats
    const int n = blockIdx.y, oct = C >> 3, cpg = G > 0 ? C / G : C;
    if (stats) {
        for (int i = threadIdx.x; i < 2 * G; i += 256) sh_g[i] = 0.0;
        __syncthreads();
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < S) ? r0 + rows_per_block : S;
    const int Wd = oct < 256 ? oct : 256, rstep = 256 / Wd, lane_row = threadIdx.x / Wd;
    const size_t cell0 = SHEET ? ((size_t)(n / Q) * (H + 1) * Wv + (size_t)(n % Q) * (W + 1)) : 0;
    for (int o = threadIdx.x % Wd; o < oct && lane_row < rstep; o += Wd) {
        float c0[8], s[8], q[8];
        vec8 a = vec8{};
        if (add_nc) a = *reinterpret_cast<const vec8*>(add_nc + (size_t)n * C + o * 8);
#pragma unroll
        for (int k = 0; k < 8; k++) { c0[k] = (bias ? bias[o * 8 + k] : 0.f) + (float)a[k]; s[k] = 0.f; q[k] = 0.f; }
        for (long long r = r0 + lane_row; r < r1; r += rstep) {
            const size_t gi = ((size_t)n * S + r) * C + o * 8;
            size_t si = gi;
            if (SHEET) { const int y = (int)(r / W), xx = (int)(r - (long long)y * W); si = (cell0 + (size_t)y * Wv + xx) * C + o * 8; }
            float f[8];
#pragma unroll
            for (int k = 0; k < 8; k++) f[k] = c0[k];
            for (int sl = 0; sl < slices; sl++) {
                const vec8 p = *reinterpret_cast<const vec8*>(part + (size_t)sl * (size_t)slice_stride + si);
#pragma unroll
                for (int k = 0; k < 8; k++) f[k] += (float)p[k];
            }
            vec8 rr = vec8{}, val;
            if (res) rr = *reinterpret_cast<const vec8*>(res + gi);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                val[k] = (T)(f[k] + (float)rr[k]);
                const float g = (float)val[k];
                s[k] += g;
                q[k] = fmaf(g, g, q[k]);
            }
            *reinterpret_cast<vec8*>(out + gi) = val;
        }
        if (stats) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int g = (o * 8 + k) / cpg;
                atomicAdd(&sh_g[2 * g], (double)s[k]);
                atomicAdd(&sh_g[2 * g + 1], (double)q[k]);
            }
        }
    }
    if (stats) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * G; i += 256) atomicAdd(&stats[(size_t)n * 2 * G + i], sh_g[i]);
    }
}

}  // namespace

extern "C" {

int gvd_conv_config(int mode, int N, int H, int W, int Cin, int Cout, int* block_n, int* tile_pixels, int* tile_width)
{
    if (mode < 0 || mode > 5 || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(-1, "gvd_conv_config: bad arguments");
    int cfg, tw32;
    if (mode >= 4) { choose(0, N, (H + 1) / 2, (W + 1) / 2, Cout, &cfg, &tw32); tw32 = 1; }   // phase upsampling: tiles of the INPUT map, 32 wide
    else choose(mode, N, H, W, Cout, &cfg, &tw32);
    if (block_n) *block_n = CFG_BN[cfg];
    if (tile_pixels) *tile_pixels = CFG_PIX[cfg];
    if (tile_width) *tile_width = mode == 1 ? 0 : (tw32 ? 32 : 16);
    return 0;
}

static int conv_launch(const void* x, const void* w_packed, const float* coef, int coef_per_n, const float* bias, const void* add_nc,
                       const void* residual, void* out, double* stats, int stats_replicas, int groups, int mode, int N, int H, int W,
                       int H_in, int W_in, int Cin, int Cout, int upsample, int silu, int is_bf16, void* stream_,
                       const void* bwd_x, const float* bwd_coef, int bwd_coef_per_n, const float* bwd_gamma, int bwd_silu,
                       int ksplit = 1, int* n_slices = nullptr)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (mode == 4 || mode == 5) {
        // 4: nearest x2 upsampling + 3x3 as four phase convolutions of the input map (kernel MODE 4): H, W are the OUTPUT dims
        // 5: its input gradient (kernel MODE 5): x = the gradient [N][H][W][Cin] at the upsampled resolution, out [N][H/2][W/2][Cout]
        if (!x || !w_packed || !out || N <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || Cin <= 0 || Cout <= 0 || (Cin & 7) || upsample || ksplit != 1 || bwd_x ||
            (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)out | (uintptr_t)residual) & 15) || (stats && (groups <= 0 || Cout % groups || stats_replicas <= 0)))
            return fail(-1, "gvd_conv_mfma (mode 4 / 5): even upsampled dims, Cin % 8 == 0, aligned tensors, no split / norm-backward form");
        const int Hl = H / 2, Wl = W / 2;
        int cfg, tw32;
        choose(0, N, Hl, Wl, Cout, &cfg, &tw32);
        const int BN = CFG_BN[cfg], PIX = CFG_PIX[cfg];
        ConvArgs a{};
        a.x = x; a.w = w_packed; a.coef = reinterpret_cast<const float2*>(coef); a.bias = bias; a.add_nc = add_nc; a.res = residual;
        a.out = out; a.stats = stats;
        a.N = N; a.H = Hl; a.W = Wl; a.Hin = Hl; a.Win = Wl; a.Cin = Cin; a.Cout = Cout; a.ups = 0; a.silu = silu ? 1 : 0; a.NS = 1;
        if ((long long)N * H * W * (mode == 4 ? Cout : Cin) >= (1LL << 31) || (long long)N * Hl * Wl * (mode == 4 ? Cin : Cout) >= (1LL << 31))
            return fail(-1, "gvd_conv_mfma: tensor too large for 32-bit offsets");
        a.nchunks = (Cin + BK - 1) / BK * (mode == 5 ? 4 : 1); a.cps = a.nchunks; a.split_stride = 0;
        a.G = groups > 0 ? groups : 1; a.cpg = Cout / a.G; a.R = stats_replicas > 0 ? stats_replicas : 1;
        a.coef_per_n = coef_per_n;
        const int th = PIX / 32;
        a.tiles_x = (Wl + 31) / 32; a.tiles_y = (Hl + th - 1) / th;
        dim3 grid((unsigned)(a.tiles_x * a.tiles_y * N), (unsigned)((Cout + BN - 1) / BN), mode == 4 ? 4u : 1u);
        a.xcd_map = xcd_rule((long long)N * (mode == 4 ? Hl * Wl : H * W) * Cin * 2, 16LL * Cin * Cout * 2, grid.x, grid.y, grid.z, false);
        const hipError_t e4 = mode == 4 ? (is_bf16 ? launch_cfg<__bf16, 4>(cfg, a, grid, stream) : launch_cfg<_Float16, 4>(cfg, a, grid, stream))
                                        : (is_bf16 ? launch_cfg<__bf16, 5>(cfg, a, grid, stream) : launch_cfg<_Float16, 5>(cfg, a, grid, stream));
        if (e4 != hipSuccess) return fail(-2, "launch k_conv_mfma (phase upsampling)", e4);
        if (n_slices) *n_slices = 1;
        return 0;
    }
    if (ksplit < 1 || (ksplit > 1 && (bias || add_nc || residual || stats || bwd_x || mode > 1 || upsample)))
        return fail(-1, "gvd_conv_mfma_splitk: a split launch is a plain stride-1 convolution (optional prologue); the sums take the rest");
    if (bwd_x) {
        if (!stats || !bwd_gamma || (bwd_silu && !bwd_coef) || (Cout & 7) || mode > 1 || upsample || ((uintptr_t)bwd_x & 15) || ((uintptr_t)bwd_coef & 15))
            return fail(-1, "gvd_conv_mfma_norm_bwd: needs stats, gamma (and the forward affine with SiLU), Cout % 8 == 0, a stride-1 mode, aligned pointers");
    }
    if (!x || !w_packed || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || mode < 0 || mode > 3)
        return fail(-1, "gvd_conv_mfma: bad arguments");
    if ((Cin & 7) || (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)out | (uintptr_t)residual) & 15))
        return fail(-1, "gvd_conv_mfma: Cin must be a multiple of 8 and tensors 16-byte aligned");
    if (mode == 1 && (H < 1 || upsample)) return fail(-1, "gvd_conv_mfma: temporal mode takes N = frames, H = samples (>= 1), W = pixels per frame");
    const int NS = mode == 1 ? H : 1;     // temporal: the samples of a batch are more pixel tiles of one launch (ConvArgs::NS)
    if (upsample < 0 || upsample > 2 || (upsample && mode != 0)) return fail(-1, "gvd_conv_mfma: upsample is 0, 1 (nearest) or 2 (zero-stuffed), stride-1 spatial mode only");
    if (stats && (groups <= 0 || Cout % groups || stats_replicas <= 0)) return fail(-1, "gvd_conv_mfma: bad statistics arguments");
    int cfg, tw32;
    choose(mode, N, H, W, Cout, &cfg, &tw32);
    const int BN = CFG_BN[cfg], PIX = CFG_PIX[cfg];
    if (mode == 1) H = 1;
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.coef = reinterpret_cast<const float2*>(coef); a.bias = bias; a.add_nc = add_nc; a.res = residual;
    a.out = out; a.stats = stats;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ups = upsample; a.silu = silu ? 1 : 0;
    if (mode >= 2) {   // stride 2: out = floor((in + pad_lo + 1 - 3) / 2) + 1 with one zero row / column behind the image
        a.pad_lo = mode == 2 ? 1 : 0;
        if (H_in <= 0 || W_in <= 0 || H != (H_in + a.pad_lo - 2) / 2 + 1 || W != (W_in + a.pad_lo - 2) / 2 + 1)
            return fail(-1, "gvd_conv_mfma: stride-2 modes need H_in, W_in with H = (H_in + pad_lo - 2) / 2 + 1");
        a.Hin = H_in; a.Win = W_in;
    } else if (upsample) {
        if ((H | W) & 1) return fail(-1, "gvd_conv_mfma: upsample needs even output dims");
        a.Hin = H >> 1; a.Win = W >> 1;
    } else {
        a.Hin = H; a.Win = W;
    }
    const long long in_elems = (long long)NS * N * a.Hin * a.Win * Cin, out_elems = (long long)NS * N * H * W * Cout;
    a.NS = NS;
    if (in_elems >= (1LL << 31) || out_elems >= (1LL << 31)) return fail(-1, "gvd_conv_mfma: tensor too large for 32-bit offsets");
    a.nchunks = (Cin + BK - 1) / BK;
    a.cps = (a.nchunks + ksplit - 1) / ksplit;
    a.split_stride = out_elems;
    const int slices = (a.nchunks + a.cps - 1) / a.cps;
    if (n_slices) *n_slices = slices;
    a.G = groups > 0 ? groups : 1; a.cpg = Cout / a.G; a.R = stats_replicas > 0 ? stats_replicas : 1;
    a.coef_per_n = coef_per_n;
    a.bx = bwd_x; a.bcoef = reinterpret_cast<const float2*>(bwd_coef); a.bgamma = bwd_gamma; a.bsilu = bwd_silu ? 1 : 0;
    a.bcoef_per_n = bwd_coef_per_n;
    dim3 grid;
    grid.y = (Cout + BN - 1) / BN;
    grid.z = (unsigned)slices;
    hipError_t e;
    if (mode != 1) {
        const int tw = (mode >= 2 || tw32) ? 32 : 16, th = PIX / tw;
        a.tiles_x = (W + tw - 1) / tw; a.tiles_y = (H + th - 1) / th;
        grid.x = (unsigned)(a.tiles_x * a.tiles_y * N);
        a.xcd_map = xcd_rule(in_elems * 2, 9LL * Cin * Cout * 2, grid.x, grid.y, grid.z, slices > 1);
        if (mode >= 2) e = is_bf16 ? launch_stride2<__bf16>(cfg, a, grid, stream) : launch_stride2<_Float16>(cfg, a, grid, stream);
        else if (tw32) e = is_bf16 ? launch_cfg<__bf16, 1>(cfg, a, grid, stream) : launch_cfg<_Float16, 1>(cfg, a, grid, stream);
        else e = is_bf16 ? launch_cfg<__bf16, 0>(cfg, a, grid, stream) : launch_cfg<_Float16, 0>(cfg, a, grid, stream);
    } else {
        int pb = PIX / N;
        if (pb < 1) return fail(-1, "gvd_conv_mfma: too many frames for one tile");
        if (pb > PB_MAX) pb = PB_MAX;
        a.PB = pb;
        a.tiles_x = (W + pb - 1) / pb; a.tiles_y = 1;
        grid.x = (unsigned)(a.tiles_x * NS);
        a.xcd_map = xcd_rule(in_elems * 2, 3LL * Cin * Cout * 2, grid.x, grid.y, grid.z, slices > 1);
        e = is_bf16 ? launch_cfg<__bf16, 2>(cfg, a, grid, stream) : launch_cfg<_Float16, 2>(cfg, a, grid, stream);
    }
    if (e != hipSuccess) return fail(-2, "launch k_conv_mfma", e);
    return 0;
}

int gvd_conv_mfma(const void* x, const void* w_packed, const float* coef, int coef_per_n, const float* bias, const void* add_nc,
                  const void* residual, void* out, double* stats, int stats_replicas, int groups, int mode, int N, int H, int W,
                  int H_in, int W_in, int Cin, int Cout, int upsample, int silu, int is_bf16, void* stream)
{
    return conv_launch(x, w_packed, coef, coef_per_n, bias, add_nc, residual, out, stats, stats_replicas, groups, mode, N, H, W,
                       H_in, W_in, Cin, Cout, upsample, silu, is_bf16, stream, nullptr, nullptr, 0, nullptr, 0);
}

int gvd_conv_mfma_splitk(const void* x, const void* w_packed, const float* coef, int coef_per_n, void* partials, int ksplit,
                         int* n_slices, int mode, int N, int H, int W, int Cin, int Cout, int silu, int is_bf16, void* stream)
{
    if (!n_slices) return fail(-1, "gvd_conv_mfma_splitk: n_slices is required");
    return conv_launch(x, w_packed, coef, coef_per_n, nullptr, nullptr, nullptr, partials, nullptr, 0, 0, mode, N, H, W, 0, 0, Cin, Cout, 0,
                       silu, is_bf16, stream, nullptr, nullptr, 0, nullptr, 0, ksplit, n_slices);
}

int gvd_conv_mfma_norm_bwd(const void* g, const void* w_packed_bwd, void* d_act, double* bwd_stats, int stats_replicas, int groups,
                           int mode, int N, int H, int W, int Cin, int Cout, const void* norm_x, const float* norm_coef,
                           int norm_coef_per_n, const float* norm_gamma, int norm_silu, int is_bf16, void* stream)
{
    return conv_launch(g, w_packed_bwd, nullptr, 0, nullptr, nullptr, nullptr, d_act, bwd_stats, stats_replicas, groups, mode, N, H, W,
                       0, 0, Cin, Cout, 0, 0, is_bf16, stream, norm_x, norm_coef, norm_coef_per_n, norm_gamma, norm_silu);
}

#ifdef GVD_CONV_TRACE
int gvd_conv_trace_read(unsigned long long* host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ctrace), sizeof(unsigned long long) * (n < 2048 * 6 ? n : 2048 * 6)) == hipSuccess ? 0 : -1;
}
#endif

int gvd_conv_sheet_in(const void* x, void* sheet, const float* coef, int silu, int N, int H, int W, int C, int Q, int is_bf16, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !sheet || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || Q <= 0 || (((uintptr_t)x | (uintptr_t)sheet | (uintptr_t)coef) & 15))
        return fail(-1, "gvd_conv_sheet_in: bad arguments (C % 8 == 0, 16-byte aligned tensors)");
    const int R = (N + Q - 1) / Q, Hv = R * (H + 1) - 1, Wv = Q * (W + 1) - 1;
    const long long total = (long long)Hv * Wv * (C >> 3);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (is_bf16) hipLaunchKernelGGL(k_sheet_in<__bf16>, dim3(blocks), dim3(256), 0, stream, (const __bf16*)x, (__bf16*)sheet, reinterpret_cast<const float2*>(coef), silu, N, H, W, C, Q, Hv, Wv);
    else hipLaunchKernelGGL(k_sheet_in<_Float16>, dim3(blocks), dim3(256), 0, stream, (const _Float16*)x, (_Float16*)sheet, reinterpret_cast<const float2*>(coef), silu, N, H, W, C, Q, Hv, Wv);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_sheet_in", e);
    return 0;
}

static int slices_out(bool sheet, const void* part, int slices, long long slice_stride, void* out, const float* bias, const void* add_nc,
                      const void* residual, double* stats, int groups, int n, long long S, int C, int H, int W, int Q, int Wv, int is_bf16,
                      hipStream_t stream)
{
    // strips of rows per block: ~1024 blocks in flight, short strips for small maps (see gn_rows_per_block in diffusion_kernels.hip)
    long long rows = ((long long)n * S + 1023) / 1024;
    rows = rows < 4 ? 4 : (rows > 128 ? 128 : rows);
    dim3 grid((unsigned)((S + rows - 1) / rows), (unsigned)n);
    const size_t sh = stats ? (size_t)groups * 16 : 0;
#define GVD_SO(T, SH) hipLaunchKernelGGL((k_slices_out<T, SH>), grid, dim3(256), sh, stream, (const T*)part, (T*)out, bias, (const T*)add_nc, (const T*)residual, stats, C, groups, S, (int)rows, slices, slice_stride, H, W, Q, Wv)
    if (is_bf16) { if (sheet) GVD_SO(__bf16, true); else GVD_SO(__bf16, false); }
    else { if (sheet) GVD_SO(_Float16, true); else GVD_SO(_Float16, false); }
#undef GVD_SO
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_slices_out", e);
    return 0;
}

int gvd_conv_sheet_out(const void* sheet, int slices, void* out, const float* bias, const void* add_nc, const void* residual, double* stats,
                       int groups, int N, int H, int W, int C, int Q, int is_bf16, void* stream_)
{
    if (!sheet || !out || slices < 1 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || Q <= 0 || (stats && (groups <= 0 || C % groups)) ||
        (((uintptr_t)sheet | (uintptr_t)out | (uintptr_t)add_nc | (uintptr_t)residual) & 15))
        return fail(-1, "gvd_conv_sheet_out: bad arguments (C % 8 == 0, 16-byte aligned tensors, C % groups == 0)");
    const int R = (N + Q - 1) / Q, Hv = R * (H + 1) - 1, Wv = Q * (W + 1) - 1;
    return slices_out(true, sheet, slices, (long long)Hv * Wv * C, out, bias, add_nc, residual, stats, groups, N, (long long)H * W, C, H, W, Q, Wv,
                      is_bf16, (hipStream_t)stream_);
}

int gvd_conv_sum_slices(const void* partials, int slices, void* out, const float* bias, const void* residual, double* stats, int groups,
                        int n_stat, long long rows, int C, int is_bf16, void* stream_)
{
    if (!partials || !out || slices < 1 || n_stat <= 0 || rows <= 0 || rows % n_stat || C <= 0 || (C & 7) || (stats && (groups <= 0 || C % groups)) ||
        (((uintptr_t)partials | (uintptr_t)out | (uintptr_t)residual) & 15))
        return fail(-1, "gvd_conv_sum_slices: bad arguments (C % 8 == 0, rows % n_stat == 0, 16-byte aligned tensors, C % groups == 0)");
    return slices_out(false, partials, slices, rows * C, out, bias, nullptr, residual, stats, groups, n_stat, rows / n_stat, C, 0, 0, 1, 0, is_bf16,
                      (hipStream_t)stream_);
}

int gvd_group_norm_merge(double* stats, const double* partial, int replicas, int merge, int N, int G, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!stats || !partial || replicas <= 0 || merge <= 0 || N <= 0 || G <= 0) return fail(-1, "gvd_group_norm_merge: bad arguments");
    const int total = N * G * 2;
    hipLaunchKernelGGL(k_gn_merge, dim3((total + 3) / 4), dim3(256), 0, stream, partial, stats, replicas, N * merge, merge, G, total);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_gn_merge", e);
    return 0;
}

int gvd_group_norm_coef(double* stats, const double* partial, int replicas, int merge, const float* gamma, const float* beta,
                        int N, int C, long long S_total, int G, float eps, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!stats || !gamma || !beta || N <= 0 || C <= 0 || G <= 0 || C % G || S_total <= 0) return fail(-1, "gvd_group_norm_coef: bad arguments");
    float2* coef = reinterpret_cast<float2*>(stats + (size_t)N * G * 2);
    if (partial) {
        if (replicas <= 0 || merge <= 0) return fail(-1, "gvd_group_norm_coef: bad replica / merge counts");
        hipLaunchKernelGGL(k_gn_merge_coef, dim3(N * G), dim3(64), 0, stream, partial, stats, replicas, N * merge, merge, gamma, beta, coef, C, G, S_total, eps);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(-2, "launch k_gn_merge_coef", e);
        return 0;
    }
    hipLaunchKernelGGL(k_gn_coef2, dim3((N * C + 255) / 256), dim3(256), 0, stream, (const double*)stats, gamma, beta, coef, N, C, G, S_total, eps);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "launch k_gn_coef", e);
    return 0;
}

}  // extern "C"
