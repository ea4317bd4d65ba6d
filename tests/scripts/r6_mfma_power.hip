// Round 6 dev tool (not part of the product): what the matrix pipes of a power-capped MI355X sustain on RANDOM operands, by MFMA shape and
// by the LDS operand traffic that accompanies them in a GEMM main loop.  Every variant runs ~0.3 s (the board's power controller settles in
// milliseconds); s_memtime of one wave gives the shader cycles of the same span, i.e. the clock the board held.
//   SHAPE 0: v_mfma_f32_32x32x16_f16 (4 independent accumulator blocks per wave)      SHAPE 1: v_mfma_f32_16x16x32_f16 (8 blocks)
//   READS  : ds_read_b128 per 32x32x16-equivalent MFMA x 10 (0 = operands stay in registers, 5 = a 128 x 128 wave tile, 7 = the 160 x 64
//            wave tile of k_gemm_nt); the fragments read are the operands of the following MFMAs
//   WAVES  : waves per SIMD (1 or 2)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tests/scripts/r6_mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_cycles[2];

template <int SHAPE, int READS, bool ZERO>
__global__ void __launch_bounds__(512) k(float* out, int trips, unsigned seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // fill this wave's 16 KB of LDS with pseudo-random halves in [-1, 1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u);
    _Float16* mine = reinterpret_cast<_Float16*>(lds + wave * 16384);
    for (int i = lane; i < 8192; i += 64) {
        s = s * 1664525u + 1013904223u;
        mine[i] = ZERO ? (_Float16)0.f : (_Float16)(((int)(s >> 9) & 0x7fff) * (2.f / 32768.f) - 1.f);
    }
    __syncthreads();
    const unsigned char* base = lds + wave * 16384 + lane * 16;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = *reinterpret_cast<const h8*>(base + i * 1024); b[i] = *reinterpret_cast<const h8*>(base + (4 + i) * 1024); }
    f16v c32[4] = {};
    f4v c16[8] = {};
    unsigned long long t0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memtime();
    int rd = 0;
    for (int t = 0; t < trips; t++) {
        // one trip = 10 MFMA 32x32x16 (or 20 MFMA 16x16x32) + READS ds_read_b128
#pragma unroll
        for (int j = 0; j < 10; j++) {
            if (j < READS) {   // refresh one fragment from LDS (rotating offsets inside the wave's 16 KB)
                const h8 f = *reinterpret_cast<const h8*>(base + ((rd + j) & 15) * 1024);
                if (j & 1) a[(j >> 1) & 3] = f; else b[(j >> 1) & 3] = f;
            }
            if (SHAPE == 0) {
                c32[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[(j + 1) & 3], c32[j & 3], 0, 0, 0);
            } else {
                c16[(2 * j) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[(j + 1) & 3], c16[(2 * j) & 7], 0, 0, 0);
                c16[(2 * j + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(j + 2) & 3], b[(j + 3) & 3], c16[(2 * j + 1) & 7], 0, 0, 0);
            }
        }
        rd += 3;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_cycles[0] = __builtin_amdgcn_s_memtime() - t0; }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int q = 0; q < 16; q++) r += c32[i][q];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) r += c16[i][q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int SHAPE, int READS, bool ZERO>
void run(const char* name, int waves_per_simd, float* out)
{
    const int threads = 256 * waves_per_simd, blocks = 256;
    const size_t smem = (size_t)threads / 64 * 16384;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, READS, ZERO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int trips = 20000;
    hipLaunchKernelGGL((k<SHAPE, READS, ZERO>), dim3(blocks), dim3(threads), smem, 0, out, trips, 1u);
    hipDeviceSynchronize();
    // size the timed launch to ~0.3 s
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, READS, ZERO>), dim3(blocks), dim3(threads), smem, 0, out, trips, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    trips = (int)(trips * 300.f / ms);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, READS, ZERO>), dim3(blocks), dim3(threads), smem, 0, out, trips, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc[2];
    hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
    const double flop = (double)trips * 10 * 32768.0 * 64 / 64 * (threads / 64) * blocks;   // 10 MFMA32-equivalents of 32*32*16*2 flop per wave per trip
    printf("%-58s waves/SIMD=%d  %7.1f ms  %7.0f TFLOP/s  clock %.3f GHz  %.1f cycles per 32x32x16-equivalent per SIMD\n", name, waves_per_simd, ms,
           flop / (ms * 1e-3) / 1e12, cyc[0] / (ms * 1e-3) / 1e9, (double)cyc[0] / ((double)trips * 10 * waves_per_simd));
    fflush(stdout);
}

int main()
{
    float* out; hipMalloc(&out, 1 << 24);
    run<0, 0, true>("32x32x16 ZERO operands, registers only", 1, out);
    run<0, 0, false>("32x32x16 random, registers only", 1, out);
    run<1, 0, false>("16x16x32 random, registers only", 1, out);
    run<0, 0, false>("32x32x16 random, registers only", 2, out);
    run<1, 0, false>("16x16x32 random, registers only", 2, out);
    run<0, 5, false>("32x32x16 random, 0.5 ds_read_b128 per MFMA", 1, out);
    run<1, 5, false>("16x16x32 random, 0.5 ds_read_b128 per MFMA-equivalent", 1, out);
    run<0, 7, false>("32x32x16 random, 0.7 ds_read_b128 per MFMA", 2, out);
    run<1, 7, false>("16x16x32 random, 0.7 ds_read_b128 per MFMA-equivalent", 2, out);
    run<0, 5, false>("32x32x16 random, 0.5 ds_read_b128 per MFMA", 2, out);
    run<0, 10, false>("32x32x16 random, 1.0 ds_read_b128 per MFMA", 2, out);
    run<0, 0, true>("32x32x16 ZERO operands, registers only", 2, out);
    return 0;
}
