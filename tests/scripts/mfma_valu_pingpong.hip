// Does the 8-wave "ping-pong" regime (MI355X_MICROARCH.md, two waves per SIMD in complementary segments) buy MFMA / VALU overlap
// on a gfx950 SIMD?  (dev tool, not part of the product; companion of mfma_valu_overlap.hip)
//
// Every run is made twice: on ONE workgroup (one CU busy: the clock stays at its maximum, so the cycle figure is a true
// pipe-sharing figure) and on 256 / 512 workgroups (every CU busy: the chip is power managed, wall time includes the clock drop).
//   MODE 0  8 waves, every wave: 16 MFMA 32x32x16 f16 per trip                      (matrix pipe alone, two waves per SIMD)
//   MODE 1  8 waves, every wave: NV v_fma_f32 + NE v_exp_f32 per trip               (softmax-like VALU alone)
//   MODE 2  8 waves, every wave: 16 MFMA then the VALU block, no barriers           (what the shipped 2 x 4-wave kernel does)
//   MODE 3  8 waves, ping-pong: waves 0-3 run [MFMA | barrier | VALU | barrier], waves 4-7 [VALU | barrier | MFMA | barrier];
//           waves 4-7 at s_setprio 1 (the guide's static priority for the younger half)
//   MODE 4  as MODE 3 without the priority
// Per trip and wave the work is identical in modes 2-4: 16 MFMAs (512 matrix cycles) + the VALU block.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/pingpong tests/scripts/mfma_valu_pingpong.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define REP 1024
#ifndef NV
#define NV 96    // plain VALU per segment (x 4 cycles)
#endif
#ifndef NE
#define NE 32    // v_exp_f32 per segment
#endif
#define FMA8(A) asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n" \
                             "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n" \
                             : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]) : "v"(s))
#define EXP8(A) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" \
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" \
                             : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]))
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, float s)
{
    float a[8], e[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; e[i] = -0.001f * (threadIdx.x + i); }
    h8 x, y;
    for (int i = 0; i < 8; i++) { x[i] = (_Float16)(0.001f * (threadIdx.x + i)); y[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f16v c[4] = {};
    const bool second = threadIdx.x >= 256;
    auto matrix = [&] {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c[j], 0, 0, 0);
    };
    auto vector = [&] {
#pragma unroll
        for (int r = 0; r < NV / 8; r++) { FMA8(a); if (r < NE / 8) EXP8(e); }
    };
    if (MODE == 3 && second) __builtin_amdgcn_s_setprio(1);
    for (int i = 0; i < REP; i++) {
        if (MODE == 0) matrix();
        else if (MODE == 1) vector();
        else if (MODE == 2) { matrix(); vector(); }
        else {
            if (!second) matrix(); else vector();
            __builtin_amdgcn_s_barrier();
            if (!second) vector(); else matrix();
            __builtin_amdgcn_s_barrier();
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; i++) r += a[i] + e[i];
    for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) r += c[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, int blocks)
{
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-66s blocks=%3d  %.3f ms  %7.1f ns per trip\n", name, blocks, ms, ms * 1e6 / REP);
    hipFree(out);
}
int main()
{
    printf("per trip and wave: 16 MFMA 32x32x16 (512 matrix cycles; two waves per SIMD -> 1024 per SIMD), %d v_fma_f32 + %d v_exp_f32\n", NV, NE);
    for (int blocks : {1, 256}) {
        run<0>("MFMA only", blocks);
        run<1>("VALU only", blocks);
        run<2>("MFMA then VALU in every wave, free running", blocks);
        run<3>("ping-pong, barriers, waves 4-7 at setprio 1", blocks);
        run<4>("ping-pong, barriers, no priority", blocks);
    }
    return 0;
}
