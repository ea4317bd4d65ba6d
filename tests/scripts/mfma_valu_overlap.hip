// Do MFMA and ordinary VALU instructions overlap on a gfx950 SIMD?  (dev tool, not part of the product)
//   MODE 0: 4 independent v_mfma_f32_32x32x16_f16 per trip (the matrix pipe alone)
//   MODE 1: 32 independent v_fma_f32 per trip (the VALU alone: 32 x 4 cycles = the duration of 4 MFMAs)
//   MODE 2: both in the SAME wave, interleaved 1 MFMA : 8 FMAs (no data dependence between them)
//   MODE 3: 512-thread blocks, waves 0-3 run MODE 0 and waves 4-7 MODE 1 (two waves per SIMD: one feeds the matrix pipe, the
//           other the VALU; roles by wave inside a block because block b lands on XCD b % 8)
// If the two pipes overlap, MODE 2 / 3 take the time of the slower of MODE 0 / 1; if they serialise, the sum.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tests/scripts/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define REP 2048
#define FMA8(A) asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n" \
                             "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n" \
                             : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]) : "v"(s))
template <int MODE>
__global__ void __launch_bounds__(MODE == 3 ? 512 : 256) k(float* out, float s)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    h8 x, y;
    for (int i = 0; i < 8; i++) { x[i] = (_Float16)(0.001f * (threadIdx.x + i)); y[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const bool mfma_wave = MODE == 0 || MODE == 2 || (MODE == 3 && threadIdx.x < 256);
    const bool valu_wave = MODE == 1 || MODE == 2 || (MODE == 3 && threadIdx.x >= 256);
    for (int i = 0; i < REP; i++) {
        if (MODE == 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0); FMA8(a);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c1, 0, 0, 0); FMA8(a);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c2, 0, 0, 0); FMA8(a);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c3, 0, 0, 0); FMA8(a);
        } else {
            if (mfma_wave) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c3, 0, 0, 0);
            }
            if (valu_wave) { FMA8(a); FMA8(a); FMA8(a); FMA8(a); }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; i++) r += a[i];
    for (int i = 0; i < 16; i++) r += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> float run(const char* name, int waves_per_simd)
{
    float* out; hipMalloc(&out, 1 << 24);
    const int threads = MODE == 3 ? 512 : 256;
    const int blocks = MODE == 3 ? 256 : 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD=%d  %.3f ms  (%.1f cycles per trip per SIMD @2.4GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / REP);
    hipFree(out);
    return ms;
}
int main()
{
    run<0>("4 MFMA 32x32x16 f16 per trip", 1);
    run<1>("32 v_fma_f32 per trip", 1);
    run<2>("same wave: 4 x (1 MFMA + 8 v_fma_f32)", 1);
    run<0>("4 MFMA per trip", 2);
    run<1>("32 v_fma_f32 per trip", 2);
    run<3>("waves 0-3 MFMA, waves 4-7 v_fma (1 + 1 per SIMD)", 2);
    run<2>("same wave mix, 2 waves per SIMD", 2);
    return 0;
}
