// Issue-rate probe: scalar vs packed fp32 FMA, v_exp_f32, v_cndmask on gfx950 (dev tool, not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tests/scripts/valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, sv = {s, s};
    for (int i = 0; i < REP; i++) {
        if (MODE == 0) {   // 8 independent scalar FMAs
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) {   // 4 independent packed FMAs (same flops as MODE 0)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv));
        } else if (MODE == 2) {   // 8 v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 4) {   // 8 v_exp_f16 (round 4: are the half-precision transcendentals cheaper?)
            asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 5) {   // 8 v_rcp_f16
            asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3\n v_rcp_f16 %4, %4\n v_rcp_f16 %5, %5\n v_rcp_f16 %6, %6\n v_rcp_f16 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 6) {   // 8 v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 7) {   // 8 v_pk_fma_f16 (two half results per lane each)
            asm volatile("v_pk_fma_f16 %0, %0, %8, %0\n v_pk_fma_f16 %1, %1, %8, %1\n v_pk_fma_f16 %2, %2, %8, %2\n v_pk_fma_f16 %3, %3, %8, %3\n"
                         "v_pk_fma_f16 %4, %4, %8, %4\n v_pk_fma_f16 %5, %5, %8, %5\n v_pk_fma_f16 %6, %6, %8, %6\n v_pk_fma_f16 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 8) {   // 8 v_cvt_f32_f16
            asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n v_cvt_f32_f16 %4, %4\n v_cvt_f32_f16 %5, %5\n v_cvt_f32_f16 %6, %6\n v_cvt_f32_f16 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 9) {   // 8 v_cvt_pkrtz_f16_f32
            asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %1, %1, %2\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n"
                         "v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %5, %5, %6\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %7, %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else {   // 8 v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, int insts_per_iter, int waves_per_simd)
{
    float* out; hipMalloc(&out, 1 << 24);
    const int blocks = 256 * waves_per_simd;   // 4 waves per block -> waves_per_simd waves on each SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)REP * insts_per_iter * waves_per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    hipFree(out);
}
int main()
{
    for (int w : {1, 4}) {
        run<0>("v_fma_f32", 8, w); run<1>("v_pk_fma_f32", 4, w); run<2>("v_exp_f32", 8, w); run<3>("v_mul_f32", 8, w);
        run<4>("v_exp_f16", 8, w); run<5>("v_rcp_f16", 8, w); run<6>("v_rcp_f32", 8, w); run<7>("v_pk_fma_f16", 8, w);
        run<8>("v_cvt_f32_f16", 8, w); run<9>("v_cvt_pkrtz", 8, w);
    }
    return 0;
}
