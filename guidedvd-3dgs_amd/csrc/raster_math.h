// raster_math.h -- per-Gaussian device math of the MI355X rasterizer (gfx950 only).
//
// Semantics follow the reference kernels (file:line cited per function, relative to
// submodules/diff-gaussian-rasterization-confidence/cuda_rasterizer/).  The translation unit is
// compiled with -ffp-contract=off and every fused multiply-add is an explicit fmaf(); the
// convention for a sum of products  a*b + c*d + e*f  is  fmaf(e,f, fmaf(a,b, c*d))  (DESIGN.md,
// "floating-point convention").  This is what makes depth bits, radii, tile rectangles and hence
// the (tile|depth) sort keys bit-exact against the independent CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GVD_BLOCK_X 16
#define GVD_BLOCK_Y 16

namespace gvd {

// auxiliary.h:22-39
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f,
                           SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                           SH_C2_4 = 0.5462742152960396f;
__device__ constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f,
                           SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                           SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                           SH_C3_6 = -0.5900435899266435f;

__device__ __forceinline__ float dot3c(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return fmaf(a2, b2, fmaf(a0, b0, a1 * b1));
}

// auxiliary.h:58-77 (transformPoint4x3 / 4x4), column-major 16-float matrices.
__device__ __forceinline__ float3 xform4x3(const float3 p, const float* __restrict__ m)
{
    return make_float3(dot3c(m[0], p.x, m[4], p.y, m[8], p.z) + m[12],
                       dot3c(m[1], p.x, m[5], p.y, m[9], p.z) + m[13],
                       dot3c(m[2], p.x, m[6], p.y, m[10], p.z) + m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const float* __restrict__ m)
{
    return make_float4(dot3c(m[0], p.x, m[4], p.y, m[8], p.z) + m[12],
                       dot3c(m[1], p.x, m[5], p.y, m[9], p.z) + m[13],
                       dot3c(m[2], p.x, m[6], p.y, m[10], p.z) + m[14],
                       dot3c(m[3], p.x, m[7], p.y, m[11], p.z) + m[15]);
}

// auxiliary.h:41-44 -- double arithmetic (the reference's literals are double), narrowed once.
__device__ __forceinline__ float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// float -> int, round toward zero, saturating, NaN -> 0 (v_cvt_i32_f32 semantics).
__device__ __forceinline__ int f2i_rz(float f) { return __float2int_rz(f); }

// auxiliary.h:46-56 (getRect); r = {minx, miny, maxx, maxy} in tile units.
__device__ __forceinline__ int4 get_rect(float px, float py, int max_radius, int gx, int gy)
{
    const float fr = (float)max_radius;
    int4 r;
    r.x = min(gx, max(0, f2i_rz((px - fr) / (float)GVD_BLOCK_X)));
    r.y = min(gy, max(0, f2i_rz((py - fr) / (float)GVD_BLOCK_Y)));
    r.z = min(gx, max(0, f2i_rz((px + fr + (float)(GVD_BLOCK_X - 1)) / (float)GVD_BLOCK_X)));
    r.w = min(gy, max(0, f2i_rz((py + fr + (float)(GVD_BLOCK_Y - 1)) / (float)GVD_BLOCK_Y)));
    return r;
}

// forward.cu:118-152 (computeCov3D); quaternion (r,x,y,z) used as given.
__device__ __forceinline__ void cov3d_from_scale_rot(const float3 scale, float mod, const float4 q, float* cov)
{
    const float s0 = mod * scale.x, s1 = mod * scale.y, s2 = mod * scale.z;
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const float R00 = fmaf(-2.f, fmaf(y, y, z * z), 1.f);
    const float R01 = 2.f * fmaf(x, y, -(r * z));
    const float R02 = 2.f * fmaf(x, z, r * y);
    const float R10 = 2.f * fmaf(x, y, r * z);
    const float R11 = fmaf(-2.f, fmaf(x, x, z * z), 1.f);
    const float R12 = 2.f * fmaf(y, z, -(r * x));
    const float R20 = 2.f * fmaf(x, z, -(r * y));
    const float R21 = 2.f * fmaf(y, z, r * x);
    const float R22 = fmaf(-2.f, fmaf(x, x, y * y), 1.f);
    const float M00 = s0 * R00, M01 = s1 * R01, M02 = s2 * R02;
    const float M10 = s0 * R10, M11 = s1 * R11, M12 = s2 * R12;
    const float M20 = s0 * R20, M21 = s1 * R21, M22 = s2 * R22;
    cov[0] = dot3c(M00, M00, M01, M01, M02, M02);
    cov[1] = dot3c(M10, M00, M11, M01, M12, M02);
    cov[2] = dot3c(M20, M00, M21, M01, M22, M02);
    cov[3] = dot3c(M10, M10, M11, M11, M12, M12);
    cov[4] = dot3c(M20, M10, M21, M11, M22, M12);
    cov[5] = dot3c(M20, M20, M21, M21, M22, M22);
}

// forward.cu:74-113 (computeCov2D); tv = view-space mean. Returns (a,b,c) incl. the +0.3 low-pass.
__device__ __forceinline__ float3 cov2d(const float3 tv, float fx, float fy, float tan_fovx, float tan_fovy,
                                        const float* c3, const float* __restrict__ vm)
{
    float tx = tv.x, ty = tv.y;
    const float tz = tv.z;
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = tx / tz;
    const float tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float j00 = fx / tz;
    const float j02 = -(fx * tx) / (tz * tz);
    const float j11 = fy / tz;
    const float j12 = -(fy * ty) / (tz * tz);
    const float T00 = fmaf(vm[2], j02, vm[0] * j00);
    const float T01 = fmaf(vm[6], j02, vm[4] * j00);
    const float T02 = fmaf(vm[10], j02, vm[8] * j00);
    const float T10 = fmaf(vm[2], j12, vm[1] * j11);
    const float T11 = fmaf(vm[6], j12, vm[5] * j11);
    const float T12 = fmaf(vm[10], j12, vm[9] * j11);
    const float V00 = c3[0], V01 = c3[1], V02 = c3[2], V11 = c3[3], V12 = c3[4], V22 = c3[5];
    const float A00 = dot3c(T00, V00, T01, V01, T02, V02);
    const float A01 = dot3c(T10, V00, T11, V01, T12, V02);
    const float A10 = dot3c(T00, V01, T01, V11, T02, V12);
    const float A11 = dot3c(T10, V01, T11, V11, T12, V12);
    const float A20 = dot3c(T00, V02, T01, V12, T02, V22);
    const float A21 = dot3c(T10, V02, T11, V12, T12, V22);
    float3 cov;
    cov.x = dot3c(A00, T00, A10, T01, A20, T02) + 0.3f;
    cov.y = dot3c(A01, T00, A11, T01, A21, T02);
    cov.z = dot3c(A01, T10, A11, T11, A21, T12) + 0.3f;
    return cov;
}

// forward.cu:20-71 (computeColorFromSH).  sh -> 3*M floats of this Gaussian (coefficient-major).
// Returns rgb; *clamp_bits gets bit c set when channel c was clamped at 0.
__device__ __forceinline__ float3 sh_to_rgb(int deg, const float3 pos, const float3 campos,
                                            const float* __restrict__ sh, uint32_t* clamp_bits)
{
    const float dx = pos.x - campos.x, dy = pos.y - campos.y, dz = pos.z - campos.z;
    const float len = sqrtf(dot3c(dx, dx, dy, dy, dz, dz));
    const float x = dx / len, y = dy / len, z = dz / len;
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; c++) res[c] = SH_C0 * sh[c];
    if (deg > 0) {
        const float k1 = SH_C1 * y, k2 = SH_C1 * z, k3 = SH_C1 * x;
#pragma unroll
        for (int c = 0; c < 3; c++)
            res[c] = fmaf(-k3, sh[9 + c], fmaf(k2, sh[6 + c], fmaf(-k1, sh[3 + c], res[c])));
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            const float k4 = SH_C2_0 * xy;
            const float k5 = SH_C2_1 * yz;
            const float k6 = SH_C2_2 * (fmaf(2.0f, zz, -xx) - yy);
            const float k7 = SH_C2_3 * xz;
            const float k8 = SH_C2_4 * (xx - yy);
#pragma unroll
            for (int c = 0; c < 3; c++)
                res[c] = fmaf(k8, sh[24 + c], fmaf(k7, sh[21 + c], fmaf(k6, sh[18 + c],
                         fmaf(k5, sh[15 + c], fmaf(k4, sh[12 + c], res[c])))));
            if (deg > 2) {
                const float k9 = SH_C3_0 * y * fmaf(3.0f, xx, -yy);
                const float k10 = SH_C3_1 * xy * z;
                const float k11 = SH_C3_2 * y * (fmaf(4.0f, zz, -xx) - yy);
                const float k12 = SH_C3_3 * z * (fmaf(-3.0f, yy, fmaf(2.0f, zz, -(3.0f * xx))));
                const float k13 = SH_C3_4 * x * (fmaf(4.0f, zz, -xx) - yy);
                const float k14 = SH_C3_5 * z * (xx - yy);
                const float k15 = SH_C3_6 * x * fmaf(-3.0f, yy, xx);
#pragma unroll
                for (int c = 0; c < 3; c++)
                    res[c] = fmaf(k15, sh[45 + c], fmaf(k14, sh[42 + c], fmaf(k13, sh[39 + c],
                             fmaf(k12, sh[36 + c], fmaf(k11, sh[33 + c], fmaf(k10, sh[30 + c],
                             fmaf(k9, sh[27 + c], res[c])))))));
            }
        }
    }
    uint32_t bits = 0;
    float3 rgb;
    {
        float v = res[0] + 0.5f; bits |= (v < 0.0f) ? 1u : 0u; rgb.x = fmaxf(v, 0.0f);
        v = res[1] + 0.5f;       bits |= (v < 0.0f) ? 2u : 0u; rgb.y = fmaxf(v, 0.0f);
        v = res[2] + 0.5f;       bits |= (v < 0.0f) ? 4u : 0u; rgb.z = fmaxf(v, 0.0f);
    }
    *clamp_bits = bits;
    return rgb;
}

// forward.cu:334-335: the per-(pixel,Gaussian) exponent, identical op sequence on host oracle and device.
__device__ __forceinline__ float gauss_power(float conA, float conB, float conC, float dx, float dy)
{
    return fmaf(-0.5f, fmaf(conA * dx, dx, (conC * dy) * dy), -((conB * dx) * dy));
}

// Conservative rectangle-level test: can this Gaussian reach alpha >= 1/255 at ANY pixel centre of the
// rectangle [x0, x0+xs] x [y0, y0+ys]?  Used to drop list entries at LDS staging time; never
// changes results (entries it drops are exactly those forward.cu:347-349 / backward.cu:505-507 skip
// at every pixel of the rectangle).  Exact minimum of the quadratic form over the rectangle, plus a rounding
// margin >> the fp32 evaluation error of gauss_power (DESIGN.md "tile culling").
__device__ __forceinline__ bool rect_may_contribute(float mx, float my, float conA, float conB, float conC,
                                                    float opacity, float x0, float y0, float xs, float ys)
{
    if (!(opacity >= (1.0f / 255.0f))) return false;         // alpha <= opacity * 1
    // the closed-form minimum below needs a positive-definite conic; otherwise keep the entry
    if (!(conA > 0.0f && conC > 0.0f && conA * conC - conB * conB > 0.0f)) return true;
    const float tau = __logf(255.0f * opacity);              // need q <= tau
    const float ax = x0 - mx, bx = (x0 + xs) - mx;           // pixel - mean, over the rectangle
    const float ay = y0 - my, by = (y0 + ys) - my;
    const float cx = fminf(fmaxf(0.0f, ax), bx);
    const float cy = fminf(fmaxf(0.0f, ay), by);
    float qmin;
    if (cx == 0.0f && cy == 0.0f) {
        qmin = 0.0f;
    } else {
        qmin = 3.0e38f;
        if (cx != 0.0f) {   // nearest vertical edge, minimise over y
            float uy = fminf(fmaxf(-conB * cx / conC, ay), by);
            qmin = 0.5f * (conA * cx * cx + conC * uy * uy) + conB * cx * uy;
        }
        if (cy != 0.0f) {   // nearest horizontal edge, minimise over x
            float ux = fminf(fmaxf(-conB * cy / conA, ax), bx);
            qmin = fminf(qmin, 0.5f * (conA * ux * ux + conC * cy * cy) + conB * ux * cy);
        }
    }
    const float Ux = fmaxf(fabsf(ax), fabsf(bx)), Uy = fmaxf(fabsf(ay), fabsf(by));
    const float mag = fabsf(conA) * Ux * Ux + fabsf(conC) * Uy * Uy + 2.0f * fabsf(conB) * Ux * Uy;
    const float margin = 1.0e-5f * mag + 1.0e-5f * fabsf(tau) + 1.0e-4f;
    // NaN-safe: any NaN in qmin keeps the entry.
    return !(qmin > tau + margin);
}

// The 16x16 tile whose first pixel is (x0, y0).
__device__ __forceinline__ bool tile_may_contribute(float mx, float my, float conA, float conB, float conC,
                                                    float opacity, float x0, float y0)
{
    return rect_may_contribute(mx, my, conA, conB, conC, opacity, x0, y0, 15.0f, 15.0f);
}

// Bit q set: the Gaussian may contribute to the 8x8 pixel quadrant q of the tile (columns 8*(q&1).., rows 8*(q>>1)..) that
// wave q of a blend workgroup owns.  The union of the four quadrants is the tile.
__device__ __forceinline__ uint32_t quad_mask(float mx, float my, float conA, float conB, float conC,
                                              float opacity, float x0, float y0)
{
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        m |= rect_may_contribute(mx, my, conA, conB, conC, opacity, x0 + 8.0f * (float)(q & 1), y0 + 8.0f * (float)(q >> 1), 7.0f, 7.0f)
                 ? (1u << q) : 0u;
    return m;
}

}  // namespace gvd
