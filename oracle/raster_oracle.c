/*
 * raster_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or
 * executed from the product path; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it).
 *
 * CPU restatement (ours, plain C, fp32) of the reference's differentiable
 * 3D-Gaussian rasterizer, submodules/diff-gaussian-rasterization-confidence:
 *   cuda_rasterizer/auxiliary.h:22-164   (SH constants, transforms, getRect, in_frustum)
 *   cuda_rasterizer/forward.cu:20-381    (SH->RGB, cov3D, cov2D, preprocess, per-tile blend)
 *   cuda_rasterizer/rasterizer_impl.cu:35-138,197-339 (keys, sort bit range, tile ranges)
 *   cuda_rasterizer/backward.cu:20-601   (blend bwd, cov2D bwd, SH bwd, cov3D bwd)
 *
 * PARITY UNPINNED against the reference binary: the reference is CUDA (no nvcc,
 * no GPU in the build container) and ships no tests or golden vectors
 * (SURVEY.md section 4 / 8c). The oracle is pinned instead by (i) the reference's own
 * Python eval_sh (utils/sh_utils.py:57-112) and covariance builder
 * (utils/general_utils.py:82-114, scene/gaussian_model.py:30-34) through
 * tests/golden/, (ii) a dense torch-autograd restatement and (iii) fp64
 * finite differences (tests/test_oracle_*.py).
 *
 * Floating-point convention.  The file is compiled with -ffp-contract=off;
 * every fused multiply-add is written as an explicit fmaf().  Where the
 * reference writes a sum of products  a*b + c*d + e*f  we use the contraction
 * an LLVM-family device compiler applies to that expression tree:
 *     fmaf(e, f, fmaf(a, b, c*d))
 * and  x - a*b -> fmaf(-a, b, x),  a*b - c -> fmaf(a, b, -c).
 * The HIP kernels (guidedvd-3dgs_amd/csrc) follow the same explicit sequence,
 * which is what makes tile keys / depth bits / radii bit-exact between the two.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16
#define BLOCK_Y 16
#define NUM_CHANNELS 3

/* auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f };

/* a0*b0 + a1*b1 + a2*b2 under the convention in the header. */
static inline float dot3c(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return fmaf(a2, b2, fmaf(a0, b0, a1 * b1));
}

/* float -> int conversion with the device semantics (round toward zero,
 * saturating, NaN -> 0) so that host and device agree on absurd inputs. */
static inline int f2i_rz(float f)
{
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:58-66 (transformPoint4x3), column-major 4x4 */
static inline void xform4x3(const float* p, const float* m, float* o)
{
    o[0] = dot3c(m[0], p[0], m[4], p[1], m[8], p[2]) + m[12];
    o[1] = dot3c(m[1], p[0], m[5], p[1], m[9], p[2]) + m[13];
    o[2] = dot3c(m[2], p[0], m[6], p[1], m[10], p[2]) + m[14];
}
/* auxiliary.h:68-77 (transformPoint4x4) */
static inline void xform4x4(const float* p, const float* m, float* o)
{
    o[0] = dot3c(m[0], p[0], m[4], p[1], m[8], p[2]) + m[12];
    o[1] = dot3c(m[1], p[0], m[5], p[1], m[9], p[2]) + m[13];
    o[2] = dot3c(m[2], p[0], m[6], p[1], m[10], p[2]) + m[14];
    o[3] = dot3c(m[3], p[0], m[7], p[1], m[11], p[2]) + m[15];
}

/* auxiliary.h:41-44: computed in double (the literals are double), then narrowed. */
static inline float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

/* auxiliary.h:46-56 */
static inline void get_rect(float px, float py, int max_radius, int gx, int gy, int* r /*minx,miny,maxx,maxy*/)
{
    float fr = (float)max_radius;
    r[0] = imin(gx, imax(0, f2i_rz((px - fr) / (float)BLOCK_X)));
    r[1] = imin(gy, imax(0, f2i_rz((py - fr) / (float)BLOCK_Y)));
    r[2] = imin(gx, imax(0, f2i_rz((px + fr + (float)(BLOCK_X - 1)) / (float)BLOCK_X)));
    r[3] = imin(gy, imax(0, f2i_rz((py + fr + (float)(BLOCK_Y - 1)) / (float)BLOCK_Y)));
}

/* forward.cu:118-152 (computeCov3D).  q = (r,x,y,z) used as given (not normalised). */
static inline void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* cov)
{
    float s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    /* GLM column vectors of R as written in the reference (column j = Rcj) */
    float R00 = fmaf(-2.f, fmaf(y, y, z * z), 1.f);
    float R01 = 2.f * fmaf(x, y, -(r * z));
    float R02 = 2.f * fmaf(x, z, r * y);
    float R10 = 2.f * fmaf(x, y, r * z);
    float R11 = fmaf(-2.f, fmaf(x, x, z * z), 1.f);
    float R12 = 2.f * fmaf(y, z, -(r * x));
    float R20 = 2.f * fmaf(x, z, -(r * y));
    float R21 = 2.f * fmaf(y, z, r * x);
    float R22 = fmaf(-2.f, fmaf(x, x, y * y), 1.f);
    /* M = S * R : M col j = (s0*Rcj[0], s1*Rcj[1], s2*Rcj[2]) */
    float M00 = s0 * R00, M01 = s1 * R01, M02 = s2 * R02;
    float M10 = s0 * R10, M11 = s1 * R11, M12 = s2 * R12;
    float M20 = s0 * R20, M21 = s1 * R21, M22 = s2 * R22;
    /* Sigma = M^T M : Sigma[j][i] = dot(M col i, M col j) */
    cov[0] = dot3c(M00, M00, M01, M01, M02, M02);
    cov[1] = dot3c(M10, M00, M11, M01, M12, M02);
    cov[2] = dot3c(M20, M00, M21, M01, M22, M02);
    cov[3] = dot3c(M10, M10, M11, M11, M12, M12);
    cov[4] = dot3c(M20, M10, M21, M11, M22, M12);
    cov[5] = dot3c(M20, M20, M21, M21, M22, M22);
}

/* forward.cu:74-113 (computeCov2D). t = view-space mean (already computed). */
static inline void cov2d(const float* tv, float fx, float fy, float tan_fovx, float tan_fovy,
                         const float* c3, const float* vm, float* cov /*a,b,c*/)
{
    float tx = tv[0], ty = tv[1], tz = tv[2];
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = tx / tz;
    const float tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;

    float j00 = fx / tz;
    float j02 = -(fx * tx) / (tz * tz);
    float j11 = fy / tz;
    float j12 = -(fy * ty) / (tz * tz);
    /* W columns: Wc0=(v0,v4,v8) Wc1=(v1,v5,v9) Wc2=(v2,v6,v10); T = W*J, T col j row i */
    float T00 = fmaf(vm[2], j02, vm[0] * j00);
    float T01 = fmaf(vm[6], j02, vm[4] * j00);
    float T02 = fmaf(vm[10], j02, vm[8] * j00);
    float T10 = fmaf(vm[2], j12, vm[1] * j11);
    float T11 = fmaf(vm[6], j12, vm[5] * j11);
    float T12 = fmaf(vm[10], j12, vm[9] * j11);
    /* A = T^T * Vrk^T : A[j][i] = sum_k T[i][k] * V[k][j], V symmetric from c3 */
    float V00 = c3[0], V01 = c3[1], V02 = c3[2], V11 = c3[3], V12 = c3[4], V22 = c3[5];
    float A00 = dot3c(T00, V00, T01, V01, T02, V02); /* j=0,i=0 */
    float A01 = dot3c(T10, V00, T11, V01, T12, V02); /* j=0,i=1 */
    float A10 = dot3c(T00, V01, T01, V11, T02, V12); /* j=1,i=0 */
    float A11 = dot3c(T10, V01, T11, V11, T12, V12); /* j=1,i=1 */
    float A20 = dot3c(T00, V02, T01, V12, T02, V22); /* j=2,i=0 */
    float A21 = dot3c(T10, V02, T11, V12, T12, V22); /* j=2,i=1 */
    /* cov[j][i] = sum_k A[k][i] * T[j][k] */
    cov[0] = dot3c(A00, T00, A10, T01, A20, T02) + 0.3f; /* [0][0] */
    cov[1] = dot3c(A01, T00, A11, T01, A21, T02);        /* [0][1] */
    cov[2] = dot3c(A01, T10, A11, T11, A21, T12) + 0.3f; /* [1][1] */
}

/* forward.cu:20-71 (computeColorFromSH).  sh = 3*M floats, coefficient-major RGB-inner. */
static inline void sh_to_rgb(int deg, const float* pos, const float* campos, const float* sh,
                             float* rgb, uint8_t* clamped)
{
    float dx = pos[0] - campos[0], dy = pos[1] - campos[1], dz = pos[2] - campos[2];
    float len = sqrtf(dot3c(dx, dx, dy, dy, dz, dz));
    float x = dx / len, y = dy / len, z = dz / len;
    float res[3];
    for (int c = 0; c < 3; c++) res[c] = SH_C0 * sh[c];
    if (deg > 0) {
        float k1 = SH_C1 * y, k2 = SH_C1 * z, k3 = SH_C1 * x;
        for (int c = 0; c < 3; c++)
            res[c] = fmaf(-k3, sh[9 + c], fmaf(k2, sh[6 + c], fmaf(-k1, sh[3 + c], res[c])));
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            float k4 = SH_C2[0] * xy;
            float k5 = SH_C2[1] * yz;
            float k6 = SH_C2[2] * (fmaf(2.0f, zz, -xx) - yy);
            float k7 = SH_C2[3] * xz;
            float k8 = SH_C2[4] * (xx - yy);
            for (int c = 0; c < 3; c++)
                res[c] = fmaf(k8, sh[24 + c], fmaf(k7, sh[21 + c], fmaf(k6, sh[18 + c],
                         fmaf(k5, sh[15 + c], fmaf(k4, sh[12 + c], res[c])))));
            if (deg > 2) {
                float k9 = SH_C3[0] * y * fmaf(3.0f, xx, -yy);
                float k10 = SH_C3[1] * xy * z;
                float k11 = SH_C3[2] * y * (fmaf(4.0f, zz, -xx) - yy);
                float k12 = SH_C3[3] * z * (fmaf(-3.0f, yy, fmaf(2.0f, zz, -(3.0f * xx))));
                float k13 = SH_C3[4] * x * (fmaf(4.0f, zz, -xx) - yy);
                float k14 = SH_C3[5] * z * (xx - yy);
                float k15 = SH_C3[6] * x * fmaf(-3.0f, yy, xx);
                for (int c = 0; c < 3; c++)
                    res[c] = fmaf(k15, sh[45 + c], fmaf(k14, sh[42 + c], fmaf(k13, sh[39 + c],
                             fmaf(k12, sh[36 + c], fmaf(k11, sh[33 + c], fmaf(k10, sh[30 + c],
                             fmaf(k9, sh[27 + c], res[c])))))));
            }
        }
    }
    for (int c = 0; c < 3; c++) {
        float v = res[c] + 0.5f;
        clamped[c] = (v < 0.0f);
        rgb[c] = fmaxf(v, 0.0f);
    }
}

/* ------------------------------------------------------------------------- */
/* forward.cu:155-256 (preprocessCUDA) for all P Gaussians.                   */
/* rects (optional, may be NULL): minx,miny,maxx,maxy per Gaussian (test aid). */
void gvdo_preprocess(int P, int D, int M,
                     const float* means3D, const float* scales, float scale_modifier,
                     const float* rotations, const float* opacities, const float* shs,
                     const float* cov3D_precomp, const float* colors_precomp,
                     const float* viewmatrix, const float* projmatrix, const float* campos,
                     int W, int H, float tan_fovx, float tan_fovy,
                     int* radii, float* means2D, float* depths, float* cov3Ds, float* rgb,
                     float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped,
                     int32_t* rects)
{
    const float focal_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:223-224 */
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        if (rects) { rects[4 * idx] = rects[4 * idx + 1] = rects[4 * idx + 2] = rects[4 * idx + 3] = 0; }
        const float* p = means3D + 3 * idx;
        float pv[3], ph[4];
        /* in_frustum, auxiliary.h:139-164 (prefiltered is always false from Python) */
        xform4x4(p, projmatrix, ph);
        float p_w = 1.0f / (ph[3] + 0.0000001f);
        float projx = ph[0] * p_w, projy = ph[1] * p_w;
        xform4x3(p, viewmatrix, pv);
        if (pv[2] <= 0.2f) continue;

        const float* c3;
        if (cov3D_precomp) c3 = cov3D_precomp + 6 * idx;
        else {
            cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx);
            c3 = cov3Ds + 6 * idx;
        }
        float cov[3];
        cov2d(pv, focal_x, focal_y, tan_fovx, tan_fovy, c3, viewmatrix, cov);
        float det = fmaf(cov[0], cov[2], -(cov[1] * cov[1]));
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conx = cov[2] * det_inv, cony = -cov[1] * det_inv, conz = cov[0] * det_inv;

        float mid = 0.5f * (cov[0] + cov[2]);
        float disc = sqrtf(fmaxf(0.1f, fmaf(mid, mid, -det)));
        float lambda1 = mid + disc;
        float lambda2 = mid - disc;
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix_x = ndc2pix(projx, W), pix_y = ndc2pix(projy, H);
        int r[4];
        get_rect(pix_x, pix_y, f2i_rz(my_radius), gx, gy, r);
        if ((r[2] - r[0]) * (r[3] - r[1]) == 0) continue;

        if (colors_precomp == NULL)
            sh_to_rgb(D, p, campos, shs + (size_t)idx * M * 3, rgb + 3 * idx, clamped + 3 * idx);

        depths[idx] = pv[2];
        radii[idx] = f2i_rz(my_radius);
        means2D[2 * idx] = pix_x;
        means2D[2 * idx + 1] = pix_y;
        conic_opacity[4 * idx] = conx;
        conic_opacity[4 * idx + 1] = cony;
        conic_opacity[4 * idx + 2] = conz;
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((r[3] - r[1]) * (r[2] - r[0]));
        if (rects) { rects[4 * idx] = r[0]; rects[4 * idx + 1] = r[1]; rects[4 * idx + 2] = r[2]; rects[4 * idx + 3] = r[3]; }
    }
}

/* rasterizer_impl.cu:54-66 (checkFrustum) */
void gvdo_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                       uint8_t* present)
{
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        float pv[3];
        xform4x3(means3D + 3 * idx, viewmatrix, pv);
        present[idx] = pv[2] > 0.2f;
    }
}

/* rasterizer_impl.cu:278-282: inclusive scan; returns num_rendered */
int64_t gvdo_scan(int P, const uint32_t* tiles_touched, uint32_t* point_offsets)
{
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
    return P > 0 ? (int64_t)acc : 0;
}

/* rasterizer_impl.cu:35-50 */
uint32_t gvdo_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* Stable LSD radix sort of (u64 key, u32 val) on bits [0, end_bit), semantics of
 * cub::DeviceRadixSort::SortPairs at rasterizer_impl.cu:304-309. */
static void radix_sort_pairs(uint64_t n, const uint64_t* kin, const uint32_t* vin,
                             uint64_t* kout, uint32_t* vout, int end_bit)
{
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    memcpy(ka, kin, sizeof(uint64_t) * n);
    memcpy(va, vin, sizeof(uint32_t) * n);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint64_t mask = (1ull << bits) - 1;
        uint64_t cnt[257] = { 0 };
        for (uint64_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
        for (uint64_t i = 0; i < n; i++) {
            uint64_t d = (ka[i] >> shift) & mask;
            kb[cnt[d]] = ka[i];
            vb[cnt[d]] = va[i];
            cnt[d]++;
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, sizeof(uint64_t) * n);
    memcpy(vout, va, sizeof(uint32_t) * n);
    free(ka); free(kb); free(va); free(vb);
}

/* rasterizer_impl.cu:70-138,290-319: duplicateWithKeys + SortPairs + identifyTileRanges.
 * All arrays sized R = point_offsets[P-1]; ranges sized 2*tiles (zero-filled here). */
void gvdo_bin(int P, int W, int H, const int* radii, const float* means2D, const float* depths,
              const uint32_t* point_offsets, uint64_t R,
              uint64_t* keys_unsorted, uint32_t* vals_unsorted,
              uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            int r[4];
            get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, r);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (int y = r[1]; y < r[3]; y++)
                for (int x = r[0]; x < r[2]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    keys_unsorted[off] = key;
                    vals_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    int bit = (int)gvdo_higher_msb((uint32_t)(gx * gy));
    radix_sort_pairs(R, keys_unsorted, vals_unsorted, keys_sorted, point_list, 32 + bit);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (uint64_t i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
}

/* forward.cu:261-381 (renderCUDA): per-pixel front-to-back blend. */
void gvdo_render(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                 const float* means2D, const float* features, const float* depths,
                 const float* conic_opacity, const float* bg,
                 float* out_color, float* out_depth, float* out_alpha, uint32_t* n_contrib)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int ty = tile / gx, tx = tile % gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= W || py >= H) continue;
                float pixfx = (float)px, pixfy = (float)py;
                float T = 1.0f, C[3] = { 0, 0, 0 }, weight = 0, Dacc = 0;
                uint32_t contributor = 0, last = 0;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    uint32_t g = point_list[k];
                    float dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
                    const float* co = conic_opacity + 4 * g;
                    float power = fmaf(-0.5f, fmaf(co[0] * dx, dx, (co[2] * dy) * dy), -((co[1] * dx) * dy));
                    if (power > 0.0f) continue;
                    float alpha = fminf(0.99f, co[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done=true: pixel stops consuming the list */
                    for (int ch = 0; ch < 3; ch++)
                        C[ch] = fmaf(features[3 * g + ch] * alpha, T, C[ch]);
                    weight = fmaf(alpha, T, weight);
                    Dacc = fmaf(depths[g] * alpha, T, Dacc);
                    T = test_T;
                    last = contributor;
                }
                size_t pid = (size_t)py * W + px;
                n_contrib[pid] = last;
                for (int ch = 0; ch < 3; ch++)
                    out_color[(size_t)ch * H * W + pid] = fmaf(T, bg[ch], C[ch]);
                out_alpha[pid] = weight;
                out_depth[pid] = Dacc;
            }
    }
}

/* backward.cu:415-601 (renderCUDA backward).  Per-(pixel,Gaussian) terms are fp32 exactly as
 * the reference computes them; the per-Gaussian sums the reference forms with float
 * atomicAdd (order-dependent) are accumulated here in fp64 and rounded once.
 * Outputs: dL_dmean2D[3P] (.z stays 0), dL_dconic[4P] (.z unused), dL_dopacity[P],
 * dL_dcolors[3P], dL_ddepths[P].  Caller zero-fills nothing: we do. */
void gvdo_render_backward(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                          const float* bg, const float* means2D, const float* conic_opacity,
                          const float* colors, const float* depths, const float* alphas,
                          const uint32_t* n_contrib, const float* dL_dpixels,
                          const float* dL_dpixel_depths, const float* dL_dalphas,
                          float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                          float* dL_dcolors, float* dL_ddepths)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    double* acc = (double*)calloc((size_t)P * 10 + 1, sizeof(double));
    const float ddelx_dx = 0.5f * W; /* 0.5 * W computed in double then narrowed: exact for ints */
    const float ddely_dy = 0.5f * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int ty = tile / gx, tx = tile % gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= W || py >= H) continue;
                size_t pid = (size_t)py * W + px;
                float pixfx = (float)px, pixfy = (float)py;
                const float T_final = 1 - alphas[pid];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = n_contrib[pid];
                float accum_rec[3] = { 0, 0, 0 }, accum_depth_rec = 0, accum_alpha_rec = 0;
                float dL_dpixel[3];
                for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpixels[(size_t)ch * H * W + pid];
                float dL_dpixel_depth = dL_dpixel_depths[pid];
                float dL_dalpha = dL_dalphas[pid];
                float last_alpha = 0, last_color[3] = { 0, 0, 0 }, last_depth = 0;
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    uint32_t g = point_list[k];
                    float dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
                    const float* co = conic_opacity + 4 * g;
                    float power = fmaf(-0.5f, fmaf(co[0] * dx, dx, (co[2] * dy) * dy), -((co[1] * dx) * dy));
                    if (power > 0.0f) continue;
                    float G = expf(power);
                    float alpha = fminf(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    float dchannel_dcolor = alpha * T;
                    float dL_dopa = 0.0f;
                    double* a = acc + (size_t)g * 10;
                    for (int ch = 0; ch < 3; ch++) {
                        float c = colors[3 * g + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        float dL_dchannel = dL_dpixel[ch];
                        dL_dopa += (c - accum_rec[ch]) * dL_dchannel;
                        float v = dchannel_dcolor * dL_dchannel;
#pragma omp atomic
                        a[6 + ch] += (double)v;
                    }
                    float c_d = depths[g];
                    accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    dL_dopa += (c_d - accum_depth_rec) * dL_dpixel_depth;
                    {
                        float v = dchannel_dcolor * dL_dpixel_depth;
#pragma omp atomic
                        a[9] += (double)v;
                    }
                    accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
                    dL_dopa += (1 - accum_alpha_rec) * dL_dalpha;
                    dL_dopa *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                    dL_dopa += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                    float dL_dG = co[3] * dL_dopa;
                    float gdx = G * dx, gdy = G * dy;
                    float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    float dG_ddely = -gdy * co[2] - gdx * co[1];
                    float v0 = dL_dG * dG_ddelx * ddelx_dx;
                    float v1 = dL_dG * dG_ddely * ddely_dy;
                    float v2 = -0.5f * gdx * dx * dL_dG;
                    float v3 = -0.5f * gdx * dy * dL_dG;
                    float v4 = -0.5f * gdy * dy * dL_dG;
                    float v5 = G * dL_dopa;
#pragma omp atomic
                    a[0] += (double)v0;
#pragma omp atomic
                    a[1] += (double)v1;
#pragma omp atomic
                    a[2] += (double)v2;
#pragma omp atomic
                    a[3] += (double)v3;
#pragma omp atomic
                    a[4] += (double)v4;
#pragma omp atomic
                    a[5] += (double)v5;
                }
            }
    }
    for (int g = 0; g < P; g++) {
        const double* a = acc + (size_t)g * 10;
        dL_dmean2D[3 * g] = (float)a[0];
        dL_dmean2D[3 * g + 1] = (float)a[1];
        dL_dmean2D[3 * g + 2] = 0.f;
        dL_dconic[4 * g] = (float)a[2];
        dL_dconic[4 * g + 1] = (float)a[3];
        dL_dconic[4 * g + 2] = 0.f;
        dL_dconic[4 * g + 3] = (float)a[4];
        dL_dopacity[g] = (float)a[5];
        dL_dcolors[3 * g] = (float)a[6];
        dL_dcolors[3 * g + 1] = (float)a[7];
        dL_dcolors[3 * g + 2] = (float)a[8];
        dL_ddepths[g] = (float)a[9];
    }
    free(acc);
}

/* backward.cu:144-274 (computeCov2DCUDA) + :346-412 (preprocessCUDA bwd) + :20-139 (SH bwd)
 * + :278-341 (cov3D bwd).  Plain fp32, reference operation order, no fused ops.
 * dL_dmean3D[3P], dL_dcov3D[6P], dL_dsh[3MP], dL_dscale[3P], dL_drot[4P] must be zero-filled
 * by the caller (rasterize_points.cu:158-167 allocates them with torch::zeros). */
void gvdo_preprocess_backward(int P, int D, int M, const float* means3D, const int* radii,
                              const float* shs, const uint8_t* clamped, const float* scales,
                              const float* rotations, float scale_modifier, const float* cov3Ds,
                              const float* view, const float* proj, float focal_x, float focal_y,
                              float tan_fovx, float tan_fovy, const float* campos,
                              const float* dL_dmean2D, const float* dL_dconics,
                              float* dL_dmeans, const float* dL_dcolor, const float* dL_ddepth,
                              float* dL_dcov, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        /* ---------------- computeCov2DCUDA ---------------- */
        const float* cov3D = cov3Ds + 6 * idx;
        const float* mean = means3D + 3 * idx;
        float dcx = dL_dconics[4 * idx], dcy = dL_dconics[4 * idx + 1], dcz = dL_dconics[4 * idx + 3];
        float t[3];
        t[0] = view[0] * mean[0] + view[4] * mean[1] + view[8] * mean[2] + view[12];
        t[1] = view[1] * mean[0] + view[5] * mean[1] + view[9] * mean[2] + view[13];
        t[2] = view[2] * mean[0] + view[6] * mean[1] + view[10] * mean[2] + view[14];
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float txtz = t[0] / t[2], tytz = t[1] / t[2];
        t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
        t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        /* GLM column-major: X[c][r] */
        float J[3][3] = { { focal_x / t[2], 0.0f, -(focal_x * t[0]) / (t[2] * t[2]) },
                          { 0.0f, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]) },
                          { 0, 0, 0 } };
        float Wm[3][3] = { { view[0], view[4], view[8] }, { view[1], view[5], view[9] }, { view[2], view[6], view[10] } };
        float Vrk[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
        float T[3][3], A[3][3], c2[3][3];
        for (int j = 0; j < 3; j++)
            for (int i = 0; i < 3; i++)
                T[j][i] = Wm[0][i] * J[j][0] + Wm[1][i] * J[j][1] + Wm[2][i] * J[j][2];
        for (int j = 0; j < 3; j++)
            for (int i = 0; i < 3; i++) /* A = T^T * Vrk^T */
                A[j][i] = T[i][0] * Vrk[0][j] + T[i][1] * Vrk[1][j] + T[i][2] * Vrk[2][j];
        for (int j = 0; j < 3; j++)
            for (int i = 0; i < 3; i++)
                c2[j][i] = A[0][i] * T[j][0] + A[1][i] * T[j][1] + A[2][i] * T[j][2];
        float a = c2[0][0] + 0.3f, b = c2[0][1], c = c2[1][1] + 0.3f;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
            dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
            dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
            dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
            dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                        (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
        float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                        (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
        float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                        (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
        float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                        (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
        float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                        (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
        float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                        (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
        float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
        float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
        float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
        float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
        float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
        float dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 +
                       (2 * focal_x * t[0]) * tz3 * dL_dJ02 + (2 * focal_y * t[1]) * tz3 * dL_dJ12;
        /* transformVec4x3Transpose; ASSIGNED (backward.cu:273) */
        float* dm = dL_dmeans + 3 * idx;
        dm[0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
        dm[1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
        dm[2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

        /* ---------------- preprocessCUDA (backward) ---------------- */
        float m[3] = { mean[0], mean[1], mean[2] };
        float m_hom_w = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
        float m_w = 1.0f / (m_hom_w + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
        float d0 = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        float d1 = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        float d2 = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dm[0] += d0; dm[1] += d1; dm[2] += d2;
        /* depth path (backward.cu:397-403) */
        float mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
        float gd = dL_ddepth[idx];
        dm[0] += (view[2] - view[3] * mul3) * gd;
        dm[1] += (view[6] - view[7] * mul3) * gd;
        dm[2] += (view[10] - view[11] * mul3) * gd;

        /* ---------------- SH backward (backward.cu:20-139) ---------------- */
        if (shs) {
            float dox = m[0] - campos[0], doy = m[1] - campos[1], doz = m[2] - campos[2];
            float len = sqrtf(dox * dox + doy * doy + doz * doz);
            float x = dox / len, y = doy / len, z = doz / len;
            const float* sh = shs + (size_t)idx * M * 3;
            float dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0.f : 1.f);
            float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
            float* dsh = dL_dsh + (size_t)idx * M * 3;
#define SHV(k, ch) sh[3 * (k) + (ch)]
#define DSH(k, w) for (int ch = 0; ch < 3; ch++) dsh[3 * (k) + ch] = (w) * dRGB[ch]
            DSH(0, SH_C0);
            if (D > 0) {
                float w1 = -SH_C1 * y, w2 = SH_C1 * z, w3 = -SH_C1 * x;
                DSH(1, w1); DSH(2, w2); DSH(3, w3);
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] = -SH_C1 * SHV(3, ch);
                    dRGBdy[ch] = -SH_C1 * SHV(1, ch);
                    dRGBdz[ch] = SH_C1 * SHV(2, ch);
                }
                if (D > 1) {
                    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    float w4 = SH_C2[0] * xy, w5 = SH_C2[1] * yz, w6 = SH_C2[2] * (2.f * zz - xx - yy);
                    float w7 = SH_C2[3] * xz, w8 = SH_C2[4] * (xx - yy);
                    DSH(4, w4); DSH(5, w5); DSH(6, w6); DSH(7, w7); DSH(8, w8);
                    for (int ch = 0; ch < 3; ch++) {
                        dRGBdx[ch] += SH_C2[0] * y * SHV(4, ch) + SH_C2[2] * 2.f * -x * SHV(6, ch) + SH_C2[3] * z * SHV(7, ch) + SH_C2[4] * 2.f * x * SHV(8, ch);
                        dRGBdy[ch] += SH_C2[0] * x * SHV(4, ch) + SH_C2[1] * z * SHV(5, ch) + SH_C2[2] * 2.f * -y * SHV(6, ch) + SH_C2[4] * 2.f * -y * SHV(8, ch);
                        dRGBdz[ch] += SH_C2[1] * y * SHV(5, ch) + SH_C2[2] * 2.f * 2.f * z * SHV(6, ch) + SH_C2[3] * x * SHV(7, ch);
                    }
                    if (D > 2) {
                        float w9 = SH_C3[0] * y * (3.f * xx - yy);
                        float w10 = SH_C3[1] * xy * z;
                        float w11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                        float w12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                        float w13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                        float w14 = SH_C3[5] * z * (xx - yy);
                        float w15 = SH_C3[6] * x * (xx - 3.f * yy);
                        DSH(9, w9); DSH(10, w10); DSH(11, w11); DSH(12, w12); DSH(13, w13); DSH(14, w14); DSH(15, w15);
                        for (int ch = 0; ch < 3; ch++) {
                            dRGBdx[ch] += (SH_C3[0] * SHV(9, ch) * 3.f * 2.f * xy +
                                           SH_C3[1] * SHV(10, ch) * yz +
                                           SH_C3[2] * SHV(11, ch) * -2.f * xy +
                                           SH_C3[3] * SHV(12, ch) * -3.f * 2.f * xz +
                                           SH_C3[4] * SHV(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                                           SH_C3[5] * SHV(14, ch) * 2.f * xz +
                                           SH_C3[6] * SHV(15, ch) * 3.f * (xx - yy));
                            dRGBdy[ch] += (SH_C3[0] * SHV(9, ch) * 3.f * (xx - yy) +
                                           SH_C3[1] * SHV(10, ch) * xz +
                                           SH_C3[2] * SHV(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                                           SH_C3[3] * SHV(12, ch) * -3.f * 2.f * yz +
                                           SH_C3[4] * SHV(13, ch) * -2.f * xy +
                                           SH_C3[5] * SHV(14, ch) * -2.f * yz +
                                           SH_C3[6] * SHV(15, ch) * -3.f * 2.f * xy);
                            dRGBdz[ch] += (SH_C3[1] * SHV(10, ch) * xy +
                                           SH_C3[2] * SHV(11, ch) * 4.f * 2.f * yz +
                                           SH_C3[3] * SHV(12, ch) * 3.f * (2.f * zz - xx - yy) +
                                           SH_C3[4] * SHV(13, ch) * 4.f * 2.f * xz +
                                           SH_C3[5] * SHV(14, ch) * (xx - yy));
                        }
                    }
                }
            }
#undef SHV
#undef DSH
            float ddx = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
            float ddy = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
            float ddz = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
            /* dnormvdv, auxiliary.h:107-117 */
            float sum2 = dox * dox + doy * doy + doz * doz;
            float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dm[0] += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
            dm[1] += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
            dm[2] += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
        }

        /* ---------------- cov3D backward (backward.cu:278-341) ---------------- */
        if (scales) {
            const float* q = rotations + 4 * idx;
            float r = q[0], x = q[1], y = q[2], z = q[3];
            /* GLM R columns */
            float R[3][3] = { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
                              { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
                              { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } };
            float s[3] = { scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1], scale_modifier * scales[3 * idx + 2] };
            float Mm[3][3]; /* M = S*R: M[j][i] = s_i * R[j][i] */
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) Mm[j][i] = s[i] * R[j][i];
            const float* g = dL_dcov + 6 * idx;
            float dS[3][3] = { { g[0], 0.5f * g[1], 0.5f * g[2] }, { 0.5f * g[1], g[3], 0.5f * g[4] }, { 0.5f * g[2], 0.5f * g[4], g[5] } };
            /* dL_dM = 2 * M * dL_dSigma  (GLM product: Res[j][i] = sum_k M[k][i]*dS[j][k]) */
            float dM[3][3];
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++)
                    dM[j][i] = (2.0f * Mm[0][i]) * dS[j][0] + (2.0f * Mm[1][i]) * dS[j][1] + (2.0f * Mm[2][i]) * dS[j][2];
            /* Rt = transpose(R), dL_dMt = transpose(dL_dM) */
            float Rt[3][3], dMt[3][3];
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) { Rt[j][i] = R[i][j]; dMt[j][i] = dM[i][j]; }
            float* ds = dL_dscale + 3 * idx;
            for (int k = 0; k < 3; k++)
                ds[k] = Rt[k][0] * dMt[k][0] + Rt[k][1] * dMt[k][1] + Rt[k][2] * dMt[k][2];
            for (int k = 0; k < 3; k++)
                for (int i = 0; i < 3; i++) dMt[k][i] *= s[k];
            float* dq = dL_drot + 4 * idx;
            dq[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
            dq[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
            dq[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
            dq[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
        }
    }
}

/* Unit-test entry points for the two pieces the reference's own Python can pin
 * (tests/test_oracle_golden.py): SH->RGB (forward.cu:20-71) and cov3D (forward.cu:118-152). */
void gvdo_sh_to_rgb_batch(int N, int deg, const float* pos, const float* campos, const float* sh,
                          float* rgb, uint8_t* clamped)
{
    for (int i = 0; i < N; i++) sh_to_rgb(deg, pos + 3 * i, campos, sh + (size_t)i * 48, rgb + 3 * i, clamped + 3 * i);
}
void gvdo_cov3d_batch(int N, const float* scales, float mod, const float* rots, float* cov)
{
    for (int i = 0; i < N; i++) cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov + 6 * i);
}
