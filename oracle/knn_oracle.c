/*
 * knn_oracle.c -- CPU restatement of the reference's `distCUDA2` (submodules/simple-knn), SURVEY 8(f) row N3.
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product).  PARITY UNPINNED against reference
 * binaries (the reference is CUDA/thrust/CUB and ships no tests or golden vectors for this op); what is restated,
 * line by line in meaning:
 *   simple_knn.cu:188-228  knn(): min/max of the cloud with init {0,0,0} (so the box always contains the origin),
 *                          10-bit-per-axis Morton code, STABLE sort of point ids by code, boxes of 1024 sorted points
 *   simple_knn.cu:46-62    prepMorton / coord2Morton: ((c - min) / (max - min)) * 1023 truncated to uint32
 *   simple_knn.cu:131-147  updateKBest<3>: strict `>` insertion, so among equal distances the candidate met FIRST in
 *                          sorted order keeps the earlier slot
 *   simple_knn.cu:149-186  boxMeanDist: self is skipped by sorted POSITION (duplicates of a point are neighbours at
 *                          distance 0); box pruning only skips boxes that cannot change the result, so the result is
 *                          "the 3 smallest (distance, sorted position) pairs", mean = (d0+d1+d2)/3.0f,
 *                          nearestIndices = ORIGINAL ids of those three
 *   spatial.cu:15-26       returns (means[P] f32, nearestIndices[P,3] i32)
 * Distances use the contraction nvcc applies to dx*dx + dy*dy + dz*dz: fma(dz,dz, fma(dy,dy, dx*dx)).
 * This oracle is an O(P^2) scan in sorted order -- obviously equal to the definition above -- for test sizes.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

static uint32_t axis_code(float c, float mn, float mx)
{
    const float t = ((c - mn) / (mx - mn)) * 1023.0f;   /* (1 << 10) - 1 as int -> float */
    return prep_morton((uint32_t)t);
}

/* codes[P], order[P] (order = original ids, stably sorted by code) */
void gvdo_knn_morton_order(const float* pts, int P, uint32_t* codes, uint32_t* order)
{
    float mn[3] = { 0.f, 0.f, 0.f }, mx[3] = { 0.f, 0.f, 0.f };   /* init {0,0,0}: simple_knn.cu:194 */
    for (int i = 0; i < P; i++)
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(mn[a], pts[3 * i + a]);
            mx[a] = fmaxf(mx[a], pts[3 * i + a]);
        }
    for (int i = 0; i < P; i++)
        codes[i] = axis_code(pts[3 * i], mn[0], mx[0]) | (axis_code(pts[3 * i + 1], mn[1], mx[1]) << 1) |
                   (axis_code(pts[3 * i + 2], mn[2], mx[2]) << 2);
    /* stable LSD radix sort, 4 x 8 bits */
    uint32_t* a = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(P > 0 ? P : 1));
    uint32_t* b = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(P > 0 ? P : 1));
    for (int i = 0; i < P; i++) a[i] = (uint32_t)i;
    for (int pass = 0; pass < 4; pass++) {
        size_t cnt[257];
        memset(cnt, 0, sizeof cnt);
        for (int i = 0; i < P; i++) cnt[((codes[a[i]] >> (8 * pass)) & 255u) + 1]++;
        for (int k = 0; k < 256; k++) cnt[k + 1] += cnt[k];
        for (int i = 0; i < P; i++) b[cnt[(codes[a[i]] >> (8 * pass)) & 255u]++] = a[i];
        uint32_t* t = a; a = b; b = t;
    }
    memcpy(order, a, sizeof(uint32_t) * (size_t)P);
    free(a);
    free(b);
}

void gvdo_knn_mean_dist(const float* pts, int P, float* mean_dists, int32_t* nearest)
{
    if (P <= 0) return;
    uint32_t* codes = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)P);
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)P);
    gvdo_knn_morton_order(pts, P, codes, order);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        const float* p = pts + 3 * (size_t)order[idx];
        float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
        int32_t bi[3] = { 0, 0, 0 };
        for (int i = 0; i < P; i++) {
            if (i == idx) continue;
            const float* q = pts + 3 * (size_t)order[i];
            const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
            float dist = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            int32_t pi = (int32_t)order[i];
            for (int j = 0; j < 3; j++) {
                if (best[j] > dist) {
                    const float t = best[j];
                    const int32_t ti = bi[j];
                    best[j] = dist; bi[j] = pi;
                    dist = t; pi = ti;
                }
            }
        }
        mean_dists[order[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
        nearest[3 * (size_t)order[idx] + 0] = bi[0];
        nearest[3 * (size_t)order[idx] + 1] = bi[1];
        nearest[3 * (size_t)order[idx] + 2] = bi[2];
    }
    free(codes);
    free(order);
}
