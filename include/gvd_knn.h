/*
 * gvd_knn.h -- C-ABI of the MI355X-native `simple-knn` replacement.
 *
 * Replaces SimpleKNN::knn / distCUDA2 (submodules/simple-knn/simple_knn.cu:188-228, spatial.cu:15-26), the other
 * native CUDA dependency of the training drivers (scene/gaussian_model.py:155,421,450,553).
 * Plain C: raw DEVICE pointers, sizes, hipStream_t as void*.  Returns 0 or a negative code (gvd_knn_last_error()).
 */
#ifndef GVD_KNN_H_INCLUDED
#define GVD_KNN_H_INCLUDED
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device scratch gvd_knn_mean_dist needs for P points. */
size_t gvd_knn_workspace_bytes(int P);

/* points [P][3] fp32 -> mean_dists [P] fp32 = mean of the squared distances to the 3 nearest OTHER points,
 * nearest_idx [P][3] int32 = their indices, nearest first.  Exact (no approximation); ties are broken by position in
 * the stable Morton-code order of the cloud, exactly as the reference's traversal does (so duplicated points are
 * each other's neighbours at distance 0).  With fewer than 4 points the unfilled slots stay at FLT_MAX / index 0,
 * as in the reference.  No host synchronisation; everything is enqueued on `stream`. */
int gvd_knn_mean_dist(const float* points, int P, float* mean_dists, int* nearest_idx,
                      void* workspace, size_t workspace_bytes, void* stream);

const char* gvd_knn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
