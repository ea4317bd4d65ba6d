/*
 * gvd_raster.h -- C-ABI of the MI355X-native differentiable Gaussian rasterizer.
 *
 * Drop-in boundary for the reference's native rasterizer
 *   submodules/diff-gaussian-rasterization-confidence/cuda_rasterizer/rasterizer.h:24-90
 *   (CudaRasterizer::Rasterizer::{markVisible,forward,backward})
 * as bound to Python by rasterize_points.cu:35-229 / ext.cpp:15-18.
 *
 * Plain C: raw DEVICE pointers (HBM), sizes, a hipStream_t passed as void*.
 * No torch types.  All float data is fp32, row-major, the reference's layouts:
 *   means3D[P,3] scales[P,3] rotations[P,4] opacities[P] shs[P,M,3] colors_precomp[P,3]
 *   cov3D_precomp[P,6] viewmatrix[16] projmatrix[16] (transposed / row-vector convention,
 *   scene/cameras.py:60-62) cam_pos[3] background[3]
 *   out_color[3,H,W] out_depth[1,H,W] out_alpha[1,H,W] radii[P] (int32)
 * A NULL pointer means "absent" exactly where the reference tests for nullptr
 * (colors_precomp / cov3D_precomp / shs / scales / rotations / radii).
 *
 * Scratch memory.  Like the reference (std::function<char*(size_t)>, rasterizer.h:32-34)
 * forward obtains its three scratch chunks through caller-supplied allocators so the host
 * framework owns them and can hand the very same chunks to backward.  The chunk layout is
 * an internal contract between gvd_raster_forward and gvd_raster_backward only
 * (documented in DESIGN.md; inspectable through gvd_raster_chunk_layout for tests).
 *
 * Threading / streams: every launch goes to `stream` (the reference used the legacy
 * default stream).  forward performs ONE host<->device sync to learn num_rendered
 * (the reference's cudaMemcpy at rasterizer_impl.cu:282; from the second render of a given
 * (P, width, height) on, stage 2 is queued speculatively BEFORE that wait, see capi.hip),
 * unless the caller passes a capacity through gvd_raster_forward_capped (no sync).
 * Environment switches (read once): GVD_RASTER_SPECULATE=0 (wait, then launch stage 2), GVD_RASTER_FUSED_SORT=0
 * (separate per-tile sort launch instead of sorting inside the forward blend kernel), GVD_RASTER_TRACE_LAUNCHES=1 (synchronise after
 * every launch and print its name to stderr first: the last line before a GPU memory fault names the kernel), GVD_RASTER_LIB (Python side: path
 * of this library).  None of them changes a result.
 * Not re-entrant on one stream from several host threads.
 *
 * Errors: functions return a negative gvd_status on failure; gvd_last_error() gives the
 * message (the reference throws std::runtime_error / AT_ERROR -> Python exception; the
 * Python host layer re-raises as RuntimeError).
 */
#ifndef GVD_RASTER_H_INCLUDED
#define GVD_RASTER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces std::function<char*(size_t)> (rasterizer.h:32-34): must return a device pointer to
 * at least `bytes` bytes (>=16-byte aligned), valid until the matching backward has run. */
typedef char* (*gvd_alloc_fn)(void* user, size_t bytes);

enum gvd_status {
    GVD_OK = 0,
    GVD_ERR_INVALID = -1,   /* bad argument (e.g. SH absent and no precomputed colours) */
    GVD_ERR_HIP = -2,       /* a HIP runtime call / kernel failed (debug!=0 localises it) */
    GVD_ERR_ALLOC = -3,     /* an allocator callback returned NULL */
    GVD_ERR_OVERFLOW = -4   /* capped forward: num_rendered exceeded the given capacity */
};

/* rasterizer.h:38-63  Rasterizer::forward.  Returns num_rendered (>= 0) or a gvd_status (< 0). */
int gvd_raster_forward(
    gvd_alloc_fn geometry_alloc, void* geometry_user,
    gvd_alloc_fn binning_alloc, void* binning_user,
    gvd_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color,
    float* out_depth,
    float* out_alpha,
    int* radii,
    int debug,
    void* stream);

/* Speculative stage 2 (MI355X addition, off by default so that the two reference-signature entry points keep the reference's
 * contract): after gvd_raster_set_speculation(1), gvd_raster_forward may lay the binning chunk out for MORE instances than
 * num_rendered (it queues scatter/sort/blend before the read-back, sized from earlier renders of the same P x width x height;
 * see capi.hip).  The capacity is encoded in the chunk's byte size -- the allocator callback is asked for exactly
 * gvd_raster_binning_bytes(capacity) bytes -- so a caller that opts in must hand that size to gvd_raster_backward_conf
 * (binning_chunk_bytes).  Nothing is keyed on chunk addresses: chunks may be cloned, offloaded and restored freely.
 * gvd_raster_binning_capacity inverts gvd_raster_binning_bytes (0xffffffff if `bytes` is not a chunk size). */
void gvd_raster_set_speculation(int on);

/* Forward -> backward contract of the binning chunk (MI355X addition).  The backward's partial records live in the binning
 * chunk: four 48-byte sub-records per (Gaussian, tile) instance -- one per 8x8 quadrant wave of k_render_bwd -- and one 4-byte
 * flag word per instance that says which of them were written.  The FLAG WORDS ARE CLEARED BY THE FORWARD (its scatter kernel
 * writes capacity x 4 bytes under its own latency, instead of a memset launch at the head of every backward).
 * gvd_raster_backward(_conf) therefore requires the binning chunk as the forward left it -- bytes may be copied / offloaded /
 * restored, but not modified.  A caller that will NOT run a backward on the chunks of its next forwards (no-grad / evaluation
 * renders) may say so with gvd_raster_expect_backward(0): the clearing stores are skipped, gvd_raster_forward lays the binning chunk
 * out WITHOUT the partial records (gvd_raster_binning_bytes_no_backward: a ninth of the size), and a backward on such a chunk is refused
 * when binning_chunk_bytes is given, undefined otherwise.  Per host thread, sticky until changed; default 1 (the reference's contract: any forward may be followed by a
 * backward). */
void gvd_raster_expect_backward(int yes);

/* k_render_bwd cuts a quadrant's walk of more than `entries` list entries into up to four units (MI355X addition; default 512,
 * 0 = never; also GVD_BWD_SPLIT in the environment).  A performance knob only: the units of a cut walk replay the part behind their
 * own with the same instructions in the same order, so every gradient is bit-identical for every setting
 * (tests/test_raster_gpu.py::test_backward_split_walks_are_bit_identical). */
void gvd_raster_set_backward_split(int entries);
uint32_t gvd_raster_binning_capacity(size_t binning_chunk_bytes);

/* Sync-free variant (MI355X addition; no reference counterpart): the caller supplies the
 * binning chunk up front, sized gvd_raster_binning_bytes(capacity, ...).  num_rendered stays
 * on the device (inside the geometry chunk); *d_status (device int32, may be NULL) receives
 * GVD_OK or GVD_ERR_OVERFLOW asynchronously.  Returns GVD_OK or a launch error. */
int gvd_raster_forward_capped(
    char* geometry_chunk, char* binning_chunk, char* image_chunk, uint32_t capacity,
    int P, int D, int M,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color,
    float* out_depth,
    float* out_alpha,
    int* radii,
    int32_t* d_status,
    int debug,
    void* stream);

/* rasterizer.h:65-89  Rasterizer::backward.  R = num_rendered returned by forward (pass the
 * capacity for chunks produced by gvd_raster_forward_capped).  Every output array is fully
 * written (no pre-zeroing needed, unlike rasterize_points.cu:158-167):
 *   dL_dmean2D[P,3] dL_dconic[P,4] dL_dopacity[P] dL_dcolor[P,3] dL_ddepth[P]
 *   dL_dmean3D[P,3] dL_dcov3D[P,6] dL_dsh[P,M,3] dL_dscale[P,3] dL_drot[P,4]
 * Per-Gaussian sums are formed without float atomics (per-instance partials reduced in a
 * fixed order), so results are run-to-run deterministic.  The binning chunk must be as the forward left it (see
 * gvd_raster_expect_backward).  Returns GVD_OK or < 0. */
int gvd_raster_backward(
    int P, int D, int M, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* alphas,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    char* geom_buffer,
    char* binning_buffer,
    char* image_buffer,
    const float* dL_dpix,
    const float* dL_dpix_depth,
    const float* dL_dalphas,
    float* dL_dmean2D,
    float* dL_dconic,
    float* dL_dopacity,
    float* dL_dcolor,
    float* dL_ddepth,
    float* dL_dmean3D,
    float* dL_dcov3D,
    float* dL_dsh,
    float* dL_dscale,
    float* dL_drot,
    int debug,
    void* stream);

/* Same as gvd_raster_backward plus the fork's per-Gaussian confidence[P] (may be NULL == all ones):
 * folds the Python-side scaling of diff_gaussian_rasterization/__init__.py:147-157 into the gather
 * kernel -- dL_dmean3D, dL_dopacity, dL_dcolor, dL_dsh, dL_dscale, dL_drot, dL_dcov3D are returned
 * already multiplied by confidence; dL_dmean2D is not (ref :149).  dL_dpix_depth / dL_dalphas may be
 * NULL (== zero gradient for that output), here and in gvd_raster_backward.
 * binning_chunk_bytes: byte size of the binning chunk as requested from the allocator callback (0 = laid out for exactly R;
 * required after gvd_raster_set_speculation(1)). */
int gvd_raster_backward_conf(
    int P, int D, int M, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* alphas,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    char* geom_buffer,
    char* binning_buffer,
    char* image_buffer,
    const float* dL_dpix,
    const float* dL_dpix_depth,
    const float* dL_dalphas,
    float* dL_dmean2D,
    float* dL_dconic,
    float* dL_dopacity,
    float* dL_dcolor,
    float* dL_ddepth,
    float* dL_dmean3D,
    float* dL_dcov3D,
    float* dL_dsh,
    float* dL_dscale,
    float* dL_drot,
    const float* confidence,
    size_t binning_chunk_bytes,
    int debug,
    void* stream);

/* rasterizer.h:28-33  Rasterizer::markVisible.  present[P] is one byte (bool) per Gaussian. */
int gvd_raster_mark_visible(int P, const float* means3D, const float* viewmatrix,
                            const float* projmatrix, uint8_t* present, void* stream);

/* Chunk sizes (the reference's required<State>(n), rasterizer_impl.h:65-71). */
size_t gvd_raster_geometry_bytes(int P, int width, int height);
size_t gvd_raster_image_bytes(int width, int height);
size_t gvd_raster_binning_bytes(uint32_t num_rendered);
/* MI355X addition: what gvd_raster_forward asks its binning allocator for under gvd_raster_expect_backward(0) -- the chunk WITHOUT the
 * backward's flag words and 4 x 48-byte partial records per instance (24 instead of 220 bytes per instance).  gvd_raster_backward(_conf)
 * refuses such a chunk (GVD_ERR_INVALID) when it is told the chunk's size.  gvd_raster_forward_capped always uses the full layout. */
size_t gvd_raster_binning_bytes_no_backward(uint32_t num_rendered);

/* Byte offsets of the sub-arrays inside the three chunks (for parity tests and debugging).
 * Names mirror rasterizer_impl.h:21-62 where a counterpart exists. */
typedef struct gvd_chunk_layout {
    /* geometry chunk */
    size_t depths;          /* f32[P]                                    */
    size_t means2D;         /* f32x2[P]                                  */
    size_t conic_opacity;   /* f32x4[P]                                  */
    size_t rgbd;            /* f32x4[P]: rgb (SH-evaluated or precomp) + view depth */
    size_t cov3D;           /* f32[6P]                                   */
    size_t clamped;         /* u32[P]: bit c set <=> channel c clamped   */
    size_t internal_radii;  /* i32[P]                                    */
    size_t tiles_touched;   /* u32[P]                                    */
    size_t point_offsets;   /* u32[P] inclusive scan of tiles_touched    */
    size_t scalars;         /* u32[8]: [0]=num_rendered [1]=max tile list length [2]=overflow */
    /* image chunk */
    size_t ranges;          /* u32x2[tiles]                              */
    size_t n_contrib;       /* u32[H*W]                                  */
    /* binning chunk */
    size_t point_list_keys; /* u64[R] sorted (tile<<32 | depth bits)     */
    size_t point_list;      /* u32[R] sorted Gaussian ids                */
    size_t bucket;          /* u64[R] per-tile (depth bits<<32 | id), sorted in place */
    /* image chunk, continued */
    size_t tile_order;      /* u32[tiles]: tile handled by blend workgroup b (longest lists first, image region b % 8) */
} gvd_chunk_layout;

void gvd_raster_chunk_layout(int P, int width, int height, uint32_t num_rendered, gvd_chunk_layout* out);

/* Per-kernel timing with HIP events on the launch stream (for bench.py's roofline leg).
 * level 1 records only the two blend kernels (2 events per launch: negligible perturbation, used inside
 * the timed region), level 2 every kernel, 0 stops; gvd_profile_read sums the elapsed time (ms) and launch count of the
 * kernel `name` ("render_fwd", "render_bwd", "preprocess", ...) and clears nothing.  Reading
 * synchronises the recorded events. */
void gvd_profile_enable(int level);
void gvd_profile_reset(void);
int gvd_profile_read(const char* name, double* total_ms, int* launches);

const char* gvd_last_error(void);
const char* gvd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GVD_RASTER_H_INCLUDED */
