// raster_kernels.h -- kernel argument blocks and launcher prototypes shared by the .hip units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gvd {

struct PreprocessArgs {
    int P, D, M, W, H, gx, gy, T, items_per_block, prefiltered;
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y;
    const float *means3D, *scales, *rotations, *opacities, *shs, *cov3D_precomp, *colors_precomp;
    const float *viewmatrix, *projmatrix, *cam_pos;
    int* radii;
    float *means2D, *depths, *cov3D, *rgbd, *conic_opacity;
    uint32_t *clamped, *tiles_touched, *hist, *block_total;
};

struct TileScanArgs {
    int T, B, gx;
    uint32_t capacity;
    const uint32_t* tile_count;
    const uint32_t* block_total;
    uint32_t *ranges, *cursor, *chunk_base, *scalars, *tile_order;
    int32_t* d_status;
    volatile uint32_t* host_mirror;
    uint32_t ticket;   // written to host_mirror[2] after the data words: the host waits for it
};

struct ScatterArgs {
    int P, gx, gy, T, items_per_block;
    uint32_t capacity;
    const uint32_t *tiles_touched, *hist, *ranges, *chunk_base;
    const float *means2D, *depths;
    const int* radii;
    uint32_t *cursor, *point_offsets;
    uint64_t* bucket;
    uint32_t* pflags;  // backward's per-instance sub-record flags (4 bytes per instance), zeroed per block; NULL: leave alone
};

struct SortArgs {
    uint32_t capacity;
    const uint32_t* ranges;
    uint64_t* bucket;
    uint32_t* point_list;
    uint64_t* keys;
};

struct RenderArgs {
    int W, H, gx, gy;
    uint32_t capacity;
    int fused_sort;       // 1: lists of <= kFusedSortMax entries are still unsorted in `bucket`; the tile's workgroup sorts its own
    uint64_t* bucket;     //    list in LDS first and writes bucket / point_list / keys (what k_sort_tiles<0> would have written)
    uint64_t* keys;
    uint32_t* point_list;
    const uint32_t *ranges, *tile_order;
    const float *means2D, *conic_opacity, *rgbd, *bg;
    float *out_color, *out_depth, *out_alpha;
    uint32_t* n_contrib;
    uint8_t* qmask;       // [4][capacity]: plane w, list position: 1 = the entry passed wave w's quadrant test (read by k_render_bwd)
    uint32_t sorted_limit; // longest list that has been sorted for this launch (by the sort kernels queued before it, or here): a speculative
                           // forward guesses the sort class; a list longer than the guess covers has NO point_list yet (uninitialised ids)
                           // and must not be walked -- the host re-runs the exact path for such a forward anyway
};

struct RenderBwdArgs {
    int W, H, gx, gy;
    uint32_t capacity;
    const uint32_t *ranges, *point_list, *n_contrib, *point_offsets, *tile_order;
    const uint32_t* scalars;  // [2] != 0: the (capped) forward overflowed its instance capacity -> no work, zero gradients
    const int* radii;
    const float *means2D, *conic_opacity, *rgbd, *bg, *alphas;
    const float *dL_dpix, *dL_dpix_depth, *dL_dalphas;
    float* partials;  // [R][4 quadrants][12]
    uint32_t* pflags; // [R]: byte q set = sub-record (instance, q) written
    const uint8_t* qmask;  // [4][capacity]: the forward's quadrant cull bits
    uint32_t split_len;    // quadrant walks longer than this many entries are cut into up to 4 units of about this length ...
    uint32_t split_positions, extra_units;   // ... for the first split_positions positions of tile_order (multiple of 8); extra_units = 12 x that, the head of the grid
};

struct GatherBwdArgs {
    int P, D, M, W, H;
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y;
    const float *means3D, *shs, *scales, *rotations, *cov3D, *viewmatrix, *projmatrix, *campos;
    const int* radii;
    const uint32_t *clamped, *point_offsets, *scalars;
    const float* partials;
    const uint32_t* pflags;
    const float* confidence;  // [P] or NULL
    int has_sh, has_scales;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_ddepth;
    float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
};

void launch_preprocess(const PreprocessArgs& a, int blocks, bool lds_hist, hipStream_t s);
void launch_colscan(uint32_t* hist, uint32_t* tile_count, int B, int T, hipStream_t s);
void launch_tilescan(const TileScanArgs& a, hipStream_t s);
void launch_scatter(const ScatterArgs& a, const TileScanArgs* fused_scan, int blocks, bool lds_hist, hipStream_t s);   // fused_scan: the launch also runs the tile scan (lds_hist only)
void launch_sort_tiles(const SortArgs& a, int T, int max_class, bool short_lists_too, hipStream_t s);
constexpr uint32_t kFusedSortMax = 2048;  // longest list a blend workgroup sorts itself (16 KiB of LDS)
void launch_render_fwd(const RenderArgs& a, int T, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);
void launch_render_bwd(const RenderBwdArgs& a, int T, hipStream_t s);
void launch_gather_bwd(const GatherBwdArgs& a, hipStream_t s);

}  // namespace gvd
