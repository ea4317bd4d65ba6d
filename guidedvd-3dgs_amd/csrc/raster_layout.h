// raster_layout.h -- host-side carving of the three scratch chunks (geometry / binning / image).
// Counterpart of GeometryState/BinningState/ImageState::fromChunk (rasterizer_impl.cu:155-195,
// rasterizer_impl.h:21-72); the layout is ours (DESIGN.md "data layout in HBM").
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace gvd {

constexpr size_t kAlign = 128;
constexpr int kMaxLdsHistTiles = 12288;  // 48 KiB of LDS tile counters per binning block
constexpr int kMaxBinBlocks = 1024;
constexpr int kPartialStride = 12;       // floats per (Gaussian, tile, quadrant) backward partial sub-record (4 per instance)

struct Layout {
    int P, W, H, gx, gy, T;
    int bin_blocks;      // B: binning blocks (each owns items_per_block consecutive Gaussians)
    int items_per_block; // multiple of 256
    int lds_hist;        // 1: per-block LDS tile histograms; 0: one global row + global atomics
    // geometry chunk
    size_t depths, means2D, conic_opacity, rgbd, cov3D, clamped, internal_radii, tiles_touched,
        point_offsets, scalars, hist, block_total, chunk_base, tile_count, cursor;
    size_t geom_bytes;
    // image chunk
    size_t ranges, n_contrib, tile_order;
    size_t img_bytes;
    // binning chunk (depends on capacity R)
    size_t keys, point_list, bucket, qmask, pflags, partials;
    size_t bin_bytes;
};

inline size_t carve(size_t& cur, size_t bytes)
{
    size_t off = (cur + kAlign - 1) & ~(kAlign - 1);
    cur = off + bytes;
    return off;
}

// `with_partials` = false: the binning chunk of a forward that no backward will follow (gvd_raster_expect_backward(0), allocator-callback
// entry point only): the flag words and the 4 x 48-byte sub-records per instance -- 196 of the chunk's 220 bytes per instance, times the
// speculative capacity's 12.5 % + 4096 of headroom -- are not laid out at all (advisor finding, round 5: ~2.2 GB at 10 M instances held
// per render in flight on a GPU shared with the diffusion model).  Such a chunk's size is not the size of any full layout (see below), so a
// backward handed it is refused by its size check instead of reading past the end.
inline Layout make_layout(int P, int W, int H, uint32_t R, bool with_partials = true)
{
    Layout L{};
    L.P = P; L.W = W; L.H = H;
    L.gx = (W + 15) / 16;
    L.gy = (H + 15) / 16;
    L.T = L.gx * L.gy;
    const int nb256 = (P + 255) / 256;
    const int k = nb256 > 0 ? (nb256 + kMaxBinBlocks - 1) / kMaxBinBlocks : 1;
    L.items_per_block = 256 * (k > 0 ? k : 1);
    L.bin_blocks = P > 0 ? (P + L.items_per_block - 1) / L.items_per_block : 1;
    L.lds_hist = (L.T <= kMaxLdsHistTiles) ? 1 : 0;
    const size_t Pz = (size_t)(P > 0 ? P : 1);
    size_t c = 0;
    L.depths = carve(c, Pz * 4);
    L.means2D = carve(c, Pz * 8);
    L.conic_opacity = carve(c, Pz * 16);
    L.rgbd = carve(c, Pz * 16);
    L.cov3D = carve(c, Pz * 24);
    L.clamped = carve(c, Pz * 4);
    L.internal_radii = carve(c, Pz * 4);
    L.tiles_touched = carve(c, Pz * 4);
    L.point_offsets = carve(c, Pz * 4);
    L.scalars = carve(c, 8 * 4);
    L.hist = carve(c, (size_t)(L.lds_hist ? L.bin_blocks : 1) * L.T * 4);
    L.block_total = carve(c, (size_t)L.bin_blocks * 4);
    L.chunk_base = carve(c, (size_t)L.bin_blocks * 4);
    L.tile_count = carve(c, (size_t)L.T * 4);
    L.cursor = carve(c, (size_t)L.T * 4);
    L.geom_bytes = c + kAlign;
    c = 0;
    L.ranges = carve(c, (size_t)L.T * 8);
    L.n_contrib = carve(c, (size_t)W * H * 4);
    L.tile_order = carve(c, (size_t)L.T * 4);
    L.img_bytes = c + kAlign;
    c = 0;
    const size_t Rz = (size_t)(R > 0 ? R : 1);
    L.keys = carve(c, Rz * 8);
    L.point_list = carve(c, Rz * 4);
    L.bucket = carve(c, Rz * 8);
    L.qmask = carve(c, Rz * 4);                           // 4 byte planes (quadrant) x list entry: the forward's conservative cull bit
    if (with_partials) {
        L.pflags = carve(c, Rz * 4);                      // byte q: quadrant q of the instance wrote its sub-record
        L.partials = carve(c, Rz * 4 * kPartialStride * 4);
        L.bin_bytes = c + kAlign;
    } else {
        L.pflags = L.partials = 0;                        // never dereferenced: the forward passes pflags = nullptr to k_scatter
        L.bin_bytes = c + kAlign + 2;                     // + 2: full layouts are multiples of 64 bytes, these are 2 mod 4 -- the size alone names the form
    }
    return L;
}

}  // namespace gvd
