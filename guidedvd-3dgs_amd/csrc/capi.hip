// capi.hip -- the C-ABI of include/gvd_raster.h: host orchestration of the HIP kernels.
// Counterpart of CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (cuda_rasterizer/rasterizer_impl.cu:141-153,197-339,343-447).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/gvd_raster.h"
#include "raster_kernels.h"
#include "raster_layout.h"


// spin-wait hint of the host's architecture (the ticket wait below): x86 `pause`, AArch64 `yield`, a compiler barrier elsewhere
static inline void gvd_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) return fail(GVD_ERR_HIP, #expr, e_);      \
    } while (0)

// CHECK_CUDA of auxiliary.h:166-173: in debug mode synchronise after each launch and report.
// GVD_RASTER_TRACE_LAUNCHES=1 (environment): the same synchronisation for every caller plus one stderr line per launch -- the last line
// printed before a GPU memory fault names the kernel.
static const bool g_trace_launches = getenv("GVD_RASTER_TRACE_LAUNCHES") != nullptr;
#define AFTER_LAUNCH(name)                                                              \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) return fail(GVD_ERR_HIP, "launch " name, e_);             \
        if (debug || g_trace_launches) {                                                \
            if (g_trace_launches) { fprintf(stderr, "[gvd] sync after %s\n", name); fflush(stderr); }   \
            e_ = hipStreamSynchronize(stream);                                          \
            if (e_ != hipSuccess) return fail(GVD_ERR_HIP, "kernel " name, e_);         \
        }                                                                               \
    } while (0)

// ---- per-kernel HIP-event timing (bench.py roofline leg) ----
struct ProfRec { const char* name; hipEvent_t a, b; };
std::mutex g_prof_mu;
int g_prof_level = 0;  // 0 off, 1 blend kernels only (cheap: used inside bench's timed region), 2 every kernel
std::vector<ProfRec> g_prof;

struct ProfScope {
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    bool on;
    ProfScope(const char* n, hipStream_t st) : s(st), name(n)
    {
        on = g_prof_level >= 2 || (g_prof_level == 1 && n[0] == 'r');  // "render_fwd" / "render_bwd"
        if (on) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_prof.push_back({ name, a, b });
        }
    }
};

// Host-side state of a forward call, one slot per (host thread, device): the host-visible mirror of {num_rendered, max tile
// list, ticket} that the tile scan writes and the host reads after its wait, and the ticket counter of the slot.  Nothing here
// is shared between threads or devices, so concurrent forwards on different streams / devices cannot read each other's R
// (the C-ABI only forbids re-entrancy on ONE stream).  Slots live as long as their thread (64 pinned bytes).
struct HostSlot { uint32_t* mirror = nullptr; uint32_t ticket = 0; };
HostSlot* host_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    static thread_local std::vector<HostSlot> slots;   // indexed by device ordinal
    if ((int)slots.size() <= dev) slots.resize((size_t)dev + 1);
    HostSlot& s = slots[(size_t)dev];
    if (!s.mirror) {
        if (hipHostMalloc((void**)&s.mirror, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { s.mirror = nullptr; return nullptr; }
        memset((void*)s.mirror, 0, 64);
    }
    return &s;
}

// ---- speculative stage 2 (exact API, no idle gap after the read-back) --------------------------------------------
// gvd_raster_forward must return the exact num_rendered, which only exists after stage 1 ran on the device.  Waiting
// for it before sizing the binning chunk leaves the GPU idle while the host wakes up, calls the allocator and queues
// stage 2 (30-80 us per render, host dependent).  Instead the binning chunk is sized from the largest num_rendered seen
// for the same (P, width, height) (+12.5 %), stage 2 is queued right behind stage 1 with that capacity, and only then
// does the host wait -- for stage 1 alone, through an event.  If the guess was too small (or the longest tile list
// needs a bigger sort class than guessed) everything is simply run again the exact way; the kernels are overflow-safe.
// The capacity a speculative chunk was laid out for travels WITH the chunk: its byte size determines it uniquely
// (gvd_raster_binning_capacity), and backward is told the size (gvd_raster_backward_conf).  Because the reference-signature
// gvd_raster_backward only receives num_rendered, speculation is opt-in: gvd_raster_set_speculation(1) by a caller that
// passes the chunk size to backward (the Python binding does).  GVD_RASTER_SPECULATE=0 turns it off altogether.
struct SpecHint { int P, W, H; uint32_t r_max, list_max; };
std::mutex g_spec_mu;
std::vector<SpecHint> g_hints;
std::atomic<int> g_spec_opt_in{0};
// k_render_bwd: quadrant walks longer than this are cut into units (raster_backward.hip); GVD_BWD_SPLIT / gvd_raster_set_backward_split
std::atomic<int> g_bwd_split{ getenv("GVD_BWD_SPLIT") ? atoi(getenv("GVD_BWD_SPLIT")) : 512 };
// gvd_raster_expect_backward: per host thread; 1 (default) = every forward prepares its binning chunk for a backward
thread_local int t_expect_backward = 1;

bool spec_enabled()
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("GVD_RASTER_SPECULATE"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1 && g_spec_opt_in.load(std::memory_order_relaxed) != 0;
}
bool spec_lookup(int P, int W, int H, SpecHint* out)
{
    std::lock_guard<std::mutex> lk(g_spec_mu);
    for (const SpecHint& h : g_hints) if (h.P == P && h.W == W && h.H == H) { *out = h; return true; }
    return false;
}
void spec_update(int P, int W, int H, uint32_t R, uint32_t max_list)
{
    std::lock_guard<std::mutex> lk(g_spec_mu);
    for (SpecHint& h : g_hints)
        if (h.P == P && h.W == W && h.H == H) { if (R > h.r_max) h.r_max = R; if (max_list > h.list_max) h.list_max = max_list; return; }
    if (g_hints.size() >= 16) g_hints.erase(g_hints.begin());
    g_hints.push_back(SpecHint{ P, W, H, R, max_list });
}
// Capacity whose binning layout occupies exactly `bytes` (bin_bytes is strictly increasing in the capacity: the last
// sub-array, the 48-byte partial records, ends the chunk unrounded).  Returns false if no capacity matches.
bool capacity_of_bytes(size_t bytes, uint32_t* cap_out, bool* compact_out = nullptr)
{
    // two forms: the full layout, and the compact one of a forward run under gvd_raster_expect_backward(0) (no partial records; full
    // sizes are multiples of 64, compact ones 2 mod 4: the size alone names the form).  keys / point_list / bucket / qmask sit at the same
    // offsets in both.
    for (int compact = 0; compact < 2; ++compact) {
        uint32_t lo = 0, hi = 0xfffffff0u;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (gvd::make_layout(0, 16, 16, mid, !compact).bin_bytes < bytes) lo = mid + 1; else hi = mid;
        }
        if (gvd::make_layout(0, 16, 16, lo, !compact).bin_bytes != bytes) continue;
        // capacities 0 and 1 share one layout (every sub-array holds at least one element): report 1, or a chunk that holds exactly one
        // instance would come back as "smaller than num_rendered requires" (found by tests/scripts/r5_raster_stress.py: P = 1, R = 1)
        *cap_out = lo ? lo : 1u;
        if (compact_out) *compact_out = compact != 0;
        return true;
    }
    return false;
}
inline int sort_class_of(uint32_t max_list) { return max_list > 16384 ? 2 : (max_list > 2048 ? 1 : 0); }

inline char* align_up(char* p) { return (char*)(((uintptr_t)p + gvd::kAlign - 1) & ~(uintptr_t)(gvd::kAlign - 1)); }

struct FwdIn {
    int P, D, M, width, height, prefiltered, debug;
    const float *background, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *viewmatrix, *projmatrix, *cam_pos;
    float scale_modifier, tan_fovx, tan_fovy;
    float *out_color, *out_depth, *out_alpha;
    int* radii;
};

// Stage 1: preprocess + column scan (+ the tile scan as its own launch when `scan_launch`; otherwise *ts_out is handed to the
// scatter launch, which then carries the scan: forward_stage2 with fused_scan).
int forward_stage1(const FwdIn& in, const gvd::Layout& L, char* geom, char* img, uint32_t capacity,
                   int32_t* d_status, volatile uint32_t* mirror, uint32_t ticket, bool scan_launch, gvd::TileScanArgs* ts_out,
                   hipStream_t stream)
{
    using namespace gvd;
    const int debug = in.debug;
    PreprocessArgs pa{};
    pa.P = in.P; pa.D = in.D; pa.M = in.M; pa.W = in.width; pa.H = in.height;
    pa.gx = L.gx; pa.gy = L.gy; pa.T = L.T; pa.items_per_block = L.items_per_block; pa.prefiltered = in.prefiltered;
    pa.scale_modifier = in.scale_modifier; pa.tan_fovx = in.tan_fovx; pa.tan_fovy = in.tan_fovy;
    pa.focal_y = in.height / (2.0f * in.tan_fovy);  // rasterizer_impl.cu:223-224
    pa.focal_x = in.width / (2.0f * in.tan_fovx);
    pa.means3D = in.means3D; pa.scales = in.scales; pa.rotations = in.rotations; pa.opacities = in.opacities;
    pa.shs = in.shs; pa.cov3D_precomp = in.cov3D_precomp; pa.colors_precomp = in.colors_precomp;
    pa.viewmatrix = in.viewmatrix; pa.projmatrix = in.projmatrix; pa.cam_pos = in.cam_pos;
    pa.radii = in.radii ? in.radii : (int*)(geom + L.internal_radii);
    pa.means2D = (float*)(geom + L.means2D); pa.depths = (float*)(geom + L.depths);
    pa.cov3D = (float*)(geom + L.cov3D); pa.rgbd = (float*)(geom + L.rgbd);
    pa.conic_opacity = (float*)(geom + L.conic_opacity);
    pa.clamped = (uint32_t*)(geom + L.clamped); pa.tiles_touched = (uint32_t*)(geom + L.tiles_touched);
    pa.hist = (uint32_t*)(geom + L.hist); pa.block_total = (uint32_t*)(geom + L.block_total);
    if (!L.lds_hist) HIP_TRY(hipMemsetAsync(geom + L.hist, 0, (size_t)L.T * 4, stream));
    {
        ProfScope ps("preprocess", stream);
        launch_preprocess(pa, L.bin_blocks, L.lds_hist != 0, stream);
    }
    AFTER_LAUNCH("preprocess");
    uint32_t* tile_count = L.lds_hist ? (uint32_t*)(geom + L.tile_count) : (uint32_t*)(geom + L.hist);
    TileScanArgs ta{};
    ta.T = L.T; ta.B = L.bin_blocks; ta.gx = L.gx; ta.capacity = capacity;
    ta.tile_count = tile_count; ta.block_total = (uint32_t*)(geom + L.block_total);
    ta.ranges = (uint32_t*)(img + L.ranges);
    ta.cursor = L.lds_hist ? nullptr : (uint32_t*)(geom + L.cursor);
    ta.chunk_base = (uint32_t*)(geom + L.chunk_base);
    ta.scalars = (uint32_t*)(geom + L.scalars);
    ta.tile_order = (uint32_t*)(img + L.tile_order);
    ta.d_status = d_status;
    ta.host_mirror = mirror;
    ta.ticket = ticket;
    if (L.lds_hist) {
        ProfScope ps("colscan", stream);
        launch_colscan((uint32_t*)(geom + L.hist), tile_count, L.bin_blocks, L.T, stream);
    }
    AFTER_LAUNCH("colscan");
    if (ts_out) *ts_out = ta;
    if (scan_launch) {
        ProfScope ps("tilescan", stream);
        launch_tilescan(ta, stream);
    }
    AFTER_LAUNCH("tilescan");
    return GVD_OK;
}

int forward_stage2(const FwdIn& in, const gvd::Layout& L, char* geom, char* bin, char* img, uint32_t capacity,
                   int max_class, const gvd::TileScanArgs* fused_scan, hipStream_t stream)
{
    using namespace gvd;
    const int debug = in.debug;
    const int* radii = in.radii ? in.radii : (const int*)(geom + L.internal_radii);
    ScatterArgs sa{};
    sa.P = in.P; sa.gx = L.gx; sa.gy = L.gy; sa.T = L.T; sa.items_per_block = L.items_per_block; sa.capacity = capacity;
    sa.tiles_touched = (const uint32_t*)(geom + L.tiles_touched); sa.hist = (const uint32_t*)(geom + L.hist);
    sa.ranges = (const uint32_t*)(img + L.ranges); sa.chunk_base = (const uint32_t*)(geom + L.chunk_base);
    sa.means2D = (const float*)(geom + L.means2D); sa.depths = (const float*)(geom + L.depths);
    sa.radii = radii; sa.cursor = (uint32_t*)(geom + L.cursor);
    sa.point_offsets = (uint32_t*)(geom + L.point_offsets); sa.bucket = (uint64_t*)(bin + L.bucket);
    // zeroed here for the backward (no memset launch there) -- unless the caller announced that none will follow
    sa.pflags = t_expect_backward ? (uint32_t*)(bin + L.pflags) : nullptr;
    {
        ProfScope ps("scatter", stream);
        launch_scatter(sa, fused_scan, L.bin_blocks, L.lds_hist != 0, stream);
    }
    AFTER_LAUNCH("scatter");
    SortArgs so{};
    so.capacity = capacity; so.ranges = (const uint32_t*)(img + L.ranges); so.bucket = (uint64_t*)(bin + L.bucket);
    so.point_list = (uint32_t*)(bin + L.point_list); so.keys = (uint64_t*)(bin + L.keys);
    static const bool fused_sort = !(getenv("GVD_RASTER_FUSED_SORT") && atoi(getenv("GVD_RASTER_FUSED_SORT")) == 0);
    if (!fused_sort || max_class >= 1) {
        ProfScope ps("sort_tiles", stream);
        launch_sort_tiles(so, L.T, max_class, !fused_sort, stream);
    }
    AFTER_LAUNCH("sort_tiles");
    RenderArgs ra{};
    ra.W = in.width; ra.H = in.height; ra.gx = L.gx; ra.gy = L.gy; ra.capacity = capacity;
    ra.fused_sort = fused_sort ? 1 : 0; ra.bucket = so.bucket; ra.keys = so.keys;
    ra.ranges = (const uint32_t*)(img + L.ranges); ra.point_list = (uint32_t*)(bin + L.point_list);
    ra.means2D = (const float*)(geom + L.means2D); ra.conic_opacity = (const float*)(geom + L.conic_opacity);
    ra.rgbd = (const float*)(geom + L.rgbd); ra.bg = in.background;
    ra.out_color = in.out_color; ra.out_depth = in.out_depth; ra.out_alpha = in.out_alpha;
    ra.n_contrib = (uint32_t*)(img + L.n_contrib);
    ra.tile_order = (const uint32_t*)(img + L.tile_order);
    ra.qmask = (uint8_t*)(bin + L.qmask);
    // what the sorts queued above cover (k_sort_tiles<1>: <= 16384, <2>: any length; the blend kernel's own sort: <= kFusedSortMax)
    ra.sorted_limit = max_class >= 2 ? 0xffffffffu : (max_class == 1 ? 16384u : kFusedSortMax);
    {
        ProfScope ps("render_fwd", stream);
        launch_render_fwd(ra, L.T, stream);
    }
    AFTER_LAUNCH("render_fwd");
    return GVD_OK;
}

int check_inputs(const FwdIn& in)
{
    if (in.P < 0 || in.width <= 0 || in.height <= 0) return fail(GVD_ERR_INVALID, "bad P/width/height");
    if (in.P > 0) {
        if (!in.means3D || !in.opacities || !in.viewmatrix || !in.projmatrix || !in.cam_pos || !in.background)
            return fail(GVD_ERR_INVALID, "null required input");
        // rasterizer_impl.cu:243-246 generalised: a colour source must exist
        if (!in.colors_precomp && (!in.shs || in.M <= 0)) return fail(GVD_ERR_INVALID, "Please provide SHs or precomputed colors!");
        if (!in.cov3D_precomp && (!in.scales || !in.rotations)) return fail(GVD_ERR_INVALID, "Please provide scale/rotation or precomputed 3D covariance!");
        if (!in.colors_precomp && (in.D < 0 || (in.D + 1) * (in.D + 1) > in.M || in.D > 3)) return fail(GVD_ERR_INVALID, "SH degree/M mismatch (degree 0..3, M >= (D+1)^2)");
    }
    if (!in.out_color || !in.out_depth || !in.out_alpha) return fail(GVD_ERR_INVALID, "null output");
    return GVD_OK;
}

int zero_outputs(const FwdIn& in, hipStream_t stream)
{
    const size_t HW = (size_t)in.width * in.height;
    HIP_TRY(hipMemsetAsync(in.out_color, 0, HW * 3 * 4, stream));
    HIP_TRY(hipMemsetAsync(in.out_depth, 0, HW * 4, stream));
    HIP_TRY(hipMemsetAsync(in.out_alpha, 0, HW * 4, stream));
    return GVD_OK;
}

}  // namespace

extern "C" {

const char* gvd_last_error(void) { return g_err.c_str(); }
const char* gvd_version(void) { return "gvd-raster 0.1 (gfx950)"; }

size_t gvd_raster_geometry_bytes(int P, int width, int height) { return gvd::make_layout(P, width, height, 0).geom_bytes; }
size_t gvd_raster_image_bytes(int width, int height) { return gvd::make_layout(0, width, height, 0).img_bytes; }
size_t gvd_raster_binning_bytes(uint32_t num_rendered) { return gvd::make_layout(0, 16, 16, num_rendered).bin_bytes; }
size_t gvd_raster_binning_bytes_no_backward(uint32_t num_rendered) { return gvd::make_layout(0, 16, 16, num_rendered, false).bin_bytes; }

void gvd_raster_chunk_layout(int P, int width, int height, uint32_t num_rendered, gvd_chunk_layout* o)
{
    const gvd::Layout L = gvd::make_layout(P, width, height, num_rendered);
    o->depths = L.depths; o->means2D = L.means2D; o->conic_opacity = L.conic_opacity; o->rgbd = L.rgbd;
    o->cov3D = L.cov3D; o->clamped = L.clamped; o->internal_radii = L.internal_radii;
    o->tiles_touched = L.tiles_touched; o->point_offsets = L.point_offsets; o->scalars = L.scalars;
    o->ranges = L.ranges; o->n_contrib = L.n_contrib;
    o->point_list_keys = L.keys; o->point_list = L.point_list; o->bucket = L.bucket; o->tile_order = L.tile_order;
}

uint32_t gvd_raster_binning_capacity(size_t binning_chunk_bytes)
{
    uint32_t cap = 0;
    return capacity_of_bytes(binning_chunk_bytes, &cap) ? cap : 0xffffffffu;
}

void gvd_raster_set_speculation(int on) { g_spec_opt_in.store(on ? 1 : 0, std::memory_order_relaxed); }
void gvd_raster_set_backward_split(int entries) { g_bwd_split.store(entries > 0 ? entries : 0, std::memory_order_relaxed); }
void gvd_raster_expect_backward(int yes) { t_expect_backward = yes ? 1 : 0; }

int gvd_raster_forward(
    gvd_alloc_fn geometry_alloc, void* geometry_user, gvd_alloc_fn binning_alloc, void* binning_user,
    gvd_alloc_fn image_alloc, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    FwdIn in{ P, D, M, width, height, prefiltered, debug, background, means3D, shs, colors_precomp, opacities,
              scales, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, scale_modifier, tan_fovx, tan_fovy,
              out_color, out_depth, out_alpha, radii };
    int rc = check_inputs(in);
    if (rc != GVD_OK) return rc;
    if (P == 0) {  // rasterize_points.cu:81: nothing runs, outputs stay torch::full(0)
        rc = zero_outputs(in, stream);
        return rc != GVD_OK ? rc : 0;
    }
    if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(GVD_ERR_INVALID, "null allocator");
    gvd::Layout L = gvd::make_layout(P, width, height, 0);
    char* geom = geometry_alloc(geometry_user, L.geom_bytes);
    char* img = image_alloc(image_user, L.img_bytes);
    if (!geom || !img) return fail(GVD_ERR_ALLOC, "geometry/image allocator returned NULL");
    geom = align_up(geom);
    img = align_up(img);
    HostSlot* slot = host_slot();
    if (!slot) return fail(GVD_ERR_HIP, "hipHostMalloc(mirror) failed");
    volatile uint32_t* mirror = slot->mirror;
    SpecHint hint;
    if (spec_enabled() && spec_lookup(P, width, height, &hint) && hint.r_max > 0 && hint.r_max < 0x60000000u) {
        {
            const uint32_t cap = hint.r_max + hint.r_max / 8 + 4096;
            const int class_spec = sort_class_of(hint.list_max + hint.list_max / 4);
            gvd::Layout Ls = gvd::make_layout(P, width, height, cap, t_expect_backward != 0);
            char* bin_s = binning_alloc(binning_user, Ls.bin_bytes);
            if (!bin_s) return fail(GVD_ERR_ALLOC, "binning allocator returned NULL");
            bin_s = align_up(bin_s);
            // the scatter launch carries the tile scan (one launch less in front of the blend); the host then waits for the
            // ticket the scan writes next to {num_rendered, max list} in the pinned mirror -- a spin on host memory instead of
            // an event: the wake-up does not wait for the end of the launch, let alone a completion signal
            const bool fuse = Ls.lds_hist != 0;
            const uint32_t ticket = ++slot->ticket;
            gvd::TileScanArgs ts{};
            rc = forward_stage1(in, Ls, geom, img, cap, nullptr, mirror, ticket, !fuse, &ts, stream);
            if (rc != GVD_OK) return rc;
            rc = forward_stage2(in, Ls, geom, bin_s, img, cap, class_spec, fuse ? &ts : nullptr, stream);
            if (rc != GVD_OK) return rc;
            {
                unsigned long long spins = 0;
                while (mirror[2] != ticket) {
                    gvd_cpu_relax();
                    if ((++spins & 0xfffffull) == 0 && hipStreamQuery(stream) != hipErrorNotReady) {   // ~every few ms: the stream died or drained
                        if (mirror[2] == ticket) break;
                        HIP_TRY(hipStreamSynchronize(stream));
                        if (mirror[2] != ticket) return fail(GVD_ERR_HIP, "tile scan finished without publishing num_rendered");
                    }
                }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            const uint32_t Rs = mirror[0], max_list_s = mirror[1];
            if (Rs > 0x7fffffffu) return fail(GVD_ERR_OVERFLOW, "num_rendered exceeds int32");
            spec_update(P, width, height, Rs, max_list_s);
            if (Rs <= cap && sort_class_of(max_list_s) <= class_spec) return (int)Rs;   // (the chunk's size encodes `cap`)
            // guessed too small: fall through and run the exact path (stage 1 again, with no capacity limit)
        }
    }
    rc = forward_stage1(in, L, geom, img, 0xffffffffu, nullptr, mirror, ++slot->ticket, true, nullptr, stream);
    if (rc != GVD_OK) return rc;
    // the one host sync of the forward (reference: cudaMemcpy at rasterizer_impl.cu:282)
    HIP_TRY(hipStreamSynchronize(stream));
    const uint32_t R = mirror[0];
    const uint32_t max_list = mirror[1];
    if (R > 0x7fffffffu) return fail(GVD_ERR_OVERFLOW, "num_rendered exceeds int32");
    spec_update(P, width, height, R, max_list);
    L = gvd::make_layout(P, width, height, R, t_expect_backward != 0);   // (no backward expected: no partial records in the chunk)
    char* bin = binning_alloc(binning_user, L.bin_bytes);
    if (!bin) return fail(GVD_ERR_ALLOC, "binning allocator returned NULL");
    bin = align_up(bin);
    const int max_class = sort_class_of(max_list);
    rc = forward_stage2(in, L, geom, bin, img, R, max_class, nullptr, stream);
    if (rc != GVD_OK) return rc;
    return (int)R;
}

int gvd_raster_forward_capped(
    char* geometry_chunk, char* binning_chunk, char* image_chunk, uint32_t capacity,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int32_t* d_status, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    FwdIn in{ P, D, M, width, height, prefiltered, debug, background, means3D, shs, colors_precomp, opacities,
              scales, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, scale_modifier, tan_fovx, tan_fovy,
              out_color, out_depth, out_alpha, radii };
    int rc = check_inputs(in);
    if (rc != GVD_OK) return rc;
    if (P == 0) {
        if (d_status) HIP_TRY(hipMemsetAsync(d_status, 0, 4, stream));
        return zero_outputs(in, stream);
    }
    if (!geometry_chunk || !binning_chunk || !image_chunk) return fail(GVD_ERR_INVALID, "null chunk");
    const gvd::Layout L = gvd::make_layout(P, width, height, capacity);
    char* geom = align_up(geometry_chunk);
    char* img = align_up(image_chunk);
    char* bin = align_up(binning_chunk);
    const bool fuse = L.lds_hist != 0;
    gvd::TileScanArgs ts{};
    rc = forward_stage1(in, L, geom, img, capacity, d_status, nullptr, 0, !fuse, &ts, stream);
    if (rc != GVD_OK) return rc;
    return forward_stage2(in, L, geom, bin, img, capacity, 2, fuse ? &ts : nullptr, stream);
}

int gvd_raster_backward_conf(
    int P, int D, int M, int R, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
    const float* confidence, size_t binning_chunk_bytes, int debug, void* stream_)
{
    using namespace gvd;
    hipStream_t stream = (hipStream_t)stream_;
    if (P == 0) return GVD_OK;
    if (P < 0 || R < 0 || !geom_buffer || !binning_buffer || !image_buffer) return fail(GVD_ERR_INVALID, "bad backward arguments");
    if (!dL_dpix || !alphas) return fail(GVD_ERR_INVALID, "null pixel gradient");
    if (!dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_ddepth || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot ||
        (M > 0 && !dL_dsh))
        return fail(GVD_ERR_INVALID, "null gradient output");
    char* geom = align_up(geom_buffer);
    char* bin = align_up(binning_buffer);
    char* img = align_up(image_buffer);
    // layout capacity of this chunk: num_rendered itself, unless the forward laid it out speculatively -- then its size says
    uint32_t cap = (uint32_t)R;
    if (binning_chunk_bytes) {
        bool compact = false;
        if (!capacity_of_bytes(binning_chunk_bytes, &cap, &compact)) return fail(GVD_ERR_INVALID, "binning_chunk_bytes is not the size of a binning chunk");
        if (compact) return fail(GVD_ERR_INVALID, "this binning chunk was laid out without the backward's partial records: its forward ran under gvd_raster_expect_backward(0)");
        if (cap < (uint32_t)R) return fail(GVD_ERR_INVALID, "binning chunk is smaller than num_rendered requires");
    }
    const Layout L = make_layout(P, width, height, cap);
    if (!radii) radii = (const int*)(geom + L.internal_radii);
    float* partials = (float*)(bin + L.partials);   // sub-records; their flag words were cleared by the forward's k_scatter
    uint32_t* pflags = (uint32_t*)(bin + L.pflags);
    RenderBwdArgs ra{};
    ra.W = width; ra.H = height; ra.gx = L.gx; ra.gy = L.gy; ra.capacity = cap;
    ra.ranges = (const uint32_t*)(img + L.ranges); ra.point_list = (const uint32_t*)(bin + L.point_list);
    ra.tile_order = (const uint32_t*)(img + L.tile_order);
    ra.n_contrib = (const uint32_t*)(img + L.n_contrib); ra.point_offsets = (const uint32_t*)(geom + L.point_offsets);
    ra.scalars = (const uint32_t*)(geom + L.scalars);
    ra.radii = radii; ra.means2D = (const float*)(geom + L.means2D); ra.conic_opacity = (const float*)(geom + L.conic_opacity);
    ra.rgbd = (const float*)(geom + L.rgbd); ra.bg = background; ra.alphas = alphas;
    ra.dL_dpix = dL_dpix; ra.dL_dpix_depth = dL_dpix_depth; ra.dL_dalphas = dL_dalphas; ra.partials = partials; ra.pflags = pflags; ra.qmask = (const uint8_t*)(bin + L.qmask);
    ra.split_len = (uint32_t)g_bwd_split.load(std::memory_order_relaxed);
    // the longest lists come first in tile_order: only the first quarter of the positions can be cut (their extra segment units head the grid)
    ra.split_positions = ra.split_len ? (uint32_t)((((L.T + 3) / 4) + 7) / 8 * 8) : 0u;
    ra.extra_units = ra.split_positions * 12u;
    {
        ProfScope ps("render_bwd", stream);
        launch_render_bwd(ra, L.T, stream);
    }
    AFTER_LAUNCH("render_bwd");
    GatherBwdArgs ga{};
    ga.P = P; ga.D = D; ga.M = M; ga.W = width; ga.H = height;
    ga.scale_modifier = scale_modifier; ga.tan_fovx = tan_fovx; ga.tan_fovy = tan_fovy;
    ga.focal_y = height / (2.0f * tan_fovy); ga.focal_x = width / (2.0f * tan_fovx);
    ga.means3D = means3D; ga.shs = shs; ga.scales = scales; ga.rotations = rotations;
    ga.cov3D = cov3D_precomp ? cov3D_precomp : (const float*)(geom + L.cov3D);
    ga.viewmatrix = viewmatrix; ga.projmatrix = projmatrix; ga.campos = campos; ga.radii = radii;
    ga.clamped = (const uint32_t*)(geom + L.clamped); ga.point_offsets = (const uint32_t*)(geom + L.point_offsets);
    ga.scalars = (const uint32_t*)(geom + L.scalars);
    ga.partials = partials; ga.pflags = pflags;
    ga.confidence = confidence;
    ga.has_sh = (shs != nullptr && M > 0 && colors_precomp == nullptr) ? 1 : 0;
    ga.has_scales = (scales != nullptr && rotations != nullptr && cov3D_precomp == nullptr) ? 1 : 0;
    ga.dL_dmean2D = dL_dmean2D; ga.dL_dconic = dL_dconic; ga.dL_dopacity = dL_dopacity; ga.dL_dcolor = dL_dcolor;
    ga.dL_ddepth = dL_ddepth; ga.dL_dmean3D = dL_dmean3D; ga.dL_dcov3D = dL_dcov3D; ga.dL_dsh = dL_dsh;
    ga.dL_dscale = dL_dscale; ga.dL_drot = dL_drot;
    {
        ProfScope ps("gather_bwd", stream);
        launch_gather_bwd(ga, stream);
    }
    AFTER_LAUNCH("gather_bwd");
    return GVD_OK;
}

int gvd_raster_backward(
    int P, int D, int M, int R, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
    int debug, void* stream_)
{
    return gvd_raster_backward_conf(P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales,
                                    scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx,
                                    tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dpix_depth,
                                    dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D,
                                    dL_dcov3D, dL_dsh, dL_dscale, dL_drot, nullptr, 0, debug, stream_);
}

int gvd_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                            uint8_t* present, void* stream_)
{
    (void)projmatrix;
    hipStream_t stream = (hipStream_t)stream_;
    if (P == 0) return GVD_OK;
    if (P < 0 || !means3D || !viewmatrix || !present) return fail(GVD_ERR_INVALID, "bad mark_visible arguments");
    gvd::launch_mark_visible(P, means3D, viewmatrix, present, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GVD_ERR_HIP, "launch mark_visible", e);
    return GVD_OK;
}

void gvd_profile_enable(int level) { g_prof_level = level; }

void gvd_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
}

int gvd_profile_read(const char* name, double* total_ms, int* launches)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0;
    int n = 0;
    for (auto& r : g_prof) {
        if (strcmp(r.name, name) != 0) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return GVD_ERR_HIP;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return GVD_ERR_HIP;
        tot += ms;
        n++;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return GVD_OK;
}

}  // extern "C"
