// guard_allocator.cpp -- a red-zone device allocator for torch (torch.cuda.memory.CUDAPluggableAllocator), TEST INFRASTRUCTURE.
//
// Verdict r5 item 6: an out-of-bounds read lived through four rounds of green parity tests because torch's caching allocator hands out
// sub-blocks of large, mapped, recycled segments -- a kernel that strays past its buffer reads (or writes) a neighbour and nobody sees it.
// Under this allocator every torch allocation -- the rasterizer's three chunks, every output, every workspace of the diffusion kernels -- is
// a hipMalloc of its own with a poisoned red zone on both sides and a poisoned body:
//   * a WRITE past either end lands in a red zone and is reported when the block is freed (and by gvd_guard_check_all());
//   * a READ past either end, or of a byte the producer never wrote, returns 0xFB bytes: -2.6e36 as fp32, -61 280 as fp16, 4 227 595 259 as
//     an index -- results that break the bit-equality / tolerance checks of the stress scripts, or fault outright when used as an index;
//   * a read far past the end leaves the mapping (each block is its own allocation) and faults.
// Usage: tests/scripts/r6_guard_run.py <script> [args] installs it before the first device allocation and runs the script under it.
// No product code knows about it.  Build: hipcc -shared -fPIC (tests/guard/build.sh); only the HIP runtime API is used.
#ifdef GVD_GUARD_HOST_FAKE
// Host-only build for the CPU test of the allocator's OWN logic (red-zone arithmetic, detection, reporting): the six HIP calls it makes are mapped to
// malloc / memset / memcpy.  tests/test_guard_allocator_gpu.py::test_guard_allocator_logic_on_the_host builds and drives this form; the real build
// (tests/guard/build.sh) uses the HIP runtime.
#include <sys/types.h>
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipMemcpyDeviceToHost = 2 };
#include <stdlib.h>
#include <string.h>
static hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 4096, n) == 0 ? hipSuccess : 2; }   // (hipMalloc returns page-aligned blocks)
static hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static hipError_t hipDeviceSynchronize() { return hipSuccess; }
static hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static hipError_t hipSetDevice(int) { return hipSuccess; }
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

constexpr unsigned char kPoison = 0xFB;

struct Block {
    char* base;
    size_t size;      // bytes the caller asked for
    size_t total;     // red zone + rounded body + red zone
    int device;
    unsigned long long serial;
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_live;
unsigned long long g_serial = 0, g_allocs = 0, g_frees = 0, g_violations = 0, g_peak = 0, g_now = 0;
size_t g_rz = 0;

size_t red_zone()
{
    if (!g_rz) {
        const char* e = getenv("GVD_GUARD_REDZONE");
        size_t v = e ? (size_t)strtoull(e, nullptr, 10) : 4096;
        g_rz = (v < 256 ? 256 : v + 255) & ~(size_t)255;   // multiples of 256 keep torch's 256-byte alignment promise of the body
    }
    return g_rz;
}

void report(const Block& b, const char* side, long long first, long long last, size_t count)
{
    ++g_violations;
    fprintf(stderr, "[gvd_guard] VIOLATION: block #%llu of %zu bytes (device %d): %zu byte(s) of the red zone %s the body were overwritten, "
                    "first at offset %+lld, last at %+lld relative to the body's %s\n",
            b.serial, b.size, b.device, count, side, first, last, side[0] == 'b' ? "start" : "end");
    const char* log = getenv("GVD_GUARD_LOG");
    if (log) {
        FILE* f = fopen(log, "a");
        if (f) {
            fprintf(f, "violation block=%llu size=%zu side=%s first=%lld last=%lld count=%zu\n", b.serial, b.size, side, first, last, count);
            fclose(f);
        }
    }
}

// red zones of one block -> host, compared with the poison.  The caller has synchronised the device.
void check_block(const Block& b)
{
    const size_t rz = red_zone();
    std::vector<unsigned char> h(rz);
    // in front of the body
    if (hipMemcpy(h.data(), b.base, rz, hipMemcpyDeviceToHost) == hipSuccess) {
        size_t first = 0, last = 0, n = 0;
        for (size_t i = 0; i < rz; ++i)
            if (h[i] != kPoison) { if (!n) first = i; last = i; ++n; }
        if (n) report(b, "before", (long long)first - (long long)rz, (long long)last - (long long)rz, n);   // negative offsets from the body's start
    }
    // behind the body: from the first byte past `size` (the rounding slack belongs to the zone) to the end of the block
    const size_t tail_off = rz + b.size;
    const size_t tail = b.total - tail_off;
    h.resize(tail);
    if (hipMemcpy(h.data(), b.base + tail_off, tail, hipMemcpyDeviceToHost) == hipSuccess) {
        size_t first = 0, last = 0, n = 0;
        for (size_t i = 0; i < tail; ++i)
            if (h[i] != kPoison) { if (!n) first = i; last = i; ++n; }
        if (n) report(b, "after", (long long)first, (long long)last, n);
    }
}

}  // namespace

extern "C" {

void* gvd_guard_malloc(ssize_t size, int device, hipStream_t stream)
{
    (void)stream;
    if (size <= 0) return nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != device) (void)hipSetDevice(device);
    const size_t rz = red_zone();
    const size_t body = ((size_t)size + 255) & ~(size_t)255;
    const size_t total = rz + body + rz;
    char* base = nullptr;
    if (hipMalloc((void**)&base, total) != hipSuccess || !base) {
        if (prev != device) (void)hipSetDevice(prev);
        return nullptr;   // torch raises its out-of-memory error
    }
    // poison zone + body + zone; synchronous with respect to the host (hipMemset on the null stream), so no later launch can overtake it
    (void)hipMemset(base, kPoison, total);
    (void)hipDeviceSynchronize();
    if (prev != device) (void)hipSetDevice(prev);
    std::lock_guard<std::mutex> lk(g_mu);
    Block b{ base, (size_t)size, total, device, ++g_serial };
    g_live[base + rz] = b;
    ++g_allocs;
    g_now += total;
    if (g_now > g_peak) g_peak = g_now;
    return base + rz;
}

void gvd_guard_free(void* ptr, ssize_t size, int device, hipStream_t stream)
{
    (void)size; (void)stream;
    if (!ptr) return;
    Block b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(ptr);
        if (it == g_live.end()) {
            fprintf(stderr, "[gvd_guard] free of a pointer this allocator did not hand out: %p\n", ptr);
            ++g_violations;
            return;
        }
        b = it->second;
        g_live.erase(it);
        ++g_frees;
        g_now -= b.total;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != device) (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();     // every kernel that could still touch the block has finished
    check_block(b);
    (void)hipFree(b.base);
    if (prev != device) (void)hipSetDevice(prev);
}

// Walk every live block now (the stress scripts call it between phases: a violation is then attributed to the phase, not to the free).
unsigned long long gvd_guard_check_all()
{
    (void)hipDeviceSynchronize();
    std::vector<Block> blocks;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        blocks.reserve(g_live.size());
        for (auto& kv : g_live) blocks.push_back(kv.second);
    }
    for (const Block& b : blocks) {
        int prev = 0;
        (void)hipGetDevice(&prev);
        if (prev != b.device) (void)hipSetDevice(b.device);
        check_block(b);
        if (prev != b.device) (void)hipSetDevice(prev);
    }
    return g_violations;
}

unsigned long long gvd_guard_violations() { return g_violations; }
unsigned long long gvd_guard_allocs() { return g_allocs; }
unsigned long long gvd_guard_frees() { return g_frees; }
unsigned long long gvd_guard_peak_bytes() { return g_peak; }
unsigned long long gvd_guard_redzone_bytes() { return red_zone(); }

// Self-test hook: scribble `n` bytes starting `offset` bytes past the END of the body of `ptr` (negative offset: before its start), from the
// host.  tests/test_guard_allocator_gpu.py uses it to prove that a violation IS reported.
int gvd_guard_scribble(void* ptr, long long offset, size_t n)
{
    Block b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(ptr);
        if (it == g_live.end()) return -1;
        b = it->second;
    }
    char* at = offset >= 0 ? (char*)ptr + b.size + offset : (char*)ptr + offset;
    if (at < b.base || at + n > b.base + b.total) return -2;
    return hipMemset(at, 0x11, n) == hipSuccess ? 0 : -3;
}

}  // extern "C"
