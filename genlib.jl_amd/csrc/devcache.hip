// devcache.hip -- see devcache.h.
#include "devcache.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <unordered_map>

namespace genphi {

namespace {

struct Block {
    void *ptr;
    size_t bytes;
};

struct DeviceState {
    std::vector<Block> idle;               // kept blocks, oldest first (a handful)
    size_t idle_bytes = 0;
    size_t keep_max = ~size_t(0);          // (set on first use: GENPHI_KEEP_MB, or 8 GiB but at most 1/16 of the device's memory)
    std::vector<hipStream_t> streams;
    PinnedRing ring;
};

struct Cache {
    std::mutex mu;
    std::map<int, DeviceState> dev;        // (node-based: references stay valid)
    std::unordered_map<void *, std::pair<size_t, int>> live;      // blocks handed out: size, device
    bool keep_set = false;                 // GENPHI_KEEP_MB given
    size_t keep_env = 0;
    Cache()
    {
        const char *e = std::getenv("GENPHI_KEEP_MB");
        if (e) { keep_set = true; keep_env = static_cast<size_t>(std::max(0L, std::atol(e))) << 20; }
    }
    // Why as much as 8 GiB: a block that goes back to the driver is cleared by it in the background with the copy engines (~65 ms per
    // GB), and until that is done every device-to-host copy of the process runs at half its rate (profiles/microbench/free_then_copy.hip).
    // A one-shot gen.phi whose plan does not fit the budget pays that on its own result copy, call after call (cfg3s, a 6 GB plan:
    // its 400 MB copy 20.4 instead of 8.4 ms with the 1 GiB budget this cache started with).
    size_t keep_max_of(DeviceState &d, int device);
};

size_t Cache::keep_max_of(DeviceState &d, int device)
{
    if (d.keep_max == ~size_t(0)) {
        if (keep_set) d.keep_max = keep_env;
        else {
            size_t total = 0;
            if (hipDeviceTotalMem(&total, device) != hipSuccess) { (void)hipGetLastError(); total = size_t(16) << 30; }
            d.keep_max = std::min(size_t(8) << 30, total / 16);
        }
    }
    return d.keep_max;
}

Cache &cache()
{
    static Cache *c = new Cache();         // (never destroyed: plans may be released during interpreter shutdown)
    return *c;
}

// request sizes are rounded so that plans of about the same shape find each other's blocks: 4 KiB steps up to 1 MiB,
// then 1/16 of the power of two below (<= 6 % over)
size_t round_size(size_t b)
{
    if (b <= (size_t(1) << 20)) return (std::max<size_t>(b, 1) + 4095) / 4096 * 4096;
    size_t p = size_t(1) << 20;
    while ((p << 1) <= b) p <<= 1;
    const size_t step = p >> 4;
    return (b + step - 1) / step * step;
}

}  // namespace

hipError_t cached_malloc(void **ptr, size_t bytes)
{
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const size_t want = round_size(bytes);
    Cache &c = cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        DeviceState &d = c.dev[device];
        size_t best = d.idle.size();
        for (size_t k = 0; k < d.idle.size(); ++k)
            if (d.idle[k].bytes >= want && d.idle[k].bytes <= want + want / 4 && (best == d.idle.size() || d.idle[k].bytes < d.idle[best].bytes)) best = k;
        if (best < d.idle.size()) {
            const Block b = d.idle[best];
            d.idle.erase(d.idle.begin() + static_cast<std::ptrdiff_t>(best));
            d.idle_bytes -= b.bytes;
            c.live[b.ptr] = {b.bytes, device};
            *ptr = b.ptr;
            return hipSuccess;
        }
    }
    e = hipMalloc(ptr, want);
    if (e != hipSuccess) {                                 // out of memory: give back what is kept and try once more
        (void)hipGetLastError();
        release_cached();
        e = hipMalloc(ptr, want);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lock(c.mu);
    c.live[*ptr] = {want, device};
    return hipSuccess;
}

hipError_t cached_free(void *ptr)
{
    if (!ptr) return hipSuccess;
    Cache &c = cache();
    std::vector<void *> evicted;                           // (the oldest kept blocks make room for the newest: what the next plan will ask for)
    bool kept = false;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.live.find(ptr);
        if (it != c.live.end()) {
            const size_t bytes = it->second.first;
            const int device = it->second.second;
            c.live.erase(it);
            DeviceState &d = c.dev[device];
            const size_t keep_max = c.keep_max_of(d, device);
            if (bytes <= keep_max) {
                while (!d.idle.empty() && (d.idle_bytes + bytes > keep_max || d.idle.size() >= 256)) {
                    evicted.push_back(d.idle.front().ptr);
                    d.idle_bytes -= d.idle.front().bytes;
                    d.idle.erase(d.idle.begin());
                }
                d.idle.push_back({ptr, bytes});
                d.idle_bytes += bytes;
                kept = true;
            }
        }
    }
    hipError_t e = hipSuccess;
    for (void *q : evicted) { const hipError_t e2 = hipFree(q); if (e == hipSuccess) e = e2; }
    if (!kept) { const hipError_t e2 = hipFree(ptr); if (e == hipSuccess) e = e2; }
    return e;
}

hipError_t cached_stream(hipStream_t *st)
{
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    Cache &c = cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        DeviceState &d = c.dev[device];
        if (!d.streams.empty()) { *st = d.streams.back(); d.streams.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

void cached_stream_release(hipStream_t st, int device)
{
    if (!st) return;
    Cache &c = cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        DeviceState &d = c.dev[device];
        if (c.keep_max_of(d, device) > 0 && d.streams.size() < 8) { d.streams.push_back(st); return; }
    }
    (void)hipStreamDestroy(st);
}

namespace {
std::mutex g_small_mu;
std::vector<void *> g_small_pinned;        // idle 4 KB pinned buffers
}  // namespace

hipError_t cached_pinned(void **ptr, size_t bytes)
{
    if (bytes > 4096) return hipErrorInvalidValue;
    {
        std::lock_guard<std::mutex> lock(g_small_mu);
        if (!g_small_pinned.empty()) { *ptr = g_small_pinned.back(); g_small_pinned.pop_back(); return hipSuccess; }
    }
    return hipHostMalloc(ptr, 4096, hipHostMallocDefault);
}

void cached_pinned_release(void *ptr)
{
    if (!ptr) return;
    {
        std::lock_guard<std::mutex> lock(g_small_mu);
        if (g_small_pinned.size() < 16) { g_small_pinned.push_back(ptr); return; }
    }
    (void)hipHostFree(ptr);
}

PinnedRing &pinned_ring(int device)
{
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    return c.dev[device].ring;
}

bool pinned_ring_reserve(PinnedRing &r, size_t n_chunks, size_t chunk_bytes, size_t n_streams)
{
    if (r.chunk_bytes < chunk_bytes) {                     // (chunks of one size: a larger request replaces them)
        for (void *q : r.chunk) (void)hipHostFree(q);
        r.chunk.clear();
        r.chunk_bytes = chunk_bytes;
    }
    while (r.chunk.size() < n_chunks) {
        void *q = nullptr;
        if (hipHostMalloc(&q, r.chunk_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
        r.chunk.push_back(q);
    }
    while (r.stream.size() < n_streams) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return false; }
        r.stream.push_back(st);
    }
    return true;
}

size_t cached_bytes()
{
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    size_t b = 0;
    for (auto &kv : c.dev) b += kv.second.idle_bytes;
    return b;
}

void release_cached()
{
    Cache &c = cache();
    int cur = -1;
    (void)hipGetDevice(&cur);
    std::vector<std::pair<int, Block>> blocks;
    std::vector<std::pair<int, hipStream_t>> streams;
    std::vector<PinnedRing *> rings;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (auto &kv : c.dev) {
            for (const Block &b : kv.second.idle) blocks.push_back({kv.first, b});
            kv.second.idle.clear(); kv.second.idle_bytes = 0;
            for (hipStream_t st : kv.second.streams) streams.push_back({kv.first, st});
            kv.second.streams.clear();
            rings.push_back(&kv.second.ring);
        }
    }
    for (auto &b : blocks) { (void)hipSetDevice(b.first); (void)hipFree(b.second.ptr); }
    for (auto &s : streams) { (void)hipSetDevice(s.first); (void)hipStreamDestroy(s.second); }
    for (PinnedRing *r : rings) {
        std::unique_lock<std::mutex> lock(r->mu, std::try_to_lock);
        if (!lock.owns_lock()) continue;                   // (a copy is using it)
        for (void *q : r->chunk) (void)hipHostFree(q);
        for (hipStream_t st : r->stream) (void)hipStreamDestroy(st);
        r->chunk.clear(); r->stream.clear(); r->chunk_bytes = 0;
    }
    {
        std::lock_guard<std::mutex> lock(g_small_mu);
        for (void *q : g_small_pinned) (void)hipHostFree(q);
        g_small_pinned.clear();
    }
    if (cur >= 0) (void)hipSetDevice(cur);
}

}  // namespace genphi
