// Caching allocator and stream cache behind ssba_pool.h.
#include "ssba_pool.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace ssba {

namespace {

struct Pool {
    std::mutex mu;
    struct Block { size_t bytes; int device; };
    std::unordered_map<void *, Block> live;                                // handed out
    std::map<std::pair<int, size_t>, std::vector<void *>> cached;          // (device, rounded size) -> free blocks
    std::map<int, std::vector<hipStream_t>> streams;
    std::unordered_map<void *, size_t> host_live;
    std::map<size_t, std::vector<void *>> host_cached;
    size_t host_cached_bytes = 0;
    size_t cached_bytes = 0, cap = 0;
    Pool() {
        const char *e = getenv("SSBA_POOL_MB");
        cap = (size_t)(e ? atol(e) : 2048) << 20;
    }
    ~Pool() {       // process exit: the runtime may already be shutting down, so nothing is freed explicitly
    }
};

Pool &pool() {
    static Pool *p = new Pool();      // intentionally leaked: see ~Pool
    return *p;
}

// round up to 1/8 of the largest power of two below the size (at least 512 B): <= 12.5 % slack, few distinct sizes
size_t rounded(size_t bytes) {
    if (bytes < 512) return 512;
    size_t p2 = 512;
    while ((p2 << 1) <= bytes) p2 <<= 1;
    const size_t g = p2 >> 3 > 512 ? p2 >> 3 : 512;
    return (bytes + g - 1) / g * g;
}

}  // namespace

hipError_t pool_malloc(void **out, size_t bytes) {
    Pool &P = pool();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t r = rounded(bytes);
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.cached.find({dev, r});
        if (it != P.cached.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            P.cached_bytes -= r;
            P.live[*out] = {r, dev};
            return hipSuccess;
        }
    }
    e = hipMalloc(out, r);
    if (e != hipSuccess) {      // out of memory with a full cache: drop the cache and retry once
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(P.mu);
            for (auto &kv : P.cached) { drop.insert(drop.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
            P.cached_bytes = 0;
        }
        for (void *p : drop) (void)hipFree(p);
        (void)hipGetLastError();
        e = hipMalloc(out, r);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lock(P.mu);
    P.live[*out] = {r, dev};
    return hipSuccess;
}

void pool_free(void *ptr) {
    if (!ptr) return;
    Pool &P = pool();
    Pool::Block b{0, 0};
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.live.find(ptr);
        if (it == P.live.end()) { (void)hipFree(ptr); return; }
        b = it->second;
        P.live.erase(it);
        if (P.cached_bytes + b.bytes <= P.cap) {
            P.cached[{b.device, b.bytes}].push_back(ptr);
            P.cached_bytes += b.bytes;
            return;
        }
    }
    (void)hipFree(ptr);
}

hipError_t pool_host_malloc(void **out, size_t bytes) {
    Pool &P = pool();
    const size_t r = rounded(bytes);
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.host_cached.find(r);
        if (it != P.host_cached.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            P.host_cached_bytes -= r;
            P.host_live[*out] = r;
            return hipSuccess;
        }
    }
    const hipError_t e = hipHostMalloc(out, r, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(P.mu);
    P.host_live[*out] = r;
    return hipSuccess;
}

void pool_host_free(void *ptr) {
    if (!ptr) return;
    Pool &P = pool();
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.host_live.find(ptr);
        if (it != P.host_live.end()) {
            const size_t r = it->second;
            P.host_live.erase(it);
            if (P.cap && P.host_cached_bytes + r <= (size_t)256 << 20) {      // pinned memory is scarcer: 256 MiB at most
                P.host_cached[r].push_back(ptr);
                P.host_cached_bytes += r;
                return;
            }
        }
    }
    (void)hipHostFree(ptr);
}

hipError_t pool_stream_acquire(hipStream_t *out) {
    Pool &P = pool();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto &v = P.streams[dev];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void pool_stream_release(hipStream_t s) {
    if (!s) return;
    Pool &P = pool();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipStreamDestroy(s); return; }
    std::lock_guard<std::mutex> lock(P.mu);
    auto &v = P.streams[dev];
    if (P.cap && v.size() < 8) v.push_back(s);
    else (void)hipStreamDestroy(s);
}

void pool_trim() {
    Pool &P = pool();
    std::vector<void *> dev, host;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        for (auto &kv : P.cached) { dev.insert(dev.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
        for (auto &kv : P.host_cached) { host.insert(host.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
        P.cached_bytes = 0;
        P.host_cached_bytes = 0;
    }
    for (void *p : dev) (void)hipFree(p);
    for (void *p : host) (void)hipHostFree(p);
}

}  // namespace ssba
