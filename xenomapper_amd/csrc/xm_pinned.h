// One book of the page-locked host memory this library holds, process-wide: what xm_strip / xm_bamdev allocated with
// hipHostMalloc (staging windows, inflated copies, line tables, text) and what callers registered through xm_host_register.
// xm_pinned_bytes() reads it (include/xenomapper_hip.h); tests/test_release_gpu.py asserts that releasing the front ends brings
// it back to zero.  Why: round 5's GPU suite died once with SIGABRT inside an unrelated 400 MB pageable host-to-device copy while
// the process-wide front ends still held every buffer earlier test modules had made them allocate, and nothing could say how
// much that was.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>

namespace xmpin {

struct Book {
    std::mutex lock;
    std::unordered_map<const void *, size_t> allocated, registered;
    uint64_t allocated_bytes = 0, registered_bytes = 0, peak_bytes = 0;
};

inline Book &book()
{
    static Book b;               // one per shared library (inline function: the translation units share it)
    return b;
}

inline hipError_t host_malloc(void **p, size_t bytes)
{
    const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e == hipSuccess && *p) {
        Book &b = book();
        std::lock_guard<std::mutex> hold(b.lock);
        b.allocated[*p] = bytes;
        b.allocated_bytes += bytes;
        if (b.allocated_bytes + b.registered_bytes > b.peak_bytes) b.peak_bytes = b.allocated_bytes + b.registered_bytes;
    }
    return e;
}

inline hipError_t host_free(void *p)
{
    if (!p) return hipSuccess;
    {
        Book &b = book();
        std::lock_guard<std::mutex> hold(b.lock);
        auto it = b.allocated.find(p);
        if (it != b.allocated.end()) { b.allocated_bytes -= it->second; b.allocated.erase(it); }
    }
    return hipHostFree(p);
}

inline void note_registered(const void *p, size_t bytes)
{
    Book &b = book();
    std::lock_guard<std::mutex> hold(b.lock);
    b.registered[p] = bytes;
    b.registered_bytes += bytes;
    if (b.allocated_bytes + b.registered_bytes > b.peak_bytes) b.peak_bytes = b.allocated_bytes + b.registered_bytes;
}

inline bool is_registered(const void *p)
{
    Book &b = book();
    std::lock_guard<std::mutex> hold(b.lock);
    return b.registered.count(p) != 0;
}

inline void note_unregistered(const void *p)
{
    Book &b = book();
    std::lock_guard<std::mutex> hold(b.lock);
    auto it = b.registered.find(p);
    if (it != b.registered.end()) { b.registered_bytes -= it->second; b.registered.erase(it); }
}

}  // namespace xmpin
