// Process-wide caches of device buffers and streams.  The reference's scripts run its drivers over whole sequences
// with --window 2 (scripts/ba_all_*.sh): thousands of small problems, one handle each.  hipMalloc / hipFree /
// hipStreamCreate cost 0.05 - 1 ms apiece, far more than such a solve, so released buffers and streams are kept
// and handed to the next handle.  SSBA_POOL_MB (default 2048) caps the cached bytes per process; 0 disables caching.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace ssba {

hipError_t pool_malloc(void **out, size_t bytes);   // on the current device
void pool_free(void *ptr);                           // no-op for nullptr; the caller has synchronised its streams
hipError_t pool_host_malloc(void **out, size_t bytes);   // pinned host memory (hipHostMalloc / hipHostFree cost ~0.1 ms each)
void pool_host_free(void *ptr);
hipError_t pool_stream_acquire(hipStream_t *out);    // non-blocking stream of the current device
void pool_stream_release(hipStream_t s);             // the caller has synchronised it
void pool_trim();                                    // hipFree / hipHostFree everything that is cached

}  // namespace ssba
