// devcache.h -- device memory, streams and pinned staging kept between calls (process-wide, per device).
//
// A one-shot gen.phi call creates a plan, uploads it, sweeps, copies the result out and destroys the plan.  Measured on an
// MI355X box (profiles/microbench/out/r05_call_wall_*.out) the allocator was most of such a call on mid-size pedigrees:
// hipStreamCreate 1.9 ms, a dozen hipMalloc + hipFree (each hipFree synchronises the device) 3 ms, and the 256 MB pinned
// staging ring of genphi_result_to_host 60-100 ms to pin and 80 ms to unpin -- per call.  The reference has no counterpart
// (its matrices are garbage-collected Julia arrays, src/compute.jl:291,301); this is the allocator a GC would be.
//
//   cached_malloc / cached_free    device blocks of released plans are kept (up to GENPHI_KEEP_MB, default 8 GiB but at most 1/16 of the device's memory, per
//                                  device; larger blocks go back to the driver at once) and handed to the next plan that asks
//                                  for about that size.  The contents of a block are undefined, as with hipMalloc.
//   cached_stream / release        non-blocking streams, kept idle between plans
//   PinnedRing                     the staging ring of genphi_result_to_host, one per device, locked for the length of a copy
//   release_cached                 everything back to the driver (C-ABI: genphi_release_cached)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>
#include <vector>

namespace genphi {

hipError_t cached_malloc(void **ptr, size_t bytes);      // on the current device
hipError_t cached_free(void *ptr);                        // nullptr allowed
hipError_t cached_stream(hipStream_t *st);                // a non-blocking stream of the current device
void cached_stream_release(hipStream_t st, int device);

struct PinnedRing {
    std::mutex mu;                 // held by the copy that uses the ring
    std::vector<void *> chunk;
    std::vector<hipStream_t> stream;
    size_t chunk_bytes = 0;
};
// small pinned host buffers (counters read back at the end of a sweep): kept in a free list
hipError_t cached_pinned(void **ptr, size_t bytes);       // bytes <= 4096
void cached_pinned_release(void *ptr);
PinnedRing &pinned_ring(int device);
// makes the ring hold >= n_chunks chunks of >= chunk_bytes and >= n_streams streams (caller holds ring.mu); false: could not pin
bool pinned_ring_reserve(PinnedRing &r, size_t n_chunks, size_t chunk_bytes, size_t n_streams);

void sparse_phi_release_kept();    // (sparse_phi.hip) the device side and the pinned buffer gen.sparse_phi keeps between calls
size_t cached_bytes();             // device bytes kept right now (all devices)
void release_cached();             // frees what is kept: device blocks, idle streams, pinned rings

}  // namespace genphi
