// Batch front-end of the hot path (see rt_trace.cpp).
#pragma once
#include "rt_objects.h"
#include "trace.h"

namespace rtamd {

// Trace M records starting at `rays` (host or device memory) with the given byte stride.
// countersOut != nullptr selects the instrumented kernels and implies a host synchronisation.
void trace_batch(Scene* s, void* rays, uint32_t M, size_t byteStride, bool occluded, const RTCIntersectContext* ctx,
                 TraceCounters* countersOut);
void trace_pointers(Scene* s, void** ptrs, uint32_t M, bool occluded, const RTCIntersectContext* ctx);

} // namespace rtamd
