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
// Entry for the rtcIntersect1/1M, rtcOccluded1/1M API calls: small host-pointer calls go through the call combiner
// (SURVEY.md section 8 row f2), everything else straight to trace_batch.
void trace_call(Scene* s, void* rays, uint32_t M, size_t byteStride, bool occluded, const RTCIntersectContext* ctx);
// persistent consumer for calls of up to 64 rays (rt_trace.cpp, trace_service.hip.h)
void service_destroy(Device* dev);
void service_quiesce(Device* dev);
static const uint32_t COMBINE_MAX_RAYS = 1024; // host-pointer calls up to this size are combined

} // namespace rtamd
