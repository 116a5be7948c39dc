/*
 * MI355X extensions to the embree3 contract.  These are additions, not replacements: the reference has
 * no counterpart because it has no device, no stream and no batch boundary.  They exist so that a caller
 * who already owns device-resident ray buffers (bench.py, a wavefront renderer) can run the hot path
 * without host staging and can time / inspect it.
 *
 * Everything here is plain C ABI: opaque handles from rtcore.h, void* for HIP handles, sized structs.
 */
#ifndef EMBREE3_AMD_RTCORE_AMD_H
#define EMBREE3_AMD_RTCORE_AMD_H

#include "rtcore.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* HIP stream (hipStream_t as void*) on which the device enqueues uploads and traversal kernels.
 * rtcIntersect1M/rtcOccluded1M on DEVICE pointers are stream-ordered on it and return without a host
 * sync; on HOST pointers they synchronise before returning (embree semantics). */
RTC_API void* rtcamdGetDeviceStream(RTCDevice device);
/* Use a caller-owned stream (e.g. the framework's current stream) instead of the library's own.
 * May be changed between calls: batches enqueued on DIFFERENT streams are independent and run concurrently (the
 * library keeps its per-launch scratch - work-queue heads, stack overflow area - in a ring of launch contexts).
 * A renderer that keeps 2-4 batches in flight this way hides the drain of one batch (a few deep rays) under the
 * start of the next; on MI355X that is worth ~1.5x in rays/s for 1 M-ray batches.  Order between batches on
 * different streams is the caller's business (HIP events), as for any other stream work. */
RTC_API void rtcamdSetDeviceStream(RTCDevice device, void* hipStream);
/* Block until everything enqueued on the device's stream has completed. */
RTC_API void rtcamdSynchronizeDevice(RTCDevice device);
/* HIP device ordinal the RTCDevice was created on (config key "gpu="). */
RTC_API int rtcamdGetDeviceOrdinal(RTCDevice device);

/* Extra readable properties for rtcGetDeviceProperty (cast to enum RTCDeviceProperty; the values lie inside that
 * enum's range and above every value the reference defines): counters of the call
 * combiner.  Host-pointer calls of up to 1024 rays (rtcIntersect1, rtcOccluded1, short 1M streams) issued
 * concurrently by several threads are traced together: the calls that arrive while a launch is in flight form the
 * next batch.  Results are those of independent calls.  (The reference answers such a call on the calling core.) */
enum RTCAMDDeviceProperty
{
  RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES = 240,     /* traversal kernel launches so far */
  RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS = 241,     /* API calls that went through the combiner */
  RTCAMD_DEVICE_PROPERTY_COMBINED_BATCHES = 242, /* batches formed out of them */
  RTCAMD_DEVICE_PROPERTY_SERVICE_CALLS = 243     /* API calls answered by the persistent small-call kernel (config key service=1) */
};

/* Layout facts of a committed scene's device acceleration structure; sizes in bytes.
 * Replaces what the reference prints with verbose=2 (kernels/bvh/bvh_statistics.cpp). */
struct RTCAMDSceneStats
{
  size_t byteSize;          /* sizeof(struct RTCAMDSceneStats), set by the caller */
  unsigned int accelKind;   /* 0 none, 1 bvh8.triangle4v (Pluecker), 2 bvh8.triangle4 (Moeller),
                               3 cbvh.box, 4 cbvh.leaf, 5 cbvh.grid, 6 gridsoa (eager) */
  unsigned int branching;   /* 8 */
  size_t nodeCount;         /* quantized BVH8 nodes */
  size_t nodeBytes;         /* bytes per node record (96) */
  size_t primCount;         /* triangles, or cBVH / GridSOA leaves */
  size_t primBytes;         /* bytes per triangle record (48) or mean bytes per subdiv leaf blob */
  size_t leafCount;         /* BVH8 leaves */
  size_t totalBytes;        /* resident HBM bytes of the accel */
  unsigned int maxDepth;
  unsigned int reserved;
};
RTC_API void rtcamdGetSceneStats(RTCScene scene, struct RTCAMDSceneStats* stats);

/* Per-batch traversal work counters, produced by an instrumented twin of the intersect kernel (same
 * code path; every wavefront stores one record, folded on the host).  The batch is traced exactly like rtcIntersect1M does.  Used to price the
 * algorithmic bytes per ray, B = 84 + nodes*nodeBytes + prims*primBytes; the
 * reference's counterpart is the EMBREE_STAT_COUNTERS build (kernels/common/stat.h:21-33). */
struct RTCAMDTraceCounters
{
  unsigned long long rays;        /* valid rays traced */
  unsigned long long nodeVisits;  /* BVH8 node records fetched */
  unsigned long long leafVisits;  /* BVH8 leaves entered */
  unsigned long long primTests;   /* triangle records fetched / cBVH or grid leaves entered */
  unsigned long long innerVisits; /* cBVH-internal 4-byte nodes decoded / GridSOA cells tested */
  unsigned long long hits;        /* rays that report a hit */
  unsigned long long stackSpills; /* pushes that overflowed the LDS stack into the HBM spill area */
  unsigned long long reserved;    /* rays that survived the root cull pre-pass (0: the pre-pass did not run); `rays` and `nodeVisits`
                                     include the rays / root visits of the pre-pass */
  /* shader-clock cycles summed over wavefronts, per phase of the traversal loop (the phases are wave-uniform):
     ray fetch, inner-node step, leaf step, pop/finish, and the whole loop.  Diagnostic: where a batch spends time. */
  unsigned long long cyclesFetch, cyclesNode, cyclesLeaf, cyclesPop, cyclesTotal;
  unsigned long long iterations, leafPhases, waves; /* loop iterations, iterations that ran the leaf phase, waves */
  /* wave timeline (100 MHz s_memrealtime): lanes with a ray summed over iterations (/(64*iterations) = occupancy of the
     loop), ~(earliest wave start), and two per-wave histograms: end time since the earliest start in 4 us buckets,
     and iterations per wave in buckets of 2.  Diagnostic: how long the drain of a batch is. */
  unsigned long long activeLaneIters, startInv;
  /* longest single ray in loop iterations (node steps + leaf steps + waiting), and the drain of the waves: time from a
     wave's last successful work-queue grab to its end, summed over waves / maximum (10 ns ticks) */
  unsigned long long maxRaySteps, drainTicksSum, drainTicksMax;
  unsigned long long waveEndHist[64], waveIterHist[64];
};
RTC_API void rtcamdIntersect1MCounted(RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHit* rayhit,
                                      unsigned int M, size_t byteStride, struct RTCAMDTraceCounters* counters);

/* The same for an any-hit batch (rtcOccluded1M semantics). */
RTC_API void rtcamdOccluded1MCounted(RTCScene scene, struct RTCIntersectContext* context, struct RTCRay* ray, unsigned int M,
                                     size_t byteStride, struct RTCAMDTraceCounters* counters);

/* Read-only views of the host copy of the committed accel (valid until the next commit / release).
 * Test infrastructure: lets an external checker walk the exact structure the kernels traverse.
 * kind: 0 = BVH8 nodes, 1 = primitive records, 2 = subdiv leaf blobs, 3 = blob offset table. */
RTC_API const void* rtcamdGetAccelData(RTCScene scene, unsigned int kind, size_t* byteSize);
/* Test hook: the encoder's restatement of the fork's leaf quantiser (quantTris<4>::setZ / estimateExtent,
 * kernels/geometry/compressed_leaf.h:193-251) on caller-supplied inputs: box = lower xyz, upper xyz of the parent box,
 * v = the four corner vertices (12 floats).  Lets tests compare it with the reference header compiled in oracle/_ref. */
/* development aid: raw wave log of the last launch (csrc/trace.h `timeline`, csrc/accel.h WaveRecord) */
RTC_API size_t rtcamdDebugReadWaveLog(RTCDevice device, void* out, size_t bytes);
/* test hook: hold (1) / release (0) the leader of the small-call combiner */
RTC_API void rtcamdDebugHoldCombiner(RTCDevice device, int hold);
RTC_API void rtcamdDebugCbvhLeafCodec(const float* box, const float* v, float extent, unsigned char* bytesOut, float* extentEstimate);
/* Test hook: runs `jobs` parallel jobs of `parts` parts each through the device's pool of staging threads (the pool behind the
 * chunked pipeline of large host-pointer batches; `threads` helpers are started if fewer exist), in `cycles` begin / end cycles,
 * and returns the number of parts that ran exactly once with the right index.  Works on a gpu=none device: CPU-side coverage of the pool. */
RTC_API unsigned long long rtcamdDebugHostPoolSelfTest(RTCDevice device, unsigned int threads, unsigned int cycles, unsigned int jobs, unsigned int parts);
/* Root reference of the BVH8 (encoding documented in DESIGN.md / csrc/accel.h). */
RTC_API unsigned int rtcamdGetAccelRoot(RTCScene scene);

#if defined(__cplusplus)
}
#endif
#endif
