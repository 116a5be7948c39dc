// Launch interface between the host object model and the HIP traversal kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

#include "accel.h"

namespace rtamd {

struct LaunchParams
{
  AccelDesc accel;
  void* rays;          // device pointer to the first RTCRayHit (intersect) or RTCRay (occluded)
  uint32_t count;      // M
  uint32_t stride;     // byteStride
  uint32_t instID;     // context->instID[0], copied into hit.instID (intersector_epilog.h:303)
  uint32_t occluded;   // 0: closest hit, 1: any hit
  void* spill;         // HBM overflow area of the per-lane LDS stacks
  uint32_t spillDepth; // entries per lane available in `spill`
  uint32_t gridBlocks; // persistent grid size the spill area was sized for
  uint32_t cbvhLevels; // fork: depth C of every cBVH blob of the scene (rtcSetSceneLevels)
  uint32_t cbvhLaneForm; // fork, lane kernel: 1 = walk blobs one ray per lane (coherent batches), 0 = quad form (trace_subdiv.hip)
  WaveRecord* counters;    // non-null selects the instrumented kernel twin; one record per wavefront
  uint32_t numCUs;         // compute units of the device (persistent grid sizing)
  uint32_t rayChunk;       // rays per work-queue grab (tuning knob, env RTAMD_CHUNK)
  uint32_t leafBatch;      // lanes waiting at a leaf before the leaf phase runs (tuning knob, env RTAMD_LEAF_BATCH)
  uint32_t blocksPerCU;    // 0 = occupancy-derived (tuning knob, env RTAMD_BLOCKS_PER_CU)
  uint32_t refillBatch;    // idle lanes needed before a wave fetches new rays (tuning knob, env RTAMD_REFILL_BATCH)
  uint32_t octSteps;       // node steps an octet stays with its ray while the queues still have rays (env RTAMD_OCT_STEPS; unlimited in the drain)
  uint32_t octLeaf;        // waiting rays from which the child-parallel leaf step runs (0 = never; env RTAMD_OCT_LEAF)
  uint32_t inlineRay;      // service kernel (DIRECT), a job of one ray: its org, tnear, dir, tfar lie in words 0..7 of the wave's exchange row 0, not only in P.rays
  uint32_t walkBatch;      // two-stage leaves: parked rays from which the walk stage runs (env RTAMD_WALK_BATCH)
  uint32_t octMax;         // lanes with node work up to which a wave runs the child-parallel node step (0 = never; env RTAMD_OCT_MAX)
  uint32_t* queues;        // TRACE_QUEUES work-queue heads, zeroed on the stream before the launch
  // Filter-function re-trace (row f3): ray i skips the triangles (geomID, primID) listed in
  // exclPairs[exclOffsets[i] .. exclOffsets[i+1]) - the candidates a host filter callback rejected in earlier rounds.
  // nullptr for ordinary launches.
  // Grid cells (eager subdivision path): the triangles of a patch share (geomID, primID), so an entry also carries the candidate's
  // distance, exclT[e] = bits of t - the kernels are deterministic, the same triangle yields the same t on the re-trace.
  const uint32_t* exclOffsets;
  const uint2* exclPairs;
  const uint32_t* exclT;
  uint32_t poolKernel;     // 1: ray-pool skeleton (trace_pool.hip.h), 0: lane-per-ray skeleton (trace_loop.hip.h)
  // Root cull pass (trace_cull.hip.h): when `survivors` is set, a streaming pre-pass has tested every ray against the root node's
  // children and appended the indices of the rays that hit at least one of them to per-work-queue lists: queue q's list starts
  // at survivors[q * perQ] (perQ = ceil(count / TRACE_QUEUES), its capacity) and holds queues[q * TRACE_QUEUE_STRIDE + 1] entries;
  // queues[q * TRACE_QUEUE_STRIDE + 2] counts the valid rays of the queue's range (for the work counters).  The traversal kernel
  // then fetches rays through the lists.  nullptr: rays are fetched by index (no pre-pass).
  uint32_t* survivors;
  unsigned long long* timeline; // development builds (-DTRACE_TIMELINE, env RTAMD_TIMELINE=1): 8 words per wavefront of the PLAIN kernel
                                // (start, first rays, last grab, end in 10 ns ticks; iterations, iterations after the last grab, lanes x
                                // iterations after the last grab, rays); nullptr otherwise
  uint32_t* overflow;      // host-mapped word, set to 1 by a kernel that had to drop a traversal-stack entry (never for a tree
                           // whose depth the builder reported correctly: the overflow area is sized for the worst case)
};

// ---- persistent consumer for small calls (trace_service.hip.h, rt_service.cpp) ---------------------------------------------
static const int SERVICE_SLOT_RAYS = 64; // one wavefront, one ray per lane
struct alignas(128) ServiceSlot // host-mapped pinned memory, one per wavefront of the service kernel
{
  uint32_t seq;        // host -> device: number of the slot's current job, written LAST (release), after seq2
  uint32_t count;      // rays of the job (<= SERVICE_SLOT_RAYS)
  uint32_t occluded;   // 0: RTCRayHit records (80 B), 1: RTCRay records (48 B)
  uint32_t instID;
  uint32_t spillDepth; // HBM stack overflow entries this scene's depth needs (<= what the service allocated)
  uint32_t pad0[3];
  AccelDesc accel;     // the committed scene's device arrays (replica of the service's GPU)
  float ray0[8];       // a job of ONE ray: org, tnear, dir, tfar of the record (words 0..6 and 8) again, so that the poll that sees the job has the ray as well
  uint32_t pad1[32 - 8 - sizeof(AccelDesc) / 4 - 8 - 1];
  uint32_t seq2;       // copy of seq in the header's second 64-byte half, written BEFORE seq (trace_service.hip.h)
  uint32_t done;       // device -> host: sequence number of the last finished job (own 128-byte line)
  uint32_t pad2[31];
  char rays[SERVICE_SLOT_RAYS * 80];
};
static_assert(sizeof(ServiceSlot) == 256 + SERVICE_SLOT_RAYS * 80 && offsetof(ServiceSlot, done) == 128 && offsetof(ServiceSlot, seq2) == 124 && offsetof(ServiceSlot, ray0) == 80 && offsetof(ServiceSlot, rays) == 256, "ServiceSlot layout");
struct ServiceParams
{
  LaunchParams base;   // tuning knobs, overflow area, overflow flag; rays / count / accel come from the slot per job
  ServiceSlot* slots;  // device address of the ring
  uint32_t numSlots;   // = wavefronts of the service kernel
  uint32_t idlePolls;  // polls without ANY served job after which the service ends by itself
  uint32_t* stop;      // host-mapped: non-zero = leave now
  uint32_t* activity;  // device word: jobs served by any wavefront
};
// start the service kernel for base.accel.kind (and base.cbvhLevels); hipErrorInvalidValue: no service kernel for this accel kind / level
hipError_t launch_service_tri(const ServiceParams& s, hipStream_t stream);    // trace_tri.hip
hipError_t launch_service_subdiv(const ServiceParams& s, hipStream_t stream); // trace_subdiv.hip

static const int TRACE_QUEUES = 64;       // work queues per launch (must equal the wavefront width: one lane scans one head)
static const int TRACE_QUEUE_STRIDE = 32; // u32 words between two work-queue heads (128 B: one L2 line each)
#ifndef TRACE_BLOCK_THREADS
#define TRACE_BLOCK_THREADS 256
#endif
static const int TRACE_BLOCK = TRACE_BLOCK_THREADS; // threads per workgroup of the lane-per-ray kernel (4 wavefronts)
// ray-pool skeleton (trace_pool.hip.h): workgroup size, ray slots per wavefront, LDS stack entries per slot, workgroups per CU
#ifndef TRACE_POOL_SLOTS_PER_WAVE
#define TRACE_POOL_SLOTS_PER_WAVE 128
#endif
#ifndef TRACE_POOL_STACK_ENTRIES
#define TRACE_POOL_STACK_ENTRIES 8
#endif
static const int TRACE_POOL_BLOCK = 128, TRACE_POOL_SLOTS = TRACE_POOL_SLOTS_PER_WAVE, TRACE_POOL_STACK = TRACE_POOL_STACK_ENTRIES;
static const int TRACE_POOL_BLOCKS_PER_CU = 8; // upper bound of resident workgroups per CU (sizes the overflow area)
#ifndef TRACE_LDS_STACK_ENTRIES
#define TRACE_LDS_STACK_ENTRIES 16
#endif
static const int TRACE_LDS_STACK = TRACE_LDS_STACK_ENTRIES; // stack entries per lane kept in LDS (8 bytes each -> 2 KiB per entry and workgroup)

// Number of workgroups of the persistent grid for `count` rays on a chip with `numCUs` compute units.
uint32_t trace_grid_blocks(uint32_t count, int numCUs, uint32_t rayChunk);

// Enqueue traversal of one batch on `stream`.  Asynchronous; errors surface through the returned hipError_t.
hipError_t launch_trace_tri(const LaunchParams& p, hipStream_t stream);    // trace_tri.hip
hipError_t launch_cull(const LaunchParams& p, hipStream_t stream);         // trace_tri.hip (trace_cull.hip.h): root cull pre-pass
hipError_t launch_trace_subdiv(const LaunchParams& p, hipStream_t stream); // trace_subdiv.hip
inline hipError_t launch_service(const ServiceParams& s, hipStream_t stream)
{
  switch (s.base.accel.kind) {
  case ACCEL_TRI_PLUECKER:
  case ACCEL_TRI_MOELLER: return launch_service_tri(s, stream);
  case ACCEL_CBVH_BOX:
  case ACCEL_CBVH_LEAF:
  case ACCEL_CBVH_GRID:
  case ACCEL_CBVH_FULL:
  case ACCEL_GRIDSOA: return launch_service_subdiv(s, stream);
  default: return hipErrorInvalidValue;
  }
}
inline hipError_t launch_trace(const LaunchParams& p, hipStream_t stream)
{
  switch (p.accel.kind) {
  case ACCEL_TRI_PLUECKER:
  case ACCEL_TRI_MOELLER: return launch_trace_tri(p, stream);
  case ACCEL_CBVH_BOX:
  case ACCEL_CBVH_LEAF:
  case ACCEL_CBVH_GRID:
  case ACCEL_CBVH_FULL:
  case ACCEL_GRIDSOA: return launch_trace_subdiv(p, stream);
  default: return hipSuccess;
  }
}

} // namespace rtamd
