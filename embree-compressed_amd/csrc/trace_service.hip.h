// Persistent consumer for SMALL calls.
//
// rtcIntersect1 / rtcOccluded1 / rtcIntersect1M with a handful of rays pay a kernel launch each (~11 us floor on the GPU plus the
// API calls around it).  The service is ONE resident kernel per accel kind: wavefront w polls slot w of a ring in host-mapped
// pinned memory; a caller copies its rays into a slot, bumps the slot's sequence number and spins on the slot's `done` word; the
// wavefront traces the slot's rays in place (the records stay in host memory, read and written over PCIe like the zero-copy path of
// rt_trace.cpp) with the SAME traversal code as every other batch - trace_body with DIRECT = true, i.e. rays [0, count) belong to
// this wavefront and there are no work queues - and publishes the sequence number.  No launch, no stream, no event per call.
// Every wave reaches its exit: the host's stop word, or `idlePolls` polls during which NO wavefront of the service served a job
// (all waves watch one device counter, so they leave within 256 polls of each other); the host restarts the service on demand and a
// job that raced with the exit is served by the new kernel (a wave starts from the slot's `done`, not from zero).
#pragma once
#include "trace_loop.hip.h"

#ifndef TRACE_SERVICE_SLEEP
#define TRACE_SERVICE_SLEEP 8 // s_sleep units (64 cycles) between two idle polls
#endif

namespace rtamd {
namespace dev {

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

template <typename Leaf, bool ROBUST>
// (two waves per SIMD: 64 wavefronts serve the whole ring, occupancy does not matter here, scratch would)
__global__ __launch_bounds__(TRACE_BLOCK, 2) void service_kernel(ServiceParams S)
{
  __shared__ uint2 ldsStack[TRACE_LDS_STACK + 1][TRACE_BLOCK];
  __shared__ __attribute__((aligned(16))) float octX[TRACE_BLOCK / 64][OCT_ROWS][OCT_WORDS];
  Leaf::prepare();
  const uint32_t wave = blockIdx.x * (TRACE_BLOCK / 64) + (threadIdx.x >> 6);
  if (wave >= S.numSlots) return;
  ServiceSlot* slot = S.slots + wave;
  const uint32_t laneId = lane_rank(~0ull);
  uint32_t last = __builtin_amdgcn_readfirstlane(ld_sys(&slot->done));
  uint32_t idle = 0, seen = 0, quiet = 0;
  for (;;) {
    // ONE access per poll: lane l reads word l of the slot's 128-byte header line (seq, count, flags, the accel's arrays) - a read of host
    // memory is a PCIe round trip (~2 us), and the first version spent five of them on the header of every job
    // The host's two 64-byte cache lines of the header each end their update with a copy of the sequence number (seq2 in word 31 first, then
    // seq in word 0, both with release order after the fields): a poll that sees BOTH copies new has both halves at least that new.
    const uint32_t h2 = ld_sys(&((const uint32_t*)slot)[laneId & 31u]);
    const uint32_t seq = (uint32_t)__builtin_amdgcn_readlane((int)h2, 0);
    if (seq != last && (uint32_t)__builtin_amdgcn_readlane((int)h2, 31) == seq) {
      auto word = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)h2, i); };
      LaunchParams P = S.base;
      AccelDesc A;
      uint32_t aw[sizeof(AccelDesc) / 4];
#pragma unroll
      for (int i = 0; i < (int)(sizeof(AccelDesc) / 4); i++) aw[i] = word(8 + i);
      __builtin_memcpy(&A, aw, sizeof(AccelDesc));
      P.accel = A;
      P.rays = slot->rays;
      P.count = min(word(1), (uint32_t)SERVICE_SLOT_RAYS);
      P.occluded = word(2);
      P.stride = P.occluded ? 48u : 80u;
      P.instID = word(3);
      P.spillDepth = min(word(4), S.base.spillDepth);
      // a job of one ray carries the ray in the header (ServiceSlot::ray0 = words 20..27): into the wave's exchange row 0 for trace_body's fetch
      P.inlineRay = P.count == 1u ? 1u : 0u;
      if (P.inlineRay) {
        if (laneId >= 20u && laneId < 28u) octX[threadIdx.x >> 6][0][laneId - 20u] = __uint_as_float(h2);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (P.occluded) trace_body<Leaf, ROBUST, true, false, true, true>(P, ldsStack, octX[threadIdx.x >> 6]);
      else trace_body<Leaf, ROBUST, false, false, true, true>(P, ldsStack, octX[threadIdx.x >> 6]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // the hit records (host memory) before the sequence number
      if (laneId == 0u) {
        __hip_atomic_store(&slot->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicAdd(S.activity, 1u);
      }
      last = seq;
      idle = 0;
      quiet = 0;
    } else {
      // (the stop word lives in host memory too: looked at every 16th idle poll, so that a poll is ONE PCIe round trip and a new job is seen sooner)
      if ((++idle & 15u) == 0u && __builtin_amdgcn_readfirstlane(ld_sys(S.stop)) != 0u) break;
      if ((idle & 255u) == 0u) {
        const uint32_t a = __builtin_amdgcn_readfirstlane(__hip_atomic_load(S.activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        quiet = a == seen ? quiet + 256u : 0u;
        seen = a;
        if (quiet >= S.idlePolls) break; // nobody has served anything for idlePolls polls: the service ends, the host restarts it on demand
      }
#if TRACE_SERVICE_SLEEP
      __builtin_amdgcn_s_sleep(TRACE_SERVICE_SLEEP);
#endif
    }
  }
}

template <typename Leaf, bool ROBUST> inline hipError_t launch_service_kernel(const ServiceParams& s, hipStream_t stream)
{
  const uint32_t blocks = (s.numSlots + (TRACE_BLOCK / 64) - 1) / (TRACE_BLOCK / 64);
  hipLaunchKernelGGL((service_kernel<Leaf, ROBUST>), dim3(blocks), dim3(TRACE_BLOCK), 0, stream, s);
  return hipGetLastError();
}

} // namespace dev
} // namespace rtamd
