// Root cull pre-pass: a streaming kernel in front of the traversal kernel.
//
// On the metric's workload (random rays through the scene's bounding box) three of four rays miss every child of the BVH8 root.
// In the persistent traversal kernel such a ray still costs a lane refill (80-byte record, three IEEE divisions) and a share of a
// full node step in a wave whose other lanes are at different depths - a third of all wave instructions of a launch went there.
// This pass does the same root test for every ray with all 64 lanes of every wave busy and nothing else to do: one coalesced
// read of the ray records, the root node from L1, and a compacted list of the rays that survive.  The traversal kernel then
// fetches only those (through the lists; per-ray order of operations and arithmetic are unchanged: a ray that fails here is
// exactly a ray whose first node step would have found no child and left its record untouched).
//
// Layout: the rays of work queue q (trace_loop.hip.h: [q * perQ, (q + 1) * perQ)) are handled by the workgroups
// blockIdx.x = q * blocksPerQ .. and appended to queue q's list, P.survivors[q * perQ ..), whose length is kept next to the
// queue head (P.queues[q * TRACE_QUEUE_STRIDE + 1]); one global atomic per workgroup of CULL_RAYS rays.
#pragma once
#include "trace_common.hip.h"

namespace rtamd {
namespace dev {

static constexpr int CULL_BLOCK = 256;
static constexpr int CULL_PER_THREAD = 4;
static constexpr int CULL_RAYS = CULL_BLOCK * CULL_PER_THREAD; // rays per workgroup

template <bool ROBUST, bool OCCLUDED, bool VEC> __global__ __launch_bounds__(CULL_BLOCK) void cull_kernel(LaunchParams P)
{
  __shared__ uint32_t wgCount, wgValid, wgBase;
  const uint32_t perQ = (P.count + (uint32_t)TRACE_QUEUES - 1u) / (uint32_t)TRACE_QUEUES;
  const uint32_t blocksPerQ = (perQ + (uint32_t)CULL_RAYS - 1u) / (uint32_t)CULL_RAYS;
  const uint32_t q = blockIdx.x / blocksPerQ, chunk = blockIdx.x - q * blocksPerQ;
  const uint32_t qLo = min(q * perQ, P.count), qHi = min(qLo + perQ, P.count);
  if (threadIdx.x == 0) { wgCount = 0u; wgValid = 0u; }
  __syncthreads();
  const QNode8* root = P.accel.nodes + P.accel.root;
  uint32_t idx[CULL_PER_THREAD];
  bool keep[CULL_PER_THREAD];
  uint32_t nValid = 0;
#pragma unroll
  for (int j = 0; j < CULL_PER_THREAD; j++) { // j-th quarter of the workgroup's block: coalesced across the lanes
    idx[j] = qLo + chunk * (uint32_t)CULL_RAYS + (uint32_t)j * CULL_BLOCK + threadIdx.x;
    keep[j] = false;
    if (idx[j] < qHi) {
      RayState r;
      load_ray<VEC>((const char*)P.rays + (size_t)idx[j] * P.stride, r);
      bool ok = r.tnear <= r.tfar; // same admission as the traversal kernel's fetch
      if (OCCLUDED) ok = ok && !(r.tfar < 0.0f);
      if (ok) {
        nValid++;
        TravRay<ROBUST> tr;
        tr.init(r);
        keep[j] = node_any_child_hit<ROBUST>(root, tr, fmaxf(r.tfar, 0.0f));
      }
    }
  }
  // compaction: one LDS atomic per wave and quarter, one global atomic per workgroup
  uint32_t slot[CULL_PER_THREAD];
#pragma unroll
  for (int j = 0; j < CULL_PER_THREAD; j++) {
    const uint64_t m = __ballot(keep[j]);
    uint32_t base = 0;
    if (m != 0ull) {
      const uint32_t lane = __builtin_amdgcn_mbcnt_hi((uint32_t)(~0ull >> 32), __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      if (lane == 0u) base = atomicAdd(&wgCount, (uint32_t)__popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
    }
    slot[j] = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  }
  if (nValid) atomicAdd(&wgValid, nValid);
  __syncthreads();
  if (threadIdx.x == 0) {
    wgBase = wgCount ? atomicAdd(&P.queues[q * TRACE_QUEUE_STRIDE + 1], wgCount) : 0u;
    if (wgValid) atomicAdd(&P.queues[q * TRACE_QUEUE_STRIDE + 2], wgValid);
  }
  __syncthreads();
  uint32_t* list = P.survivors + (size_t)q * perQ + wgBase;
#pragma unroll
  for (int j = 0; j < CULL_PER_THREAD; j++)
    if (keep[j]) list[slot[j]] = idx[j];
}

template <bool ROBUST, bool OCCLUDED> inline hipError_t launch_cull_vec(const LaunchParams& p, hipStream_t stream)
{
  const uint32_t perQ = (p.count + (uint32_t)TRACE_QUEUES - 1u) / (uint32_t)TRACE_QUEUES;
  const uint32_t blocks = (uint32_t)TRACE_QUEUES * ((perQ + (uint32_t)CULL_RAYS - 1u) / (uint32_t)CULL_RAYS);
  const bool vec = (p.stride % 16 == 0) && (((uintptr_t)p.rays) % 16 == 0);
  if (vec) hipLaunchKernelGGL((cull_kernel<ROBUST, OCCLUDED, true>), dim3(blocks), dim3(CULL_BLOCK), 0, stream, p);
  else hipLaunchKernelGGL((cull_kernel<ROBUST, OCCLUDED, false>), dim3(blocks), dim3(CULL_BLOCK), 0, stream, p);
  return hipGetLastError();
}

} // namespace dev
} // namespace rtamd
