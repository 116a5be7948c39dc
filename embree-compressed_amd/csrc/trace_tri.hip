// Triangle leaves of the quantized BVH8: Triangle4v / Pluecker (robust) and Triangle4 / Moeller (fast).
// 48-byte records fetched as three dwordx4; every group of 4 records from the leaf start is one block with the
// reference's 4-wide SIMD semantics.
//   block loop            kernels/geometry/intersector_iterators.h:32-36 (ArrayIntersector1)
//   epilog / tie rules    kernels/geometry/intersector_epilog.h:226-307 (closest), :388-450 (any hit)
#include "trace_loop.hip.h"
#include "trace_pool.hip.h"

namespace rtamd {
namespace dev {

// Records requested per memory round trip inside a block of 4 (1, 2 or 4): more hides latency, fewer saves registers.
#ifndef TRI_FETCH
#define TRI_FETCH 4
#endif

template <bool PLUECKER> struct TriLeaf
{
  static __device__ __forceinline__ void prepare() {}

  template <bool OCCLUDED, bool COUNT>
  static __device__ __forceinline__ bool intersect(const LaunchParams& P, uint32_t ref, RayState& r, WorkCounters& wc, uint32_t rayIdx)
  {
    const TriRecord* __restrict__ prims = P.accel.prims;
    const uint32_t first = ref & ((1u << TRI_START_BITS) - 1u);
    const uint32_t count = (ref >> TRI_START_BITS) & 31u;
    for (uint32_t b = 0; b < count; b += 4) {
      const float tfarBlock = r.tfar; // all lanes of a block see the tfar at block entry
      const uint32_t nb = min(4u, count - b);
      bool found = false;
      TriHit best;
      uint32_t bestPrim = 0, bestGeom = 0;
      best.t = RT_INF;
      // all records of the block are requested before the first one is used (one memory round trip per block instead
      // of one per triangle); slots past the leaf end re-read the last record and are skipped below
      for (uint32_t g = 0; g < nb; g += TRI_FETCH) {
        float4 A[TRI_FETCH], B[TRI_FETCH], C[TRI_FETCH];
#pragma unroll
        for (uint32_t k = 0; k < TRI_FETCH; k++) {
          const float4* tp = (const float4*)(prims + first + b + min(g + k, nb - 1u));
          A[k] = tp[0]; B[k] = tp[1]; C[k] = tp[2];
        }
#pragma unroll
        for (uint32_t k = 0; k < TRI_FETCH; k++) {
          if (g + k >= nb) break;
          if (COUNT) wc.prims++;
          TriHit h;
          bool ok = PLUECKER ? pluecker(r, A[k], B[k], C[k], tfarBlock, h) : moeller(r, A[k], B[k], C[k], tfarBlock, h);
          if (ok && P.exclOffsets) { // filter re-trace: a candidate the host filter rejected before stays rejected
            const uint32_t e1 = P.exclOffsets[rayIdx + 1];
            for (uint32_t e = P.exclOffsets[rayIdx]; e < e1; e++) {
              const uint2 x = P.exclPairs[e];
              if (x.x == __float_as_uint(A[k].w) && x.y == __float_as_uint(B[k].w)) ok = false;
            }
          }
          if (ok) {
            if (OCCLUDED) return true; // Occluded1EpilogM: any valid lane (no ray mask, no filter)
            // select_min over valid lanes, lowest lane wins ties (vfloat4_sse2.h:654-659)
            if (!found || h.t < best.t) {
              best = h;
              bestGeom = __float_as_uint(A[k].w);
              bestPrim = __float_as_uint(B[k].w);
              found = true;
            }
          }
        }
      }
      if (found) { // Intersect1EpilogM, intersector_epilog.h:293-305
        r.tfar = best.t;
        r.ngx = best.ngx; r.ngy = best.ngy; r.ngz = best.ngz;
        r.u = best.u; r.v = best.v;
        r.primID = bestPrim; r.geomID = bestGeom;
        r.hit = 1u;
      }
    }
    return false;
  }
};

} // namespace dev

hipError_t launch_trace_tri(const LaunchParams& p, hipStream_t stream)
{
  // Triangle4v <-> robust traversal, Triangle4 <-> fast traversal (bvh_intersector1_bvh8.cpp:27-29)
  if (p.poolKernel) {
    if (p.accel.kind == ACCEL_TRI_PLUECKER) return dev::launch_leaf_pool<dev::TriLeaf<true>, true>(p, stream);
    return dev::launch_leaf_pool<dev::TriLeaf<false>, false>(p, stream);
  }
  if (p.accel.kind == ACCEL_TRI_PLUECKER) return dev::launch_leaf<dev::TriLeaf<true>, true>(p, stream);
  return dev::launch_leaf<dev::TriLeaf<false>, false>(p, stream);
}

} // namespace rtamd
