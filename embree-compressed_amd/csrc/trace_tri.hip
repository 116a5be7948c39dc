// Triangle leaves of the quantized BVH8: Triangle4v / Pluecker (robust) and Triangle4 / Moeller (fast).
// 48-byte records fetched as three dwordx4; every group of 4 records from the leaf start is one block with the
// reference's 4-wide SIMD semantics.
//   block loop            kernels/geometry/intersector_iterators.h:32-36 (ArrayIntersector1)
//   epilog / tie rules    kernels/geometry/intersector_epilog.h:226-307 (closest), :388-450 (any hit)
#include "trace_loop.hip.h"
#include "trace_pool.hip.h"
#include "trace_service.hip.h"
#include "trace_cull.hip.h"

namespace rtamd {
namespace dev {

// Records requested per memory round trip inside a block of 4 (1, 2 or 4): more hides latency, fewer saves registers.
#ifndef TRI_FETCH
#define TRI_FETCH 4
#endif

template <bool PLUECKER> struct TriLeaf
{
  static constexpr bool OCTET = true;
  static constexpr bool CONST_NG = false;
  static constexpr int GROUP = 8;
  static constexpr bool HIT_IN_MEMORY = false;
  // Both forms stay in the lane kernel.  Measured with the lane-per-ray form dropped (127 VGPRs, four waves per SIMD): random
  // rays 18.0 -> 19.2 Grays/s in flight, 0.098 -> 0.089 ms alone, but rays that visit many full leaves lose: camera rays 11.4 ->
  // 10.8, bounce rays 10.3 -> 9.4, shadow rays 15.9 -> 13.2 Grays/s (8 lanes per ray test a 4-triangle leaf at half occupancy;
  // with 64 rays at a leaf the lane-per-ray block loop is the better use of the wave).
  static constexpr bool OCTET_ONLY = false;
  static constexpr int MIN_WAVES = TRACE_MIN_WAVES_PER_SIMD;
  static __device__ __forceinline__ bool octet_ok(const LaunchParams&) { return true; }
  static __device__ __forceinline__ void prepare() {}

  // Child-parallel form (trace_loop.hip.h): the 8 lanes of an octet test 8 consecutive records of the leaf of the ray in
  // exchange row `x`, i.e. two of the reference's blocks of 4 per pass: lanes 0-3 block A, lanes 4-7 block B.  Block
  // semantics are kept exactly: A is tested against the tfar at its entry, its minimum-t lane (lowest lane on ties) wins;
  // B's depth test `Ts <= absDen * tfar` is then evaluated with the tfar A left behind (so an equal-t hit of the later
  // block replaces the earlier one, triangle_intersector_pluecker.h:117), its winner chosen the same way.  The winning
  // lane writes the hit into the row (words 0..7 = t, Ng, u, v, geomID, primID; word 9 = 1).
  template <bool OCCLUDED, bool COUNT>
  static __device__ __forceinline__ void octet_pass(const LaunchParams& P, float* x, bool valid, uint32_t lid, WorkCounters& wc)
  {
    const TriRecord* __restrict__ prims = P.accel.prims;
    const uint32_t k = lid & 7u, sh = lid & 56u;
    RayState r;
    r.ox = x[0]; r.oy = x[1]; r.oz = x[2]; r.tnear = x[3];
    r.dx = x[4]; r.dy = x[5]; r.dz = x[6]; r.tfar = x[7];
    const uint32_t ref = __float_as_uint(x[8]);
    const uint32_t first = ref & ((1u << TRI_START_BITS) - 1u);
    uint32_t cnt = valid ? (ref >> TRI_START_BITS) & 31u : 0u;
    float tfar = r.tfar;
    for (uint32_t b = 0; __ballot(b < cnt) != 0ull; b += 8u) {
      const bool present = b + k < cnt;
      const float4* tp = (const float4*)(prims + first + (present ? b + k : 0u));
      const float4 A = tp[0], B = tp[1], C = tp[2];
      if (COUNT && present) wc.prims++;
      TriHit h;
      h.t = RT_INF; h.Ts = 0.f; h.absDen = 0.f;
      bool ok = (PLUECKER ? pluecker(r, A, B, C, tfar, h) : moeller(r, A, B, C, tfar, h)) && present;
      if (ok && P.exclOffsets) { // filter re-trace: a candidate the host filter rejected before stays rejected
        const uint32_t rayIdx = __float_as_uint(x[10]);
        const uint32_t e1 = P.exclOffsets[rayIdx + 1];
        for (uint32_t e = P.exclOffsets[rayIdx]; e < e1; e++) {
          const uint2 q = P.exclPairs[e];
          if (q.x == __float_as_uint(A.w) && q.y == __float_as_uint(B.w)) ok = false;
        }
      }
      const uint32_t m8 = (uint32_t)(__ballot(ok) >> sh) & 0xffu;
      if (OCCLUDED) { // Occluded1EpilogM: any valid lane
        if (m8 != 0u) {
          if (k == 0u) x[9] = __uint_as_float(1u);
          cnt = 0u;
        }
        continue;
      }
      // block A (also evaluated, unused, in the lanes of block B: the quads reduce separately)
      float tq = ok ? h.t : RT_INF;
      tq = fminf(tq, __uint_as_float(dpp_u32<DPP_XOR1>(__float_as_uint(tq))));
      tq = fminf(tq, __uint_as_float(dpp_u32<DPP_XOR2>(__float_as_uint(tq))));
      const float tqm = __uint_as_float(dpp_u32<DPP_HALF_MIRROR>(__float_as_uint(tq)));
      const float tA = k < 4u ? tq : tqm; // minimum of block A, in all 8 lanes
      const bool hasA = (m8 & 0x0fu) != 0u;
      const float tfarB = hasA ? tA : tfar;
      const bool okB = ok && k >= 4u && (h.Ts <= h.absDen * tfarB);
      float tb = okB ? h.t : RT_INF;
      tb = fminf(tb, __uint_as_float(dpp_u32<DPP_XOR1>(__float_as_uint(tb))));
      tb = fminf(tb, __uint_as_float(dpp_u32<DPP_XOR2>(__float_as_uint(tb))));
      const float tbm = __uint_as_float(dpp_u32<DPP_HALF_MIRROR>(__float_as_uint(tb)));
      const float tB = k >= 4u ? tb : tbm; // minimum of block B, in all 8 lanes
      const uint32_t wA = (uint32_t)(__ballot(ok && k < 4u && h.t == tA) >> sh) & 0x0fu;
      const uint32_t wB = (uint32_t)(__ballot(okB && h.t == tB) >> sh) & 0xf0u;
      const uint32_t winner = wB != 0u ? (uint32_t)__ffs(wB) - 1u : (wA != 0u ? (uint32_t)__ffs(wA) - 1u : 8u);
      if (k == winner) { // Intersect1EpilogM, intersector_epilog.h:293-305
        x[0] = h.t; x[1] = h.ngx; x[2] = h.ngy; x[3] = h.ngz; x[4] = h.u; x[5] = h.v;
        x[6] = A.w; x[7] = B.w;
        x[9] = __uint_as_float(1u);
      }
      tfar = wB != 0u ? tB : (wA != 0u ? tA : tfar);
    }
  }

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

hipError_t launch_cull(const LaunchParams& p, hipStream_t stream)
{
  // every subdivision accel and Triangle4v traverse robustly, Triangle4 with the fast test (same choice as the traversal kernels)
  const bool robust = p.accel.kind != ACCEL_TRI_MOELLER;
  if (robust) return p.occluded ? dev::launch_cull_vec<true, true>(p, stream) : dev::launch_cull_vec<true, false>(p, stream);
  return p.occluded ? dev::launch_cull_vec<false, true>(p, stream) : dev::launch_cull_vec<false, false>(p, stream);
}

hipError_t launch_service_tri(const ServiceParams& s, hipStream_t stream)
{
#ifdef TRACE_DEV_METRIC_ONLY
  return hipErrorInvalidValue;
#else
  if (s.base.accel.kind == ACCEL_TRI_PLUECKER) return dev::launch_service_kernel<dev::TriLeaf<true>, true>(s, stream);
  return dev::launch_service_kernel<dev::TriLeaf<false>, false>(s, stream);
#endif
}

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
