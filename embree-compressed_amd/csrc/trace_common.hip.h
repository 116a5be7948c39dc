// Device-side skeleton shared by all traversal kernels (gfx950): ray state, per-ray node-test constants,
// the quantized-BVH8 while-while traversal loop with the per-lane LDS stack, and the triangle edge tests.
//
// One ray per lane, 64 rays per wavefront, persistent grid-stride over the batch.  Each lane keeps its
// traversal stack in LDS (entry-major layout stack[entry][lane]: bank = lane, so the lanes of a wave never
// conflict whatever their individual stack depths) with an HBM overflow area for pathological depths.
// Node records are 96 bytes fetched as six dwordx4.  No MFMA: branchy slab / edge-function arithmetic.
//
// Restated reference algorithms (semantics kept, code new):
//   traversal loop        kernels/bvh/bvh_intersector1.cpp:40-126 (intersect), :128-209 (occluded)
//   TravRay precompute    kernels/bvh/node_intersector1.h:33-105 (fast), :108-177 (robust)
//   slab tests            node_intersector1.h:249-286 (fast, FMA form), :334-349 (robust)
//   child ordering        kernels/bvh/bvh_traverser1.h:549-666 + kernels/common/stack_item.h:39-80
//   Pluecker test         kernels/geometry/triangle_intersector_pluecker.h:79-132, finalize :41-51
//   Moeller test          kernels/geometry/triangle_intersector_moeller.h:75-113, finalize :42-48
//   FMA placement         common/math/vec3.h:193-212 (dot, cross, stable_triangle_normal)
// Compiled with -ffp-contract=off: every fused multiply-add below is explicit, exactly where the
// reference's AVX2 build has one.
//
// A kernel is this skeleton instantiated with a Leaf policy:
//   struct Leaf { static __device__ bool intersect<OCCLUDED,COUNT>(const LaunchParams&, uint32_t ref, RayState&, WorkCounters&); }
// returning true when an any-hit query is finished (ray occluded).
#pragma once
#include <hip/hip_runtime.h>

#include "accel.h"
#include "trace.h"

namespace rtamd {
namespace dev {

struct WorkCounters
{
  unsigned long long nodes = 0, leaves = 0, prims = 0, inner = 0, hits = 0, rays = 0, spills = 0;
};

#define RT_INF __builtin_huge_valf()

__device__ __forceinline__ float msub(float a, float b, float c) { return __builtin_fmaf(a, b, -c); } // a*b-c fused
__device__ __forceinline__ float madd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
  return madd(ax, bx, madd(ay, by, az * bz)); // vec3.h:193
}
__device__ __forceinline__ float xorf(float a, uint32_t signbit) { return __uint_as_float(__float_as_uint(a) ^ signbit); }

struct RayState
{
  float ox, oy, oz, tnear;
  float dx, dy, dz;
  float tfar;
  // hit
  float ngx, ngy, ngz, u, v;
  uint32_t primID, geomID;
  uint32_t hit; // 0/1.  Per-lane flags live in vector registers on purpose: a `bool` that is live across divergent
                // branches is kept as a 64-bit lane mask in scalar registers and costs three scalar instructions at
                // every join of every branch it crosses; the loop had ~100 of those per iteration.
};

// ---------------------------------------------------------------------------------------------------
// per-ray node-test constants
// ---------------------------------------------------------------------------------------------------
template <bool ROBUST> struct TravRay;

template <> struct TravRay<true> // node_intersector1.h:108-129
{
  float ox, oy, oz;
  float rnx, rny, rnz; // rdir_near; rdir_far = rdir_near*(1+3ulp) is re-derived per node (3 multiplies instead of 3 registers)
  float tnear;
  __device__ __forceinline__ bool negx() const { return !(rnx >= 0.0f); }
  __device__ __forceinline__ bool negy() const { return !(rny >= 0.0f); }
  __device__ __forceinline__ bool negz() const { return !(rnz >= 0.0f); }
  // seven words, for kernels that park the per-ray constants outside the registers (trace_pool.hip.h)
  static constexpr int WORDS = 7;
  __device__ __forceinline__ void store(float* f, int stride) const
  {
    f[0] = ox; f[stride] = oy; f[2 * stride] = oz; f[3 * stride] = rnx; f[4 * stride] = rny; f[5 * stride] = rnz; f[6 * stride] = tnear;
  }
  __device__ __forceinline__ void load(const float* f, int stride)
  {
    ox = f[0]; oy = f[stride]; oz = f[2 * stride]; rnx = f[3 * stride]; rny = f[4 * stride]; rnz = f[5 * stride]; tnear = f[6 * stride];
  }
  __device__ __forceinline__ void init(const RayState& r)
  {
    ox = r.ox; oy = r.oy; oz = r.oz;
    // zero_fix: |d| < 1e-18 -> +1e-18 (vec3fa.h:163-165), then a true division
    const float zx = fabsf(r.dx) < 1e-18f ? 1e-18f : r.dx;
    const float zy = fabsf(r.dy) < 1e-18f ? 1e-18f : r.dy;
    const float zz = fabsf(r.dz) < 1e-18f ? 1e-18f : r.dz;
    rnx = 1.0f / zx; rny = 1.0f / zy; rnz = 1.0f / zz;
    tnear = fmaxf(r.tnear, 0.0f);
  }
  __device__ __forceinline__ float nearT(float px, float py, float pz) const
  {
    return fmaxf(fmaxf((px - ox) * rnx, (py - oy) * rny), (pz - oz) * rnz);
  }
  __device__ __forceinline__ float farT(float px, float py, float pz) const
  {
    const float ulp3 = 1.0f + 3.0f * 1.1920929e-7f; // 1+3*FLT_EPSILON
    const float rfx = rnx * ulp3, rfy = rny * ulp3, rfz = rnz * ulp3;
    return fminf(fminf((px - ox) * rfx, (py - oy) * rfy), (pz - oz) * rfz);
  }
};

template <> struct TravRay<false> // node_intersector1.h:33-57, AVX2 form with org_rdir
{
  float rx, ry, rz;    // rdir = rcp_safe(dir)
  float orx, ory, orz; // org*rdir
  float tnear;
  __device__ __forceinline__ bool negx() const { return !(rx >= 0.0f); }
  __device__ __forceinline__ bool negy() const { return !(ry >= 0.0f); }
  __device__ __forceinline__ bool negz() const { return !(rz >= 0.0f); }
  static constexpr int WORDS = 7;
  __device__ __forceinline__ void store(float* f, int stride) const
  {
    f[0] = rx; f[stride] = ry; f[2 * stride] = rz; f[3 * stride] = orx; f[4 * stride] = ory; f[5 * stride] = orz; f[6 * stride] = tnear;
  }
  __device__ __forceinline__ void load(const float* f, int stride)
  {
    rx = f[0]; ry = f[stride]; rz = f[2 * stride]; orx = f[3 * stride]; ory = f[4 * stride]; orz = f[5 * stride]; tnear = f[6 * stride];
  }
  __device__ __forceinline__ void init(const RayState& r)
  {
    const float zx = fabsf(r.dx) < 1e-18f ? 1e-18f : r.dx;
    const float zy = fabsf(r.dy) < 1e-18f ? 1e-18f : r.dy;
    const float zz = fabsf(r.dz) < 1e-18f ? 1e-18f : r.dz;
    rx = 1.0f / zx; ry = 1.0f / zy; rz = 1.0f / zz; // reference: rcpps + one Newton step (vec3fa.h:133-168)
    orx = r.ox * rx; ory = r.oy * ry; orz = r.oz * rz;
    tnear = fmaxf(r.tnear, 0.0f);
  }
  __device__ __forceinline__ float nearT(float px, float py, float pz) const
  {
    return fmaxf(fmaxf(msub(px, rx, orx), msub(py, ry, ory)), msub(pz, rz, orz));
  }
  __device__ __forceinline__ float farT(float px, float py, float pz) const
  {
    return fminf(fminf(msub(px, rx, orx), msub(py, ry, ory)), msub(pz, rz, orz));
  }
};

__device__ __forceinline__ float q2f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); } // v_cvt_f32_ubyteK

// ---------------------------------------------------------------------------------------------------
// does the ray hit any child of this node?  Same decode and slab arithmetic as the lane-per-ray node step of trace_loop.hip.h
// (a child counts iff tN <= tF there), used by the root cull pre-pass (trace_cull.hip.h).
// ---------------------------------------------------------------------------------------------------
template <bool ROBUST> __device__ __forceinline__ bool node_any_child_hit(const QNode8* node, const TravRay<ROBUST>& tr, float travFar)
{
  const uint4* np = (const uint4*)node;
  const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4], n5 = np[5];
  const float ox = __uint_as_float(n0.x), oy = __uint_as_float(n0.y), oz = __uint_as_float(n0.z);
  const float sx = __uint_as_float((n0.w & 0xffu) << 23);
  const float sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23);
  const float sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
  const bool ngx = tr.negx(), ngy = tr.negy(), ngz = tr.negz();
  const uint32_t nx0 = ngx ? n3.z : n3.x, nx1 = ngx ? n3.w : n3.y;
  const uint32_t fx0 = ngx ? n3.x : n3.z, fx1 = ngx ? n3.y : n3.w;
  const uint32_t ny0 = ngy ? n4.z : n4.x, ny1 = ngy ? n4.w : n4.y;
  const uint32_t fy0 = ngy ? n4.x : n4.z, fy1 = ngy ? n4.y : n4.w;
  const uint32_t nz0 = ngz ? n5.z : n5.x, nz1 = ngz ? n5.w : n5.y;
  const uint32_t fz0 = ngz ? n5.x : n5.z, fz1 = ngz ? n5.y : n5.w;
  const uint32_t cref[8] = {n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w};
  bool any = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int kk = k & 3;
    const float npx = madd(q2f(k < 4 ? nx0 : nx1, kk), sx, ox);
    const float npy = madd(q2f(k < 4 ? ny0 : ny1, kk), sy, oy);
    const float npz = madd(q2f(k < 4 ? nz0 : nz1, kk), sz, oz);
    const float fpx = madd(q2f(k < 4 ? fx0 : fx1, kk), sx, ox);
    const float fpy = madd(q2f(k < 4 ? fy0 : fy1, kk), sy, oy);
    const float fpz = madd(q2f(k < 4 ? fz0 : fz1, kk), sz, oz);
    const float tN = fmaxf(tr.nearT(npx, npy, npz), tr.tnear);
    const float tF = fminf(tr.farT(fpx, fpy, fpz), travFar);
    any |= (tN <= tF) & (cref[k] != REF_EMPTY);
  }
  return any;
}

// ---------------------------------------------------------------------------------------------------
// child order of a node with exactly FOUR hit children
// ---------------------------------------------------------------------------------------------------
// traverseClosestHit (bvh_traverser1.h:566-635) pushes the hit children in ascending child index and sorts them on the stack:
// two children by one compare, three by a 3-comparator network, five and more by an insertion sort - all of which visit the
// nearer child first and, on equal distances, the HIGHER child index first (strict compares, the later push stays on top).
// Exactly four children go through the 5-comparator network of stack_item.h:47-61,
//   sort(s1,s2,s3,s4): if (s2<s1) xchg(s2,s1); if (s4<s3) xchg(s4,s3); if (s3<s1) xchg(s3,s1); if (s4<s2) xchg(s4,s2); if (s3<s2) xchg(s3,s2);
// (s1 = top of stack = highest index), which is not stable: on equal distances its order differs from "higher index first"
// in 56 of the 256 tie patterns.  Quantized nodes share planes between siblings, so equal entry distances are not exotic (the
// first version of the kernels ranked all cases by "nearer, then higher index": 1 ray per million on the metric scene reached two
// blobs of one patch in the other order).  sort4_network returns, for the hit child with index-order position q (0 = lowest
// child index of the four), its visiting position (0 = entered now, 3 = visited last) exactly as that network leaves the stack.
#define RTAMD_CX(da, qa, db, qb)                                                                    \
  {                                                                                                  \
    const bool sw = da < db;                                                                         \
    const uint32_t td = sw ? db : da, tq = sw ? qb : qa;                                             \
    db = sw ? da : db; qb = sw ? qa : qb;                                                            \
    da = td; qa = tq;                                                                                \
  }
__device__ __forceinline__ uint32_t sort4_network(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, uint32_t q)
{
  uint32_t d1 = e3, d2 = e2, d3 = e1, d4 = e0; // s1 = last pushed = highest child index
  uint32_t q1 = 3u, q2 = 2u, q3 = 1u, q4 = 0u;
  RTAMD_CX(d2, q2, d1, q1) // if (s2.dist < s1.dist) xchg(s2, s1)
  RTAMD_CX(d4, q4, d3, q3)
  RTAMD_CX(d3, q3, d1, q1)
  RTAMD_CX(d4, q4, d2, q2)
  RTAMD_CX(d3, q3, d2, q2)
  return q == q1 ? 0u : (q == q2 ? 1u : (q == q3 ? 2u : 3u));
}
#undef RTAMD_CX
// Lane-per-ray form: ranks of the four hit children (mask has exactly four bits) re-done by the network.
__device__ __forceinline__ void rank4_by_network(uint32_t mask, const uint32_t dist[8], uint32_t rank[8])
{
  uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t q = (uint32_t)__popc(mask & ((1u << k) - 1u));
    const bool h = (mask >> k) & 1u;
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = (h && q == (uint32_t)j) ? dist[k] : e[j];
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t q = (uint32_t)__popc(mask & ((1u << k) - 1u));
    if ((mask >> k) & 1u) rank[k] = sort4_network(e[0], e[1], e[2], e[3], q);
  }
}

// ---------------------------------------------------------------------------------------------------
// triangle tests on one record; they return the un-finalized hit like PlueckerHitM / MoellerTrumboreHitM
// ---------------------------------------------------------------------------------------------------
struct TriHit
{
  float t, u, v, ngx, ngy, ngz;
  float Ts, absDen; // operands of the depth test `Ts <= absDen * tfar` (the octet leaf step re-evaluates it with a newer tfar)
};

// Pluecker, watertight, no backface culling (triangle_intersector_pluecker.h:79-132).
__device__ __forceinline__ bool pluecker(const RayState& r, const float4 A, const float4 B, const float4 C, float tfarBlock, TriHit& h)
{
  const float v0x = A.x - r.ox, v0y = A.y - r.oy, v0z = A.z - r.oz;
  const float v1x = B.x - r.ox, v1y = B.y - r.oy, v1z = B.z - r.oz;
  const float v2x = C.x - r.ox, v2y = C.y - r.oy, v2z = C.z - r.oz;
  const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
  const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
  const float e2x = v1x - v2x, e2y = v1y - v2y, e2z = v1z - v2z;
  // U = dot(cross(e0, v2+v0), D) ; cross(a,b) = (msub(a.y,b.z,a.z*b.y), msub(a.z,b.x,a.x*b.z), msub(a.x,b.y,a.y*b.x))
  float sx = v2x + v0x, sy = v2y + v0y, sz = v2z + v0z;
  const float U = dot3(msub(e0y, sz, e0z * sy), msub(e0z, sx, e0x * sz), msub(e0x, sy, e0y * sx), r.dx, r.dy, r.dz);
  sx = v0x + v1x; sy = v0y + v1y; sz = v0z + v1z;
  const float V = dot3(msub(e1y, sz, e1z * sy), msub(e1z, sx, e1x * sz), msub(e1x, sy, e1y * sx), r.dx, r.dy, r.dz);
  sx = v1x + v2x; sy = v1y + v2y; sz = v1z + v2z;
  const float W = dot3(msub(e2y, sz, e2z * sy), msub(e2z, sx, e2x * sz), msub(e2x, sy, e2y * sx), r.dx, r.dy, r.dz);
  const float minUVW = fminf(fminf(U, V), W);
  const float maxUVW = fmaxf(fmaxf(U, V), W);
  if (!((minUVW >= 0.0f) | (maxUVW <= 0.0f))) return false;

  // Ng = stable_triangle_normal(e0,e1,e2) (vec3.h:200-212)
  const float ab_x = e0z * e1y, ab_y = e0x * e1z, ab_z = e0y * e1x;
  const float bc_x = e1z * e2y, bc_y = e1x * e2z, bc_z = e1y * e2x;
  const float cab_x = msub(e0y, e1z, ab_x), cab_y = msub(e0z, e1x, ab_y), cab_z = msub(e0x, e1y, ab_z);
  const float cbc_x = msub(e1y, e2z, bc_x), cbc_y = msub(e1z, e2x, bc_y), cbc_z = msub(e1x, e2y, bc_z);
  const float ngx = fabsf(ab_x) < fabsf(bc_x) ? cab_x : cbc_x;
  const float ngy = fabsf(ab_y) < fabsf(bc_y) ? cab_y : cbc_y;
  const float ngz = fabsf(ab_z) < fabsf(bc_z) ? cab_z : cbc_z;
  const float dn = dot3(ngx, ngy, ngz, r.dx, r.dy, r.dz);
  const float den = dn + dn; // twice()
  const float absDen = fabsf(den);
  const uint32_t sgnDen = __float_as_uint(den) & 0x80000000u;

  const float tn = dot3(v0x, v0y, v0z, ngx, ngy, ngz);
  const float T = tn + tn;
  const float Ts = xorf(T, sgnDen);
  h.Ts = Ts; h.absDen = absDen;
  if (!(absDen * r.tnear < Ts)) return false;
  if (!(Ts <= absDen * tfarBlock)) return false;
  if (!(den != 0.0f)) return false;

  // finalize (:41-51): reference uses rcp = rcpps + Newton step; a correctly rounded division is used here
  const float rcpDen = 1.0f / den;
  h.t = T * rcpDen;
  const float UVW = U + V + W;
  const float rcpUVW = fabsf(UVW) < 1e-18f ? 0.0f : 1.0f / UVW;
  h.u = U * rcpUVW;
  h.v = V * rcpUVW;
  h.ngx = ngx; h.ngy = ngy; h.ngz = ngz;
  return true;
}

// Moeller-Trumbore on (v0, e1=v0-v1, e2=v2-v0) (triangle_intersector_moeller.h:75-113,120-123).
__device__ __forceinline__ bool moeller(const RayState& r, const float4 A, const float4 B, const float4 C, float tfarBlock, TriHit& h)
{
  // Ng = cross(e2, e1)
  const float ngx = msub(C.y, B.z, C.z * B.y), ngy = msub(C.z, B.x, C.x * B.z), ngz = msub(C.x, B.y, C.y * B.x);
  const float cx = A.x - r.ox, cy = A.y - r.oy, cz = A.z - r.oz;
  // R = cross(C, D)
  const float rx = msub(cy, r.dz, cz * r.dy), ry = msub(cz, r.dx, cx * r.dz), rz = msub(cx, r.dy, cy * r.dx);
  const float den = dot3(ngx, ngy, ngz, r.dx, r.dy, r.dz);
  const float absDen = fabsf(den);
  const uint32_t sgnDen = __float_as_uint(den) & 0x80000000u;
  const float U = xorf(dot3(rx, ry, rz, C.x, C.y, C.z), sgnDen);
  const float V = xorf(dot3(rx, ry, rz, B.x, B.y, B.z), sgnDen);
  if (!((den != 0.0f) & (U >= 0.0f) & (V >= 0.0f) & (U + V <= absDen))) return false;
  const float T = xorf(dot3(ngx, ngy, ngz, cx, cy, cz), sgnDen);
  h.Ts = T; h.absDen = absDen;
  if (!((absDen * r.tnear < T) & (T <= absDen * tfarBlock))) return false;
  const float rcpAbsDen = 1.0f / absDen;
  h.t = T * rcpAbsDen;
  h.u = U * rcpAbsDen;
  h.v = V * rcpAbsDen;
  h.ngx = ngx; h.ngy = ngy; h.ngz = ngz;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// ray record I/O.  RTCRayHit: [org.xyz tnear][dir.xyz time][tfar mask id flags][Ng.xyz u][v primID geomID instID]
// ---------------------------------------------------------------------------------------------------
// Ray records are read once and hit records written once per batch: they COULD be streamed with the non-temporal hint (global_load /
// store ... nt) so that 80 MB of rays per million do not evict the accel from the 4 MiB L2 of an XCD.  Measured round 3
// (profiles/r03_layout_nt_ab.txt): nt loads make the metric kernel 2 % SLOWER alone (0.1421 -> 0.1451 ms) and 3 % in flight
// (12.0 -> 11.66 Grays/s), nt 16-byte stores cost 12-20 B of scratch per lane at the 128-VGPR limit: plain accesses stay the default.
#ifndef TRACE_RAY_NT
#define TRACE_RAY_NT 0
#endif
#ifndef TRACE_HIT_NT
#define TRACE_HIT_NT 0
#endif
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream4(const float4* p)
{
#if TRACE_RAY_NT
  const f32x4_nt v = __builtin_nontemporal_load((const f32x4_nt*)p);
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ float ld_stream(const float* p)
{
#if TRACE_RAY_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_stream4(float4* p, float4 v)
{
#if TRACE_HIT_NT
  f32x4_nt w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, (f32x4_nt*)p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st_stream(float* p, float v)
{
#if TRACE_HIT_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

template <bool VEC> __device__ __forceinline__ void load_ray(const char* p, RayState& r)
{
  if (VEC) {
    const float4 a = ld_stream4((const float4*)p);
    const float4 b = ld_stream4((const float4*)p + 1);
    const float c = ld_stream((const float*)p + 8);
    r.ox = a.x; r.oy = a.y; r.oz = a.z; r.tnear = a.w;
    r.dx = b.x; r.dy = b.y; r.dz = b.z; r.tfar = c;
  } else {
    const float* f = (const float*)p;
    r.ox = ld_stream(f); r.oy = ld_stream(f + 1); r.oz = ld_stream(f + 2); r.tnear = ld_stream(f + 3);
    r.dx = ld_stream(f + 4); r.dy = ld_stream(f + 5); r.dz = ld_stream(f + 6); r.tfar = ld_stream(f + 8);
  }
}

template <bool VEC> __device__ __forceinline__ void store_hit(char* p, const RayState& r, uint32_t instID)
{
  st_stream((float*)p + 8, r.tfar);
  if (VEC) {
    st_stream4((float4*)p + 3, make_float4(r.ngx, r.ngy, r.ngz, r.u));
    st_stream4((float4*)p + 4, make_float4(r.v, __uint_as_float(r.primID), __uint_as_float(r.geomID), __uint_as_float(instID)));
  } else {
    float* f = (float*)p;
    st_stream(f + 12, r.ngx); st_stream(f + 13, r.ngy); st_stream(f + 14, r.ngz); st_stream(f + 15, r.u); st_stream(f + 16, r.v);
    st_stream(f + 17, __uint_as_float(r.primID)); st_stream(f + 18, __uint_as_float(r.geomID)); st_stream(f + 19, __uint_as_float(instID));
  }
}


} // namespace dev
} // namespace rtamd
