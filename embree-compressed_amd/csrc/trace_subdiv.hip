// Subdivision-surface leaves of the quantized BVH8 on gfx950:
//   GridCellLeaf   eager path: one 3x3-vertex cell = 8 Pluecker triangles with patch-uv mapping
//                  (kernels/geometry/grid_soa_intersector1.h:44-117, Gather3x3 grid_soa.h:198-245, MapUV :137-156,
//                   decodeUV :248-257, Intersect1EpilogMU intersector_epilog.h:460-530)
//   CbvhLeaf<MODE> fork path: traversal of one compressed per-sub-grid BVH blob
//                  (kernels/geometry/compressed.h:454-752, node decode compressed_node.h:488-510,
//                   helpers compressed_help.h:86-308, leaf z decode compressed_leaf.h:93-111)
// The cBVH's implicit complete quadtree has a compile-time depth C here (the kernel is instantiated per C), so its
// depth-first, nearest-first traversal is a chain of C nested loops that keeps every level's parent box in registers
// instead of the reference's 16-entry box stack.  The visiting order is the reference's: ascending uint(tNear),
// equal distances -> lower child index first (compressed.h:690-749); cells use the tNear/tFar of the node decode
// that produced them (compressed.h:544-549,617); the traversal ray's tfar is never tightened (compressed.h:523).
// Scalar arithmetic of the fork is written without fused operations (the reference leaves contraction to its
// compiler; this path is "parity unpinned").
#include "trace_loop.hip.h"
#include "trace_pool.hip.h"
#include "trace_service.hip.h"

namespace rtamd {
namespace dev {

// ---------------------------------------------------------------------------------------------------
// eager: grid cell
// ---------------------------------------------------------------------------------------------------
struct RelV
{
  float x, y, z;
};

// Pluecker on vertices already relative to the ray origin; returns un-mapped barycentrics.
__device__ __forceinline__ bool pluecker_rel(const RayState& r, const RelV a, const RelV b, const RelV c, float tfarBlock, TriHit& h)
{
  const float e0x = c.x - a.x, e0y = c.y - a.y, e0z = c.z - a.z;
  const float e1x = a.x - b.x, e1y = a.y - b.y, e1z = a.z - b.z;
  const float e2x = b.x - c.x, e2y = b.y - c.y, e2z = b.z - c.z;
  float sx = c.x + a.x, sy = c.y + a.y, sz = c.z + a.z;
  const float U = dot3(msub(e0y, sz, e0z * sy), msub(e0z, sx, e0x * sz), msub(e0x, sy, e0y * sx), r.dx, r.dy, r.dz);
  sx = a.x + b.x; sy = a.y + b.y; sz = a.z + b.z;
  const float V = dot3(msub(e1y, sz, e1z * sy), msub(e1z, sx, e1x * sz), msub(e1x, sy, e1y * sx), r.dx, r.dy, r.dz);
  sx = b.x + c.x; sy = b.y + c.y; sz = b.z + c.z;
  const float W = dot3(msub(e2y, sz, e2z * sy), msub(e2z, sx, e2x * sz), msub(e2x, sy, e2y * sx), r.dx, r.dy, r.dz);
  const float minUVW = fminf(fminf(U, V), W);
  const float maxUVW = fmaxf(fmaxf(U, V), W);
  if (!((minUVW >= 0.0f) | (maxUVW <= 0.0f))) return false;
  const float ab_x = e0z * e1y, ab_y = e0x * e1z, ab_z = e0y * e1x;
  const float bc_x = e1z * e2y, bc_y = e1x * e2z, bc_z = e1y * e2x;
  const float cab_x = msub(e0y, e1z, ab_x), cab_y = msub(e0z, e1x, ab_y), cab_z = msub(e0x, e1y, ab_z);
  const float cbc_x = msub(e1y, e2z, bc_x), cbc_y = msub(e1z, e2x, bc_y), cbc_z = msub(e1x, e2y, bc_z);
  const float ngx = fabsf(ab_x) < fabsf(bc_x) ? cab_x : cbc_x;
  const float ngy = fabsf(ab_y) < fabsf(bc_y) ? cab_y : cbc_y;
  const float ngz = fabsf(ab_z) < fabsf(bc_z) ? cab_z : cbc_z;
  const float dn = dot3(ngx, ngy, ngz, r.dx, r.dy, r.dz);
  const float den = dn + dn;
  const float absDen = fabsf(den);
  const uint32_t sgnDen = __float_as_uint(den) & 0x80000000u;
  const float tn = dot3(a.x, a.y, a.z, ngx, ngy, ngz);
  const float T = tn + tn;
  const float Ts = xorf(T, sgnDen);
  if (!(absDen * r.tnear < Ts)) return false;
  if (!(Ts <= absDen * tfarBlock)) return false;
  if (!(den != 0.0f)) return false;
  const float rcpDen = 1.0f / den;
  h.t = T * rcpDen;
  const float UVW = U + V + W;
  const float rcpUVW = fabsf(UVW) < 1e-18f ? 0.0f : 1.0f / UVW;
  h.u = U * rcpUVW;
  h.v = V * rcpUVW;
  h.ngx = ngx; h.ngy = ngy; h.ngz = ngz;
  return true;
}

// Filter re-trace (row f3, LaunchParams::exclOffsets): a triangle of patch (geomID, primID) whose distance is bit-equal to one the
// host filter rejected for this ray in an earlier round stays rejected (Intersect1EpilogMU offers the candidates of a cell one by
// one, intersector_epilog.h:488-509; here the host does, between passes).
__device__ __forceinline__ bool cell_candidate_excluded(const LaunchParams& P, uint32_t rayIdx, uint32_t geomID, uint32_t primID, float t)
{
  const uint32_t e1 = P.exclOffsets[rayIdx + 1];
  for (uint32_t e = P.exclOffsets[rayIdx]; e < e1; e++) {
    const uint2 q = P.exclPairs[e];
    if (q.x == geomID && q.y == primID && P.exclT[e] == __float_as_uint(t)) return true;
  }
  return false;
}

struct GridCellLeaf
{
  static constexpr bool OCTET = true;
  static constexpr bool OCTET_ONLY = true; // lane kernel: cells are always tested 8 lanes per ray (intersect() below serves the ray-pool kernel)
  static constexpr int GROUP = 8;
  static constexpr bool HIT_IN_MEMORY = false;
  static constexpr bool CONST_NG = false;
  static constexpr int MIN_WAVES = TRACE_MIN_WAVES_PER_SIMD;
  static __device__ __forceinline__ bool octet_ok(const LaunchParams&) { return true; }
  static __device__ __forceinline__ void prepare() {}

  // Child-parallel form (trace_loop.hip.h): lane 8g+k tests triangle k of the cell of the ray in exchange row `x`
  // (words 0..7 = org, tnear, dir, tfar; word 8 = leaf ref).  Same vertex differences, same Pluecker test against the tfar
  // at cell entry, same winner (minimum t, lowest triangle on ties, select_min vfloat4_sse2.h:654-659) as intersect()
  // below; the winning lane maps uv and writes the hit into the row (words 0..7 = t, Ng, u, v, geomID, primID; word 9 = 1).
  template <bool OCCLUDED, bool COUNT>
  static __device__ __forceinline__ void octet_pass(const LaunchParams& P, float* x, bool valid, uint32_t lid, WorkCounters& wc)
  {
    const uint32_t k = lid & 7u;
    RayState r;
    r.ox = x[0]; r.oy = x[1]; r.oz = x[2]; r.tnear = x[3];
    r.dx = x[4]; r.dy = x[5]; r.dz = x[6]; r.tfar = x[7];
    const uint32_t idx = __float_as_uint(x[8]) & 0x7FFFFFFFu;
    const float* gp = (const float*)(P.accel.blobs + (size_t)idx * sizeof(GridCell));
    // Gather3x3 lane -> vertex indices (grid_soa.h:218-223), one nibble per triangle
    const uint32_t i0 = (0x74634130u >> (4u * k)) & 15u, i1 = (0x55442211u >> (4u * k)) & 15u, i2 = (0x87765443u >> (4u * k)) & 15u;
    const RelV a = RelV{gp[i0] - r.ox, gp[9u + i0] - r.oy, gp[18u + i0] - r.oz};
    const RelV b = RelV{gp[i1] - r.ox, gp[9u + i1] - r.oy, gp[18u + i1] - r.oz};
    const RelV c = RelV{gp[i2] - r.ox, gp[9u + i2] - r.oy, gp[18u + i2] - r.oz};
    TriHit h;
    h.t = RT_INF;
    bool ok = pluecker_rel(r, a, b, c, r.tfar, h) && valid;
    if (ok && P.exclOffsets) ok = !cell_candidate_excluded(P, __float_as_uint(x[10]), __float_as_uint(gp[36]), __float_as_uint(gp[37]), h.t);
    const uint32_t mask8 = (uint32_t)(__ballot(ok) >> (lid & 56u)) & 0xffu;
    if (COUNT && valid && k == 0u) {
      wc.prims++;
      wc.inner += (OCCLUDED && mask8) ? (unsigned long long)__ffs(mask8) : 8ull; // the lane-per-ray any-hit loop stops at the first valid triangle
    }
    if (OCCLUDED) {
      if (valid && k == 0u && mask8 != 0u) x[9] = __uint_as_float(1u);
      return;
    }
    float tmin = ok ? h.t : RT_INF;
    tmin = fminf(tmin, __uint_as_float(dpp_u32<DPP_XOR1>(__float_as_uint(tmin))));
    tmin = fminf(tmin, __uint_as_float(dpp_u32<DPP_XOR2>(__float_as_uint(tmin))));
    tmin = fminf(tmin, __uint_as_float(dpp_u32<DPP_HALF_MIRROR>(__float_as_uint(tmin))));
    const uint32_t win8 = (uint32_t)(__ballot(ok && h.t == tmin) >> (lid & 56u)) & 0xffu;
    if (ok && win8 != 0u && k == (uint32_t)__ffs(win8) - 1u) {
      // MapUV (grid_soa.h:148-155): uv = u*uv1 + v*uv2 + (1-u-v)*uv0 on the 16-bit decoded vertex uvs
      const uint32_t w0 = __float_as_uint(gp[27u + i0]), w1 = __float_as_uint(gp[27u + i1]), w2 = __float_as_uint(gp[27u + i2]);
      const float s = 8.0f / 0x10000;
      const float u0 = (float)(w0 & 0xffffu) * s, v0 = (float)(w0 >> 16) * s;
      const float u1 = (float)(w1 & 0xffffu) * s, v1 = (float)(w1 >> 16) * s;
      const float u2 = (float)(w2 & 0xffffu) * s, v2 = (float)(w2 >> 16) * s;
      const float bu = h.u, bv = h.v;
      const float bw = (1.0f - bu) - bv;
      x[0] = h.t; x[1] = h.ngx; x[2] = h.ngy; x[3] = h.ngz;
      x[4] = (bu * u1 + bv * u2) + bw * u0;
      x[5] = (bu * v1 + bv * v2) + bw * v0;
      x[6] = gp[36];
      x[7] = gp[37];
      x[9] = __uint_as_float(1u);
    }
  }

  template <bool OCCLUDED, bool COUNT>
  static __device__ __forceinline__ bool intersect(const LaunchParams& P, uint32_t ref, RayState& r, WorkCounters& wc, uint32_t rayIdx)
  {
    const uint32_t idx = ref & 0x7FFFFFFFu;
    const float4* gp = (const float4*)(P.accel.blobs + (size_t)idx * sizeof(GridCell));
    float f[40];
#pragma unroll
    for (int k = 0; k < 10; k++) {
      const float4 q = gp[k];
      f[4 * k] = q.x; f[4 * k + 1] = q.y; f[4 * k + 2] = q.z; f[4 * k + 3] = q.w;
    }
    if (COUNT) wc.prims++;
    // f[0..8] px, f[9..17] py, f[18..26] pz, f[27..35] packed uv, f[36] geomID, f[37] primID
    RelV p[9];
#pragma unroll
    for (int k = 0; k < 9; k++) p[k] = RelV{f[k] - r.ox, f[9 + k] - r.oy, f[18 + k] - r.oz};
    // Gather3x3 lane -> (v0,v1,v2) vertex indices r*3+c (grid_soa.h:218-223)
    constexpr int T0[8] = {0, 3, 1, 4, 3, 6, 4, 7};
    constexpr int T1[8] = {1, 1, 2, 2, 4, 4, 5, 5};
    constexpr int T2[8] = {3, 4, 4, 5, 6, 7, 7, 8};
    const float tfarBlock = r.tfar;
    bool found = false;
    TriHit best;
    best.t = RT_INF;
    int bestLane = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) {
      if (COUNT) wc.inner++;
      TriHit h;
      if (pluecker_rel(r, p[T0[l]], p[T1[l]], p[T2[l]], tfarBlock, h)) {
        if (P.exclOffsets && cell_candidate_excluded(P, rayIdx, __float_as_uint(f[36]), __float_as_uint(f[37]), h.t)) continue;
        if (OCCLUDED) return true;
        if (!found || h.t < best.t) { // select_min: lowest lane among equal minima
          best = h;
          bestLane = l;
          found = true;
        }
      }
    }
    if (found) {
      // MapUV (grid_soa.h:148-155): uv = u*uv1 + v*uv2 + (1-u-v)*uv0 on the 16-bit decoded vertex uvs
      uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
      for (int l = 0; l < 8; l++)
        if (l == bestLane) {
          w0 = __float_as_uint(f[27 + T0[l]]);
          w1 = __float_as_uint(f[27 + T1[l]]);
          w2 = __float_as_uint(f[27 + T2[l]]);
        }
      const float s = 8.0f / 0x10000;
      const float u0 = (float)(w0 & 0xffffu) * s, v0 = (float)(w0 >> 16) * s;
      const float u1 = (float)(w1 & 0xffffu) * s, v1 = (float)(w1 >> 16) * s;
      const float u2 = (float)(w2 & 0xffffu) * s, v2 = (float)(w2 >> 16) * s;
      const float bu = best.u, bv = best.v;
      const float bw = (1.0f - bu) - bv;
      r.u = (bu * u1 + bv * u2) + bw * u0;
      r.v = (bu * v1 + bv * v2) + bw * v0;
      r.tfar = best.t;
      r.ngx = best.ngx; r.ngy = best.ngy; r.ngz = best.ngz;
      r.geomID = __float_as_uint(f[36]);
      r.primID = __float_as_uint(f[37]);
      r.hit = 1u;
    }
    return false;
  }
};

// ---------------------------------------------------------------------------------------------------
// fork: cBVH blob
// ---------------------------------------------------------------------------------------------------
// MODE_FULL (subdiv_accel=bvh4.compressed.full, compressed.h:40,774): the box mode's cells and hits over quadtree nodes that hold
// their four child boxes as floats (96 B, plane-major: lx[4] ux[4] ly[4] uy[4] lz[4] uz[4]) instead of a 4-byte code of planes
// relative to the parent box - nothing to decode and nothing inherited from the parent, so a descent carries only the node index.
enum { MODE_BOX = 0, MODE_LEAF = 1, MODE_GRID = 2, MODE_FULL = 3 };
#ifndef TRACE_CBVH_TWO_STAGE
#define TRACE_CBVH_TWO_STAGE 1 // quad form: frustum tests and walks of blob visits run as separate, separately batched stages (trace_loop.hip.h)
#endif
#ifndef TRACE_CBVH_PREFETCH
#define TRACE_CBVH_PREFETCH 0 // measured r2 (profiles/r02_prefetch_and_priority_ab.txt): 2-3 % slower alone and in flight - the later lines are not what a blob visit waits for
#endif

// Node decode tables (compressed_node.h:488-510): border planes, mid planes, and their complements 1-x (the same
// fp32 subtraction the reference performs per decode, done once).  They live in LDS: a lookup is one ds_read (~100
// cycles) instead of a global load per plane; with eight lookups per node and up to five dependent levels per blob the
// global-memory version put several extra L1/L2 round trips on every ray's critical path.
__constant__ float c_tblBorder[8] = {0.000f, 0.005f, 0.010f, 0.050f, 0.100f, 0.200f, 0.400f, 0.600f};
__constant__ float c_tblMid[8] = {0.00f, 0.40f, 0.48f, 0.49f, 0.50f, 0.51f, 0.52f, 0.60f};
enum { TBL_BORDER = 0, TBL_MID = 8, TBL_ONE_MINUS_MID = 16, TBL_ONE_MINUS_BORDER = 24 };
__device__ __forceinline__ float* cbvh_tables()
{
  __shared__ float tbl[32];
  return tbl;
}
__device__ __forceinline__ void cbvh_tables_init()
{
  float* t = cbvh_tables();
  if (threadIdx.x < 8) {
    const float b = c_tblBorder[threadIdx.x], m = c_tblMid[threadIdx.x];
    t[TBL_BORDER + threadIdx.x] = b;
    t[TBL_MID + threadIdx.x] = m;
    t[TBL_ONE_MINUS_MID + threadIdx.x] = 1.f - m;
    t[TBL_ONE_MINUS_BORDER + threadIdx.x] = 1.f - b;
  }
  __syncthreads();
}

__device__ __forceinline__ uint32_t compact1by1(uint32_t x)
{
  x &= 0x55555555u;
  x = (x ^ (x >> 1)) & 0x33333333u;
  x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
  x = (x ^ (x >> 4)) & 0x00ff00ffu;
  x = (x ^ (x >> 8)) & 0x0000ffffu;
  return x;
}

// Kept small on purpose (the blob walk is what holds this kernel at the register limit): node / cell / grid addresses are
// constant offsets from the header (the level count is a template parameter), rdir_far, the local origin, rcp_edges and extent
// are re-derived or re-read from the (L1-hot) header where they are used.
struct CbvhCtx
{
  const CbvhHeader* H;
  RayState* r;
  // projected ray (compressed.h:470-508) and its robust TravRay<4,4,true> constants (:522-523)
  float ox, oy, oz, dx, dy, dz;
  float rnx, rny, rnz; // rdir_near; rdir_far = rdir_near * (1+3ulp) is formed per node
  float travFar; // travRay.tfar: fixed for the whole blob
  float tfar;    // local tfar, shrinks with hits
  float near, zFactor;
  uint32_t special; // 0/1 in a vector register (see RayState::hit)
  const float* row; // quad form only: the ray's exchange row (org at words 0..2), so that the origin need not stay in registers
#ifdef RTAMD_TRACE_RAY
  uint32_t dbgRay; // development aid: -DRTAMD_TRACE_RAY=<index> prints the cell tests of that ray (tools/fork_diff.py)
#endif
};
template <int LEVELS> struct CbvhGeom
{
  static constexpr uint32_t ELEMS = ((1u << (2 * LEVELS)) - 1u) / 3u; // inner nodes of the complete quadtree
  // section offsets of the blob: accel.h (line 0 = CbvhHeader, then CbvhMid + nodes, then cells / grid, then CbvhTail)
  static __device__ __forceinline__ const uint32_t* nodes(const CbvhHeader* H) { return (const uint32_t*)((const uint8_t*)H + CBVH_NODES_OFFSET); }
  static __device__ __forceinline__ const uint8_t* leaves(const CbvhHeader* H) { return (const uint8_t*)H + cbvh_payload_offset(LEVELS, MODE_LEAF); }
  static __device__ __forceinline__ const float* grid(const CbvhHeader* H) { return (const float*)((const uint8_t*)H + cbvh_payload_offset(LEVELS, MODE_GRID)); }
  static __device__ __forceinline__ const float* fullNode(const CbvhHeader* H, uint32_t curr) { return (const float*)((const uint8_t*)H + CBVH_NODES_OFFSET) + 24u * curr; }
};
__device__ __forceinline__ const CbvhMid* cbvh_mid(const CbvhHeader* H) { return (const CbvhMid*)((const uint8_t*)H + CBVH_HEADER_BYTES); }
template <int MODE, int LEVELS> __device__ __forceinline__ const CbvhTail* cbvh_tail(const CbvhHeader* H)
{
  return (const CbvhTail*)((const uint8_t*)H + cbvh_tail_offset(LEVELS, MODE));
}
typedef float f32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));

__device__ __forceinline__ void project3(const float* m, float x, float y, float z, float& ox, float& oy, float& oz)
{
  const float px = (m[0] * x + m[1] * y) + m[2];
  const float py = (m[3] * x + m[4] * y) + m[5];
  const float pw = (m[6] * x + m[7] * y) + m[8];
  ox = px / pw; oy = py / pw; oz = z;
}

// 2-D line intersection, compressed_help.h:93-106
__device__ __forceinline__ float intersect_line(float p2x, float p2y, float p3x, float p3y, float ox, float oy, float dx, float dy)
{
  const float vx = p2x - ox, vy = p2y - oy;
  const float lx = p3x - p2x, ly = p3y - p2y;
  const float t1 = (ly * vx - lx * vy) / (ly * dx - lx * dy);
  const float t2 = (dx * vy - dy * vx) / (lx * dy - ly * dx);
  return (t2 >= 0.f && t2 <= 1.f) ? t1 : __builtin_nanf("");
}

// commit a hit found inside the blob (compressed.h:570-591 / :631-653)
template <int MODE, int LEVELS, bool ROW = false> __device__ __forceinline__ void cbvh_commit(CbvhCtx& c, float u, float v, float t)
{
  RayState& r = *c.r;
  const CbvhHeader* H = c.H;
  const CbvhMid* M = cbvh_mid(H);
  r.u = M->uv0x + u * M->uv1x;
  r.v = M->uv0y + v * M->uv1y;
  r.ngx = 1.f; r.ngy = 0.f; r.ngz = 0.f; // dummy normal
  r.geomID = M->geomID;
  r.primID = M->primID;
  r.hit = 1u;
  c.tfar = t;
  if (c.special) {
    // flat frame: un-project the local hit point and measure the distance in the rotated world frame (:583-587)
    float px, py, pz;
    project3(cbvh_tail<MODE, LEVELS>(H)->iproj, c.ox + c.dx * t, c.oy + c.dy * t, c.oz + c.dz * t, px, py, pz);
    const float* S = H->space; // lOrg again (same fma chain as at blob entry)
    const float rox = ROW ? c.row[0] : r.ox, roy = ROW ? c.row[1] : r.oy, roz = ROW ? c.row[2] : r.oz;
    const float lox = madd(rox, S[0], madd(roy, S[1], roz * S[2]));
    const float loy = madd(rox, S[3], madd(roy, S[4], roz * S[5]));
    const float loz = madd(rox, S[6], madd(roy, S[7], roz * S[8]));
    const float ex = px - lox, ey = py - loy, ez = pz - loz;
    r.tfar = sqrtf(madd(ex, ex, madd(ey, ey, ez * ez)));
  } else
    r.tfar = t / c.zFactor + c.near;
}

// intersect_patch, compressed_help.h:135-229, split in two: patch_candidate evaluates everything that does not depend on the
// distance found so far and returns 0 = no hit, 1 = hit at the entry point t1 (cell too small, or entry point inside the slab:
// accepted whatever tt is), 2 = secant hit at t inside [t1, t2] (accepted iff t < tt); intersect_patch adds that last test.
// The quad form (below) evaluates the candidates of the four cells of a node in parallel and resolves them in visiting order.
__device__ __forceinline__ uint32_t patch_candidate(uint32_t idx, float rcp_edges, float dz, float t1, float t2, float v0, float v1, float v2,
                                                    float v3, float blx, float bly, float bhx, float bhy, const CbvhCtx& c, float& u, float& v,
                                                    float& t)
{
  const float px = t1 * c.dx + c.ox, py = t1 * c.dy + c.oy, pz = t1 * c.dz + c.oz;
  const float p2x = t2 * c.dx + c.ox, p2y = t2 * c.dy + c.oy, p2z = t2 * c.dz + c.oz;
  const float lenX = 1.0f / (bhx - blx), lenY = 1.0f / (bhy - bly); // reference: rcp()
  const float fx1 = (px - blx) * lenX, fy1 = (py - bly) * lenY;
  const float mx = (float)compact1by1(idx), my = (float)compact1by1(idx >> 1);
  if (t2 - t1 < 1.0E-6f) { // too small a patch
    t = t1;
    u = (fx1 + mx) * rcp_edges;
    v = (fy1 + my) * rcp_edges;
    return 1u;
  }
  const float fx2 = (p2x - blx) * lenX, fy2 = (p2y - bly) * lenY;
  const float dx1 = 1.f - fx1, dy1 = 1.f - fy1;
  float z1 = v0 * dx1 * dy1 + v1 * fx1 * dy1 + v2 * dx1 * fy1 + v3 * fx1 * fy1;
  const float dx2 = 1.f - fx2, dy2 = 1.f - fy2;
  float z2 = v0 * dx2 * dy2 + v1 * fx2 * dy2 + v2 * dx2 * fy2 + v3 * fx2 * fy2;
  if (pz >= z1 && pz <= z1 + dz) { // entry point inside the slab
    t = t1;
    u = (fx1 + mx) * rcp_edges;
    v = (fy1 + my) * rcp_edges;
    return 1u;
  }
  if (pz > z1 + dz) { z1 += dz; z2 += dz; }
  const float alpha = p2z - z2, beta = z1 - pz;
  const float ts = (t1 * alpha + t2 * beta) / (alpha + beta);
  const float d = (ts - t1) / (t2 - t1);
  const float fx = fx2 - fx1, fy = fy2 - fy1;
  if (ts >= t1 && ts <= t2) {
    u = (fx * d + fx1 + mx) * rcp_edges;
    v = (fy * d + fy1 + my) * rcp_edges;
    t = ts;
    return 2u;
  }
  return 0u;
}

__device__ __forceinline__ bool intersect_patch(uint32_t idx, float rcp_edges, float dz, float t1, float t2, float v0, float v1, float v2,
                                                float v3, float blx, float bly, float bhx, float bhy, const CbvhCtx& c, float& u, float& v,
                                                float& tt)
{
  float t, cu, cv;
  const uint32_t kind = patch_candidate(idx, rcp_edges, dz, t1, t2, v0, v1, v2, v3, blx, bly, bhx, bhy, c, cu, cv, t);
  if (kind == 1u || (kind == 2u && t < tt)) {
    u = cu; v = cv; tt = t;
    return true;
  }
  return false;
}

// intersect_triangle, compressed_help.h:232-275 (world space, updates the ray in place)
__device__ __forceinline__ bool grid_triangle(const float* v0, const float* v1, const float* v2, RayState& r)
{
  const float a = v0[0] - v1[0], b = v0[1] - v1[1], cc = v0[2] - v1[2];
  const float d = v0[0] - v2[0], e = v0[1] - v2[1], f = v0[2] - v2[2];
  const float g = r.dx, h = r.dy, i = r.dz;
  const float j = v0[0] - r.ox, k = v0[1] - r.oy, l = v0[2] - r.oz;
  float common1 = e * i - h * f, common2 = g * f - d * i, common3 = d * h - e * g;
  const float M = 1.0f / (a * common1 + b * common2 + cc * common3); // reference: rcp()
  float beta = j * common1 + k * common2 + l * common3;
  common1 = a * k - j * b; common2 = j * cc - a * l; common3 = b * l - k * cc;
  float gamma = i * common1 + h * common2 + g * common3;
  float tt = -(f * common1 + e * common2 + d * common3);
  beta *= M; gamma *= M; tt *= M;
  if (tt > 0 && tt < r.tfar && tt >= r.tnear)
    if (beta > 0 && gamma > 0 && beta + gamma <= 1) {
      r.tfar = tt;
      r.u = beta;
      r.v = gamma;
      return true;
    }
  return false;
}

// `zz` (leaf mode): the cell's two height bytes, z12 in bits 0..7 and z34 in bits 8..15
template <int MODE, int LEVELS, bool COUNT>
__device__ __forceinline__ void cbvh_cell(CbvhCtx& c, uint32_t idx, uint32_t zz, float tN, float tF, float blx, float bly, float blz,
                                          float bhx, float bhy, float bhz, WorkCounters& wc)
{
  if (MODE == MODE_LEAF) { // compressed.h:539-593
    if (tN >= c.tfar) return;
    const float dimZ = bhz - blz;
    const float range = (1.f + 2.f * c.H->extent) * dimZ;
    const float dz = 0.0625f * range; // getDelta() = rcp(16) (compressed_leaf.h:109-111)
    const float rcpF = 0.0625f * range;
    const float off = blz - dimZ * c.H->extent;
    const uint32_t z12 = zz & 0xffu, z34 = (zz >> 8) & 0xffu;
    const float z1 = off + rcpF * (float)(z12 >> 4), z2 = off + rcpF * (float)(z12 & 0xf);
    const float z3 = off + rcpF * (float)(z34 >> 4), z4 = off + rcpF * (float)(z34 & 0xf);
    float u, v, t = c.tfar;
    if (COUNT) wc.inner++;
    if (intersect_patch(idx, c.H->rcp_edges, dz, tN, tF, z1, z2, z3, z4, blx, bly, bhx, bhy, c, u, v, t)) cbvh_commit<MODE, LEVELS>(c, u, v, t);
  } else if (MODE == MODE_GRID) { // compressed.h:597-611, compressed_help.h:278-308
    const uint32_t x = compact1by1(idx), y = compact1by1(idx >> 1);
    const uint32_t w = (1u << LEVELS) + 1u; // grid_width
    const float* g0 = CbvhGeom<LEVELS>::grid(c.H) + 3 * (y * w + x);
    const float* g1 = g0 + 3;
    const float* g2 = g0 + 3 * w;
    const float* g3 = g2 + 3;
    const float q0[3] = {g0[0], g0[1], g0[2]}, q1[3] = {g1[0], g1[1], g1[2]}, q2[3] = {g2[0], g2[1], g2[2]}, q3[3] = {g3[0], g3[1], g3[2]};
    RayState& r = *c.r;
    if (COUNT) wc.inner++;
    const bool hit1 = grid_triangle(q0, q1, q2, r);
    const bool hit2 = grid_triangle(q3, q2, q1, r);
    if (hit1 || hit2) {
      const float uu = hit2 ? ((float)x + (1.f - r.u)) * c.H->rcp_edges : ((float)x + r.u) * c.H->rcp_edges;
      const float vv = hit2 ? ((float)y + (1.f - r.v)) * c.H->rcp_edges : ((float)y + r.v) * c.H->rcp_edges;
      r.ngx = 1.f; r.ngy = 0.f; r.ngz = 0.f;
      const CbvhMid* M = cbvh_mid(c.H);
      r.u = M->uv0x + uu * M->uv1x;
      r.v = M->uv0y + vv * M->uv1y;
      r.geomID = M->geomID;
      r.primID = M->primID;
      r.hit = 1u;
      c.tfar = (r.tfar - c.near) * c.zFactor;
    }
  } else { // voxel, compressed.h:614-654
    const float is = tN;
    if (is <= c.tfar) {
      const float u = (((c.ox + c.dx * is) - blx) / (bhx - blx) + (float)compact1by1(idx)) * c.H->rcp_edges;
      const float v = (((c.oy + c.dy * is) - bly) / (bhy - bly) + (float)compact1by1(idx >> 1)) * c.H->rcp_edges;
      if (COUNT) wc.inner++;
      cbvh_commit<MODE, LEVELS>(c, u, v, is);
    }
  }
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// One inner node with REM levels below it (REM == 1: its children are cells); `w` is the node's word (bytes xz, x,
// yz, y).  The four children of node `curr` are the consecutive words 4*curr+1 .. 4*curr+4 (or, below the last inner
// level, the consecutive 2-byte cells 4*curr+1-elems ..): they are requested with ONE load as soon as the node is
// entered, so the round trip overlaps the decode and the slab test of this node instead of following them per child.
template <int MODE, int LEVELS, int REM, bool COUNT>
__device__ __forceinline__ void cbvh_node(CbvhCtx& c, uint32_t curr, uint32_t w, float blx, float bly, float blz, float bhx, float bhy,
                                          float bhz, WorkCounters& wc)
{
  uint32_t cw[4] = {0u, 0u, 0u, 0u};
  float tN[4], tF[4];
  uint32_t d[4];
  uint32_t mask = 0;
  float lx0, lx1, ux0, ux1, ly0, ly1, uy0, uy1, lz, uz; // coded modes: two x columns, two y rows, one z slab
  f32x4_a16 FX0, FX1, FY0, FY1;                         // full mode: the four children's own x / y planes (cells need them for u, v)
  if (COUNT) wc.inner++;
  const bool negx = !(c.rnx >= 0.f), negy = !(c.rny >= 0.f), negz = !(c.rnz >= 0.f);
  const float ulp3 = 1.0f + 3.0f * 1.1920929e-7f;
  const float rfx = c.rnx * ulp3, rfy = c.rny * ulp3, rfz = c.rnz * ulp3; // rdir_far (node_intersector1.h:116-118)
  if constexpr (MODE == MODE_FULL) {
    // Node<flavor::ref>::getNode (compressed_node.h:700-714): the stored boxes; intersectNodeRobust (node_intersector1.h:351-368) per child
    const f32x4_a16* N = (const f32x4_a16*)CbvhGeom<LEVELS>::fullNode(c.H, curr);
    FX0 = N[0]; FX1 = N[1]; FY0 = N[2]; FY1 = N[3];
    const f32x4_a16 FZ0 = N[4], FZ1 = N[5];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float nX = ((negx ? FX1[k] : FX0[k]) - c.ox) * c.rnx, fX = ((negx ? FX0[k] : FX1[k]) - c.ox) * rfx;
      const float nY = ((negy ? FY1[k] : FY0[k]) - c.oy) * c.rny, fY = ((negy ? FY0[k] : FY1[k]) - c.oy) * rfy;
      const float nZ = ((negz ? FZ1[k] : FZ0[k]) - c.oz) * c.rnz, fZ = ((negz ? FZ0[k] : FZ1[k]) - c.oz) * rfz;
      tN[k] = fmaxf(fmaxf(nX, nY), fmaxf(nZ, 0.f));
      tF[k] = fminf(fminf(fX, fY), fminf(fZ, c.travFar));
      const bool h = tN[k] <= tF[k];
      d[k] = h ? __float_as_uint(tN[k]) : 0xFFFFFFFFu;
      mask |= h ? (1u << k) : 0u;
    }
    lx0 = lx1 = ux0 = ux1 = ly0 = ly1 = uy0 = uy1 = lz = uz = 0.f;
  } else {
    if constexpr (REM > 1) {
      const u32x4_a4 q = *(const u32x4_a4*)(CbvhGeom<LEVELS>::nodes(c.H) + (4u * curr + 1u));
      cw[0] = q.x; cw[1] = q.y; cw[2] = q.z; cw[3] = q.w;
    } else if constexpr (MODE == MODE_LEAF) {
      const u32x2_a4 q = *(const u32x2_a4*)(CbvhGeom<LEVELS>::leaves(c.H) + 2u * (4u * curr + 1u - CbvhGeom<LEVELS>::ELEMS)); // 4 cells x 2 bytes
      cw[0] = q.x; cw[1] = q.y;
    }
    // getNode, compressed_node.h:488-510
    const float dimX = bhx - blx, dimY = bhy - bly, dimZ = bhz - blz;
    const float* T = cbvh_tables();
    lx0 = T[TBL_BORDER + ((w >> 5) & 7)] * dimX + blx;            // X1: children 0,2 lower
    lx1 = T[TBL_MID + ((w >> 2) & 7)] * dimX + blx;               // X2: children 1,3 lower
    ux0 = T[TBL_ONE_MINUS_MID + ((w >> 13) & 7)] * dimX + blx;    // X3: children 0,2 upper
    ux1 = T[TBL_ONE_MINUS_BORDER + ((w >> 10) & 7)] * dimX + blx; // X4: children 1,3 upper
    ly0 = T[TBL_BORDER + ((w >> 21) & 7)] * dimY + bly;           // Y1: children 0,1 lower
    ly1 = T[TBL_MID + ((w >> 18) & 7)] * dimY + bly;              // Y2: children 2,3 lower
    uy0 = T[TBL_ONE_MINUS_MID + ((w >> 29) & 7)] * dimY + bly;    // Y3: children 0,1 upper
    uy1 = T[TBL_ONE_MINUS_BORDER + ((w >> 26) & 7)] * dimY + bly; // Y4: children 2,3 upper
    lz = (float)(w & 3) * 0.25f * dimZ + blz;              // table3 = k/4
    uz = (1.f - (float)((w >> 16) & 3) * 0.25f) * dimZ + blz;

    // intersectNodeRobust (node_intersector1.h:351-368) on the two x columns, two y rows, one z slab
    const float nX0 = ((negx ? ux0 : lx0) - c.ox) * c.rnx, fX0 = ((negx ? lx0 : ux0) - c.ox) * rfx;
    const float nX1 = ((negx ? ux1 : lx1) - c.ox) * c.rnx, fX1 = ((negx ? lx1 : ux1) - c.ox) * rfx;
    const float nY0 = ((negy ? uy0 : ly0) - c.oy) * c.rny, fY0 = ((negy ? ly0 : uy0) - c.oy) * rfy;
    const float nY1 = ((negy ? uy1 : ly1) - c.oy) * c.rny, fY1 = ((negy ? ly1 : uy1) - c.oy) * rfy;
    const float nZ = ((negz ? uz : lz) - c.oz) * c.rnz, fZ = ((negz ? lz : uz) - c.oz) * rfz;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      tN[k] = fmaxf(fmaxf((k & 1) ? nX1 : nX0, (k & 2) ? nY1 : nY0), fmaxf(nZ, 0.f));
      tF[k] = fminf(fminf((k & 1) ? fX1 : fX0, (k & 2) ? fY1 : fY0), fminf(fZ, c.travFar));
      const bool h = tN[k] <= tF[k];
      d[k] = h ? __float_as_uint(tN[k]) : 0xFFFFFFFFu;
      mask |= h ? (1u << k) : 0u;
    }
  }
  if (mask == 0) return;
  // nearest first, equal distances -> lower index first (compressed.h:690-749)
  uint32_t rank[4] = {0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = a + 1; b < 4; b++) {
      const uint32_t bFirst = d[b] < d[a] ? 1u : 0u; // tie -> a (lower index) first
      rank[a] += bFirst;
      rank[b] += 1u - bFirst;
    }
  const int nhit = __popc(mask);
  for (int k = 0; k < nhit; k++) {
    int rr = 0;
#pragma unroll
    for (int q = 1; q < 4; q++) rr = ((mask >> q) & 1u) && rank[q] == (uint32_t)k ? q : rr;
    if (!(((mask >> rr) & 1u) && rank[rr] == (uint32_t)k)) rr = 0;
    float cbx0 = (rr & 1) ? lx1 : lx0, cbx1 = (rr & 1) ? ux1 : ux0;
    float cby0 = (rr & 2) ? ly1 : ly0, cby1 = (rr & 2) ? uy1 : uy0;
    if constexpr (MODE == MODE_FULL && REM == 1) {
      cbx0 = FX0[0]; cbx1 = FX1[0]; cby0 = FY0[0]; cby1 = FY1[0];
#pragma unroll
      for (int q = 1; q < 4; q++) { cbx0 = rr == q ? FX0[q] : cbx0; cbx1 = rr == q ? FX1[q] : cbx1; cby0 = rr == q ? FY0[q] : cby0; cby1 = rr == q ? FY1[q] : cby1; }
    }
    const uint32_t child = 4u * curr + 1u + (uint32_t)rr;
    uint32_t cword;
    if constexpr (REM == 1) cword = ((rr & 2) ? cw[1] : cw[0]) >> ((rr & 1) * 16);
    else cword = rr == 0 ? cw[0] : rr == 1 ? cw[1] : rr == 2 ? cw[2] : cw[3];
    if constexpr (REM == 1) {
      float tn = tN[0], tf = tF[0];
#pragma unroll
      for (int q = 1; q < 4; q++) { tn = rr == q ? tN[q] : tn; tf = rr == q ? tF[q] : tf; }
      cbvh_cell<MODE, LEVELS, COUNT>(c, child - CbvhGeom<LEVELS>::ELEMS, cword, tn, tf, cbx0, cby0, lz, cbx1, cby1, uz, wc);
    } else
      cbvh_node<MODE, LEVELS, REM - 1, COUNT>(c, child, cword, cbx0, cby0, lz, cbx1, cby1, uz, wc);
  }
}

// ---------------------------------------------------------------------------------------------------
// fork: cBVH blob, QUAD form (lane kernel): four lanes per ray, lane q of a quad = child q of the current quadtree node
// ---------------------------------------------------------------------------------------------------
// The blob walk above keeps one ray per lane: three nested levels of parent boxes, child words and visiting orders in registers
// (the kernel sat at the 168-VGPR limit of three waves per SIMD) and ~1500 dependent wave instructions per visit, most of them
// with a handful of lanes active because the rays of a wave are at different depths of different blobs.  In the quad form the
// four children of a node are decoded and slab-tested by the four lanes of a quad at once, the four frustum edges, the
// projections of the entry / exit point and the three reciprocals of the local direction go one per lane, the four cells of a
// last-level node are evaluated in parallel (patch_candidate) and then resolved in visiting order, because a hit shortens the
// local tfar the later cells see.  Control flow is uniform within a quad; values move between its lanes by DPP (fixed patterns)
// or ds_bpermute (the child chosen at run time).  Arithmetic, visiting order (nearest first, equal distances -> lower index
// first, compressed.h:690-749) and every quirk are those of the lane-per-ray form - which the ray-pool kernel still runs, so the
// two are compared byte for byte by tests/test_gpu_properties.py - hence of the reference.
enum : int { DPP_Q0 = 0x00, DPP_Q1 = 0x55, DPP_Q2 = 0xAA, DPP_Q3 = 0xFF }; // quad_perm: broadcast lane 0..3 of the quad
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) { return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v))); }
// value `v` of lane `src` (0..3, uniform within the quad) of this lane's quad
__device__ __forceinline__ uint32_t quad_get(uint32_t v, uint32_t src, uint32_t lid)
{
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lid & ~3u) + src) << 2), (int)v);
}
__device__ __forceinline__ float quad_getf(float v, uint32_t src, uint32_t lid) { return __uint_as_float(quad_get(__float_as_uint(v), src, lid)); }
__device__ __forceinline__ uint32_t quad_ballot(bool p, uint32_t lid) { return (uint32_t)(__ballot(p) >> (lid & 60u)) & 0xFu; }

// Row layout while a quad walks a blob (words of the ray's exchange row): 0..2 ray origin (kept for the flat-frame distance), 3 world
// distance of the hit so far, 4 `near`, 5 zFactor, 6 `special`, 7 / 8 u / v of the hit so far, 9 hit flag.  The commit-only part of
// the context and the hit itself live there instead of in seven registers; all four lanes write the same values.
enum : int { QR_TFAR = 3, QR_NEAR = 4, QR_ZFACTOR = 5, QR_SPECIAL = 6, QR_U = 7, QR_V = 8, QR_HIT = 9 };
template <int MODE, int LEVELS>
__device__ __forceinline__ void quad_commit(CbvhCtx& c, float u, float v, float t) // cbvh_commit for the leaf / box modes of the quad form
{
  const CbvhHeader* H = c.H;
  const CbvhMid* M = cbvh_mid(H);
  float* x = (float*)c.row;
  x[QR_U] = M->uv0x + u * M->uv1x;
  x[QR_V] = M->uv0y + v * M->uv1y;
  x[QR_HIT] = __uint_as_float(1u);
  c.tfar = t;
  if (__float_as_uint(x[QR_SPECIAL]) != 0u) {
    // flat frame: un-project the local hit point and measure the distance in the rotated world frame (:583-587)
    float px, py, pz;
    project3(cbvh_tail<MODE, LEVELS>(H)->iproj, c.ox + c.dx * t, c.oy + c.dy * t, c.oz + c.dz * t, px, py, pz);
    const float* S = H->space; // lOrg again (same fma chain as at blob entry)
    const float rox = x[0], roy = x[1], roz = x[2];
    const float lox = madd(rox, S[0], madd(roy, S[1], roz * S[2]));
    const float loy = madd(rox, S[3], madd(roy, S[4], roz * S[5]));
    const float loz = madd(rox, S[6], madd(roy, S[7], roz * S[8]));
    const float ex = px - lox, ey = py - loy, ez = pz - loz;
    x[QR_TFAR] = sqrtf(madd(ex, ex, madd(ey, ey, ez * ez)));
  } else
    x[QR_TFAR] = t / x[QR_ZFACTOR] + x[QR_NEAR];
}

template <int MODE, int LEVELS, int REM, bool COUNT>
__device__ __forceinline__ void quad_node(CbvhCtx& c, uint32_t lid, uint32_t curr, uint32_t w, float blx, float bly, float blz, float bhx, float bhy,
                                          float bhz, WorkCounters& wc)
{
  const uint32_t q = lid & 3u;
  // this lane's child: its node word / its cell's two height bytes, requested before anything else (one 16-byte / 8-byte access per quad)
  uint32_t cw = 0u;
  float lx, ux, ly, uy, lz, uz;
  if (COUNT && q == 0u) wc.inner++;
  if constexpr (MODE == MODE_FULL) {
    // Node<flavor::ref>::getNode (compressed_node.h:700-714): child q's stored box, six 16-byte accesses per quad
    const float* N = CbvhGeom<LEVELS>::fullNode(c.H, curr) + q;
    lx = N[0]; ux = N[4]; ly = N[8]; uy = N[12]; lz = N[16]; uz = N[20];
  } else {
    if constexpr (REM > 1) cw = CbvhGeom<LEVELS>::nodes(c.H)[4u * curr + 1u + q];
    else if constexpr (MODE == MODE_LEAF) cw = ((const uint16_t*)CbvhGeom<LEVELS>::leaves(c.H))[4u * curr + 1u + q - CbvhGeom<LEVELS>::ELEMS];
    // getNode, compressed_node.h:488-510, the planes of child q only (x column = q & 1, y row = q >> 1)
    const float dimX = bhx - blx, dimY = bhy - bly, dimZ = bhz - blz;
    const float* T = cbvh_tables();
    const uint32_t xl = (q & 1u) ? TBL_MID + ((w >> 2) & 7u) : TBL_BORDER + ((w >> 5) & 7u);
    const uint32_t xu = (q & 1u) ? TBL_ONE_MINUS_BORDER + ((w >> 10) & 7u) : TBL_ONE_MINUS_MID + ((w >> 13) & 7u);
    const uint32_t yl = (q & 2u) ? TBL_MID + ((w >> 18) & 7u) : TBL_BORDER + ((w >> 21) & 7u);
    const uint32_t yu = (q & 2u) ? TBL_ONE_MINUS_BORDER + ((w >> 26) & 7u) : TBL_ONE_MINUS_MID + ((w >> 29) & 7u);
    lx = T[xl] * dimX + blx; ux = T[xu] * dimX + blx;
    ly = T[yl] * dimY + bly; uy = T[yu] * dimY + bly;
    lz = (float)(w & 3) * 0.25f * dimZ + blz;
    uz = (1.f - (float)((w >> 16) & 3) * 0.25f) * dimZ + blz;
  }
  // intersectNodeRobust (node_intersector1.h:351-368) for this child
  const bool negx = !(c.rnx >= 0.f), negy = !(c.rny >= 0.f), negz = !(c.rnz >= 0.f);
  const float ulp3 = 1.0f + 3.0f * 1.1920929e-7f;
  const float rfx = c.rnx * ulp3, rfy = c.rny * ulp3, rfz = c.rnz * ulp3;
  const float nX = ((negx ? ux : lx) - c.ox) * c.rnx, fX = ((negx ? lx : ux) - c.ox) * rfx;
  const float nY = ((negy ? uy : ly) - c.oy) * c.rny, fY = ((negy ? ly : uy) - c.oy) * rfy;
  const float nZ = ((negz ? uz : lz) - c.oz) * c.rnz, fZ = ((negz ? lz : uz) - c.oz) * rfz;
  const float tN = fmaxf(fmaxf(nX, nY), fmaxf(nZ, 0.f));
  const float tF = fminf(fminf(fX, fY), fminf(fZ, c.travFar));
  const bool h = tN <= tF;
  const uint32_t nhit = (uint32_t)__popc(quad_ballot(h, lid));
  if (nhit == 0u) return;
  // visiting position of this lane's child: children that are nearer, or equally near with a lower index, come first
  // (compressed.h:690-749); 4 = not hit.  (One register per level across the descent: no lane masks, no separate hit flag.)
  const uint32_t d = h ? __float_as_uint(tN) : 0xFFFFFFFFu;
  const uint32_t d1 = dpp_u32<DPP_XOR1>(d), d2 = dpp_u32<DPP_XOR2>(d), d3 = dpp_u32<DPP_XOR3>(d);
  uint32_t pos = 0u;
  pos += d1 < d + ((q ^ 1u) < q ? 1u : 0u) ? 1u : 0u;
  pos += d2 < d + ((q ^ 2u) < q ? 1u : 0u) ? 1u : 0u;
  pos += d3 < d + ((q ^ 3u) < q ? 1u : 0u) ? 1u : 0u;
  pos = h ? pos : 4u;

  if constexpr (REM > 1) {
    for (uint32_t k = 0; k < nhit; k++) {
      const uint32_t src = (uint32_t)__ffs(quad_ballot(pos == k, lid)) - 1u; // the child visited k-th
      if constexpr (MODE == MODE_FULL) { // the child node holds its own children's boxes: only its index goes down
        quad_node<MODE, LEVELS, REM - 1, COUNT>(c, lid, 4u * curr + 1u + src, 0u, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, wc);
        continue;
      }
      const float cbx0 = quad_getf(lx, src, lid), cbx1 = quad_getf(ux, src, lid);
      const float cby0 = quad_getf(ly, src, lid), cby1 = quad_getf(uy, src, lid);
      const uint32_t cword = quad_get(cw, src, lid);
      quad_node<MODE, LEVELS, REM - 1, COUNT>(c, lid, 4u * curr + 1u + src, cword, cbx0, cby0, lz, cbx1, cby1, uz, wc);
    }
  } else if constexpr (MODE == MODE_LEAF) { // compressed.h:539-593
    // candidates of the (up to) four cells in parallel; none of this depends on the distance found so far
    float ct = 0.f, cu = 0.f, cv = 0.f;
    uint32_t kind = 0u;
    if (pos != 4u) {
      const float range = (1.f + 2.f * c.H->extent) * (uz - lz);
      const float dz = 0.0625f * range; // getDelta() = rcp(16) (compressed_leaf.h:109-111)
      const float rcpF = 0.0625f * range;
      const float off = lz - (uz - lz) * c.H->extent;
      const uint32_t z12 = cw & 0xffu, z34 = (cw >> 8) & 0xffu;
      const float z1 = off + rcpF * (float)(z12 >> 4), z2 = off + rcpF * (float)(z12 & 0xf);
      const float z3 = off + rcpF * (float)(z34 >> 4), z4 = off + rcpF * (float)(z34 & 0xf);
      kind = patch_candidate(4u * curr + 1u + q - CbvhGeom<LEVELS>::ELEMS, c.H->rcp_edges, dz, tN, tF, z1, z2, z3, z4, lx, ly, ux, uy, c, cu, cv, ct);
    }
    // resolution in visiting order: a cell is skipped once its box entry lies behind the hit so far (`tN >= tfar`), a secant
    // hit counts only in front of it (`t < tt`)
    for (uint32_t k = 0; k < nhit; k++) {
      const bool tested = pos == k && !(tN >= c.tfar);
      if (COUNT && tested) wc.inner++;
      const uint32_t acc = quad_ballot(tested && (kind == 1u || (kind == 2u && ct < c.tfar)), lid);
      if (acc != 0u) {
        const uint32_t src = (uint32_t)__ffs(acc) - 1u;
        quad_commit<MODE, LEVELS>(c, quad_getf(cu, src, lid), quad_getf(cv, src, lid), quad_getf(ct, src, lid));
      }
    }
  } else if constexpr (MODE == MODE_BOX || MODE == MODE_FULL) { // voxel, compressed.h:614-654: the box entry point is the hit
    float cu = 0.f, cv = 0.f;
    if (pos != 4u) {
      const uint32_t idx = 4u * curr + 1u + q - CbvhGeom<LEVELS>::ELEMS;
      cu = (((c.ox + c.dx * tN) - lx) / (ux - lx) + (float)compact1by1(idx)) * c.H->rcp_edges;
      cv = (((c.oy + c.dy * tN) - ly) / (uy - ly) + (float)compact1by1(idx >> 1)) * c.H->rcp_edges;
    }
    for (uint32_t k = 0; k < nhit; k++) {
      const bool take = pos == k && tN <= c.tfar;
      if (COUNT && take) wc.inner++;
      const uint32_t acc = quad_ballot(take, lid);
      if (acc != 0u) {
        const uint32_t src = (uint32_t)__ffs(acc) - 1u;
        quad_commit<MODE, LEVELS>(c, quad_getf(cu, src, lid), quad_getf(cv, src, lid), quad_getf(tN, src, lid));
      }
    }
  } else { // MODE_GRID (compressed.h:597-611): true triangles on the world ray, one cell after the other on all four lanes
    for (uint32_t k = 0; k < nhit; k++) {
      const uint32_t src = (uint32_t)__ffs(quad_ballot(pos == k, lid)) - 1u;
      WorkCounters dummy;
      cbvh_cell<MODE_GRID, LEVELS, COUNT>(c, 4u * curr + 1u + src - CbvhGeom<LEVELS>::ELEMS, 0u, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, q == 0u ? wc : dummy);
    }
  }
}

// QUAD = true: in the lane kernel blobs are walked in the quad form only (four lanes per ray, octet_pass below; 128 VGPRs, four waves
// per SIMD): the form for incoherent rays, most of which miss - the metric's workload.  QUAD = false: one ray per lane (intersect()
// below, 168 VGPRs, three waves): the better use of the wave when most lanes wait at blobs at once, i.e. for coherent camera
// rays (measured round 2, 1920x1080 primary rays: 0.754 ms against 0.975 ms for the quad form; random rays: 0.159 ms against
// 0.148 ms).  The caller's RTCIntersectContext::flags decide (RTC_INTERSECT_CONTEXT_FLAG_COHERENT -> QUAD = false), as in the
// reference, where that flag selects the coherent traversal kernels (rtcore.cpp:411-417).  The ray-pool kernel always uses intersect().
template <int MODE, int LEVELS, bool QUAD = true> struct CbvhLeaf
{
  static constexpr bool OCTET = QUAD;
  static constexpr bool OCTET_ONLY = QUAD;
  static constexpr int GROUP = 4;
  static constexpr bool CONST_NG = true; // dummy normal (1,0,0): written at store time by the lane kernel
  static constexpr bool HIT_IN_MEMORY = QUAD; // quad form: a hit is written to the ray record when it is found; only tfar stays in registers
  static __device__ __forceinline__ bool octet_ok(const LaunchParams&) { return true; }
  // bytes of a blob of this mode and level (cbvh_blob_bytes, cbvh_encode.cpp)
  static constexpr uint32_t BLOB_BYTES = cbvh_stride(LEVELS, MODE);

  // Local ray and frustum test of the quad form: rotate the ray into the blob's frame (compressed.h:458-459, on all four lanes), then
  // intersect_frustum (compressed_help.h:109-133): edge q on lane q (t1x: corners 1-3, t2x: 2-4, t1y: 1-2, t2y: 3-4; corner j at
  // box[2j], box[2j+1]), every lane combines the four values in the reference's order.  false = the ray leaves the blob here.
  static __device__ __forceinline__ bool quad_frustum(const CbvhHeader* H, const RayState& r, uint32_t q, float& lox, float& loy, float& loz, float& ldx,
                                                      float& ldy, float& ldz, float& near, float& far)
  {
    const float* S = H->space;
    lox = madd(r.ox, S[0], madd(r.oy, S[1], r.oz * S[2]));
    loy = madd(r.ox, S[3], madd(r.oy, S[4], r.oz * S[5]));
    loz = madd(r.ox, S[6], madd(r.oy, S[7], r.oz * S[8]));
    ldx = madd(r.dx, S[0], madd(r.dy, S[1], r.dz * S[2]));
    ldy = madd(r.dx, S[3], madd(r.dy, S[4], r.dz * S[5]));
    ldz = madd(r.dx, S[6], madd(r.dy, S[7], r.dz * S[8]));
    const float* B = H->box;
    const uint32_t a = 2u + ((0x4020u >> (4u * q)) & 15u), b = 2u + ((0x6264u >> (4u * q)) & 15u); // start / end corner of edge q
    const float tq = intersect_line(B[a], B[a + 1u], B[b], B[b + 1u], lox, loy, ldx, ldy);
    const float t1x = dpp_f32<DPP_Q0>(tq), t2x = dpp_f32<DPP_Q1>(tq), t1y = dpp_f32<DPP_Q2>(tq), t2y = dpp_f32<DPP_Q3>(tq);
    const float rz = 1.0f / (fabsf(ldz) < 1e-18f ? 1e-18f : ldz); // rcp_safe
    const float orz = loz * rz;
    const float t1z = B[0] * rz - orz, t2z = B[1] * rz - orz;
    const float near1 = fminf(fminf(t1x, t2x), fminf(t1y, t2y));
    const float far1 = fmaxf(fmaxf(t1x, t2x), fmaxf(t1y, t2y));
    near = fmaxf(fmaxf(fminf(t1z, t2z), near1), r.tnear);
    far = fminf(fminf(fmaxf(t1z, t2z), far1), r.tfar);
    return near <= far && near1 == near1 && far1 == far1;
  }

  // Two-stage blob visits (trace_loop.hip.h): the frustum test alone, for the ray in exchange row `x`; lane 0 of the quad sets word 9
  // when the ray passes.  On the metric's rays 28 % of the visits end here; the walk (octet_pass) runs later, for the rays that
  // passed, batched over several leaf phases - it repeats the test on the same ray and header, so the two decisions agree.
  static constexpr bool TWO_STAGE = QUAD && TRACE_CBVH_TWO_STAGE != 0;
  static constexpr bool TWO_STAGE_TEST = TRACE_CBVH_TWO_STAGE == 1; // 2 = development variant: arrivals are parked without the test stage
  static __device__ __forceinline__ void frustum_pass(const LaunchParams& P, float* x, bool valid, uint32_t lid)
  {
    if (!valid) return; // uniform within the quad
    const uint32_t q = lid & 3u;
    RayState r;
    r.ox = x[0]; r.oy = x[1]; r.oz = x[2]; r.tnear = x[3];
    r.dx = x[4]; r.dy = x[5]; r.dz = x[6]; r.tfar = x[7];
    const uint32_t idx = __float_as_uint(x[8]) & 0x7FFFFFFFu;
    const CbvhHeader* H = (const CbvhHeader*)(P.accel.blobs + (size_t)idx * P.accel.blobStride);
    float lox, loy, loz, ldx, ldy, ldz, near, far;
    const bool pass = quad_frustum(H, r, q, lox, loy, loz, ldx, ldy, ldz, near, far);
    if (q == 0u && pass) x[9] = __uint_as_float(1u);
  }

  // Quad form of intersect() below for the ray in exchange row `x` (words 0..7 = org, tnear, dir, tfar; word 8 = leaf ref); lane
  // q of the quad `lid >> 2`.  On a hit lane 0 of the quad writes tfar, u, v, geomID, primID into words 0, 4..7 and sets word 9.
  template <bool OCCLUDED, bool COUNT>
  static __device__ __forceinline__ void octet_pass(const LaunchParams& P, float* x, bool valid, uint32_t lid, WorkCounters& wc)
  {
    if (!valid) return; // uniform within the quad
    const uint32_t q = lid & 3u;
    RayState r;
    r.ox = x[0]; r.oy = x[1]; r.oz = x[2]; r.tnear = x[3];
    r.dx = x[4]; r.dy = x[5]; r.dz = x[6]; r.tfar = x[7];
    r.hit = 0u;
    const uint32_t idx = __float_as_uint(x[8]) & 0x7FFFFFFFu;
    const CbvhHeader* H = (const CbvhHeader*)(P.accel.blobs + (size_t)idx * P.accel.blobStride);
#if TRACE_CBVH_PREFETCH
    // The header fills the first two 128-byte lines of a blob; the deeper nodes and the cells lie in the next ones, and the walk
    // would meet each of them as a separate dependent miss.  Their first touch is issued here, together with the header's.
    uint32_t pf = 0u;
    if (!OCCLUDED) pf = ((const uint32_t*)H)[min(64u + 32u * (q & 1u), BLOB_BYTES / 4u - 1u)];
#endif
    if (OCCLUDED) { // the fork's occluded() stub (compressed.h:754-756): see intersect() below
      const CbvhTail* T = cbvh_tail<MODE, LEVELS>(H);
      const float zx = fabsf(r.dx) < 1e-18f ? 1e-18f : r.dx, zy = fabsf(r.dy) < 1e-18f ? 1e-18f : r.dy, zz = fabsf(r.dz) < 1e-18f ? 1e-18f : r.dz;
      const float ulp3 = 1.0f + 3.0f * 1.1920929e-7f;
      const float rnx = 1.0f / zx, rny = 1.0f / zy, rnz = 1.0f / zz;
      const float nx = ((rnx >= 0.f ? T->wlo[0] : T->whi[0]) - r.ox) * rnx, fx = ((rnx >= 0.f ? T->whi[0] : T->wlo[0]) - r.ox) * (rnx * ulp3);
      const float ny = ((rny >= 0.f ? T->wlo[1] : T->whi[1]) - r.oy) * rny, fy = ((rny >= 0.f ? T->whi[1] : T->wlo[1]) - r.oy) * (rny * ulp3);
      const float nz = ((rnz >= 0.f ? T->wlo[2] : T->whi[2]) - r.oz) * rnz, fz = ((rnz >= 0.f ? T->whi[2] : T->wlo[2]) - r.oz) * (rnz * ulp3);
      const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, fmaxf(r.tnear, 0.f)));
      const float tf = fminf(fminf(fx, fy), fminf(fz, fmaxf(r.tfar, 0.f)));
      if (q == 0u && tn <= tf) x[9] = __uint_as_float(1u);
      return;
    }
    CbvhCtx c;
    c.H = H;
    c.r = &r;
    c.row = x;
    const uint32_t rootWord = H->rootWord; // line 0 (accel.h): nothing outside it is read before the frustum test
    float lox, loy, loz, ldx, ldy, ldz, near, far;
    if (!quad_frustum(H, r, q, lox, loy, loz, ldx, ldy, ldz, near, far)) return;
    if (COUNT && q == 0u) wc.prims++; // counters of a cBVH accel: leaves = blob visits, prims = visits that pass the frustum test (walks), inner = quadtree nodes entered + cells tested
#if TRACE_CBVH_PREFETCH
    asm volatile("" ::"v"(pf)); // (loads return in order: this costs no wait beyond the header's)
#endif
    c.near = near;
    // projected ray between entry and exit point (:470-508): lane 0 / 1 = x / y of the entry point, lane 2 / 3 = of the exit point
    float tx, ty, tz;
    {
      const float tp = (q & 2u) ? far : near;
      const float X = lox + ldx * tp, Y = loy + ldy * tp;
      const float* m = H->proj;
      const uint32_t row = 3u * (q & 1u);
      const float pn = (m[row] * X + m[row + 1u] * Y) + m[row + 2u];
      const float pw = (m[6] * X + m[7] * Y) + m[8];
      const float pq = pn / pw;
      c.ox = dpp_f32<DPP_Q0>(pq); c.oy = dpp_f32<DPP_Q1>(pq); tx = dpp_f32<DPP_Q2>(pq); ty = dpp_f32<DPP_Q3>(pq);
      c.oz = loz + ldz * near;
      tz = loz + ldz * far;
    }
    c.dx = tx - c.ox; c.dy = ty - c.oy; c.dz = tz - c.oz;
    c.special = 0u;
    c.zFactor = 0.f;
    const float g_epsilon = 1.0E-4f;
    if (fabsf(c.dx) < g_epsilon && fabsf(c.dy) < g_epsilon && fabsf(c.dz) < g_epsilon) {
      c.dz = copysignf(1.f, ldz);
      c.oz -= c.dz;
      c.zFactor = 3.402823466e+38f;
      c.tfar = 3.402823466e+38f;
    } else if (fabsf(c.dz) < g_epsilon) {
      c.special = 1u;
      const float len2 = madd(c.dx, c.dx, madd(c.dy, c.dy, c.dz * c.dz));
      c.tfar = sqrtf(len2);
      const float rl = 1.0f / sqrtf(len2); // reference: rsqrt + Newton step
      c.dx *= rl; c.dy *= rl; c.dz *= rl;
    } else {
      const float len2 = madd(c.dx, c.dx, madd(c.dy, c.dy, c.dz * c.dz));
      const float rl = 1.0f / sqrtf(len2);
      c.dx *= rl; c.dy *= rl; c.dz *= rl;
      c.zFactor = ldz / c.dz;
      c.tfar = (r.tfar - near) * c.zFactor;
    }
    c.travFar = c.tfar;
    { // rdir_near of the local ray: one reciprocal per lane
      const float dq = q == 0u ? c.dx : (q == 1u ? c.dy : c.dz);
      const float rq = 1.0f / (fabsf(dq) < 1e-18f ? 1e-18f : dq);
      c.rnx = dpp_f32<DPP_Q0>(rq); c.rny = dpp_f32<DPP_Q1>(rq); c.rnz = dpp_f32<DPP_Q2>(rq);
    }
    if constexpr (MODE != MODE_GRID) {
      // what only a commit needs moves into the row (see quad_commit); every lane of the quad writes the same values
      x[QR_NEAR] = c.near; x[QR_ZFACTOR] = c.zFactor; x[QR_SPECIAL] = __uint_as_float(c.special);
    }
    // root: local frame box xy in [-1,1], z from the leaf data (:517-519)
    quad_node<MODE, LEVELS, LEVELS, COUNT>(c, lid, 0u, rootWord, -1.f, -1.f, H->box[0], 1.f, 1.f, H->box[1], wc);
    if constexpr (MODE != MODE_GRID) {
      if (q == 0u && __float_as_uint(x[QR_HIT]) != 0u) { // results into the words the kernel skeleton reads (ids from the header)
        const float t = x[QR_TFAR], u = x[QR_U], v = x[QR_V];
        x[0] = t; x[4] = u; x[5] = v;
        x[6] = __uint_as_float(cbvh_mid(H)->geomID); x[7] = __uint_as_float(cbvh_mid(H)->primID);
      }
    } else if (q == 0u && r.hit) {
      x[0] = r.tfar; x[4] = r.u; x[5] = r.v;
      x[6] = __uint_as_float(cbvh_mid(H)->geomID); x[7] = __uint_as_float(cbvh_mid(H)->primID);
      x[9] = __uint_as_float(1u);
    }
  }

  // quad form: ~10 VGPRs per quadtree level (119 / 130 / 142 / 152 / 162 for C = 1..5 when bounded at three waves per SIMD);
  // up to three levels are compiled for FOUR waves per SIMD (<= 128 VGPRs)
#ifndef TRACE_CBVH_MIN_WAVES
#define TRACE_CBVH_MIN_WAVES(levels) ((levels) <= 3 ? 4 : TRACE_MIN_WAVES_PER_SIMD)
#endif
  // (grid mode walks with the world ray and its hit in registers - its triangle tests feed each other through ray.tfar - and stays at three;
  // the lane-per-ray form keeps every level's parent box in registers: three waves up to C = 3, two beyond)
  // (full mode, quad form: a descent carries only the node index - 113..117 VGPRs at every level: four waves throughout)
  static constexpr int MIN_WAVES = !QUAD ? (LEVELS >= 4 ? 2 : TRACE_MIN_WAVES_PER_SIMD) : (MODE == MODE_GRID ? (LEVELS >= 4 ? 2 : TRACE_MIN_WAVES_PER_SIMD) : (MODE == MODE_FULL ? 4 : TRACE_CBVH_MIN_WAVES(LEVELS)));
  static __device__ __forceinline__ void prepare() { cbvh_tables_init(); }

  template <bool OCCLUDED, bool COUNT>
  static __device__ __forceinline__ bool intersect(const LaunchParams& P, uint32_t ref, RayState& r, WorkCounters& wc, uint32_t rayIdxDbg)
  {
    const uint32_t idx = ref & 0x7FFFFFFFu;
    const uint8_t* blob = P.accel.blobs + (size_t)idx * P.accel.blobStride;
    const CbvhHeader* H = (const CbvhHeader*)blob;
    if (OCCLUDED) {
      // the fork's occluded() is a stub returning true for every leaf the outer traversal reaches
      // (compressed.h:754-756): the leaf is reached iff the robust slab test of its bounds passes
      const CbvhTail* T = cbvh_tail<MODE, LEVELS>(H);
      const float zx = fabsf(r.dx) < 1e-18f ? 1e-18f : r.dx, zy = fabsf(r.dy) < 1e-18f ? 1e-18f : r.dy, zz = fabsf(r.dz) < 1e-18f ? 1e-18f : r.dz;
      const float ulp3 = 1.0f + 3.0f * 1.1920929e-7f;
      const float rnx = 1.0f / zx, rny = 1.0f / zy, rnz = 1.0f / zz;
      const float nx = ((rnx >= 0.f ? T->wlo[0] : T->whi[0]) - r.ox) * rnx, fx = ((rnx >= 0.f ? T->whi[0] : T->wlo[0]) - r.ox) * (rnx * ulp3);
      const float ny = ((rny >= 0.f ? T->wlo[1] : T->whi[1]) - r.oy) * rny, fy = ((rny >= 0.f ? T->whi[1] : T->wlo[1]) - r.oy) * (rny * ulp3);
      const float nz = ((rnz >= 0.f ? T->wlo[2] : T->whi[2]) - r.oz) * rnz, fz = ((rnz >= 0.f ? T->whi[2] : T->wlo[2]) - r.oz) * (rnz * ulp3);
      const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, fmaxf(r.tnear, 0.f)));
      const float tf = fminf(fminf(fx, fy), fminf(fz, fmaxf(r.tfar, 0.f)));
      return tn <= tf;
    }

    CbvhCtx c;
#ifdef RTAMD_TRACE_RAY
    c.dbgRay = rayIdxDbg;
#endif
    c.H = H;
    const uint32_t rootWord = H->rootWord; // line 0 (accel.h), requested together with the rest of the header: no extra round trip after the frustum test
    c.r = &r;
    // rotate the ray into the local frame (:458-459; xfmPoint/xfmVector are fma chains, linearspace3.h:168-169)
    const float* S = H->space;
    const float lox = madd(r.ox, S[0], madd(r.oy, S[1], r.oz * S[2]));
    const float loy = madd(r.ox, S[3], madd(r.oy, S[4], r.oz * S[5]));
    const float loz = madd(r.ox, S[6], madd(r.oy, S[7], r.oz * S[8]));
    const float ldx = madd(r.dx, S[0], madd(r.dy, S[1], r.dz * S[2]));
    const float ldy = madd(r.dx, S[3], madd(r.dy, S[4], r.dz * S[5]));
    const float ldz = madd(r.dx, S[6], madd(r.dy, S[7], r.dz * S[8]));

    // intersect_frustum, compressed_help.h:109-133
    float near = r.tnear, far = r.tfar;
    {
      const float* B = H->box;
      const float rz = 1.0f / (fabsf(ldz) < 1e-18f ? 1e-18f : ldz); // rcp_safe
      const float orz = loz * rz;
      const float t1z = B[0] * rz - orz, t2z = B[1] * rz - orz;
      const float t1x = intersect_line(B[2], B[3], B[6], B[7], lox, loy, ldx, ldy);
      const float t2x = intersect_line(B[4], B[5], B[8], B[9], lox, loy, ldx, ldy);
      const float t1y = intersect_line(B[2], B[3], B[4], B[5], lox, loy, ldx, ldy);
      const float t2y = intersect_line(B[6], B[7], B[8], B[9], lox, loy, ldx, ldy);
      const float near1 = fminf(fminf(t1x, t2x), fminf(t1y, t2y));
      const float far1 = fmaxf(fmaxf(t1x, t2x), fmaxf(t1y, t2y));
      near = fmaxf(fmaxf(fminf(t1z, t2z), near1), near);
      far = fminf(fminf(fmaxf(t1z, t2z), far1), far);
      if (!(near <= far && near1 == near1 && far1 == far1)) return false;
    }
    if (COUNT) wc.prims++; // visits that pass the frustum test (see octet_pass)
    c.near = near;
#ifdef RTAMD_TRACE_RAY
    if (rayIdxDbg == RTAMD_TRACE_RAY) printf("  GPU blob prim %u: lOrg %a %a %a lDir %a %a %a near %a far %a ray.tfar %a\n", cbvh_mid(H)->primID, lox, loy, loz, ldx, ldy, ldz, near, far, r.tfar);
#endif

    // projected ray between entry and exit point (:470-508)
    float tx, ty, tz;
    project3(H->proj, lox + ldx * near, loy + ldy * near, loz + ldz * near, c.ox, c.oy, c.oz);
    project3(H->proj, lox + ldx * far, loy + ldy * far, loz + ldz * far, tx, ty, tz);
    c.dx = tx - c.ox; c.dy = ty - c.oy; c.dz = tz - c.oz;
    c.special = 0u;
    c.zFactor = 0.f;
    const float g_epsilon = 1.0E-4f;
    if (fabsf(c.dx) < g_epsilon && fabsf(c.dy) < g_epsilon && fabsf(c.dz) < g_epsilon) {
      c.dz = copysignf(1.f, ldz);
      c.oz -= c.dz;
      c.zFactor = 3.402823466e+38f;
      c.tfar = 3.402823466e+38f;
    } else if (fabsf(c.dz) < g_epsilon) {
      c.special = 1u;
      const float len2 = madd(c.dx, c.dx, madd(c.dy, c.dy, c.dz * c.dz));
      c.tfar = sqrtf(len2);
      const float rl = 1.0f / sqrtf(len2); // reference: rsqrt + Newton step
      c.dx *= rl; c.dy *= rl; c.dz *= rl;
    } else {
      const float len2 = madd(c.dx, c.dx, madd(c.dy, c.dy, c.dz * c.dz));
      const float rl = 1.0f / sqrtf(len2);
      c.dx *= rl; c.dy *= rl; c.dz *= rl;
      c.zFactor = ldz / c.dz;
      c.tfar = (r.tfar - near) * c.zFactor;
    }
    c.travFar = c.tfar;
    {
      const float ulp3 = 1.0f + 3.0f * 1.1920929e-7f;
      const float zx = fabsf(c.dx) < 1e-18f ? 1e-18f : c.dx, zy = fabsf(c.dy) < 1e-18f ? 1e-18f : c.dy, zz = fabsf(c.dz) < 1e-18f ? 1e-18f : c.dz;
      c.rnx = 1.0f / zx; c.rny = 1.0f / zy; c.rnz = 1.0f / zz;
    }
    // root: local frame box xy in [-1,1], z from the leaf data (:517-519)
    cbvh_node<MODE, LEVELS, LEVELS, COUNT>(c, 0u, rootWord, -1.f, -1.f, H->box[0], 1.f, 1.f, H->box[1], wc);
    return false;
  }
};

template <int MODE> hipError_t launch_cbvh(const LaunchParams& p, hipStream_t stream, uint32_t levels)
{
#ifdef TRACE_DEV_METRIC_ONLY
  // Development builds (`make OUT=lib_wdev EXTRA=-DTRACE_DEV_METRIC_ONLY`, with the other objects copied from lib/: seconds instead of four
  // minutes): only the metric's kernel is instantiated - cbvh.leaf, C = 3, quad form.  Everything else fails with hipErrorInvalidValue.
  if constexpr (MODE == MODE_LEAF) { if (levels == 3 && !p.poolKernel && !p.cbvhLaneForm) return launch_leaf<CbvhLeaf<MODE, 3, true>, true>(p, stream); }
  return hipErrorInvalidValue;
#else
  if (p.poolKernel) switch (levels) {
    case 1: return launch_leaf_pool<CbvhLeaf<MODE, 1, false>, true>(p, stream);
    case 2: return launch_leaf_pool<CbvhLeaf<MODE, 2, false>, true>(p, stream);
    case 3: return launch_leaf_pool<CbvhLeaf<MODE, 3, false>, true>(p, stream);
    case 4: return launch_leaf_pool<CbvhLeaf<MODE, 4, false>, true>(p, stream);
    case 5: return launch_leaf_pool<CbvhLeaf<MODE, 5, false>, true>(p, stream);
    default: return hipErrorInvalidValue;
    }
  if (p.cbvhLaneForm) switch (levels) { // coherent batches: one ray per lane
    case 1: return launch_leaf<CbvhLeaf<MODE, 1, false>, true>(p, stream);
    case 2: return launch_leaf<CbvhLeaf<MODE, 2, false>, true>(p, stream);
    case 3: return launch_leaf<CbvhLeaf<MODE, 3, false>, true>(p, stream);
    case 4: return launch_leaf<CbvhLeaf<MODE, 4, false>, true>(p, stream);
    case 5: return launch_leaf<CbvhLeaf<MODE, 5, false>, true>(p, stream);
    default: return hipErrorInvalidValue;
    }
  switch (levels) {
  case 1: return launch_leaf<CbvhLeaf<MODE, 1, true>, true>(p, stream);
  case 2: return launch_leaf<CbvhLeaf<MODE, 2, true>, true>(p, stream);
  case 3: return launch_leaf<CbvhLeaf<MODE, 3, true>, true>(p, stream);
  case 4: return launch_leaf<CbvhLeaf<MODE, 4, true>, true>(p, stream);
  case 5: return launch_leaf<CbvhLeaf<MODE, 5, true>, true>(p, stream);
  default: return hipErrorInvalidValue;
  }
#endif
}

} // namespace dev

// service kernels (trace_service.hip.h): grid cells, and the fork's four modes in the quad form at the compression levels the tutorials use
// (C = 2: the framework's default, 3: bomberman.ecs, 4: displacement_geometry); other levels keep the call combiner
template <int MODE> static hipError_t launch_service_cbvh(const ServiceParams& s, hipStream_t stream)
{
  switch (s.base.cbvhLevels) {
#ifndef TRACE_DEV_METRIC_ONLY
  case 2: return dev::launch_service_kernel<dev::CbvhLeaf<MODE, 2, true>, true>(s, stream);
  case 4: return dev::launch_service_kernel<dev::CbvhLeaf<MODE, 4, true>, true>(s, stream);
#endif
  case 3: return dev::launch_service_kernel<dev::CbvhLeaf<MODE, 3, true>, true>(s, stream);
  default: return hipErrorInvalidValue;
  }
}

hipError_t launch_service_subdiv(const ServiceParams& s, hipStream_t stream)
{
  switch (s.base.accel.kind) {
#ifndef TRACE_DEV_METRIC_ONLY
  case ACCEL_GRIDSOA: return dev::launch_service_kernel<dev::GridCellLeaf, true>(s, stream);
  case ACCEL_CBVH_BOX: return launch_service_cbvh<dev::MODE_BOX>(s, stream);
  case ACCEL_CBVH_GRID: return launch_service_cbvh<dev::MODE_GRID>(s, stream);
  case ACCEL_CBVH_FULL: return launch_service_cbvh<dev::MODE_FULL>(s, stream);
#endif
  case ACCEL_CBVH_LEAF: return launch_service_cbvh<dev::MODE_LEAF>(s, stream);
  default: return hipErrorInvalidValue;
  }
}

hipError_t launch_trace_subdiv(const LaunchParams& p, hipStream_t stream)
{
  switch (p.accel.kind) {
#ifndef TRACE_DEV_METRIC_ONLY
  case ACCEL_GRIDSOA:
    if (p.poolKernel) return dev::launch_leaf_pool<dev::GridCellLeaf, true>(p, stream);
    return dev::launch_leaf<dev::GridCellLeaf, true>(p, stream);
#endif
  case ACCEL_CBVH_BOX: return dev::launch_cbvh<dev::MODE_BOX>(p, stream, p.cbvhLevels);
  case ACCEL_CBVH_LEAF: return dev::launch_cbvh<dev::MODE_LEAF>(p, stream, p.cbvhLevels);
  case ACCEL_CBVH_GRID: return dev::launch_cbvh<dev::MODE_GRID>(p, stream, p.cbvhLevels);
  case ACCEL_CBVH_FULL: return dev::launch_cbvh<dev::MODE_FULL>(p, stream, p.cbvhLevels);
  default: return hipErrorInvalidValue;
  }
}

} // namespace rtamd
