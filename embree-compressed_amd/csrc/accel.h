// Device-resident acceleration-structure records shared by the host builders and the HIP kernels.
//
// Everything the kernels read lives in four flat HBM arrays per committed scene:
//   nodes   : QNode8[]       96-byte quantized BVH8 nodes, index 0 = root (if the root is inner)
//   prims   : TriRecord[]    48-byte triangle records (v0,v1,v2 or v0,e1,e2 + ids), leaf-contiguous
//   blobs   : bytes          cBVH / GridSOA leaf blobs for subdivision geometry (16-byte aligned each)
// The reference keeps the same information behind 64-bit tagged pointers (kernels/bvh/bvh.h:150-396,
// AlignedNode :433-594, QuantizedNode :1150-1324, Triangle4v kernels/geometry/trianglev.h:24-162).
#pragma once
#include <stdint.h>
#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

namespace rtamd {

// ---- child / node references (32 bit) -------------------------------------------------------------
// bit 31      : leaf flag
// inner       : bits 0..30 = node index
// triangle leaf: bits 26..30 = triangle count (1..28, i.e. up to 7 blocks of 4 like bvh.h:140), bits 0..25 = first record
// subdiv leaf : bits 0..30 = blob index (one blob per leaf, like encodeTypedLeaf(ptr,1) bvh_builder_subdiv.cpp:728)
static const uint32_t REF_EMPTY = 0xFFFFFFFFu; // no child (reference: BVH::emptyNode, bvh.h:117-132)
static const uint32_t REF_LEAF = 0x80000000u;
static const uint32_t TRI_LEAF_MAX = 28;
static const uint32_t TRI_START_BITS = 26;

inline __host__ __device__ uint32_t make_tri_leaf(uint32_t first, uint32_t count)
{
  return REF_LEAF | (count << TRI_START_BITS) | first;
}

// ---- quantized BVH8 node, 96 bytes = 6 x dwordx4 ----------------------------------------------------
// Child i box, per axis a:  lo = fmaf(float(qlo[a][i]), scale[a], origin[a]),  hi likewise with qhi,
// scale[a] = as_float(uint32(exp[a]) << 23)  (a power of two, or 0.0 for a flat axis).
// The builder guarantees lo <= exact child lower and hi >= exact child upper under exactly this fp32
// formula, so the box test is conservative.  Empty children have child == REF_EMPTY and an inverted box.
struct alignas(16) QNode8
{
  float origin[3];
  uint8_t exp[3];
  uint8_t pad;
  uint32_t child[8];
  uint8_t q[6][8]; // lo_x, hi_x, lo_y, hi_y, lo_z, hi_z  (same plane order as AlignedNode, bvh.h:588-593)
};
static_assert(sizeof(QNode8) == 96, "QNode8 must be 96 bytes");

// ---- triangle record, 48 bytes = 3 x dwordx4 ----------------------------------------------------------
// Pluecker accel (robust):  a = v0, b = v1, c = v2            (TriangleMv, trianglev.h:156-161)
// Moeller  accel (default): a = v0, b = e1 = v0-v1, c = e2 = v2-v0   (TriangleM, triangle.h:52-53)
// Records of one leaf are contiguous; every group of 4 from the leaf start is one "block" and keeps the
// reference's 4-wide SIMD semantics (all 4 tested against the tfar at block entry, lowest lane wins ties).
struct alignas(16) TriRecord
{
  float ax, ay, az;
  uint32_t geomID;
  float bx, by, bz;
  uint32_t primID;
  float cx, cy, cz;
  uint32_t pad;
};
static_assert(sizeof(TriRecord) == 48, "TriRecord must be 48 bytes");

enum AccelKind : uint32_t
{
  ACCEL_NONE = 0,
  ACCEL_TRI_PLUECKER = 1, // tri_accel=bvh8.triangle4v, or RTC_SCENE_FLAG_ROBUST (scene.cpp:158-164,204)
  ACCEL_TRI_MOELLER = 2,  // default / tri_accel=bvh8.triangle4 / qbvh8.triangle4 (scene.cpp:130-211)
  ACCEL_CBVH_BOX = 3,     // subdiv_accel=bvh4.compressed.box
  ACCEL_CBVH_LEAF = 4,    // subdiv_accel=bvh4.compressed.leaf
  ACCEL_CBVH_GRID = 5,    // subdiv_accel=bvh4.compressed.grid
  ACCEL_GRIDSOA = 6       // eager subdiv (default subdiv accel)
};

// What a kernel launch needs to know about one committed scene.
struct AccelDesc
{
  const QNode8* nodes;
  const TriRecord* prims;
  const uint8_t* blobs;
  const uint32_t* blobOffsets; // blob index -> byte offset / 16
  uint32_t root;               // REF_EMPTY for an empty scene
  uint32_t kind;               // AccelKind
  uint32_t robust;             // 1: robust node test (TravRay<...,true>), 0: fast test
  uint32_t pad;
};

// Work counters of the instrumented kernels (mirrors RTCAMDTraceCounters).
struct TraceCounters
{
  unsigned long long rays, nodeVisits, leafVisits, primTests, innerVisits, hits, stackSpills, reserved;
};

} // namespace rtamd
