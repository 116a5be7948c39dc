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

// ---- eager subdivision leaf: one 3x3-vertex cell (2x2 quads = 8 triangles), 160 bytes = 10 x dwordx4 ------
// Replaces the inner leaves of GridSOA (kernels/geometry/grid_soa.h:267-286, :85-90): the reference stores whole
// <=9x9 sub-grids in SoA form and a private BVH4 down to 3x3-vertex cells; here every cell is self-contained and the
// scene BVH8 goes straight down to cells.  p[r*3+c] is the vertex in row r (v direction), column c (u direction);
// uv[] holds the reference's packed patch coordinates (v16<<16 | u16, scale 8/65536, grid_soa.cpp:48-52).
struct alignas(16) GridCell
{
  float px[9], py[9], pz[9];
  uint32_t uv[9];
  uint32_t geomID, primID;
  uint32_t pad[2];
};
static_assert(sizeof(GridCell) == 160, "GridCell must be 160 bytes");

// ---- fork: compressed per-sub-grid BVH ("cBVH") blob -------------------------------------------------------------
// Header (CBVH_HEADER_BYTES) followed by elems 4-byte nodes, then (leaf mode) 4^C 2-byte height patches, then (grid
// mode) (2^C+1)^2 float3 vertices; blob stride is a multiple of 16.  Mirrors CompressedBVH's members
// (kernels/geometry/compressed.h:408-433) with offsets instead of host pointers, the 3x3 inverse of proj precomputed
// (the reference inverts at run time, compressed.h:585,647), and the leaf's world bounds kept for the any-hit stub.
struct alignas(16) CbvhHeader
{
  uint32_t geomID, primID;
  float uv0x, uv0y, uv1x, uv1y; // uv[0], uv[1]
  float rcp_edges;
  float extent;
  uint32_t elems;       // (4^C-1)/3 inner nodes
  uint32_t grid_width;  // 2^C+1
  uint32_t levels;      // C
  uint32_t pad0;
  float space[9];       // rows of the 3x3 world->local matrix: l = (dot(row0,p), dot(row1,p), dot(row2,p))
  float proj[9];        // row-major homography
  float iproj[9];       // row-major inverse homography
  float box[10];        // frustum: z slab + four 2-D corner points (compressed.h:278-292)
  float wlo[3], whi[3]; // world-space bounds handed to the outer BVH (bounds_o)
  float pad1;
};
static const uint32_t CBVH_HEADER_BYTES = 224;
// bvh4.compressed.full: a quadtree node holds its four child boxes as floats (the reference's NodeStorage<flavor::ref,32,32,32>,
// compressed_node.h:371-389, 24 floats); here plane-major so that a quad reads one plane of its four children with one access:
// lx[4], ux[4], ly[4], uy[4], lz[4], uz[4]
static const uint32_t CBVH_FULL_NODE_BYTES = 96;
static_assert(sizeof(CbvhHeader) == CBVH_HEADER_BYTES, "CbvhHeader must be 224 bytes");

enum AccelKind : uint32_t
{
  ACCEL_NONE = 0,
  ACCEL_TRI_PLUECKER = 1, // tri_accel=bvh8.triangle4v, or RTC_SCENE_FLAG_ROBUST (scene.cpp:158-164,204)
  ACCEL_TRI_MOELLER = 2,  // default / tri_accel=bvh8.triangle4 / qbvh8.triangle4 (scene.cpp:130-211)
  ACCEL_CBVH_BOX = 3,     // subdiv_accel=bvh4.compressed.box
  ACCEL_CBVH_LEAF = 4,    // subdiv_accel=bvh4.compressed.leaf
  ACCEL_CBVH_GRID = 5,    // subdiv_accel=bvh4.compressed.grid
  ACCEL_GRIDSOA = 6,      // eager subdiv (default subdiv accel)
  ACCEL_CBVH_FULL = 7     // subdiv_accel=bvh4.compressed.full: the fork's box mode over UNcompressed quadtree nodes (compressed.h:40,774)
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
  uint32_t blobStride;          // bytes per leaf blob (GridCell: 160; cBVH: header + nodes + leaves/grid)
};

// What one wavefront of an instrumented kernel reports (plain stores into its own slot: atomics on shared words
// serialise in L2 at ~70 ns each and slow the very batch they are meant to describe).
struct WaveRecord
{
  unsigned long long start, end;               // s_memrealtime ticks (100 MHz)
  unsigned long long iterations, leafPhases, laneIters;
  unsigned long long cyclesFetch, cyclesNode, cyclesLeaf, cyclesPop, cyclesTotal;
  unsigned long long rays, nodes, leaves, prims, inner, hits, spills;
  unsigned long long lastGrab, maxRaySteps;
  unsigned long long valid;
};
static const uint32_t WAVE_LOG_CAPACITY = 16384; // wave records per launch (two launches per batch: triangles, subdiv)

// Work counters of the instrumented kernels (mirrors RTCAMDTraceCounters).
struct TraceCounters
{
  unsigned long long rays, nodeVisits, leafVisits, primTests, innerVisits, hits, stackSpills, reserved;
  unsigned long long cyclesFetch, cyclesNode, cyclesLeaf, cyclesPop, cyclesTotal, iterations, leafPhases, waves;
  unsigned long long activeLaneIters, startInv;
  unsigned long long maxRaySteps, drainTicksSum, drainTicksMax;
  unsigned long long waveEndHist[64], waveIterHist[64];
};

} // namespace rtamd
