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
// Mirrors CompressedBVH's members (kernels/geometry/compressed.h:408-433) with offsets instead of host pointers, the 3x3 inverse
// of proj precomputed (the reference inverts at run time, compressed.h:585,647), and the leaf's world bounds kept for the any-hit
// stub.  Round 3 layout, ordered by WHEN a visit needs a field, in 128-byte lines (the blob stride is a multiple of 128 and the
// blob array is 128-byte aligned, so a line of a blob is a line of L2 / HBM):
//   line 0  (CbvhHeader, bytes 0..127)   everything up to and including the frustum test and the projected ray: space, box, proj, a
//           copy of the root node's word, rcp_edges, extent.  The visits that end at the frustum test (compressed_help.h:109-133; 28 % of the
//           metric's, nearly all of a grazing ray's) and the test stage of the two-stage visits touch this ONE line (round 2: the same fields lay in two or three lines of a 448-byte
//           record that started on a line boundary only every other blob).
//   line 1  (CbvhMid, bytes 128..159, then the nodes from byte 160)   ids + uv window (commit only) and the 4-byte node words
//           (compressed_node.h:261-295); C = 3: 32 + 84 bytes, one line.
//   then    (leaf mode) 4^C 2-byte height patches (compressed_leaf.h:21-47), or (grid mode) (2^C+1)^2 float3 vertices, 16-byte
//           aligned; C = 3 leaf mode: bytes 256..383, exactly line 2.
//   tail    (CbvhTail, 64 bytes, 16-byte aligned)   iproj (flat-frame commits only) and the world bounds (any-hit stub only).
// bomberman L6/C3 leaf mode: 512 B per blob (reference 396), 46 528 blobs = 23.8 MB.
struct alignas(16) CbvhHeader
{
  float space[9];       // rows of the 3x3 world->local matrix: l = (dot(row0,p), dot(row1,p), dot(row2,p))
  float box[10];        // frustum: z slab + four 2-D corner points (compressed.h:278-292)
  float proj[9];        // row-major homography
  uint32_t rootWord;    // copy of the first node word (coded modes), so that a visit rejected by the frustum test never leaves line 0
  float rcp_edges;
  float extent;
  uint32_t levels;      // C
};
struct alignas(16) CbvhMid
{
  uint32_t geomID, primID;
  float uv0x, uv0y, uv1x, uv1y; // uv[0], uv[1]
  uint32_t elems;       // (4^C-1)/3 inner nodes
  uint32_t grid_width;  // 2^C+1
};
struct alignas(16) CbvhTail
{
  float iproj[9];       // row-major inverse homography
  float wlo[3], whi[3]; // world-space bounds handed to the outer BVH (bounds_o)
  float pad;
};
static const uint32_t CBVH_HEADER_BYTES = 128, CBVH_MID_BYTES = 32, CBVH_TAIL_BYTES = 64, CBVH_NODES_OFFSET = CBVH_HEADER_BYTES + CBVH_MID_BYTES;
// bvh4.compressed.full: a quadtree node holds its four child boxes as floats (the reference's NodeStorage<flavor::ref,32,32,32>,
// compressed_node.h:371-389, 24 floats); here plane-major so that a quad reads one plane of its four children with one access:
// lx[4], ux[4], ly[4], uy[4], lz[4], uz[4]
static const uint32_t CBVH_FULL_NODE_BYTES = 96;
static_assert(sizeof(CbvhHeader) == CBVH_HEADER_BYTES && sizeof(CbvhMid) == CBVH_MID_BYTES && sizeof(CbvhTail) == CBVH_TAIL_BYTES, "cBVH blob sections");
// section offsets of a blob of compression level C; mode: 0 box, 1 leaf, 2 grid, 3 full (CbvhMode / MODE_* of the kernels)
constexpr uint32_t cbvh_elems(uint32_t C) { return ((1u << (2u * C)) - 1u) / 3u; }
constexpr uint32_t cbvh_payload_offset(uint32_t C, uint32_t mode) { return (CBVH_NODES_OFFSET + cbvh_elems(C) * (mode == 3u ? CBVH_FULL_NODE_BYTES : 4u) + 15u) & ~15u; } // cells / grid
constexpr uint32_t cbvh_tail_offset(uint32_t C, uint32_t mode)
{
  return (cbvh_payload_offset(C, mode) + (mode == 1u ? 2u << (2u * C) : (mode == 2u ? 12u * ((1u << C) + 1u) * ((1u << C) + 1u) : 0u)) + 15u) & ~15u;
}
constexpr uint32_t cbvh_stride(uint32_t C, uint32_t mode) { return (cbvh_tail_offset(C, mode) + CBVH_TAIL_BYTES + 127u) & ~127u; }
static_assert(cbvh_payload_offset(3, 1) == 256 && cbvh_tail_offset(3, 1) == 384 && cbvh_stride(3, 1) == 512, "C = 3 leaf mode: header | mid + nodes | cells | tail, one line each");

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
