// Host encoder of the fork's compressed per-sub-grid BVH ("cBVH").
//
// Restates, as new code, the build-time half of the fork:
//   CompressedBVH ctor            kernels/geometry/compressed.h:49-337   (frame, homography, hierarchy, frustum box, leaves)
//   Node<com,man,man2,3,3,2>      kernels/geometry/compressed_node.h:397-512, storage :261-295, tables :22-39,80-114
//   quantTris<4>                  kernels/geometry/compressed_leaf.h:21-113,117-251
//   ComputeLinearEstimate/project kernels/geometry/compressed_help.h:54-90   (Eigen's 8x8 fullPivLu -> own full-pivot solve)
//   Morton helpers                kernels/geometry/compressed_help.h:19-50
// The codec functions below are shared with the unit tests through rtcamd test hooks; the DEVICE decode in
// trace_subdiv.hip repeats cbvh_decode_child() operation for operation.
#pragma once
#include "accel.h"
#include "subdiv_tess.h"

namespace rtamd {

enum CbvhMode { CBVH_BOX = 0, CBVH_LEAF = 1, CBVH_GRID = 2, CBVH_FULL = 3 /* box semantics, float child boxes instead of 4-byte codes */ };

struct Box3f
{
  float lo[3], hi[3];
};

// 4-byte node: bytes xz,x,yz,y (compressed_node.h:261-295)
struct CbvhNode
{
  uint8_t xz, x, yz, y;
};

extern const float CBVH_TABLE_BORDER[8]; // table1: {0,.005,.01,.05,.1,.2,.4,.6}
extern const float CBVH_TABLE_MID[8];    // table2: {0,.4,.48,.49,.5,.51,.52,.6}
extern const float CBVH_TABLE_Z[4];      // table3: {0,.25,.5,.75}

void cbvh_encode_node(const Box3f& parent, const Box3f child[4], CbvhNode& out);    // Node::setAABB
Box3f cbvh_decode_child(const CbvhNode& n, const Box3f& parent, int loc);           // Node::getAABB
inline uint32_t cbvh_morton_x(uint32_t code);
inline uint32_t cbvh_morton_y(uint32_t code);

size_t cbvh_blob_bytes(unsigned C, CbvhMode mode);
void cbvh_debug_leaf_codec(const float box[6], const float v[12], float extent, uint8_t bytesOut[2], float* extentEstimate);

// Encode the sub-grid [x0,x1]x[y0,y1] (x1-x0 == y1-y0 == 2^C) of a tessellated patch into `blob`
// (cbvh_blob_bytes(C,mode) bytes, 16-byte aligned) and return the world bounds for the outer BVH.
void cbvh_encode(const PatchGrid& pg, unsigned x0, unsigned x1, unsigned y0, unsigned y1, unsigned C, CbvhMode mode, uint8_t* blob,
                 Box3& boundsOut);

// https://fgiesen.wordpress.com/2009/12/13/decoding-morton-codes/ (compressed_help.h:32-50)
inline uint32_t cbvh_compact1by1(uint32_t x)
{
  x &= 0x55555555u;
  x = (x ^ (x >> 1)) & 0x33333333u;
  x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
  x = (x ^ (x >> 4)) & 0x00ff00ffu;
  x = (x ^ (x >> 8)) & 0x0000ffffu;
  return x;
}
inline uint32_t cbvh_morton_x(uint32_t code) { return cbvh_compact1by1(code); }
inline uint32_t cbvh_morton_y(uint32_t code) { return cbvh_compact1by1(code >> 1); }

} // namespace rtamd
