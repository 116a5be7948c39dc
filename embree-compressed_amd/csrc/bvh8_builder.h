// Host builder: binned-SAH binary tree collapsed into 8-wide nodes, then quantized into QNode8 records.
//
// This is new code, not a restatement: the reference's builders (kernels/builders/*, bvh_builder_sah.cpp)
// are outside the hot path and tree shape does not change which hit is closest.
// What is kept from the reference are the leaf-size conventions of BVHNBuilderSAH<8,...,Triangle4v>
// (bvh_builder_sah.cpp:651-658: SAH block size 4, leaves of up to 7 blocks) and of the subdiv builders
// (bvh_builder_subdiv.cpp:845-851: exactly one primitive per leaf).
#pragma once
#include <functional>

#include "accel.h"
#include "rt_common.h"

namespace rtamd {

struct BuildPrim
{
  Box3 box;
  uint32_t id; // index into the caller's primitive list
};

struct BuildSettings
{
  uint32_t blockSize = 4;    // SAH cost counts ceil(n/blockSize) intersections
  uint32_t minLeaf = 4;      // never split at or below this size
  uint32_t maxLeaf = 28;     // always split above this size
  float travCost = 1.0f;
  float intCost = 1.0f;
  unsigned threads = 1;      // host threads for the upper levels of the binary tree (subtrees above 64 k primitives)
};

struct BuildResult
{
  std::vector<QNode8> nodes;
  uint32_t root = REF_EMPTY;
  uint32_t maxDepth = 0;
  size_t leafCount = 0;
};

// makeLeaf(prims, begin, end) -> leaf reference (REF_LEAF | payload); prims[begin,end) is the final
// leaf order (the builder permutes the array in place).
using MakeLeafFn = std::function<uint32_t(const BuildPrim* prims, size_t begin, size_t end)>;

BuildResult build_bvh8(std::vector<BuildPrim>& prims, const BuildSettings& settings, const MakeLeafFn& makeLeaf);

// Quantize the boxes of up to 8 children into one node; exported for the unit tests of the codec.
void quantize_node(const Box3* childBoxes, const uint32_t* childRefs, int n, QNode8& out);
// Decode child i of a node with exactly the arithmetic the kernels use.
Box3 dequantize_child(const QNode8& node, int i);

} // namespace rtamd
