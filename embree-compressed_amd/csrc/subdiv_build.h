// Subdivision-surface accel construction (tessellation -> cBVH / GridSOA leaf blobs -> BVH8 over the leaves).
// Reference: kernels/bvh/bvh_builder_subdiv.cpp:47-216 (eager), :684-884 (fork, oriented/compressed).
#pragma once
#include "rt_objects.h"

namespace rtamd {
void build_subdiv_accel(Scene* scene);
}
