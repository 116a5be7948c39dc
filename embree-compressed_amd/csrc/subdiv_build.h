// Subdivision-surface accel construction (tessellation -> cBVH / GridSOA leaf blobs -> BVH8 over the leaves).
// Reference: kernels/bvh/bvh_builder_subdiv.cpp:47-216 (eager), :684-884 (fork, oriented/compressed).
#pragma once
#include "rt_objects.h"

#include <functional>

namespace rtamd {
void build_subdiv_accel(Scene* scene);
// commit-time host parallelism ("threads=N" device config, default = hardware threads, at most 64)
unsigned host_threads(const Device* dev);
void parallel_for_range(size_t count, unsigned threads, const std::function<void(size_t, size_t)>& body);
}
