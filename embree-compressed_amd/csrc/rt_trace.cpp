// Batch front-end of the hot path: host/device pointer handling, staging, kernel launches.
// Replaces RayStreamFilter::filterAOS (kernels/bvh/bvh_intersector_stream_filters.cpp:24-165) and the
// per-ray dispatch through Accel::Intersectors (kernels/common/accel.h:264-267).
#include "rt_trace.h"
#include <chrono>

namespace rtamd {

uint32_t trace_grid_blocks(uint32_t count, int numCUs, uint32_t rayChunk)
{
  // upper bound of the resident set: 20 waves per CU; a wave takes `rayChunk` rays per queue grab
  const uint32_t resident = (uint32_t)numCUs * 5u * (256u / TRACE_BLOCK);
  const uint32_t perBlock = std::max(1u, rayChunk) * (TRACE_BLOCK / 64u);
  const uint32_t need = (count + perBlock - 1) / perBlock;
  return need < resident ? (need ? need : 1u) : resident;
}

// Rays a wave takes from a work queue per grab.  Large batches: Device::tuneChunk (256: one atomic per 256 rays keeps the queue heads
// cold).  Batches that cannot give every resident wavefront (16 per CU) such a share are cut finer, down to 32 rays per wave, so that a
// mid-size batch - a 16 k .. 128 k chunk of the host pipeline, a test batch, a combined group of small calls - spreads over the whole chip
// instead of count / 256 wavefronts (round 3 finding, profiles/r03_deep_subset_probe.txt: 1000 rays ran on FOUR wavefronts).
// With other batches in flight the coarse share wins again from ~100 k rays on (every wavefront pays its deepest ray's iterations, the chip
// is issue bound: fewer, fuller wavefronts per batch), so launch_on switches back to tuneChunk there.  Measured round 3, cbvh.leaf, kernel
// alone / four batches in flight, fine vs 256-ray chunks (profiles/r03_chunk_ab.txt): 4 k rays 43 vs 105 us / 222 vs 105 Mrays/s, 16 k 46 vs
// 89 us / 779 vs 525, 64 k 61 vs 98 us / 1.90 vs 1.84 Grays/s, 128 k 73 vs 98 us / 2.97 vs 3.52, 250 k 104 vs 107 us / 4.8 vs 5.7, 500 k 0 % / -7 %.
static uint32_t ray_chunk_for(const Device* dev, uint32_t M, int numCUs)
{
  if (dev->tuneChunkFixed) return dev->tuneChunk;
  // the share that gives every resident wavefront exactly ONE grab: a queue owns ceil(M / TRACE_QUEUES) rays and serves waves / TRACE_QUEUES wavefronts.
  // (1 M rays on 4096 wavefronts: 245, not 256 - 64 grabs per queue instead of 61 full ones and a rest of 9 rays; a wave that needs a second grab late
  // doubles its life: 240 rays per grab cost +5 %, 245 gain 1.2 % over 256, profiles/r03_chunk_busy_ab.txt)
  const uint32_t waves = (uint32_t)numCUs * 16u;
  const uint32_t perQ = (M + (uint32_t)TRACE_QUEUES - 1u) / (uint32_t)TRACE_QUEUES;
  const uint32_t grabs = std::max(1u, waves / (uint32_t)TRACE_QUEUES);
  const uint32_t share = (perQ + grabs - 1u) / grabs;
  return std::min(dev->tuneChunk, std::max(32u, share));
}

// -1: plain host memory; otherwise the HIP ordinal the allocation lives on
static int pointer_device(const void* p)
{
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // plain (unregistered) host memory
    return -1;
  }
  return (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) ? attr.device : -1;
}
static bool is_device_pointer(const void* p) { return pointer_device(p) >= 0; }

// One traversal launch of `A` over M records at dRays (memory of shard `si`'s GPU) on that shard's stream.
// The calling thread's current HIP device must be the shard's (GpuShard::use()).
// cullCountsOut (counted batches): host buffer of TRACE_QUEUES * TRACE_QUEUE_STRIDE words that receives the launch's queue words
// after the kernels (word 1 of a queue = rays that survived the root cull pre-pass, word 2 = valid rays the pre-pass tested)
static void launch_on(Scene* s, const Accel& A, size_t si, void* dRays, uint32_t M, uint32_t stride, bool occluded, uint32_t instID,
                      WaveRecord* dCounters, const uint32_t* exclOffsets = nullptr, const uint2* exclPairs = nullptr, uint32_t* cullCountsOut = nullptr,
                      bool coherent = false, const uint32_t* exclT = nullptr, hipStream_t onStream = nullptr)
{
  Device* dev = s->device;
  Device::GpuShard& sh = *dev->shards[si];
  if (A.kind == ACCEL_NONE || A.root == REF_EMPTY) return;
  LaunchParams p;
  p.accel = A.desc(si);
  p.rays = dRays;
  p.count = M;
  p.stride = stride;
  p.instID = instID;
  p.occluded = occluded ? 1u : 0u;
  p.rayChunk = ray_chunk_for(dev, M, sh.numCUs);
  p.gridBlocks = trace_grid_blocks(M, sh.numCUs, p.rayChunk);
  // Ray-pool skeleton for very large batches - where it still pays.  Since the two-stage blob visits and the batched leaf passes of round 3 the lane
  // kernel is the faster one at EVERY size for grid cells and the cBVH box / leaf / full modes (4 M rays, one stream: cbvh.leaf 12.0 vs 10.2 Grays/s,
  // eager 6 M 14.5 vs 10.7); the pool keeps triangles (6 M: 16.8 vs 14.8) and the cBVH grid mode (11.0 vs 8.3).  profiles/r03_lane_pool_ab.txt
  const bool poolPays = A.kind == ACCEL_TRI_PLUECKER || A.kind == ACCEL_TRI_MOELLER || A.kind == ACCEL_CBVH_GRID;
  p.poolKernel = dev->tunePoolKernel == 2u ? (poolPays && M >= dev->tunePoolMinRays ? 1u : 0u) : dev->tunePoolKernel;
  // worst-case stack: 7 siblings per level plus the entry being expanded.  The overflow area is sized for it, so a push
  // can only be dropped if the tree is deeper than the builder reported; the kernels then raise `overflow` (below).
  const uint32_t worst = 7u * (A.maxDepth + 1u) + 2u;
  size_t spillBytes;
  if (p.poolKernel) { // one overflow column per ray slot of every resident wavefront
    p.gridBlocks = (uint32_t)sh.numCUs * TRACE_POOL_BLOCKS_PER_CU;
    p.spillDepth = worst > (uint32_t)TRACE_POOL_STACK ? worst - TRACE_POOL_STACK : 0u;
    spillBytes = (size_t)p.gridBlocks * (TRACE_POOL_BLOCK / 64) * TRACE_POOL_SLOTS * (size_t)p.spillDepth * 8u + 16u;
  } else {
    p.spillDepth = worst > (uint32_t)TRACE_LDS_STACK ? worst - TRACE_LDS_STACK : 0u;
    spillBytes = (size_t)p.gridBlocks * TRACE_BLOCK * (size_t)p.spillDepth * 8u + 16u;
  }
  p.counters = dCounters;
  p.cbvhLevels = s->compressionLevel;
  // cBVH blob walk: quad form (four lanes per ray, two-stage visits) or one ray per lane.  Coherent batches (RTC_INTERSECT_CONTEXT_FLAG_COHERENT,
  // e.g. the primary rays of viewer_stream_device.cpp:305) keep most lanes at blobs at once; since the two-stage visits the quad form is the faster
  // one for them too when the batch has the chip (1920x1080 camera rays alone: 0.630 vs 0.768 ms), the lane form still wins with several batches
  // in flight (6.8 vs 6.2 Grays/s) - decided below, once the launch context says how many others are running.  RTAMD_CBVH_FORM=quad|lane overrides.
  p.cbvhLaneForm = dev->tuneCbvhForm == 2u ? 0u : dev->tuneCbvhForm;
  p.numCUs = (uint32_t)sh.numCUs;
  p.leafBatch = dev->tuneLeafBatch;
  p.refillBatch = dev->tuneRefillBatch;
  p.octMax = dev->tuneOctMax;
  p.walkBatch = dev->tuneWalkBatch;
  p.inlineRay = 0u;
  p.octSteps = dev->tuneOctSteps;
  // waiting rays from which the child-parallel leaf phase runs: triangle leaves 16, grid cells 24 (measured optima), cBVH blobs
  // (quad form, 16 rays per pass) 16
  p.octLeaf = dev->tuneOctLeaf != 0xFFFFFFFFu ? dev->tuneOctLeaf : (A.kind == ACCEL_GRIDSOA ? 24u : 16u);
  p.exclOffsets = exclOffsets;
  p.exclPairs = exclPairs;
  p.exclT = exclT;
  p.overflow = sh.overflowDev;
  static const bool timeline = getenv("RTAMD_TIMELINE") != nullptr; // development aid, see trace.h
  p.timeline = (timeline && !dCounters) ? (unsigned long long*)sh.countersDev : nullptr;
  // {context, queue heads, launch, event} as one unit: concurrent callers on device-resident batches must not pick the same
  // context (its event still reads "finished" until the new launch has recorded it).  The stream is read once.
  std::lock_guard<std::mutex> seq(sh.seqMutex);
  const hipStream_t stream = onStream ? onStream : sh.stream;
  unsigned busyOther = 0;
  Device::LaunchCtx& ctx = sh.acquireLaunchCtx(spillBytes, &busyOther, stream);
  p.spill = ctx.spill;
  if (dev->tuneCbvhForm == 2u && coherent && busyOther >= 2u) p.cbvhLaneForm = 1u;
  // A batch alone on the chip is fastest with two workgroups per CU; when two or more batches are running on other
  // streams a leaner grid is better: every wave pays its deepest ray's iterations, so fewer waves per batch waste fewer
  // instructions (measured: 11.2 -> 12.0 Grays/s with four batches in flight; alone 0.174 -> 0.237 ms, hence adaptive).
  // (The grid-cell kernel, whose leaves are always tested 8 lanes per ray, needs 118 VGPRs: four waves per SIMD fit, and a batch
  // alone on the chip is 10 % faster with four workgroups per CU; 0.169 -> 0.151 ms.)
  const bool octOnly = A.kind == ACCEL_GRIDSOA || ((A.kind == ACCEL_CBVH_BOX || A.kind == ACCEL_CBVH_LEAF || A.kind == ACCEL_CBVH_FULL) && !p.cbvhLaneForm); // four waves per SIMD
  const uint32_t aloneBlocks = octOnly ? dev->tuneAloneBlocksOct : 2u;
  // with that fourth wave slot two workgroups per CU per batch are also the better grid in flight (eager, 40 steps: random rays
  // 13.3 -> 13.8 Grays/s, shadow rays 9.3 -> 9.8, camera rays 7.1 -> 7.7)
  const uint32_t busyBlocks = octOnly ? dev->tuneBusyBlocksOct : 1u;
  p.blocksPerCU = (dev->tuneBlocksAuto ? (busyOther >= 2u ? busyBlocks : (busyOther == 1u ? 2u : aloneBlocks)) : dev->tuneBlocksPerCU) * (256u / TRACE_BLOCK); // knob unit: 4 waves
  // With two or more other batches running, fewer and fuller wavefronts still (up to tuneChunkBusy = 512 rays per grab) as long as the batch
  // keeps half of the chip's wavefront slots busy: 1 M rays 12.8 -> 13.05 Grays/s, eager 14.2 -> 14.7, triangles 18.3 -> 18.9; a flat 512
  // loses 16-19 % at 131 k - 250 k rays (profiles/r03_chunk_busy_ab.txt)
  uint32_t busyChunk = dev->tuneChunk;
  if (busyOther >= 2u) busyChunk = std::min(std::max(dev->tuneChunk, (M / ((uint32_t)sh.numCUs * 8u)) & ~63u), std::max(dev->tuneChunk, dev->tuneChunkBusy));
  if (busyOther >= 1u && M >= 100000u && !dev->tuneChunkFixed && p.rayChunk < busyChunk && !p.poolKernel) {
    p.rayChunk = busyChunk; // in flight: coarse shares (see ray_chunk_for); the grid only shrinks, the overflow area was sized for the larger one
    p.gridBlocks = trace_grid_blocks(M, sh.numCUs, p.rayChunk);
  }
  p.queues = (uint32_t*)ctx.queues;
  // Root cull pre-pass (trace_cull.hip.h): large batches on the lane kernel whose root is an inner node.  Filter re-traces
  // (exclusion lists) are small and skip it.
  p.survivors = nullptr;
  if (dev->tuneCull && !p.poolKernel && !exclOffsets && M >= dev->tuneCullMinRays && !(A.root & REF_LEAF)) {
    const size_t need = ((size_t)(M + TRACE_QUEUES - 1) / TRACE_QUEUES) * TRACE_QUEUES * 4u;
    if (need > ctx.survivorsBytes) { // first batch of this size on this context (an allocation synchronises the device)
      HIP_CHECK(hipStreamSynchronize(stream));
      if (ctx.survivors) HIP_CHECK(hipFree(ctx.survivors));
      ctx.survivors = nullptr;
      ctx.survivorsBytes = 0;
      HIP_CHECK(hipMalloc(&ctx.survivors, need + need / 4));
      ctx.survivorsBytes = need + need / 4;
    }
    p.survivors = (uint32_t*)ctx.survivors;
  }
  HIP_CHECK(hipMemsetAsync(ctx.queues, 0, TRACE_QUEUES * TRACE_QUEUE_STRIDE * 4, stream)); // heads, survivor counts, valid-ray counts
  if (p.timeline) HIP_CHECK(hipMemsetAsync(p.timeline, 0, (size_t)WAVE_LOG_CAPACITY * 64, stream));
  if (p.survivors) HIP_CHECK(launch_cull(p, stream));
  HIP_CHECK(launch_trace(p, stream));
  if (cullCountsOut && p.survivors) HIP_CHECK(hipMemcpyAsync(cullCountsOut, ctx.queues, TRACE_QUEUES * TRACE_QUEUE_STRIDE * 4, hipMemcpyDeviceToHost, stream));
  else if (cullCountsOut) memset(cullCountsOut, 0, TRACE_QUEUES * TRACE_QUEUE_STRIDE * 4);
  HIP_CHECK(hipEventRecord(ctx.done, stream));
  dev->statLaunches++;
}

// ---- filter callbacks (row f3) -----------------------------------------------------------------------------------------
// Filter functions are host function pointers (intersector_epilog.h:251-291, filter.h:27-130): the device cannot call
// them.  Two-phase scheme: the kernel finds the closest candidate of every ray; the host runs the geometry's filter and
// then the context filter on it with the reference's argument protocol (ray.tfar = candidate distance, N = 1); an
// accepted candidate is the ray's result; a rejected one is put on the ray's exclusion list and the ray is traced again
// (only those rays, compacted), the kernel skipping listed candidates, until every ray has an accepted hit or none.
// For pure accept/reject filters this is the reference's result: rejected candidates never shorten the ray there either,
// so the closest accepted candidate wins.  The callbacks see the candidates of a ray in order of distance instead of
// traversal order, each at most once.  Occlusion filters run the same loop on closest candidates (any accepted candidate
// = occluded).
// Subdivision geometry (round 2):
//  * eager grid cells (GridSOAIntersector1 -> Intersect1EpilogMU / Occluded1EpilogMU, grid_soa_intersector1.h:61,83,
//    intersector_epilog.h:460-600: every triangle of a patch is offered to the filter on its own, with the PATCH's geomID / primID):
//    a candidate is identified by (geomID, primID, bits of t) - kernels are deterministic, the same triangle gives the same t
//    when the ray is traced again.  Two triangles of one patch hit at a bit-identical distance (a ray through their shared edge)
//    are rejected together, where the reference would offer both.
//  * the fork's compressed modes never call a filter: CompressedBVHIntersector1::intersect writes the hit itself and occluded()
//    is a stub (compressed.h:454-756, no runIntersectionFilter1 anywhere in compressed*.h).  Hits on such an accel are accepted
//    without a callback, geometry and context filter alike; for any-hit queries the stub pass runs first, unfiltered.
static const unsigned FILTER_MAX_ROUNDS = 256;

static void trace_filtered(Scene* s, void* rays, uint32_t M, size_t byteStride, bool occluded, const RTCIntersectContext* ctx)
{
  Device* dev = s->device;
  RTCIntersectContext localCtx;
  if (!ctx) { memset(&localCtx, 0, sizeof(localCtx)); localCtx.instID[0] = RTC_INVALID_GEOMETRY_ID; ctx = &localCtx; }
  const uint32_t instID = ctx->instID[0];
  const uint32_t recIn = occluded ? (uint32_t)sizeof(RTCRay) : (uint32_t)sizeof(RTCRayHit);
  const bool haveSubdiv = s->subdivAccel.kind != ACCEL_NONE && s->subdivAccel.root != REF_EMPTY;
  const bool forkAccel = haveSubdiv && s->subdivAccel.kind != ACCEL_GRIDSOA; // no filter calls on these (see above)
  std::lock_guard<std::mutex> lock(dev->launchMutex);
  // the host filter loop runs on the first shard (its rounds are latency bound, not throughput bound)
  Device::GpuShard& sh = dev->primary();
  sh.use();

  // the caller's records, on the host
  const bool devPtr = is_device_pointer(rays);
  const size_t span = (size_t)(M - 1) * byteStride + recIn;
  std::vector<char> mirror;
  char* src = (char*)rays;
  if (devPtr) {
    mirror.resize(span);
    HIP_CHECK(hipMemcpyAsync(mirror.data(), rays, span, hipMemcpyDeviceToHost, sh.stream));
    HIP_CHECK(hipStreamSynchronize(sh.stream));
    src = mirror.data();
  }
  std::vector<RTCRayHit> W(M);
  std::vector<uint32_t> act;
  act.reserve(M);
  for (uint32_t i = 0; i < M; i++) {
    memcpy(&W[i].ray, src + (size_t)i * byteStride, sizeof(RTCRay));
    if (occluded) {
      memset(&W[i].hit, 0, sizeof(RTCHit));
      W[i].hit.geomID = W[i].hit.primID = W[i].hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
    } else
      memcpy(&W[i].hit, src + (size_t)i * byteStride + sizeof(RTCRay), sizeof(RTCHit));
    if (W[i].ray.tnear <= W[i].ray.tfar && !(occluded && W[i].ray.tfar < 0.0f)) act.push_back(i);
  }
  if (occluded && forkAccel && !act.empty()) {
    // the stub any-hit pass of the fork's accel, unfiltered; the filter loop below then only sees the triangle accel
    const uint32_t K = (uint32_t)act.size();
    const size_t bytes = (size_t)K * sizeof(RTCRay);
    sh.ensureStaging(bytes);
    RTCRay* h = (RTCRay*)sh.stageHost;
    for (uint32_t k = 0; k < K; k++) h[k] = W[act[k]].ray;
    HIP_CHECK(hipMemcpyAsync(sh.stageDev, h, bytes, hipMemcpyHostToDevice, sh.stream));
    launch_on(s, s->subdivAccel, 0, sh.stageDev, K, (uint32_t)sizeof(RTCRay), true, instID, nullptr);
    HIP_CHECK(hipMemcpyAsync(h, sh.stageDev, bytes, hipMemcpyDeviceToHost, sh.stream));
    HIP_CHECK(hipStreamSynchronize(sh.stream));
    std::vector<uint32_t> rest;
    for (uint32_t k = 0; k < K; k++) {
      if (h[k].tfar < 0.0f) W[act[k]].ray.tfar = -std::numeric_limits<float>::infinity();
      else rest.push_back(act[k]);
    }
    act.swap(rest);
  }
  struct Rejected { uint32_t geomID, primID, tbits; };
  std::vector<std::vector<Rejected>> exclTri(M), exclSub(M);
  std::vector<uint32_t> offT, offS, tS, next;
  std::vector<uint2> pairsT, pairsS;
  void* dExcl = nullptr;
  size_t dExclBytes = 0;
  auto freeExcl = [&]() { if (dExcl) hipFree(dExcl); dExcl = nullptr; };
  auto a16 = [](size_t n) { return (n + 15) & ~(size_t)15; };
  try {
    for (unsigned round = 0; !act.empty() && round < FILTER_MAX_ROUNDS; round++) {
      const uint32_t K = (uint32_t)act.size();
      const size_t bytes = (size_t)K * sizeof(RTCRayHit);
      sh.ensureStaging(bytes);
      RTCRayHit* h = (RTCRayHit*)sh.stageHost;
      offT.assign(K + 1, 0);
      offS.assign(K + 1, 0);
      pairsT.clear(); pairsS.clear(); tS.clear();
      for (uint32_t k = 0; k < K; k++) {
        h[k] = W[act[k]];
        offT[k] = (uint32_t)pairsT.size();
        offS[k] = (uint32_t)pairsS.size();
        for (const Rejected& e : exclTri[act[k]]) pairsT.push_back(make_uint2(e.geomID, e.primID));
        for (const Rejected& e : exclSub[act[k]]) { pairsS.push_back(make_uint2(e.geomID, e.primID)); tS.push_back(e.tbits); }
      }
      offT[K] = (uint32_t)pairsT.size();
      offS[K] = (uint32_t)pairsS.size();
      const uint32_t *dOffT = nullptr, *dOffS = nullptr, *dTS = nullptr;
      const uint2 *dPairsT = nullptr, *dPairsS = nullptr;
      if (!pairsT.empty() || !pairsS.empty()) {
        const size_t offBytes = a16((size_t)(K + 1) * 4);
        const size_t oPT = 2 * offBytes, oPS = oPT + a16(pairsT.size() * sizeof(uint2)), oTS = oPS + a16(pairsS.size() * sizeof(uint2));
        const size_t need = oTS + a16(tS.size() * 4);
        if (need > dExclBytes) {
          HIP_CHECK(hipStreamSynchronize(sh.stream));
          freeExcl();
          dExclBytes = need * 2;
          HIP_CHECK(hipMalloc(&dExcl, dExclBytes));
        }
        char* D = (char*)dExcl;
        if (!pairsT.empty()) {
          HIP_CHECK(hipMemcpyAsync(D, offT.data(), (size_t)(K + 1) * 4, hipMemcpyHostToDevice, sh.stream));
          HIP_CHECK(hipMemcpyAsync(D + oPT, pairsT.data(), pairsT.size() * sizeof(uint2), hipMemcpyHostToDevice, sh.stream));
          dOffT = (const uint32_t*)D;
          dPairsT = (const uint2*)(D + oPT);
        }
        if (!pairsS.empty()) {
          HIP_CHECK(hipMemcpyAsync(D + offBytes, offS.data(), (size_t)(K + 1) * 4, hipMemcpyHostToDevice, sh.stream));
          HIP_CHECK(hipMemcpyAsync(D + oPS, pairsS.data(), pairsS.size() * sizeof(uint2), hipMemcpyHostToDevice, sh.stream));
          HIP_CHECK(hipMemcpyAsync(D + oTS, tS.data(), tS.size() * 4, hipMemcpyHostToDevice, sh.stream));
          dOffS = (const uint32_t*)(D + offBytes);
          dPairsS = (const uint2*)(D + oPS);
          dTS = (const uint32_t*)(D + oTS);
        }
      }
      HIP_CHECK(hipMemcpyAsync(sh.stageDev, h, bytes, hipMemcpyHostToDevice, sh.stream));
      launch_on(s, s->triAccel, 0, sh.stageDev, K, (uint32_t)sizeof(RTCRayHit), false, instID, nullptr, dOffT, dPairsT);
      if (!(occluded && forkAccel))
        launch_on(s, s->subdivAccel, 0, sh.stageDev, K, (uint32_t)sizeof(RTCRayHit), false, instID, nullptr, dOffS, dPairsS, nullptr, false, dTS);
      HIP_CHECK(hipMemcpyAsync(h, sh.stageDev, bytes, hipMemcpyDeviceToHost, sh.stream));
      HIP_CHECK(hipStreamSynchronize(sh.stream));
      next.clear();
      for (uint32_t k = 0; k < K; k++) {
        const uint32_t i = act[k];
        const RTCRayHit& got = h[k];
        const bool found = got.hit.geomID != RTC_INVALID_GEOMETRY_ID &&
                           (got.ray.tfar != W[i].ray.tfar || got.hit.primID != W[i].hit.primID || got.hit.geomID != W[i].hit.geomID);
        if (!found) continue; // miss: the caller's record stays as it is
        // the hit reports instID in geomID when instanced; instancing is not on this path, so geomID is the geometry
        Geometry* geo = got.hit.geomID < s->geometries.size() ? s->geometries[got.hit.geomID] : nullptr;
        const bool onSubdiv = geo && geo->type == RTC_GEOMETRY_TYPE_SUBDIVISION;
        const bool unfiltered = onSubdiv && forkAccel;
        RTCFilterFunctionN fn = geo && !unfiltered ? (occluded ? geo->occludedFilter : geo->intersectFilter) : nullptr;
        RTCFilterFunctionN cfn = unfiltered ? nullptr : ctx->filter;
        bool accepted = true;
        RTCRayHit cand = W[i];
        cand.ray.tfar = got.ray.tfar; // filter.h / intersector_epilog.h:277-279: the callback sees tfar = candidate distance
        RTCHit hit = got.hit;
        if (fn || cfn) {
          int mask = -1;
          RTCFilterFunctionNArguments a;
          a.valid = &mask;
          a.geometryUserPtr = geo ? geo->userPtr : nullptr;
          a.context = ctx;
          a.ray = (RTCRayN*)&cand.ray;
          a.hit = (RTCHitN*)&hit;
          a.N = 1;
          if (fn) fn(&a);
          if (mask != 0 && cfn) cfn(&a);
          accepted = mask != 0;
        }
        if (accepted) {
          if (occluded) W[i].ray.tfar = -std::numeric_limits<float>::infinity();
          else { W[i].ray = cand.ray; W[i].hit = hit; } // copyHitToRay
        } else {
          uint32_t tb;
          memcpy(&tb, &got.ray.tfar, 4);
          (onSubdiv ? exclSub : exclTri)[i].push_back(Rejected{got.hit.geomID, got.hit.primID, tb});
          next.push_back(i);
        }
      }
      act.swap(next);
    }
  } catch (...) {
    freeExcl();
    throw;
  }
  freeExcl();
  // outputs: tfar, and the hit for rtcIntersect
  for (uint32_t i = 0; i < M; i++) {
    char* dst = src + (size_t)i * byteStride;
    memcpy(dst + 32, &W[i].ray.tfar, 4);
    if (!occluded) memcpy(dst + sizeof(RTCRay), &W[i].hit, sizeof(RTCHit));
  }
  if (devPtr) {
    HIP_CHECK(hipMemcpyAsync(rays, mirror.data(), span, hipMemcpyHostToDevice, sh.stream));
    HIP_CHECK(hipStreamSynchronize(sh.stream));
  }
}

// ---- large host-pointer batches: chunked pipeline ---------------------------------------------------------------------------
// The callers of the drop-in API pass HOST records (viewer_stream_device.cpp:288-341).  One such batch used to be: memcpy into
// pinned memory, H2D, traversal, D2H, scatter - one after the other, the copies on one host thread (~0.1 Grays/s; the traversal is
// 1-2 % of that).  Here the range of every shard is cut into chunks of Device::tunePipeChunk rays; chunk k is gathered into pinned
// memory by the host pool while chunks k-1 and k-2 are on the GPU (upload, traversal, download on two alternating internal streams:
// PCIe is full duplex), and chunk k-2 is scattered back when its event has fired.  The results are those of the unpipelined path:
// a stream is M independent rays (tests/test_gpu_host_pipeline.py compares the two byte for byte).
static void trace_host_pipelined(Scene* s, char* rays, uint32_t M, size_t byteStride, bool occluded, uint32_t instID, bool coherent, uint32_t rec)
{
  Device* dev = s->device;
  const size_t G = M < 2u * dev->shards.size() ? 1 : dev->shards.size();
  // chunk size by batch size (tools/pcie_probe.py and a sweep of 16 k .. 200 k rays on MI355X: 32 k rays in 16 k chunks 0.29 ms against 0.41 ms
  // unpipelined, 64 k in 32 k chunks 0.41 / 0.70 ms, 128 k 0.69 / 1.20 ms, 200 k 0.99 / 1.79 ms, 1 M in 128 k chunks 2.84 / 9.4 ms)
  const uint32_t CH = dev->tunePipeChunk ? dev->tunePipeChunk : (M < 49152u ? 16384u : (M < 300000u ? 32768u : (M < 600000u ? 65536u : 131072u)));
  const unsigned LAG = 2;
  if (dev->hostPool.threads.empty()) {
    const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
    const unsigned want = dev->tuneHostThreads ? dev->tuneHostThreads : std::min(8u, hw / 2u);
    if (want > 1) dev->hostPool.start(want - 1); // the calling thread is the last worker
  }
  const size_t workers = dev->hostPool.threads.size() + 1;
  struct Lane { Device::GpuShard* sh; size_t g; uint32_t lo, n, chunks; };
  std::vector<Lane> lanes;
  uint32_t maxChunks = 0;
  for (size_t g = 0; g < G; g++) {
    Lane L;
    L.sh = dev->shards[g].get();
    L.g = g;
    L.lo = (uint32_t)((uint64_t)M * g / G);
    L.n = (uint32_t)((uint64_t)M * (g + 1) / G) - L.lo;
    if (L.n == 0) continue;
    L.chunks = (L.n + CH - 1) / CH;
    maxChunks = std::max(maxChunks, L.chunks);
    L.sh->use();
    L.sh->ensureStaging((size_t)L.n * rec);
    // the two internal streams exist from the first pipelined batch on: created with the device they took hardware queues away from
    // the caller's streams (measured: four device-resident batches in flight on four streams fell from 11.4 to 9.3 Grays/s, at most
    // three kernels overlapped)
    for (hipStream_t& ps : L.sh->pipeStream)
      if (!ps) HIP_CHECK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    while (L.sh->pipeEvents.size() < L.chunks) {
      hipEvent_t e;
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      L.sh->pipeEvents.push_back(e);
    }
    lanes.push_back(L);
  }
  // rays [a, b) of a lane, split over the workers
  auto for_parts = [&](uint32_t a, uint32_t b, const std::function<void(uint32_t, uint32_t)>& body) {
    const uint32_t n = b - a;
    const size_t parts = std::min<size_t>(workers, std::max<uint32_t>(1u, n / 4096u));
    dev->hostPool.run(parts, [&](size_t p) { body(a + (uint32_t)((uint64_t)n * p / parts), a + (uint32_t)((uint64_t)n * (p + 1) / parts)); });
  };
  struct Hot { Device::HostPool& p; Hot(Device::HostPool& q) : p(q) { p.begin(); } ~Hot() { p.end(); } } hot(dev->hostPool); // helpers poll for the duration of the call
  // RTAMD_PIPE_TRACE=1: where the calling thread's time goes (gather / enqueue / waiting for a chunk's event / scatter), per call
  static const bool pipeTrace = getenv("RTAMD_PIPE_TRACE") != nullptr;
  double tGather = 0, tEnq = 0, tWait = 0, tScatter = 0;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (uint32_t k = 0; k < maxChunks + LAG; k++) {
    for (Lane& L : lanes) {
      if (k >= L.chunks) continue;
      const double t0 = pipeTrace ? now() : 0.0;
      const uint32_t a = k * CH, b = std::min(L.n, a + CH);
      char* h = (char*)L.sh->stageHost;
      const char* src = rays + (size_t)L.lo * byteStride;
      for_parts(a, b, [&](uint32_t x, uint32_t y) {
        if (byteStride == rec) memcpy(h + (size_t)x * rec, src + (size_t)x * rec, (size_t)(y - x) * rec);
        else
          for (uint32_t i = x; i < y; i++) memcpy(h + (size_t)i * rec, src + (size_t)i * byteStride, rec);
      });
      const double t1 = pipeTrace ? now() : 0.0;
      L.sh->use();
      const hipStream_t st = L.sh->pipeStream[k & 1u];
      char* d = (char*)L.sh->stageDev + (size_t)a * rec;
      const size_t bytes = (size_t)(b - a) * rec;
      HIP_CHECK(hipMemcpyAsync(d, h + (size_t)a * rec, bytes, hipMemcpyHostToDevice, st));
      launch_on(s, s->triAccel, L.g, d, b - a, rec, occluded, instID, nullptr, nullptr, nullptr, nullptr, coherent, nullptr, st);
      launch_on(s, s->subdivAccel, L.g, d, b - a, rec, occluded, instID, nullptr, nullptr, nullptr, nullptr, coherent, nullptr, st);
      HIP_CHECK(hipMemcpyAsync(h + (size_t)a * rec, d, bytes, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipEventRecord(L.sh->pipeEvents[k], st));
      if (pipeTrace) { const double t2 = now(); tGather += t1 - t0; tEnq += t2 - t1; }
    }
    if (k < LAG) continue;
    const uint32_t j = k - LAG;
    for (Lane& L : lanes) {
      if (j >= L.chunks) continue;
      L.sh->use();
      const double t0 = pipeTrace ? now() : 0.0;
      HIP_CHECK(hipEventSynchronize(L.sh->pipeEvents[j]));
      const double t1 = pipeTrace ? now() : 0.0;
      const uint32_t a = j * CH, b = std::min(L.n, a + CH);
      const char* h = (const char*)L.sh->stageHost;
      char* dst0 = rays + (size_t)L.lo * byteStride;
      for_parts(a, b, [&](uint32_t x, uint32_t y) { // only tfar (byte 32) and the hit record (bytes 48..79) are outputs
        // A miss leaves a record untouched, and most incoherent rays miss: a record whose outputs came back unchanged is not
        // written (reading the caller's cache line is cheaper than dirtying it: 1 M random rays 1.5 -> ~0.9 ms of scatter)
        for (uint32_t i = x; i < y; i++) {
          char* dst = dst0 + (size_t)i * byteStride;
          const char* src = h + (size_t)i * rec;
          if (memcmp(dst + 32, src + 32, 4) != 0) memcpy(dst + 32, src + 32, 4);
          if (!occluded && memcmp(dst + 48, src + 48, 32) != 0) memcpy(dst + 48, src + 48, 32);
        }
      });
      if (pipeTrace) { const double t2 = now(); tWait += t1 - t0; tScatter += t2 - t1; }
    }
  }
  if (pipeTrace) fprintf(stderr, "embree3-amd: pipelined host batch of %u rays: gather %.2f ms, enqueue %.2f ms, waiting for events %.2f ms, scatter %.2f ms (%zu host threads, %u chunks)\n", M, tGather, tEnq, tWait, tScatter, workers, maxChunks);
  for (Lane& L : lanes) L.sh->checkOverflow();
}

void trace_batch(Scene* s, void* rays, uint32_t M, size_t byteStride, bool occluded, const RTCIntersectContext* ctx,
                 TraceCounters* countersOut)
{
  Device* dev = s->device;
  if (s->modified) RT_THROW(RTC_ERROR_INVALID_OPERATION, "scene got not committed"); // scene.cpp:25,54
  if (M == 0) return;
  dev->useDevice();
  if (byteStride > 0xFFFFFFFFull) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "byteStride too large");
  if (((uintptr_t)rays) & 3) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "ray not aligned to 4 bytes"); // rtcore.cpp:413
  // (geometry filters on subdivision meshes are called on the eager accel only: the fork's intersector never calls one, see trace_filtered)
  const bool subdivFilters = s->subdivFilter && s->subdivAccel.kind == ACCEL_GRIDSOA;
  if (!countersOut && ((ctx && ctx->filter) || subdivFilters || (occluded ? s->triOccludedFilter : s->triIntersectFilter))) {
    trace_filtered(s, rays, M, byteStride, occluded, ctx);
    return;
  }
  const uint32_t instID = ctx ? ctx->instID[0] : RTC_INVALID_GEOMETRY_ID;
  const bool coherent = ctx && (ctx->flags & RTC_INTERSECT_CONTEXT_FLAG_COHERENT);
  const uint32_t rec = occluded ? (uint32_t)sizeof(RTCRay) : (uint32_t)sizeof(RTCRayHit);

  // instrumented twin: every wavefront stores one WaveRecord; first half of the log = triangle launch, second = subdiv.
  // Counted batches run on ONE shard (the first, or the one the device pointer lives on) and one at a time.
  WaveRecord* dCounters = nullptr;
  WaveRecord* dCounters2 = nullptr;
  const size_t logBytes = 2 * (size_t)WAVE_LOG_CAPACITY * sizeof(WaveRecord);
  std::unique_lock<std::mutex> countLock(dev->launchMutex, std::defer_lock);
  size_t countShard = 0;
  std::vector<uint32_t> cullWords; // counted batches: queue words of the triangle launch, then of the subdivision launch
  uint32_t *cull1 = nullptr, *cull2 = nullptr;
  if (countersOut) {
    cullWords.assign(2 * (size_t)TRACE_QUEUES * TRACE_QUEUE_STRIDE, 0u);
    cull1 = cullWords.data();
    cull2 = cull1 + (size_t)TRACE_QUEUES * TRACE_QUEUE_STRIDE;
  }

  const int ptrDev = pointer_device(rays);
  if (ptrDev >= 0) {
    // device-resident stream: trace in place on the GPU the records live on, stream-ordered, no host synchronisation
    size_t si = dev->shards.size();
    for (size_t i = 0; i < dev->shards.size() && si == dev->shards.size(); i++)
      if (dev->shards[i]->ordinal == ptrDev) si = i;
    if (si == dev->shards.size()) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "the ray buffer lives on a GPU this RTCDevice does not use (gpu= / gpus=)");
    Device::GpuShard& sh = *dev->shards[si];
    sh.use();
    sh.checkOverflow(); // report of an earlier asynchronous batch
    if (countersOut) {
      countLock.lock();
      countShard = si;
      dCounters = (WaveRecord*)sh.countersDev;
      dCounters2 = dCounters + WAVE_LOG_CAPACITY;
      HIP_CHECK(hipMemsetAsync(dCounters, 0, logBytes, sh.stream));
    }
    launch_on(s, s->triAccel, si, rays, M, (uint32_t)byteStride, occluded, instID, dCounters, nullptr, nullptr, cull1, coherent);
    launch_on(s, s->subdivAccel, si, rays, M, (uint32_t)byteStride, occluded, instID, dCounters2, nullptr, nullptr, cull2, coherent);
  } else {
    // Host records: staged through pinned memory.  With several shards the M rays are split into contiguous ranges
    // [g*M/G, (g+1)*M/G), one per shard: H2D, traversal and D2H of the ranges run concurrently on the shards' own streams
    // and write disjoint slices of the caller's buffer (no exchange step, no collective).
    if (!countLock.owns_lock()) countLock.lock();
    if (!countersOut && M >= dev->tunePipeMinRays) {
      trace_host_pipelined(s, (char*)rays, M, byteStride, occluded, instID, coherent, rec);
      return;
    }
    const size_t G = (countersOut || M < 2u * dev->shards.size()) ? 1 : dev->shards.size();
    std::vector<uint32_t> lo(G + 1);
    for (size_t g = 0; g <= G; g++) lo[g] = (uint32_t)((uint64_t)M * g / G);
    for (size_t g = 0; g < G; g++) {
      Device::GpuShard& sh = *dev->shards[g];
      const uint32_t n = lo[g + 1] - lo[g];
      if (n == 0) continue;
      sh.use();
      const size_t bytes = (size_t)n * rec;
      sh.ensureStaging(bytes);
      char* h = (char*)sh.stageHost;
      const char* src = (const char*)rays + (size_t)lo[g] * byteStride;
      if (byteStride == rec) memcpy(h, src, bytes);
      else
        for (uint32_t i = 0; i < n; i++) memcpy(h + (size_t)i * rec, src + (size_t)i * byteStride, rec);
      // Small batches (single rays and the combiner's groups, row f2): the kernels read and write the pinned staging buffer in place
      // over PCIe - two copies and their DMA latency less per call; a few KB of rays cost nothing over the bus.
      void* dRays = sh.stageDev;
      const bool zeroCopy = !countersOut && n <= dev->tuneZeroCopyMax;
      if (zeroCopy) HIP_CHECK(hipHostGetDevicePointer(&dRays, h, 0));
      else HIP_CHECK(hipMemcpyAsync(sh.stageDev, h, bytes, hipMemcpyHostToDevice, sh.stream));
      if (countersOut) {
        countShard = g;
        dCounters = (WaveRecord*)sh.countersDev;
        dCounters2 = dCounters + WAVE_LOG_CAPACITY;
        HIP_CHECK(hipMemsetAsync(dCounters, 0, logBytes, sh.stream));
      }
      launch_on(s, s->triAccel, g, dRays, n, rec, occluded, instID, dCounters, nullptr, nullptr, cull1, coherent);
      launch_on(s, s->subdivAccel, g, dRays, n, rec, occluded, instID, dCounters2, nullptr, nullptr, cull2, coherent);
      if (!zeroCopy) HIP_CHECK(hipMemcpyAsync(h, sh.stageDev, bytes, hipMemcpyDeviceToHost, sh.stream));
    }
    for (size_t g = 0; g < G; g++) {
      Device::GpuShard& sh = *dev->shards[g];
      const uint32_t n = lo[g + 1] - lo[g];
      if (n == 0) continue;
      sh.use();
      HIP_CHECK(hipStreamSynchronize(sh.stream));
      // only tfar (byte 32) and the hit record (bytes 48..79) are outputs
      const char* h = (const char*)sh.stageHost;
      for (uint32_t i = 0; i < n; i++) {
        char* dst = (char*)rays + (size_t)(lo[g] + i) * byteStride;
        const char* src = h + (size_t)i * rec;
        memcpy(dst + 32, src + 32, 4);
        if (!occluded) memcpy(dst + 48, src + 48, 32);
      }
    }
    for (size_t g = 0; g < G; g++) dev->shards[g]->checkOverflow();
  }

  if (countersOut) {
    std::vector<WaveRecord> log(2 * (size_t)WAVE_LOG_CAPACITY);
    Device::GpuShard& csh = *dev->shards[countShard];
    csh.use();
    HIP_CHECK(hipMemcpyAsync(log.data(), dCounters, logBytes, hipMemcpyDeviceToHost, csh.stream));
    HIP_CHECK(hipStreamSynchronize(csh.stream));
    csh.checkOverflow();
    TraceCounters& c = *countersOut;
    memset(&c, 0, sizeof(c));
    unsigned long long first = ~0ull;
    for (const WaveRecord& w : log)
      if (w.valid) first = std::min(first, w.start);
    c.startInv = ~first;
    for (const WaveRecord& w : log) {
      if (!w.valid) continue;
      c.rays += w.rays; c.nodeVisits += w.nodes; c.leafVisits += w.leaves; c.primTests += w.prims;
      c.innerVisits += w.inner; c.hits += w.hits; c.stackSpills += w.spills;
      c.cyclesFetch += w.cyclesFetch; c.cyclesNode += w.cyclesNode; c.cyclesLeaf += w.cyclesLeaf; c.cyclesPop += w.cyclesPop;
      c.cyclesTotal += w.cyclesTotal;
      c.iterations += w.iterations; c.leafPhases += w.leafPhases; c.waves += 1; c.activeLaneIters += w.laneIters;
      c.maxRaySteps = std::max(c.maxRaySteps, w.maxRaySteps);
      c.drainTicksSum += w.end - w.lastGrab;
      c.drainTicksMax = std::max(c.drainTicksMax, w.end - w.lastGrab);
      c.waveEndHist[std::min<unsigned long long>((w.end - first) / 400ull, 63ull)] += 1; // 4 us buckets of 10 ns ticks
      c.waveIterHist[std::min<unsigned long long>(w.iterations / 2ull, 63ull)] += 1;
    }
    // root cull pre-pass: it visited the root once for every valid ray; the traversal kernel saw (and counted) the survivors only
    for (int l = 0; l < 2; l++) {
      unsigned long long survivors = 0, valid = 0;
      for (int q = 0; q < TRACE_QUEUES; q++) {
        survivors += cullWords[((size_t)l * TRACE_QUEUES + q) * TRACE_QUEUE_STRIDE + 1];
        valid += cullWords[((size_t)l * TRACE_QUEUES + q) * TRACE_QUEUE_STRIDE + 2];
      }
      if (valid) {
        c.rays += valid - survivors;
        c.nodeVisits += valid;
        c.reserved += survivors; // rays that survived the root cull (reported as `cullSurvivors` by the Python binding)
      }
    }
  }
}

// ---- persistent consumer for small calls (row f2) -------------------------------------------------------------
// See trace_service.hip.h.  One resident kernel per RTCDevice, for the accel kind of the first scene that makes a small call; a caller owns
// one slot of the ring for the duration of its call (threads are dealt slots round-robin; two threads that share a slot take turns).
struct Device::Service
{
  static const uint32_t SLOTS = 64;          // wavefronts of the service kernel = calls in flight
  static const uint32_t SPILL_DEPTH = 512;   // HBM stack overflow entries per lane the service can offer (scenes that need more keep the combiner)
  uint32_t kind = 0, levels = 0;
  hipStream_t stream = nullptr;
  ServiceSlot* slotsHost = nullptr;
  ServiceSlot* slotsDev = nullptr;
  uint32_t* stopHost = nullptr;
  uint32_t* stopDev = nullptr;
  uint32_t* activityDev = nullptr;
  void* spillDev = nullptr;
  LaunchParams base;
  std::atomic<uint32_t> slotLock[SLOTS];
  uint32_t slotSeq[SLOTS];
  std::atomic<uint32_t> nextSlot{0};
  std::atomic<uint64_t> lastSubmitNs{0};
  std::atomic<uint64_t> starts{0};
  bool failed = false;
};

// every live service, so that a process that exits without releasing its RTCDevice still stops the resident kernels BEFORE the runtime tears
// down the host-mapped ring they poll (a kernel reading freed host memory faults the GPU)
static std::mutex g_serviceRegistryMutex;
static std::vector<Device::Service*> g_serviceRegistry;
static void service_stop_all_at_exit()
{
  std::lock_guard<std::mutex> g(g_serviceRegistryMutex);
  for (Device::Service* sv : g_serviceRegistry) {
    if (sv->stopHost) __atomic_store_n(sv->stopHost, 1u, __ATOMIC_RELEASE);
    if (sv->stream) (void)hipStreamSynchronize(sv->stream);
  }
  g_serviceRegistry.clear();
}
static void service_register(Device::Service* sv)
{
  std::lock_guard<std::mutex> g(g_serviceRegistryMutex);
  static bool hooked = false;
  if (!hooked) { atexit(service_stop_all_at_exit); hooked = true; }
  g_serviceRegistry.push_back(sv);
}
static void service_unregister(Device::Service* sv)
{
  std::lock_guard<std::mutex> g(g_serviceRegistryMutex);
  g_serviceRegistry.erase(std::remove(g_serviceRegistry.begin(), g_serviceRegistry.end(), sv), g_serviceRegistry.end());
}

static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void service_destroy(Device* dev)
{
  Device::Service* sv = dev->service;
  if (!sv) return;
  dev->service = nullptr;
  service_unregister(sv);
  if (dev->gpu >= 0) (void)hipSetDevice(dev->gpu);
  if (sv->stopHost) __atomic_store_n(sv->stopHost, 1u, __ATOMIC_RELEASE);
  if (sv->stream) { (void)hipStreamSynchronize(sv->stream); (void)hipStreamDestroy(sv->stream); }
  if (sv->slotsHost) (void)hipHostFree(sv->slotsHost);
  if (sv->stopHost) (void)hipHostFree(sv->stopHost);
  if (sv->activityDev) (void)hipFree(sv->activityDev);
  if (sv->spillDev) (void)hipFree(sv->spillDev);
  delete sv;
}

// Stop the service kernel and wait for it (it restarts on demand): before the library allocates or frees device memory, which synchronises
// the whole device and would otherwise wait for the resident kernel's idle exit.  Jobs in flight are finished first; a job that arrives in
// between is served after the restart its caller triggers.
void service_quiesce(Device* dev)
{
  if (!dev->service) return;
  std::lock_guard<std::mutex> g(dev->serviceMutex);
  Device::Service* sv = dev->service;
  if (!sv || sv->failed || !sv->stream) return;
  if (dev->gpu >= 0) (void)hipSetDevice(dev->gpu);
  __atomic_store_n(sv->stopHost, 1u, __ATOMIC_RELEASE);
  (void)hipStreamSynchronize(sv->stream);
  __atomic_store_n(sv->stopHost, 0u, __ATOMIC_RELEASE);
}

// (re)start the service kernel if it is not running; serviceMutex held
static void service_start_locked(Device* dev, Device::Service* sv)
{
  const hipError_t q = hipStreamQuery(sv->stream);
  if (q == hipErrorNotReady) { (void)hipGetLastError(); return; } // running
  if (q != hipSuccess) HIP_CHECK(q);
  ServiceParams sp;
  sp.base = sv->base;
  sp.slots = sv->slotsDev;
  sp.numSlots = Device::Service::SLOTS;
  sp.idlePolls = 16384u; // ~30 ms without a single job: the kernel leaves the GPU to itself
  sp.stop = sv->stopDev;
  sp.activity = sv->activityDev;
  HIP_CHECK(launch_service(sp, sv->stream));
  sv->starts++;
}

// Trace a call of up to 64 rays through the service.  false: not applicable (the caller falls back to the combiner).
static bool service_trace(Scene* s, char* rays, uint32_t M, size_t byteStride, bool occluded, uint32_t instID)
{
  Device* dev = s->device;
  if (!dev->tuneService || dev->gpu < 0 || dev->shards.size() != 1 || M > (uint32_t)SERVICE_SLOT_RAYS) return false;
  const bool tri = s->triAccel.kind != ACCEL_NONE && s->triAccel.root != REF_EMPTY, sub = s->subdivAccel.kind != ACCEL_NONE && s->subdivAccel.root != REF_EMPTY;
  if (tri == sub) return false; // two accels (AccelN) or none: the general path
  const Accel& A = tri ? s->triAccel : s->subdivAccel;
  const uint32_t worst = 7u * (A.maxDepth + 1u) + 2u;
  const uint32_t need = worst > (uint32_t)TRACE_LDS_STACK ? worst - TRACE_LDS_STACK : 0u;
  if (need > Device::Service::SPILL_DEPTH) return false;
  Device::GpuShard& sh = dev->primary();
  Device::Service* sv = dev->service;
  if (!sv) {
    std::lock_guard<std::mutex> g(dev->serviceMutex);
    if (!dev->service) {
      sh.use();
      std::unique_ptr<Device::Service> n(new Device::Service);
      n->kind = A.kind;
      n->levels = s->compressionLevel;
      for (uint32_t i = 0; i < Device::Service::SLOTS; i++) { n->slotLock[i].store(0u); n->slotSeq[i] = 0u; }
      LaunchParams& p = n->base;
      memset(&p, 0, sizeof(p));
      p.accel = A.desc(0);
      p.cbvhLevels = s->compressionLevel;
      p.numCUs = (uint32_t)sh.numCUs;
      p.rayChunk = (uint32_t)SERVICE_SLOT_RAYS;
      p.leafBatch = dev->tuneLeafBatch;
      p.refillBatch = dev->tuneRefillBatch;
      p.octMax = dev->tuneOctMax;
      p.walkBatch = dev->tuneWalkBatch;
      p.octSteps = dev->tuneOctSteps;
      p.octLeaf = dev->tuneOctLeaf != 0xFFFFFFFFu ? dev->tuneOctLeaf : (A.kind == ACCEL_GRIDSOA ? 24u : 16u);
      p.overflow = sh.overflowDev;
      p.spillDepth = Device::Service::SPILL_DEPTH;
      p.gridBlocks = Device::Service::SLOTS / (TRACE_BLOCK / 64);
      try {
        HIP_CHECK(hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking));
        HIP_CHECK(hipHostMalloc((void**)&n->slotsHost, sizeof(ServiceSlot) * Device::Service::SLOTS, hipHostMallocMapped));
        memset(n->slotsHost, 0, sizeof(ServiceSlot) * Device::Service::SLOTS);
        HIP_CHECK(hipHostGetDevicePointer((void**)&n->slotsDev, n->slotsHost, 0));
        HIP_CHECK(hipHostMalloc((void**)&n->stopHost, 128, hipHostMallocMapped));
        memset(n->stopHost, 0, 128);
        HIP_CHECK(hipHostGetDevicePointer((void**)&n->stopDev, n->stopHost, 0));
        HIP_CHECK(hipMalloc((void**)&n->activityDev, 128));
        HIP_CHECK(hipMemset(n->activityDev, 0, 128));
        HIP_CHECK(hipMalloc(&n->spillDev, (size_t)Device::Service::SLOTS * 64u * Device::Service::SPILL_DEPTH * 8u + 16u));
        p.spill = n->spillDev;
        service_register(n.get());
        service_start_locked(dev, n.get());
      } catch (...) { // no service kernel for this accel kind / level, or out of memory: the combiner serves the calls
        n->failed = true;
        (void)hipGetLastError();
      }
      dev->service = n.release();
    }
    sv = dev->service;
  }
  if (sv->failed || sv->kind != A.kind || (sv->levels != s->compressionLevel && A.kind != ACCEL_TRI_PLUECKER && A.kind != ACCEL_TRI_MOELLER && A.kind != ACCEL_GRIDSOA)) return false;

  // a slot: threads are dealt slots round-robin once; a shared slot is taken in turns
  static thread_local uint32_t mySlot = 0xFFFFFFFFu;
  static thread_local const Device::Service* mySlotOf = nullptr;
  if (mySlotOf != sv) { mySlot = sv->nextSlot.fetch_add(1u) % Device::Service::SLOTS; mySlotOf = sv; }
  unsigned spins = 0;
  for (;;) {
    uint32_t expect = 0u;
    if (sv->slotLock[mySlot].compare_exchange_weak(expect, 1u, std::memory_order_acquire)) break;
    if ((++spins & 255u) == 0u) std::this_thread::yield();
  }
  struct Unlock { std::atomic<uint32_t>& l; ~Unlock() { l.store(0u, std::memory_order_release); } } unlock{sv->slotLock[mySlot]};

  const uint64_t t0 = now_ns();
  if (t0 - sv->lastSubmitNs.load(std::memory_order_relaxed) > 10000000ull) { // quiet for 10 ms: the kernel may have left, look before the job goes in
    std::lock_guard<std::mutex> g(dev->serviceMutex);
    sh.use();
    service_start_locked(dev, sv);
  }
  sv->lastSubmitNs.store(t0, std::memory_order_relaxed);
  ServiceSlot& slot = sv->slotsHost[mySlot];
  const uint32_t rec = occluded ? (uint32_t)sizeof(RTCRay) : (uint32_t)sizeof(RTCRayHit);
  for (uint32_t i = 0; i < M; i++) memcpy(slot.rays + (size_t)i * rec, rays + (size_t)i * byteStride, rec);
  if (M == 1u) { // the ray itself rides in the polled header line (trace_service.hip.h)
    memcpy(slot.ray0, rays, 28);
    memcpy(slot.ray0 + 7, rays + 32, 4);
  }
  slot.count = M;
  slot.occluded = occluded ? 1u : 0u;
  slot.instID = instID;
  slot.spillDepth = need;
  slot.accel = A.desc(0);
  const uint32_t seq = ++sv->slotSeq[mySlot];
  __atomic_store_n(&slot.seq2, seq, __ATOMIC_RELEASE);
  __atomic_store_n(&slot.seq, seq, __ATOMIC_RELEASE);
  spins = 0;
  uint64_t lastCheck = t0;
  while (__atomic_load_n(&slot.done, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((++spins & 127u) == 0u) std::this_thread::yield(); // more callers than cores: let the others put their jobs in
    if ((spins & 4095u) == 0u) {
      const uint64_t t = now_ns();
      if (t - lastCheck > 2000000ull) { // 2 ms without an answer: has the kernel left (idle exit raced with this job)?  restart it
        std::lock_guard<std::mutex> g(dev->serviceMutex);
        sh.use();
        service_start_locked(dev, sv);
        lastCheck = t;
      }
      if (t - t0 > 5000000000ull) RT_THROW(RTC_ERROR_UNKNOWN, "the small-call service kernel does not answer");
    }
  }
  for (uint32_t i = 0; i < M; i++) {
    char* dst = rays + (size_t)i * byteStride;
    const char* src = slot.rays + (size_t)i * rec;
    memcpy(dst + 32, src + 32, 4);
    if (!occluded) memcpy(dst + 48, src + 48, 32);
  }
  dev->statServiceCalls++;
  return true;
}

// ---- call combiner (row f2) -----------------------------------------------------------------------------------------
// The reference answers rtcIntersect1 in ~1 us on the calling core; here a call costs a staging copy, a kernel launch
// and a synchronisation (~60 us) whatever its size.  Harness threads of an embree application call concurrently
// (SURVEY.md section 8b "Threading"), so the calls that arrive while a launch is in flight are traced together by the
// next leader: T calling threads then see ~T rays per launch instead of one.  Results are those of independent calls
// (a stream is M independent single-ray calls); no timer, no extra thread, no CPU traversal.
static void combine_process(Device* dev, std::vector<Device::SmallCall*>& batch)
{
  // group by (scene, kind, instID); each group becomes one contiguous batch
  std::vector<char> done(batch.size(), 0);
  for (size_t i = 0; i < batch.size(); i++) {
    if (done[i]) continue;
    Device::SmallCall* a = batch[i];
    std::vector<Device::SmallCall*> group;
    size_t total = 0;
    for (size_t j = i; j < batch.size(); j++) {
      Device::SmallCall* b = batch[j];
      if (!done[j] && b->scene == a->scene && b->occluded == a->occluded && b->instID == a->instID) {
        done[j] = 1;
        group.push_back(b);
        total += b->M;
      }
    }
    const uint32_t rec = a->occluded ? (uint32_t)sizeof(RTCRay) : (uint32_t)sizeof(RTCRayHit);
    try {
      std::vector<char> tmp(total * rec + 16);
      char* base = (char*)(((uintptr_t)tmp.data() + 15) & ~(uintptr_t)15);
      size_t k = 0;
      for (Device::SmallCall* c : group)
        for (uint32_t r = 0; r < c->M; r++, k++) memcpy(base + k * rec, c->base + (size_t)r * c->stride, rec);
      RTCIntersectContext ctx;
      memset(&ctx, 0, sizeof(ctx));
      ctx.instID[0] = a->instID;
      trace_batch(a->scene, base, (uint32_t)total, rec, a->occluded, &ctx, nullptr);
      k = 0;
      for (Device::SmallCall* c : group)
        for (uint32_t r = 0; r < c->M; r++, k++) {
          char* dst = c->base + (size_t)r * c->stride;
          memcpy(dst + 32, base + k * rec + 32, 4);
          if (!c->occluded) memcpy(dst + 48, base + k * rec + 48, 32);
        }
      dev->statCombinedBatches++;
    } catch (const rtc_error& e) {
      for (Device::SmallCall* c : group) { c->error = e.code; c->message = e.msg; }
    } catch (const std::exception& e) {
      for (Device::SmallCall* c : group) { c->error = RTC_ERROR_UNKNOWN; c->message = e.what(); }
    }
  }
}

void trace_call(Scene* s, void* rays, uint32_t M, size_t byteStride, bool occluded, const RTCIntersectContext* ctx)
{
  Device* dev = s->device;
  if (M == 0 || M > COMBINE_MAX_RAYS || (ctx && ctx->filter) || s->triIntersectFilter || s->triOccludedFilter || s->subdivFilter || s->modified || (((uintptr_t)rays) & 3) ||
      byteStride > 0xFFFFFFFFull || is_device_pointer(rays)) {
    trace_batch(s, rays, M, byteStride, occluded, ctx, nullptr); // large, device-resident, or about to raise its own error
    return;
  }
  if (service_trace(s, (char*)rays, M, byteStride, occluded, ctx ? ctx->instID[0] : RTC_INVALID_GEOMETRY_ID)) return;
  Device::SmallCall call;
  call.scene = s;
  call.base = (char*)rays;
  call.M = M;
  call.stride = byteStride;
  call.occluded = occluded;
  call.instID = ctx ? ctx->instID[0] : RTC_INVALID_GEOMETRY_ID;
  std::unique_lock<std::mutex> lk(dev->combMutex);
  dev->combPending.push_back(&call);
  dev->statCombinedCalls++; // counted once the call is pending (the held-leader test waits on this count)
  while (!call.done) {
    if (!dev->combBusy) { // become the leader for everything that is pending now (own call included)
      dev->combBusy = true;
      while (dev->combHold.load(std::memory_order_acquire)) { // test hook (rtcamdDebugHoldCombiner): the leader waits, the others queue up behind it
        lk.unlock();
        std::this_thread::yield();
        lk.lock();
      }
      std::vector<Device::SmallCall*> batch;
      batch.swap(dev->combPending);
      lk.unlock();
      combine_process(dev, batch);
      lk.lock();
      for (Device::SmallCall* c : batch) c->done = true;
      dev->combBusy = false;
      dev->combCv.notify_all();
    } else
      dev->combCv.wait(lk);
  }
  lk.unlock();
  if (call.error != RTC_ERROR_NONE) throw rtc_error(call.error, call.message);
}

void trace_pointers(Scene* s, void** ptrs, uint32_t M, bool occluded, const RTCIntersectContext* ctx)
{
  // rtcIntersect1Mp / rtcOccluded1Mp: gather the pointed-to records into one batch (filterAOP, filters.cpp:167-)
  const uint32_t rec = occluded ? (uint32_t)sizeof(RTCRay) : (uint32_t)sizeof(RTCRayHit);
  std::vector<char> tmp((size_t)M * rec + 16);
  char* base = (char*)(((uintptr_t)tmp.data() + 15) & ~(uintptr_t)15);
  for (uint32_t i = 0; i < M; i++) memcpy(base + (size_t)i * rec, ptrs[i], rec);
  trace_batch(s, base, M, rec, occluded, ctx, nullptr);
  for (uint32_t i = 0; i < M; i++) {
    memcpy((char*)ptrs[i] + 32, base + (size_t)i * rec + 32, 4);
    if (!occluded) memcpy((char*)ptrs[i] + 48, base + (size_t)i * rec + 48, 32);
  }
}

} // namespace rtamd
