// The traversal kernel skeleton: persistent wavefronts with dynamic ray fetch and per-lane LDS stacks.
//
// MI355X design (not a port of the reference's per-ray CPU loop):
//  * Persistent grid (2 workgroups of 4 waves per CU by default; the third wave slot of a SIMD is left to other batches
//    in flight).  Rays are pulled from 64 work queues (heads in separate 128-byte lines; a wave starts at its home
//    queue and finds the next live one with a single vector load of all heads) in chunks of 256, so that the atomic
//    traffic stays far below what one counter word sustains (~70 ns per RMW and address) and the batch is balanced
//    dynamically: random rays differ by more than 10x in traversal cost and a static assignment leaves most SIMDs
//    idle in the tail.
//  * A lane whose ray has finished is refilled from the wave's chunk (ballot + mbcnt rank), so finished rays do
//    not park lanes ("wavefront ballot for active-ray compaction").
//  * if-if traversal step: every iteration each lane handles ONE event: an inner node, a leaf, or a pop.  Leaves
//    are expensive and rare, so lanes that reached a leaf wait until leafBatch lanes want one (or no lane can do
//    node work): the leaf code then runs with many lanes active instead of a handful.
//  * Per-lane traversal stack in LDS, entry-major (stack[entry][lane]: bank = lane, never a conflict, whatever
//    the depth mix), with an HBM overflow area for pathological depths.
// Per-ray semantics are untouched by any of this: each lane performs exactly the reference's depth-first sequence
// for its ray (kernels/bvh/bvh_intersector1.cpp:40-126 closest hit, :128-209 any hit; child order
// kernels/bvh/bvh_traverser1.h:549-666 with the tie rules of kernels/common/stack_item.h:39-80).
//
// A kernel is this skeleton instantiated with a Leaf policy:
//   struct Leaf { static __device__ void prepare();   // once per workgroup, before the loop (LDS tables)
//                 template<bool OCCLUDED,bool COUNT> static __device__ bool intersect(const LaunchParams&, uint32_t ref, RayState&, WorkCounters&, uint32_t rayIdx); }
// intersect returns true when an any-hit query is finished (ray occluded).
#pragma once
#include "trace_common.hip.h"

namespace rtamd {
namespace dev {

// Minimum resident waves per SIMD requested from the register allocator (second __launch_bounds__ argument).
// Measured on MI355X (1 M rays, bomberman L6/C3): asking for 4 or 5 waves makes the subdivision kernels spill
// 76-168 bytes per lane and costs 12-48 % (cbvh.leaf 5.44 -> 4.81 -> 3.92 Grays/s), so the bound stays at 3
// (<= 168 VGPRs, no scratch); the triangle kernels need 103 VGPRs and get 4 waves either way.
#ifndef TRACE_MIN_WAVES_PER_SIMD
#define TRACE_MIN_WAVES_PER_SIMD 3
#endif
// 1: the instrumented twin also reads the shader clock around every phase (cyclesFetch/Node/Leaf/Pop); the reads
// serialise the scalar memory pipe and slow that twin down ~2x, so the default build leaves those four counters at 0
// waves per SIMD the instrumented twins are compiled for (see trace_kernel below)
#ifndef TRACE_COUNT_MIN_WAVES
#define TRACE_COUNT_MIN_WAVES(Leaf) 2
#endif
#ifndef TRACE_PHASE_STAMPS
#define TRACE_PHASE_STAMPS 0
#endif
// rays a wave takes from a queue per atomic and lanes that must wait at a leaf before the leaf code runs come from
// LaunchParams::rayChunk / leafBatch (defaults 256 / 32, Device::tuneChunk / tuneLeafBatch)
static constexpr uint32_t QUEUE_STRIDE = TRACE_QUEUE_STRIDE; // queue heads live in separate 128-byte lines

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) // number of set bits of `mask` below this lane
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---- child-parallel ("octet") node step ----------------------------------------------------------------------
// When at most TRACE_OCT_MAX lanes of a wave have node work, the node step is not run lane-per-ray (one lane tests the
// 8 children of its node one after the other: ~600 dependent instructions whatever the number of busy lanes) but with 8
// lanes per ray: lane 8g+k tests child k of the node of the g-th ray.  The rays' node-test constants travel through a
// small LDS exchange area, the visiting order of the hit children comes from seven DPP compares inside the octet, the
// stacked children are written straight into the owning lane's LDS stack column.  Arithmetic, child order and stack
// contents are exactly those of the lane-per-ray step below (same TravRay functions, same tie rule), so results do not
// change; what changes is the critical path of the few deep rays that bound a batch's drain (measured on MI355X: a lone
// ray needs ~1.1 us per lane-per-ray node step) and the instructions those nearly empty waves take from everybody else.
#ifndef TRACE_OCT_MAX
#define TRACE_OCT_MAX 30 // exchange rows per wave: 4 waves x 30 rows x 48 B + 17 stack rows + the cBVH tables keep a workgroup at 40 704 B of LDS, four per CU
#endif
static constexpr int OCT_ROWS = TRACE_OCT_MAX > 0 ? TRACE_OCT_MAX : 1; // rays per wave the exchange area holds
static constexpr int OCT_WORDS = 12; // TravRay (7) + travFar + cur + sp + owner thread + ray.tfar
#ifndef TRACE_OCT_PIPE
#define TRACE_OCT_PIPE 1
#endif
// passes of 8 rays whose node loads are in flight together.  Measured: 2 or 4 make the register allocator spill 136-168 bytes
// per lane in the subdivision kernels (the whole loop then runs 20 % slower), so one pass at a time it is
static constexpr int OCT_PIPE = TRACE_OCT_PIPE;
enum : int { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_XOR3 = 0x1B, DPP_HALF_MIRROR = 0x141 }; // quad_perm / row_half_mirror
// leaves whose visit splits into a cheap test and a walk declare `static constexpr bool TWO_STAGE` (and frustum_pass); see the leaf step
template <typename Leaf, typename = void> struct leaf_two_stage { static constexpr bool value = false; };
template <typename Leaf> struct leaf_two_stage<Leaf, decltype((void)Leaf::TWO_STAGE)> { static constexpr bool value = Leaf::TWO_STAGE; };

template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// DIRECT (service kernel, trace_service.hip.h): the rays [0, P.count) belong to THIS wavefront - no work queues, nothing to grab
template <typename Leaf, bool ROBUST, bool OCCLUDED, bool COUNT, bool VEC, bool DIRECT = false>
__device__ __forceinline__ void trace_body(const LaunchParams& P, uint2 (*ldsStack)[TRACE_BLOCK], float (*octX)[OCT_WORDS])
{
  const uint32_t tid = threadIdx.x;
  const uint32_t gthread = blockIdx.x * TRACE_BLOCK + tid;
  // overflow column of this lane in the HBM spill area: formed where it is used (rare path) from an opaque copy of the thread
  // index - as a loop invariant it would be hoisted and hold two VGPRs across the whole loop
  auto spill_col = [&]() -> uint2* {
    uint32_t g = gthread;
    asm volatile("" : "+v"(g));
    return (uint2*)P.spill + (size_t)g * P.spillDepth;
  };
  const QNode8* __restrict__ nodes = P.accel.nodes;
  uint32_t* __restrict__ queues = P.queues;

  // TRACE_QUEUES work queues; queue q owns the contiguous rays [q*perQ, (q+1)*perQ).  One RMW on a queue head costs
  // ~70 ns serialized per address on MI355X (measured: 8 queues x 128-ray chunks made the grabs alone cost ~60 us per
  // 1 M rays), so the heads are spread over 64 addresses in separate 128-byte lines; a wave starts at its home queue
  // and, when that is drained, finds the next non-empty queue with ONE vector load of all heads.
  const uint32_t perQ = (P.count + (uint32_t)TRACE_QUEUES - 1u) / (uint32_t)TRACE_QUEUES;
  const uint32_t laneId = lane_rank(~0ull);
  // Entries queue q hands out: its share of the batch, or - after the root cull pre-pass (trace_cull.hip.h) - the survivors the
  // pre-pass appended to the queue's list (count next to the queue head); the entries are then ray indices read from the list.
  const uint32_t* __restrict__ survivors = P.survivors;
  auto queue_len = [&](uint32_t q) -> uint32_t {
    if (survivors) return __hip_atomic_load(&queues[q * QUEUE_STRIDE + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lo = min(q * perQ, P.count);
    return min(lo + perQ, P.count) - lo;
  };
  uint32_t qCur = (blockIdx.x * (TRACE_BLOCK / 64) + (tid >> 6)) & (uint32_t)(TRACE_QUEUES - 1); // wave-uniform
  uint32_t poolNext = 0, poolEnd = DIRECT ? P.count : 0u; // wave-uniform: rays [poolNext, poolEnd) belong to this wave
  // Staggered exhaustion: a quarter of the waves treats a queue as closed once 85 % of it are handed out, another quarter
  // at 92 %, the rest drains it.  The waves then enter their drain (few deep rays, few active lanes) at different times
  // instead of all at once; the SIMD slots of the early leavers are free for the other batches in flight.  Measured on
  // cbvh.leaf, 1 M rays: +2.8 % with four batches in flight (11.17 vs 10.88 Grays/s over 40 steps), -1 % with two or three,
  // alone unchanged.
  // (All waves at 100 %: baseline; half of the waves at 85 %: +2 % / -6 % alone; four levels 70..100 %: +1 % / -10 %.)
  const uint32_t wIdx = blockIdx.x * (TRACE_BLOCK / 64) + (tid >> 6);
  // (r2, second session, 4096 waves of the quad-form kernel: a THIRD class at 96 % - only a quarter of the waves drains the queues - takes a
  // batch alone from 0.148 to 0.143 ms and four in flight from 11.5 to 11.6 Grays/s; levels swept with the metric-only dev build:
  // 85/92/96 0.1426-0.1444 ms, 80/90/96 0.1431-0.1434, 88/94/98 0.1434-0.1456, 75/85/95 0.1429-0.1440 but -1.7 % in flight, 85/92/100 (before)
  // 0.1471-0.1494, no staggering at all 0.1605)
#ifndef TRACE_STAG_A
#define TRACE_STAG_A 85u
#define TRACE_STAG_B 92u
#define TRACE_STAG_C 96u
#endif
  // (the third class only in the kernels that run four workgroups per CU - grid cells, cBVH quad form: one 256-ray chunk per wave; the triangle
  // kernels, two workgroups per CU and two chunks per wave, lose with it: alone 0.102 -> 0.107 ms)
  const uint32_t stag = (wIdx & 3u) == 1u ? TRACE_STAG_A : ((wIdx & 3u) == 3u ? TRACE_STAG_B : (((wIdx & 3u) == 2u && Leaf::OCTET_ONLY) ? TRACE_STAG_C : 100u));
  bool exhausted = P.accel.root == REF_EMPTY;

  WorkCounters wc;
  RayState r;
  TravRay<ROBUST> tr;
  float travFar = 0.f;
  uint32_t sp = 0, cur = REF_EMPTY, rayIdx = 0;
  // lane state bits (vector register, see RayState::hit): ST_ACTIVE = the lane owns a ray, ST_POP = its next event is a pop
  // ST_WALK (two-stage leaves): the ray passed the cheap first stage of its leaf and waits for the walk
  enum : uint32_t { ST_ACTIVE = 1u, ST_POP = 2u, ST_WALK = 4u };
  uint32_t st = 0u;
  r.hit = 0u;

  auto push = [&](uint32_t ref, uint32_t dist, uint32_t slot) {
    if (slot < (uint32_t)TRACE_LDS_STACK) ldsStack[slot][tid] = make_uint2(ref, dist);
    else {
      if (slot - TRACE_LDS_STACK < P.spillDepth) spill_col()[slot - TRACE_LDS_STACK] = make_uint2(ref, dist);
      else __hip_atomic_store(P.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // entry dropped: the host raises an error
      if (COUNT) wc.spills++;
    }
  };
  // overflow entries: nontemporal loads, so that the compiler keeps this rare global path apart from the LDS read
  // (merged, both become one flat access with flat latency on every pop)
  auto pop_spill = [&](uint32_t slot) -> uint2 {
    if (!(slot - TRACE_LDS_STACK < P.spillDepth)) return make_uint2(REF_EMPTY, 0x7f800000u);
    const uint32_t* e = (const uint32_t*)(spill_col() + (slot - TRACE_LDS_STACK));
    return make_uint2(__builtin_nontemporal_load(e), __builtin_nontemporal_load(e + 1));
  };

  // phase stamps of the instrumented twin: s_memtime around the wave-uniform phases (diagnostic only)
  unsigned long long tFetch = 0, tNode = 0, tLeaf = 0, tPop = 0, nIter = 0, nLeafPhase = 0, tStamp = 0, tBegin = 0;
  auto stamp = [&](unsigned long long& acc) {
    if (COUNT && TRACE_PHASE_STAMPS) {
      const unsigned long long now = __builtin_readcyclecounter();
      acc += now - tStamp;
      tStamp = now;
    }
  };
  unsigned long long laneIters = 0, rtBegin = 0, rtLastGrab = 0;
  uint32_t raySteps = 0, maxRaySteps = 0;
  if (COUNT) {
    tBegin = tStamp = __builtin_readcyclecounter();
    rtBegin = __builtin_amdgcn_s_memrealtime();
  }
#ifdef TRACE_TIMELINE
  // development build: the PLAIN kernel's wave timeline (trace.h `timeline`): wave-uniform scalars only
  unsigned long long tlBegin = 0, tlFirst = 0, tlGrab = 0;
  uint32_t tlIter = 0, tlDrainIter = 0, tlDrainLanes = 0, tlRays = 0;
  if (!COUNT && P.timeline) tlBegin = __builtin_amdgcn_s_memrealtime();
#endif

  for (;;) {
    if (COUNT) nIter++;
#ifdef TRACE_TIMELINE
    if (!COUNT && P.timeline) {
      tlIter++;
      if (exhausted) { tlDrainIter++; tlDrainLanes += (uint32_t)__popcll(__ballot((st & ST_ACTIVE) != 0u)); }
    }
#endif
    // ---- refill idle lanes ---------------------------------------------------------------------------
    const uint64_t idleMask = __ballot(!(st & ST_ACTIVE));
    // refilling a handful of lanes costs as many instructions as refilling all 64: wait until refillBatch lanes are
    // idle (or until nothing else can run)
    if (idleMask != 0ull && !exhausted && (__popcll(idleMask) >= (int)P.refillBatch || idleMask == ~0ull)) {
      if (DIRECT && poolNext == poolEnd) exhausted = true;
      else if (poolNext == poolEnd) { // take a new chunk (one lane does the atomic, the result is wave-uniform)
        for (;;) {
          const uint32_t qLo = qCur * perQ; // first entry of the queue (ray index, or position in the survivor lists)
          const uint32_t qLen = queue_len(qCur);
          uint32_t base = 0xFFFFFFFFu;
          bool open = true;
          if (stag != 100u) { // this wave treats the queue as closed once `stag` percent of it are handed out
            uint32_t pre = 0;
            if (laneId == 0u) pre = __hip_atomic_load(&queues[qCur * QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pre = __builtin_amdgcn_readfirstlane(pre);
            open = pre < (uint32_t)((uint64_t)qLen * stag / 100u);
          }
          if (open) {
            if (laneId == 0u) base = atomicAdd(&queues[qCur * QUEUE_STRIDE], P.rayChunk);
            base = __builtin_amdgcn_readfirstlane(base);
          }
          if (base < qLen) {
            if (COUNT) rtLastGrab = __builtin_amdgcn_s_memrealtime();
#ifdef TRACE_TIMELINE
            if (!COUNT && P.timeline) { tlGrab = __builtin_amdgcn_s_memrealtime(); if (!tlFirst) tlFirst = tlGrab; tlRays += min(P.rayChunk, qLen - base); }
#endif
            poolNext = qLo + base;
            poolEnd = min(poolNext + P.rayChunk, qLo + qLen);
            break;
          }
          // drained: lane l reads head l (heads only grow, so a stale value can only under-report "drained"), the
          // ballot marks the queues that still have rays, take the next one cyclically after qCur
          const uint32_t head = __hip_atomic_load(&queues[laneId * QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t myLen = queue_len(laneId);
          const uint64_t live = __ballot(laneId < (uint32_t)TRACE_QUEUES && head < (uint32_t)((uint64_t)myLen * stag / 100u));
          if (live == 0ull) { exhausted = true; break; }
          const uint64_t rot = (live >> qCur) | (qCur ? (live << (64u - qCur)) : 0ull); // bit k = queue (qCur+k)&63
          qCur = (qCur + (uint32_t)__builtin_ctzll(rot)) & (uint32_t)(TRACE_QUEUES - 1);
        }
      }
      if (poolNext != poolEnd) {
        const uint32_t mine = poolNext + lane_rank(idleMask);
        if (!(st & ST_ACTIVE) && mine < poolEnd) {
          rayIdx = survivors ? survivors[mine] : mine;
          const char* rp = (const char*)P.rays + (size_t)rayIdx * P.stride;
          if (DIRECT && P.inlineRay) { // (service kernel: the ray came with the polled slot header - no second read of host memory)
            const float* x = octX[0];
            r.ox = x[0]; r.oy = x[1]; r.oz = x[2]; r.tnear = x[3];
            r.dx = x[4]; r.dy = x[5]; r.dz = x[6]; r.tfar = x[7];
          } else
            load_ray<VEC>(rp, r);
          r.hit = 0u;
          // stream front-end: rays with tnear > tfar are skipped (bvh_intersector_stream_filters.cpp:156);
          // occluded: already-occluded rays return early (bvh_intersector1.cpp:132-134)
          bool ok = r.tnear <= r.tfar;
          if (OCCLUDED) ok = ok && !(r.tfar < 0.0f);
          if (ok) {
            if (COUNT) { wc.rays++; raySteps = 0; }
            tr.init(r);
            travFar = fmaxf(r.tfar, 0.0f); // tray.tfar
            sp = 0;
            cur = P.accel.root;
            st = ST_ACTIVE;
          }
        }
        poolNext = min(poolNext + (uint32_t)__popcll(idleMask), poolEnd);
      }
    }
    stamp(tFetch);
    if (COUNT) {
      laneIters += (unsigned long long)__popcll(__ballot((st & ST_ACTIVE) != 0u));
      if (st & ST_ACTIVE) { raySteps++; maxRaySteps = max(maxRaySteps, raySteps); }
    }
    if (__ballot((st & ST_ACTIVE) != 0u) == 0ull) {
      if (exhausted) break;
      continue;
    }

    // ---- inner node step ---------------------------------------------------------------------------------
    const bool atNode = st == ST_ACTIVE && !(cur & REF_LEAF);
    const uint64_t nodeMask = __ballot(atNode);
    const uint32_t nNode = (uint32_t)__popcll(nodeMask);
    // the octet step writes LDS stack slots only: a ray whose push could reach the overflow area takes the other path
    const bool useOct = TRACE_OCT_MAX > 0 && nNode != 0u && nNode <= min((uint32_t)TRACE_OCT_MAX, P.octMax) &&
                        __ballot(atNode && sp + 7u > (uint32_t)TRACE_LDS_STACK) == 0ull;
    if (useOct) {
      {
        // lane-constant values of this block (lane index, octet row, tie constants) are derived from an opaque copy of the
        // lane id: otherwise the compiler hoists them out of the traversal loop and keeps ~12 registers alive through the
        // lane-per-ray node step and the leaf code, which are at the register limit already (measured: 64-80 B of scratch)
        uint32_t lid = laneId;
        asm volatile("" : "+v"(lid));
        const uint32_t myRow = lane_rank(nodeMask);
        if (atNode) {
          float* x = octX[myRow];
          tr.store(x, 1);
          x[7] = travFar;
          x[8] = __uint_as_float(cur);
          x[9] = __uint_as_float(sp);
          x[10] = __uint_as_float(tid);
          x[11] = r.tfar;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t k = lid & 7u;
        // Passes of 8 rays.  An octet stays with its ray for up to `maxSteps` node steps (descend into the nearest hit
        // child, push the others, pop when nothing is hit - the loop below IS the traversal loop of
        // bvh_intersector1.cpp:60-105 for that ray) and hands it back to the owning lane when the ray reaches a leaf, is
        // finished, or could overflow the LDS part of its stack: in the drain a deep ray then pays one memory round trip
        // and ~130 instructions per node instead of a full iteration of the wave's loop with three LDS exchanges.  While
        // the queues still have rays the octets return after P.octSteps steps so that idle lanes are refilled in time.
        const uint32_t maxSteps = exhausted ? 0xFFFFu : P.octSteps;
        for (uint32_t base = 0; base < nNode; base += 8u) {
          const uint32_t row = base + (lid >> 3);
          const uint32_t rowc = min(row, nNode - 1u);
          const float* x = octX[rowc];
          TravRay<ROBUST> t;
          t.load(x, 1);
          const float oFar = x[7];
          uint32_t oCur = __float_as_uint(x[8]);
          uint32_t oSp = __float_as_uint(x[9]);
          const uint32_t oTid = __float_as_uint(x[10]);
          const float rTfar = x[11]; // ray.tfar, for the distance cull of popped entries
          const bool ngx = t.negx(), ngy = t.negy(), ngz = t.negz();
          bool done = !(row < nNode); // uniform within an octet
          uint32_t steps = 0;
          while (__ballot(!done) != 0ull) {
            if (!done) {
              if (COUNT && k == 0u) wc.nodes++;
              const unsigned char* nb = (const unsigned char*)(nodes + oCur);
              const uint4 n0 = *(const uint4*)nb;
              const uint32_t cref = ((const uint32_t*)nb)[4u + k];
              // plane bytes: lower[0..7] at 48/64/80, upper[0..7] eight bytes further (x / y / z)
              const uint32_t qnx = nb[48u + (ngx ? 8u : 0u) + k], qfx = nb[48u + (ngx ? 0u : 8u) + k];
              const uint32_t qny = nb[64u + (ngy ? 8u : 0u) + k], qfy = nb[64u + (ngy ? 0u : 8u) + k];
              const uint32_t qnz = nb[80u + (ngz ? 8u : 0u) + k], qfz = nb[80u + (ngz ? 0u : 8u) + k];
              const float ox = __uint_as_float(n0.x), oy = __uint_as_float(n0.y), oz = __uint_as_float(n0.z);
              const float sx = __uint_as_float((n0.w & 0xffu) << 23);
              const float sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23);
              const float sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
              const float npx = madd((float)qnx, sx, ox), npy = madd((float)qny, sy, oy), npz = madd((float)qnz, sz, oz);
              const float fpx = madd((float)qfx, sx, ox), fpy = madd((float)qfy, sy, oy), fpz = madd((float)qfz, sz, oz);
              const float tN = fmaxf(t.nearT(npx, npy, npz), t.tnear);
              const float tF = fminf(t.farT(fpx, fpy, fpz), oFar);
              const bool h = (tN <= tF) & (cref != REF_EMPTY);
              const uint32_t dist = h ? __float_as_uint(tN) : 0x7FFFFFFFu; // non-hit: above every distance, below 2^31 (the ranking takes the sign of differences)
              // all 8 lanes of an octet are in here together (`done` is uniform within the octet), so the ballot bits and
              // the DPP partners of a lane are always live
              const uint32_t mask8 = (uint32_t)(__ballot(h) >> (lid & 56u)) & 0xffu;
              const uint32_t nhit = (uint32_t)__popc(mask8);
              uint32_t rank;
              if (OCCLUDED) rank = (uint32_t)__popc(mask8 >> (k + 1u)); // traverseAnyHit: highest index first
              else {
                // children visited before this one: strictly nearer, or equally near with a higher index (the tie rule
                // of the lane-per-ray step); non-hit lanes carry 0xFFFFFFFF and never count.  The partner of lane k under
                // "xor x" is child k^x; whether that index is higher is a per-lane constant.
                const uint32_t m = dpp_u32<DPP_HALF_MIRROR>(dist); // child k^7
                const uint32_t d1 = dpp_u32<DPP_XOR1>(dist), d2 = dpp_u32<DPP_XOR2>(dist), d3 = dpp_u32<DPP_XOR3>(dist);
                const uint32_t d4 = dpp_u32<DPP_XOR3>(m), d5 = dpp_u32<DPP_XOR2>(m), d6 = dpp_u32<DPP_XOR1>(m);
                // [d < dist + c] as the sign bit of the 32-bit difference (all operands are below 2^31): no v_cmp -> vcc -> v_addc chain
                rank = (d1 - (dist + ((k ^ 1u) > k ? 1u : 0u))) >> 31;
                rank += (d2 - (dist + ((k ^ 2u) > k ? 1u : 0u))) >> 31;
                rank += (d3 - (dist + ((k ^ 3u) > k ? 1u : 0u))) >> 31;
                rank += (d4 - (dist + ((k ^ 4u) > k ? 1u : 0u))) >> 31;
                rank += (d5 - (dist + ((k ^ 5u) > k ? 1u : 0u))) >> 31;
                rank += (d6 - (dist + ((k ^ 6u) > k ? 1u : 0u))) >> 31;
                rank += (m - (dist + ((k ^ 7u) > k ? 1u : 0u))) >> 31;
                // exactly four hit children with equal distances among them: the reference's 5-comparator network orders them
                // differently (trace_common.hip.h, sort4_network); nhit is uniform within the octet
                if (nhit == 4u) {
                  const bool tie = h && ((d1 == dist) | (d2 == dist) | (d3 == dist) | (d4 == dist) | (d5 == dist) | (d6 == dist) | (m == dist));
                  if (((uint32_t)(__ballot(tie) >> (lid & 56u)) & 0xffu) != 0u) { // rare
                    uint32_t mm = mask8;
                    const uint32_t i0 = (uint32_t)__ffs(mm) - 1u; mm &= mm - 1u;
                    const uint32_t i1 = (uint32_t)__ffs(mm) - 1u; mm &= mm - 1u;
                    const uint32_t i2 = (uint32_t)__ffs(mm) - 1u; mm &= mm - 1u;
                    const uint32_t i3 = (uint32_t)__ffs(mm) - 1u;
                    const uint32_t ob = lid & 56u; // first lane of this octet
                    const uint32_t e0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((ob + i0) << 2), (int)dist);
                    const uint32_t e1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((ob + i1) << 2), (int)dist);
                    const uint32_t e2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((ob + i2) << 2), (int)dist);
                    const uint32_t e3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((ob + i3) << 2), (int)dist);
                    if (h) rank = sort4_network(e0, e1, e2, e3, (uint32_t)__popc(mask8 & ((1u << k) - 1u)));
                  }
                }
              }
              if (nhit != 0u) {
                const uint32_t top = oSp + nhit - 1u;
                if (h && rank != 0u) ldsStack[top - rank][oTid] = make_uint2(cref, dist);
                // the nearest child's reference, to all lanes of the octet (exactly one lane contributes; a child
                // reference is never 0: node 0 is the root)
                uint32_t v = (h && rank == 0u) ? cref : 0u;
                v |= dpp_u32<DPP_XOR1>(v);
                v |= dpp_u32<DPP_XOR2>(v);
                v |= dpp_u32<DPP_HALF_MIRROR>(v);
                oCur = v;
                oSp = top;
              } else {
                // pop (bvh_intersector1.cpp:80-88): all lanes of the octet read the same entries
                for (;;) {
                  if (oSp == 0u) { oCur = REF_EMPTY; done = true; break; } // stack empty: the ray is finished
                  oSp--;
                  const uint2 e = ldsStack[oSp][oTid];
                  if (e.x == REF_EMPTY) continue;
                  if (!OCCLUDED && __uint_as_float(e.y) > rTfar) continue;
                  oCur = e.x;
                  break;
                }
              }
              steps++;
              // back to the owning lane: at a leaf, after maxSteps steps, or when the next push could leave the LDS stack
              if ((oCur & REF_LEAF) || steps >= maxSteps || oSp + 7u > (uint32_t)TRACE_LDS_STACK) done = true;
            }
          }
          if (row < nNode && k == 0u) {
            octX[row][8] = __uint_as_float(oCur);
            octX[row][9] = __uint_as_float(oSp);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (atNode) {
          cur = __float_as_uint(octX[myRow][8]);
          sp = __float_as_uint(octX[myRow][9]);
          if (cur == REF_EMPTY) st |= ST_POP;
        }
      }
    }
    else if (atNode) {
      if (COUNT) wc.nodes++;
      const uint4* np = (const uint4*)(nodes + cur);
      const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4], n5 = np[5];
      const float ox = __uint_as_float(n0.x), oy = __uint_as_float(n0.y), oz = __uint_as_float(n0.z);
      const float sx = __uint_as_float((n0.w & 0xffu) << 23);
      const float sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23);
      const float sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
      // near / far plane bytes per axis: words .x,.y = lower[0..7], .z,.w = upper[0..7]
      const bool ngx = tr.negx(), ngy = tr.negy(), ngz = tr.negz();
      const uint32_t nx0 = ngx ? n3.z : n3.x, nx1 = ngx ? n3.w : n3.y;
      const uint32_t fx0 = ngx ? n3.x : n3.z, fx1 = ngx ? n3.y : n3.w;
      const uint32_t ny0 = ngy ? n4.z : n4.x, ny1 = ngy ? n4.w : n4.y;
      const uint32_t fy0 = ngy ? n4.x : n4.z, fy1 = ngy ? n4.y : n4.w;
      const uint32_t nz0 = ngz ? n5.z : n5.x, nz1 = ngz ? n5.w : n5.y;
      const uint32_t fz0 = ngz ? n5.x : n5.z, fz1 = ngz ? n5.y : n5.w;
      const uint32_t cref[8] = {n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w};

      uint32_t dist[8];
      uint32_t mask = 0;
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
        const bool h = (tN <= tF) & (cref[k] != REF_EMPTY);
        // non-hit: distinct sentinels 0x7FFFFFF8..0x7FFFFFFF - above every distance (tN >= 0 is at most +inf = 0x7F800000) and, like the distances,
        // below 2^31, so that the ranking below can take the sign of a 32-bit difference; distinct for the tie detection
        dist[k] = h ? __float_as_uint(tN) : (0x7FFFFFF8u + (uint32_t)k);
        mask |= h ? (1u << k) : 0u;
      }
      const int nhit = __popc(mask);
#ifdef RTAMD_TRACE_RAY
      if (rayIdx == RTAMD_TRACE_RAY)
        printf("  GPU node %u (lane step) sp %u mask %02x dist %08x %08x %08x %08x %08x %08x %08x %08x refs %08x %08x %08x %08x %08x %08x %08x %08x tfar %a\n", cur, sp, mask,
               dist[0], dist[1], dist[2], dist[3], dist[4], dist[5], dist[6], dist[7], cref[0], cref[1], cref[2], cref[3], cref[4], cref[5], cref[6], cref[7], r.tfar);
#endif
      if (nhit == 0) st |= ST_POP;
      else if (nhit == 1) {
        const int k = __ffs(mask) - 1;
        uint32_t c = cref[0];
#pragma unroll
        for (int j = 1; j < 8; j++) c = (k == j) ? cref[j] : c;
        cur = c;
      } else {
        // rank[k] = number of hit children visited before child k; rank 0 is entered now, the others are stacked
        // so that they pop in rank order.
        uint32_t rank[8];
        if (OCCLUDED) {
          // traverseAnyHit (bvh_traverser1.h:638-666): descend into the highest-index hit child, stack the rest
          // in ascending index order, no sorting
#pragma unroll
          for (int k = 0; k < 8; k++) rank[k] = (uint32_t)__popc(mask >> (k + 1));
        } else {
          // traverseClosestHit: visit order = ascending uint(tNear), equal distances -> higher child index first
          // (strict compares in bvh_traverser1.h:590-591 and stack_item.h:39-80; four hit children: see below); non-hit
          // children carry sentinels above every distance and sort behind every hit child
          // (round 3) aFirst = [dist[a] < dist[b]] is the sign bit of the 32-bit difference (all values are below 2^31): two VALU
          // instructions without the v_cmp -> vcc -> v_addc chain and its hazard slots (100 s_nop in the 770 instructions of this step)
#pragma unroll
          for (int k = 0; k < 8; k++) rank[k] = (uint32_t)(7 - k); // child k is the lower index of 7-k pairs: counts them as "b first" ...
#pragma unroll
          for (int a = 0; a < 8; a++) {
#pragma unroll
            for (int b = a + 1; b < 8; b++) {
              const uint32_t aFirst = (dist[a] - dist[b]) >> 31; // tie -> 0 -> b (higher index) first
              rank[b] += aFirst;
              rank[a] -= aFirst; // ... and takes back the pairs a wins
            }
          }
          // exactly four hit children: the reference's 5-comparator network decides ties differently (trace_common.hip.h)
          if (__ballot(nhit == 4) != 0ull) {
            bool tie = false; // non-hit children carry distinct sentinels, so any equality is a tie between hit children
#pragma unroll
            for (int a = 0; a < 8; a++)
#pragma unroll
              for (int b = a + 1; b < 8; b++) tie |= dist[a] == dist[b];
            if (nhit == 4 && tie) rank4_by_network(mask, dist, rank);
          }
        }
        const uint32_t top = sp + (uint32_t)nhit - 1u;
        uint32_t next = REF_EMPTY;
        if (top <= (uint32_t)TRACE_LDS_STACK) {
          // common case, branch-free: every entry lands in LDS; children that are not stacked write to the scratch row
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const bool h = (mask >> k) & 1u;
            const bool stacked = h && rank[k] != 0u;
            next = (h && rank[k] == 0u) ? cref[k] : next;
            ldsStack[stacked ? top - rank[k] : (uint32_t)TRACE_LDS_STACK][tid] = make_uint2(cref[k], dist[k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) {
            if (mask & (1u << k)) {
              if (rank[k] == 0u) next = cref[k];
              else push(cref[k], dist[k], top - rank[k]);
            }
          }
        }
        sp = top;
        cur = next;
      }
    }

    stamp(tNode);
    // ---- leaf step: run only when enough lanes wait at a leaf, or when nobody has node work -------------------
    // (lanes whose node step just ended at a leaf count as waiting: they need no extra iteration to get there)
    const bool atLeafNow = st == ST_ACTIVE && (cur & REF_LEAF);
    const bool atNodeNext = st == ST_ACTIVE && !(cur & REF_LEAF);
    // Two-stage leaves (cBVH blobs, quad form): a visit is a cheap test (the frustum test: ~150 wave instructions for 16 rays, no
    // divergence; it ends 28 % of the visits of the metric's rays) and a long, divergent walk.  Run together, a pass lasts as long
    // as its longest walk, the lanes of the rays that left at the test idle, and a pass starts as soon as the wave runs out of node
    // work - typically with 10 of 16 quads filled.  Run apart, the tests go through as the rays arrive and the rays that passed are
    // parked (ST_WALK) until P.walkBatch of them wait - over several iterations - or the wave has nothing else to do; then full walk
    // passes serve them.  Per ray the order of events is unchanged (a parked ray does nothing else), so the results are, byte for
    // byte.  Measured round 3 (profiles/r03_two_stage_ab.txt): kernel alone 144 -> 137 us, four batches in flight +5 %; parking
    // without the test stage gains nothing.
    constexpr bool TWO_STAGE = !OCCLUDED && leaf_two_stage<Leaf>::value;
    bool walkNow = false, walkForced = false;
    if constexpr (TWO_STAGE) {
      const uint64_t newMask = __ballot(atLeafNow);
      const bool nodeWork = __ballot(atNodeNext) != 0ull;
      const uint32_t nNew = (uint32_t)__popcll(newMask);
      if constexpr (!Leaf::TWO_STAGE_TEST) { // development variant: no test stage, arrivals are parked at once
        if (atLeafNow) { if (COUNT) wc.leaves++; st |= ST_WALK; }
      } else
      if (nNew != 0u && (nNew >= max(P.octLeaf, 1u) || !nodeWork)) {
        if (COUNT) nLeafPhase++;
        uint32_t lid = laneId;
        asm volatile("" : "+v"(lid));
        const uint32_t myRow = lane_rank(newMask);
        const uint32_t nRows = min(nNew, (uint32_t)OCT_ROWS);
        const bool inPhase = atLeafNow && myRow < nRows;
        if (inPhase) {
          if (COUNT) wc.leaves++;
          float* x = octX[myRow];
          x[0] = r.ox; x[1] = r.oy; x[2] = r.oz; x[3] = r.tnear;
          x[4] = r.dx; x[5] = r.dy; x[6] = r.dz; x[7] = r.tfar;
          x[8] = __uint_as_float(cur);
          x[9] = __uint_as_float(0u); // set by the quad when the ray passes
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr uint32_t GROUP = (uint32_t)Leaf::GROUP;
        for (uint32_t base = 0; base < nRows; base += 64u / GROUP) {
          const uint32_t row = base + lid / GROUP;
          Leaf::frustum_pass(P, octX[min(row, nRows - 1u)], row < nRows, lid);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (inPhase) st |= __float_as_uint(octX[myRow][9]) != 0u ? ST_WALK : ST_POP;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      // the walk: enough parked rays, or nothing else left to do in this iteration (no node work, no new arrival waiting for its test,
      // nobody about to pop) - every iteration makes progress, parked rays cannot wait for ever
      const uint64_t parked = __ballot(st == (ST_ACTIVE | ST_WALK));
      if (parked != 0ull) {
        const bool other = nodeWork || __ballot(st == (ST_ACTIVE | ST_POP) || (st == ST_ACTIVE && (cur & REF_LEAF))) != 0ull;
        walkForced = !other;
        walkNow = (uint32_t)__popcll(parked) >= P.walkBatch || walkForced;
      }
    }
    const bool atLeaf = TWO_STAGE ? (walkNow && st == (ST_ACTIVE | ST_WALK)) : atLeafNow; // the rays of this iteration's leaf step
    const uint64_t leafMask = __ballot(atLeaf);
    if (leafMask != 0ull) {
      const bool nodeWork = !TWO_STAGE && __ballot(atNodeNext) != 0ull; // (two-stage leaves: decided above)
      const uint32_t nLeaf = (uint32_t)__popcll(leafMask);
      bool leafDone = false;
      if constexpr (Leaf::OCTET) {
        // Child-parallel leaf step (leaves whose primitives map onto the 8 lanes of an octet): lane 8g+k tests primitive k
        // of the leaf of the g-th waiting ray.  A pass of 8 rays costs about a fifth of the lane-per-ray leaf code and runs
        // on full wavefronts, so the lanes need not wait for P.leafBatch companions: 8 waiting rays are enough.
        // (OCTET_ONLY leaves have no lane-per-ray form in this kernel: at most OCT_ROWS waiting rays go through per phase, the
        // others wait for the next one; their lane-per-ray code - 40 live registers for a grid cell - is not even compiled in)
        if ((Leaf::OCTET_ONLY || (P.octLeaf != 0u && Leaf::octet_ok(P) && nLeaf <= (uint32_t)OCT_ROWS)) &&
            (nLeaf >= (Leaf::OCTET_ONLY ? max(P.octLeaf, 1u) : P.octLeaf) || !nodeWork)) {
          leafDone = true;
          if (COUNT) nLeafPhase++;
          uint32_t lid = laneId;
          asm volatile("" : "+v"(lid)); // see the node step: keeps the lane constants of this block out of the loop's live set
          const uint32_t myRow = lane_rank(leafMask);
          uint32_t nRows = min(nLeaf, (uint32_t)OCT_ROWS);
          // (two-stage leaves: full passes only while the wave has other work - the remainder stays parked; in flight +2 %)
          if (TWO_STAGE && !walkForced && nRows > 64u / (uint32_t)Leaf::GROUP) nRows = 64u / (uint32_t)Leaf::GROUP;
          const bool inPhase = atLeaf && myRow < nRows;
          if (inPhase) {
            if (COUNT && !TWO_STAGE) wc.leaves++;
            float* x = octX[myRow];
            x[0] = r.ox; x[1] = r.oy; x[2] = r.oz; x[3] = r.tnear;
            x[4] = r.dx; x[5] = r.dy; x[6] = r.dz; x[7] = r.tfar;
            x[8] = __uint_as_float(cur);
            x[9] = __uint_as_float(0u); // result flag
            x[10] = __uint_as_float(rayIdx);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // Leaf::GROUP lanes per ray: 8 (a block pair of triangles, the 8 triangles of a grid cell) or 4 (the quadtree of a cBVH blob)
          constexpr uint32_t GROUP = (uint32_t)Leaf::GROUP;
          for (uint32_t base = 0; base < nRows; base += 64u / GROUP) {
            const uint32_t row = base + lid / GROUP;
            Leaf::template octet_pass<OCCLUDED, COUNT>(P, octX[min(row, nRows - 1u)], row < nRows, lid, wc);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (inPhase) {
            const float* x = octX[myRow];
            if (__float_as_uint(x[9]) != 0u) {
              r.hit = 1u;
              if (OCCLUDED) {
                r.tfar = -RT_INF; // bvh_intersector1.cpp:198-201
                sp = 0;           // any hit found: terminate this ray
              } else if constexpr (Leaf::HIT_IN_MEMORY) {
                // the hit goes to the ray record now (a later, nearer hit overwrites it; nothing is stored when the ray ends): five
                // registers less across the loop for kernels at a register limit.  The ray's tfar in memory shrinks as it goes.
                RayState hr;
                hr.tfar = x[0]; hr.u = x[4]; hr.v = x[5];
                hr.geomID = __float_as_uint(x[6]); hr.primID = __float_as_uint(x[7]);
                float one = 1.f, zero = 0.f;
                asm volatile("" : "+v"(one), "+v"(zero));
                hr.ngx = one; hr.ngy = zero; hr.ngz = zero;
                store_hit<VEC>((char*)P.rays + (size_t)rayIdx * P.stride, hr, P.instID);
                r.tfar = hr.tfar;
              } else {
                r.tfar = x[0]; r.ngx = x[1]; r.ngy = x[2]; r.ngz = x[3]; r.u = x[4]; r.v = x[5];
                r.geomID = __float_as_uint(x[6]); r.primID = __float_as_uint(x[7]);
              }
            }
            travFar = OCCLUDED ? travFar : r.tfar; // tray.tfar = ray.tfar (bvh_intersector1.cpp:117)
            st = ST_ACTIVE | ST_POP;
          }
        }
      }
      if (!Leaf::OCTET_ONLY && !leafDone && (nLeaf >= P.leafBatch || !nodeWork)) {
        if (COUNT) nLeafPhase++;
        if constexpr (!Leaf::OCTET_ONLY) if (atLeaf) {
          if (COUNT) wc.leaves++;
          if (Leaf::template intersect<OCCLUDED, COUNT>(P, cur, r, wc, rayIdx)) {
            r.tfar = -RT_INF; // bvh_intersector1.cpp:198-201
            r.hit = 1u;
            sp = 0;           // any hit found: terminate this ray
          }
          travFar = OCCLUDED ? travFar : r.tfar; // tray.tfar = ray.tfar (bvh_intersector1.cpp:117)
          st |= ST_POP;
        }
      }
    }

    stamp(tLeaf);
    // ---- pop -------------------------------------------------------------------------------------------------------
    if (st == (ST_ACTIVE | ST_POP)) {
      bool finished = false;
      for (;;) {
        if (sp == 0) { finished = true; break; }
        sp--;
        uint2 e;
        if (sp < (uint32_t)TRACE_LDS_STACK) e = ldsStack[sp][tid];
        else e = pop_spill(sp);
        if (e.x == REF_EMPTY) continue;                           // entry lost to an exhausted spill area
#ifdef RTAMD_TRACE_RAY
        if (rayIdx == RTAMD_TRACE_RAY) printf("  GPU pop sp %u ref %08x dist %08x tfar %a\n", sp, e.x, e.y, r.tfar);
#endif
        if (!OCCLUDED && __uint_as_float(e.y) > r.tfar) continue; // bvh_intersector1.cpp:86
        cur = e.x;
        break;
      }
      st = ST_ACTIVE;
      if (finished) {
        if (r.hit) {
          char* rp = (char*)P.rays + (size_t)rayIdx * P.stride;
          if (COUNT) wc.hits++;
          if (OCCLUDED) ((float*)rp)[8] = r.tfar;
          else if constexpr (Leaf::HIT_IN_MEMORY) {
            // already stored when it was found
          } else {
            // leaves that report a constant normal (the fork's dummy Ng = (1,0,0), compressed.h:575,638) do not keep it in
            // registers across the loop: three VGPRs less in the kernels that sit at the register limit
            if constexpr (Leaf::CONST_NG) {
              float one = 1.f, zero = 0.f;
              asm volatile("" : "+v"(one), "+v"(zero)); // opaque: otherwise the constants are hoisted out of the loop and live (spilled) across it
              r.ngx = one; r.ngy = zero; r.ngz = zero;
            }
            store_hit<VEC>(rp, r, P.instID);
          }
        }
        st = 0u;
      }
    }
    stamp(tPop);
  }

#ifdef TRACE_TIMELINE
  if (!COUNT && P.timeline) {
    const uint32_t waveIdx = blockIdx.x * (TRACE_BLOCK / 64) + (tid >> 6);
    if (laneId == 0u && waveIdx < WAVE_LOG_CAPACITY) {
      unsigned long long* o = P.timeline + (size_t)waveIdx * 8u;
      o[0] = tlBegin; o[1] = tlFirst; o[2] = tlGrab; o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = tlIter; o[5] = tlDrainIter; o[6] = tlDrainLanes; o[7] = tlRays;
    }
  }
#endif
  if (COUNT) {
    // per-lane work counters are reduced over the wave; lane 0 stores the wave's record into its own slot
    unsigned long long v[7] = {wc.rays, wc.nodes, wc.leaves, wc.prims, wc.inner, wc.hits, wc.spills};
#pragma unroll
    for (int i = 0; i < 7; i++)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxRaySteps = max(maxRaySteps, (uint32_t)__shfl_xor((int)maxRaySteps, off, 64));
    const uint32_t waveIdx = blockIdx.x * (TRACE_BLOCK / 64) + (tid >> 6);
    if (laneId == 0u && waveIdx < WAVE_LOG_CAPACITY) {
      WaveRecord rec;
      rec.start = rtBegin;
      rec.end = __builtin_amdgcn_s_memrealtime();
      rec.iterations = nIter; rec.leafPhases = nLeafPhase; rec.laneIters = laneIters;
      rec.cyclesFetch = tFetch; rec.cyclesNode = tNode; rec.cyclesLeaf = tLeaf; rec.cyclesPop = tPop;
      rec.cyclesTotal = (unsigned long long)__builtin_readcyclecounter() - tBegin;
      rec.rays = v[0]; rec.nodes = v[1]; rec.leaves = v[2]; rec.prims = v[3]; rec.inner = v[4]; rec.hits = v[5]; rec.spills = v[6];
      rec.lastGrab = rtLastGrab ? rtLastGrab : rtBegin;
      rec.maxRaySteps = maxRaySteps;
      rec.valid = 1ull;
      P.counters[waveIdx] = rec;
    }
  }
}

template <typename Leaf, bool ROBUST, bool OCCLUDED, bool COUNT, bool VEC>
// The instrumented twin carries ~25 more live registers (counters, time stamps).  It is compiled for 2 waves per SIMD (<= 256
// VGPRs, no scratch) because that is the occupancy a batch alone on the chip runs at anyway (two workgroups per CU), so the
// twin's timeline stays close to the plain kernel's; bounded at 3 waves it spills 76-180 bytes per lane.
// Correctness does not depend on this: round 1 blamed wrong records of two uncommitted working-tree builds on that scratch;
// round 2 re-ran the twins of commits f962206 and af9d1b8 (3 waves, 112-136 B scratch) and of this file with
// -D'TRACE_COUNT_MIN_WAVES(L)=L::MIN_WAVES' and '=4' (profiles/r02_twin_scratch_check.txt): byte-identical to the plain kernel in
// every run, full GPU suite green.  The sources of the two failing builds were never committed; their failing rays all sat in
// the last 15 % of a work queue and the hit counters exceeded the stored hits (rays traced twice, others never), which points at
// the work hand-out those experimental builds carried (guided hand-out), not at register spills.
__global__ __launch_bounds__(TRACE_BLOCK, COUNT ? TRACE_COUNT_MIN_WAVES(Leaf) : Leaf::MIN_WAVES) void trace_kernel(LaunchParams P)
{
  __shared__ uint2 ldsStack[TRACE_LDS_STACK + 1][TRACE_BLOCK]; // + one scratch row for the branch-free pushes
  __shared__ __attribute__((aligned(16))) float octX[TRACE_BLOCK / 64][OCT_ROWS][OCT_WORDS]; // octet node step: per-wave exchange rows
  Leaf::prepare();
  trace_body<Leaf, ROBUST, OCCLUDED, COUNT, VEC>(P, ldsStack, octX[threadIdx.x >> 6]);
}

template <typename Leaf, bool ROBUST, bool OCCLUDED, bool COUNT>
inline hipError_t launch_vec(const LaunchParams& p, hipStream_t stream)
{
  const bool vec = (p.stride % 16 == 0) && (((uintptr_t)p.rays) % 16 == 0);
  // persistent grid = what is resident at once for THIS instantiation (its VGPR budget decides), capped by the
  // host's bound (which sized the spill area) and by the work available
  static int occVec = 0, occGen = 0;
  int& occ = vec ? occVec : occGen;
  if (occ == 0) {
    hipError_t e = vec ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, trace_kernel<Leaf, ROBUST, OCCLUDED, COUNT, true>, TRACE_BLOCK, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, trace_kernel<Leaf, ROBUST, OCCLUDED, COUNT, false>, TRACE_BLOCK, 0);
    if (e != hipSuccess || occ <= 0) occ = 1;
  }
  uint32_t blocks = (p.blocksPerCU ? std::min<uint32_t>(p.blocksPerCU, (uint32_t)occ) : (uint32_t)occ) * p.numCUs;
  if (blocks > p.gridBlocks) blocks = p.gridBlocks;
  if (vec) hipLaunchKernelGGL((trace_kernel<Leaf, ROBUST, OCCLUDED, COUNT, true>), dim3(blocks), dim3(TRACE_BLOCK), 0, stream, p);
  else hipLaunchKernelGGL((trace_kernel<Leaf, ROBUST, OCCLUDED, COUNT, false>), dim3(blocks), dim3(TRACE_BLOCK), 0, stream, p);
  return hipGetLastError();
}

template <typename Leaf, bool ROBUST>
inline hipError_t launch_leaf(const LaunchParams& p, hipStream_t stream)
{
  const bool cnt = p.counters != nullptr;
  if (p.occluded) return cnt ? launch_vec<Leaf, ROBUST, true, true>(p, stream) : launch_vec<Leaf, ROBUST, true, false>(p, stream);
  return cnt ? launch_vec<Leaf, ROBUST, false, true>(p, stream) : launch_vec<Leaf, ROBUST, false, false>(p, stream);
}

} // namespace dev
} // namespace rtamd
