// Alternative traversal skeleton: a POOL of rays per wavefront, phases that run on full wavefronts.
//
// The lane-per-ray loop (trace_loop.hip.h) keeps one ray in every lane and lets the lanes of a wave be in different
// phases: measured on MI355X the fetch code then runs with ~18 of 64 lanes, the node step with ~29, the leaf code with
// ~16 - and with several batches in flight the VALU pipes are saturated by those partly filled instructions.
// Here a wave owns POOL_R ray slots in LDS (all per-ray state: ray, node-test constants, current node, stack) and three
// slot queues (free, at-node, at-leaf).  Every iteration it picks ONE phase and runs it for up to 64 rays taken from the
// matching queue, so the phase code executes on full wavefronts as long as the pool is full:
//   fetch: 64 free slots are loaded with the next 64 rays of the wave's chunk            -> at-node
//   node : 64 rays do one BVH8 node step                                                  -> at-node / at-leaf / pop
//   leaf : 64 rays intersect their leaf; a hit is written to the ray record at once       -> pop
//   pop  : inline after node / leaf: next stack entry, or the ray is finished and its slot is free again
// Each ray still performs exactly the reference's depth-first sequence (same node test, same child order, same leaf
// code, same stack discipline as trace_loop.hip.h), so results are bit-identical to the lane-per-ray kernel (forced with
// RTAMD_KERNEL=pool it passes the whole GPU suite; tests/test_gpu_properties.py compares the two on 4 M rays).
//
// USED FOR LARGE BATCHES ONLY (>= 2.5 M rays per launch, Device::tunePoolMinRays).  Measured on MI355X (bomberman cbvh.leaf, 1 M random rays): 0.188 ms alone (lane kernel
// 0.176), 8.6 Grays/s with four batches in flight (10.3).  The phases do run fuller (lane utilisation 0.30 vs 0.27) but
// the wave instructions per launch hardly change (44.0 M vs 46.0 M): a wave's instruction count is set by the number
// of loop iterations it needs, and that is the step count of its DEEPEST ray (41 iterations for 488 rays per wave,
// ~22 would do if every phase were full), not the amount of work.  Where the drain is amortised the fuller phases do
// pay: 0.401 vs 0.423 ms at 4 M rays, 1.213 vs 1.365 ms at 16 M (67.7 vs 78.5 us per million rays).  The next step is
// fewer, larger pools (rays of a whole workgroup or CU behind one set of queues), so that a deep ray
// keeps one wavefront iterating instead of every wavefront one.
#pragma once
#include "trace_common.hip.h"
#include "trace_loop.hip.h"

namespace rtamd {
namespace dev {

#ifndef TRACE_POOL_SLOTS_PER_WAVE
#define TRACE_POOL_SLOTS_PER_WAVE 128
#endif
#ifndef TRACE_POOL_STACK_ENTRIES
#define TRACE_POOL_STACK_ENTRIES 8
#endif
// geometry sweep on MI355X (cbvh.leaf, 4 M rays alone / in flight): 128 slots x 8 entries 0.402 ms / 12.2 Grays/s;
// 112 x 6: 0.414 / 11.6; 96 x 6: 0.435 / 11.1; 160 x 8: 0.441 / 10.9; 128 x 12: 0.461 / 10.2
static constexpr int POOL_R = TRACE_POOL_SLOTS_PER_WAVE;     // ray slots per wavefront
static constexpr int POOL_STACK = TRACE_POOL_STACK_ENTRIES;  // stack entries per slot kept in LDS (the rest goes to the HBM overflow area)
static_assert(POOL_R == TRACE_POOL_SLOTS && POOL_STACK == TRACE_POOL_STACK, "trace.h sizes the overflow area from these");
static constexpr int POOL_BLOCK = 128; // 2 wavefronts per workgroup: 39 KB of LDS each, four workgroups per CU
static constexpr int POOL_WAVES = POOL_BLOCK / 64;
// per-slot words: ray (8) + TravRay (7) + travFar + cur + sp + ray index
enum { PW_OX = 0, PW_OY, PW_OZ, PW_TNEAR, PW_DX, PW_DY, PW_DZ, PW_TFAR, PW_TR0, PW_TRAVFAR = PW_TR0 + 7, PW_CUR, PW_SP, PW_IDX, PW_WORDS };

struct PoolLds
{
  float state[POOL_WAVES][PW_WORDS][POOL_R];
  uint2 stack[POOL_WAVES][POOL_STACK][POOL_R];
  uint32_t qFree[POOL_WAVES][POOL_R], qNode[POOL_WAVES][POOL_R], qLeaf[POOL_WAVES][POOL_R];
};

template <typename Leaf, bool ROBUST, bool OCCLUDED, bool COUNT, bool VEC>
__global__ __launch_bounds__(POOL_BLOCK, 2) void trace_pool_kernel(LaunchParams P)
{
  __shared__ PoolLds L;
  Leaf::prepare();
  const uint32_t tid = threadIdx.x, wave = tid >> 6, laneId = lane_rank(~0ull);
  float(*S)[POOL_R] = L.state[wave];
  uint2(*K)[POOL_R] = L.stack[wave];
  uint32_t* qFree = L.qFree[wave];
  uint32_t* qNode = L.qNode[wave];
  uint32_t* qLeaf = L.qLeaf[wave];
  const QNode8* __restrict__ nodes = P.accel.nodes;
  uint32_t* __restrict__ queues = P.queues;
  // overflow area: one column of P.spillDepth entries per slot of every wave of the grid
  uint2* __restrict__ spillBase = (uint2*)P.spill + (size_t)(blockIdx.x * POOL_WAVES + wave) * POOL_R * P.spillDepth;

  for (uint32_t s = laneId; s < (uint32_t)POOL_R; s += 64u) qFree[s] = s;
  uint32_t nFree = POOL_R, nNode = 0, nLeaf = 0; // wave-uniform queue fill levels

  const uint32_t perQ = (P.count + (uint32_t)TRACE_QUEUES - 1u) / (uint32_t)TRACE_QUEUES;
  uint32_t qCur = (blockIdx.x * POOL_WAVES + wave) & (uint32_t)(TRACE_QUEUES - 1);
  uint32_t poolNext = 0, poolEnd = 0;
  bool exhausted = P.accel.root == REF_EMPTY;

  WorkCounters wc;
  unsigned long long nIter = 0, nLeafPhase = 0, laneIters = 0, rtBegin = 0, rtLastGrab = 0, tBegin = 0;
  if (COUNT) { tBegin = __builtin_readcyclecounter(); rtBegin = __builtin_amdgcn_s_memrealtime(); }

  auto push_q = [&](uint32_t* q, uint32_t& n, bool pred, uint32_t slot) {
    const uint64_t m = __ballot(pred);
    if (pred) q[n + lane_rank(m)] = slot;
    n += (uint32_t)__popcll(m);
  };
  auto spill_of = [&](uint32_t slot) -> uint2* { return spillBase + (size_t)slot * P.spillDepth; };
  auto stack_write = [&](uint32_t slot, uint32_t pos, uint32_t ref, uint32_t dist) {
    if (pos < (uint32_t)POOL_STACK) K[pos][slot] = make_uint2(ref, dist);
    else {
      if (pos - POOL_STACK < P.spillDepth) spill_of(slot)[pos - POOL_STACK] = make_uint2(ref, dist);
      else __hip_atomic_store(P.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // entry dropped: the host raises an error
      if (COUNT) wc.spills++;
    }
  };
  // pop until an entry survives the distance cull; false: the stack is empty, the ray is finished
  auto pop_next = [&](uint32_t slot, uint32_t& sp, uint32_t& cur, float tfar) -> bool {
    for (;;) {
      if (sp == 0) return false;
      sp--;
      uint2 e;
      if (sp < (uint32_t)POOL_STACK) e = K[sp][slot];
      else if (sp - POOL_STACK < P.spillDepth) {
        const uint32_t* w = (const uint32_t*)(spill_of(slot) + (sp - POOL_STACK));
        e = make_uint2(__builtin_nontemporal_load(w), __builtin_nontemporal_load(w + 1));
      } else e = make_uint2(REF_EMPTY, 0x7f800000u);
      if (e.x == REF_EMPTY) continue;
      if (!OCCLUDED && __uint_as_float(e.y) > tfar) continue; // bvh_intersector1.cpp:86
      cur = e.x;
      return true;
    }
  };
  // after a node step or a leaf: route the ray to its next phase
  auto route = [&](bool have, bool alive, uint32_t slot, uint32_t cur) {
    push_q(qNode, nNode, have && alive && !(cur & REF_LEAF), slot);
    push_q(qLeaf, nLeaf, have && alive && (cur & REF_LEAF), slot);
    push_q(qFree, nFree, have && !alive, slot);
  };

  for (;;) {
    if (COUNT) { nIter++; laneIters += nNode + nLeaf; }
    // ---- phase choice (wave-uniform) --------------------------------------------------------------------------
    enum { PH_FETCH, PH_NODE, PH_LEAF, PH_EXIT };
    int phase;
    const bool canFetch = !exhausted && nFree > 0;
    if (canFetch && nFree >= 64u) phase = PH_FETCH;
    else if (nNode >= 64u) phase = PH_NODE;
    else if (nLeaf >= 64u) phase = PH_LEAF;
    else if (canFetch) phase = PH_FETCH;
    else if (nNode >= nLeaf && nNode > 0u) phase = PH_NODE;
    else if (nLeaf > 0u) phase = PH_LEAF;
    else if (exhausted) phase = PH_EXIT;
    else phase = PH_FETCH;
    if (phase == PH_EXIT) break;

    if (phase == PH_FETCH) {
      if (poolNext == poolEnd) { // take a new chunk (same queues as trace_loop.hip.h)
        for (;;) {
          const uint32_t qLo = min(qCur * perQ, P.count);
          const uint32_t qHi = min(qLo + perQ, P.count);
          uint32_t base = 0;
          if (laneId == 0u) base = atomicAdd(&queues[qCur * QUEUE_STRIDE], P.rayChunk);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base < qHi - qLo) {
            if (COUNT) rtLastGrab = __builtin_amdgcn_s_memrealtime();
            poolNext = qLo + base;
            poolEnd = min(poolNext + P.rayChunk, qHi);
            break;
          }
          const uint32_t myLo = min(laneId * perQ, P.count), myHi = min(myLo + perQ, P.count);
          const uint32_t head = __hip_atomic_load(&queues[laneId * QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t live = __ballot(laneId < (uint32_t)TRACE_QUEUES && head < myHi - myLo);
          if (live == 0ull) { exhausted = true; break; }
          const uint64_t rot = (live >> qCur) | (qCur ? (live << (64u - qCur)) : 0ull);
          qCur = (qCur + (uint32_t)__builtin_ctzll(rot)) & (uint32_t)(TRACE_QUEUES - 1);
        }
      }
      if (poolNext != poolEnd) {
        const uint32_t n = min(min(64u, nFree), poolEnd - poolNext);
        const bool have = laneId < n;
        uint32_t slot = 0;
        if (have) slot = qFree[nFree - 1u - laneId];
        nFree -= n;
        bool ok = false;
        if (have) {
          const uint32_t rayIdx = poolNext + laneId;
          RayState r;
          load_ray<VEC>((const char*)P.rays + (size_t)rayIdx * P.stride, r);
          ok = r.tnear <= r.tfar; // stream front-end: rays with tnear > tfar are skipped
          if (OCCLUDED) ok = ok && !(r.tfar < 0.0f);
          if (ok) {
            if (COUNT) wc.rays++;
            TravRay<ROBUST> tr;
            tr.init(r);
            S[PW_OX][slot] = r.ox; S[PW_OY][slot] = r.oy; S[PW_OZ][slot] = r.oz; S[PW_TNEAR][slot] = r.tnear;
            S[PW_DX][slot] = r.dx; S[PW_DY][slot] = r.dy; S[PW_DZ][slot] = r.dz; S[PW_TFAR][slot] = r.tfar;
            tr.store(&S[PW_TR0][slot], POOL_R);
            S[PW_TRAVFAR][slot] = fmaxf(r.tfar, 0.0f);
            S[PW_CUR][slot] = __uint_as_float(P.accel.root);
            S[PW_SP][slot] = __uint_as_float(0u);
            S[PW_IDX][slot] = __uint_as_float(rayIdx);
          }
        }
        poolNext += n;
        route(have, ok, slot, P.accel.root);
      }
      continue;
    }

    if (phase == PH_NODE) {
      const uint32_t n = min(64u, nNode);
      const bool have = laneId < n;
      uint32_t slot = 0;
      if (have) slot = qNode[nNode - 1u - laneId];
      nNode -= n;
      bool alive = false;
      uint32_t cur = REF_EMPTY;
      if (have) {
        if (COUNT) wc.nodes++;
        TravRay<ROBUST> tr;
        tr.load(&S[PW_TR0][slot], POOL_R);
        const float travFar = S[PW_TRAVFAR][slot];
        cur = __float_as_uint(S[PW_CUR][slot]);
        const uint32_t spw = __float_as_uint(S[PW_SP][slot]); // bit 31: the ray already has a hit (counters only)
        uint32_t sp = spw & 0x7FFFFFFFu;
        const uint4* np = (const uint4*)(nodes + cur);
        const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4], n5 = np[5];
        const float ox = __uint_as_float(n0.x), oy = __uint_as_float(n0.y), oz = __uint_as_float(n0.z);
        const float sx = __uint_as_float((n0.w & 0xffu) << 23);
        const float sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23);
        const float sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
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
          dist[k] = h ? __float_as_uint(tN) : (0x7FFFFFF8u + (uint32_t)k); // distinct sentinels above every distance and below 2^31 (trace_loop.hip.h)
          mask |= h ? (1u << k) : 0u;
        }
        const int nhit = __popc(mask);
        if (nhit == 0) {
          alive = pop_next(slot, sp, cur, OCCLUDED ? 0.f : S[PW_TFAR][slot]);
        } else if (nhit == 1) {
          const int k = __ffs(mask) - 1;
          uint32_t c = cref[0];
#pragma unroll
          for (int j = 1; j < 8; j++) c = (k == j) ? cref[j] : c;
          cur = c;
          alive = true;
        } else {
          // same ordering rules as trace_loop.hip.h (bvh_traverser1.h:549-666, stack_item.h:39-80)
          uint32_t rank[8];
          if (OCCLUDED) {
#pragma unroll
            for (int k = 0; k < 8; k++) rank[k] = (uint32_t)__popc(mask >> (k + 1));
          } else {
#pragma unroll
            for (int k = 0; k < 8; k++) rank[k] = (uint32_t)(7 - k);
#pragma unroll
            for (int a = 0; a < 8; a++) {
#pragma unroll
              for (int b = a + 1; b < 8; b++) {
                const uint32_t aFirst = (dist[a] - dist[b]) >> 31; // [dist[a] < dist[b]], sign of the difference (trace_loop.hip.h)
                rank[b] += aFirst;
                rank[a] -= aFirst;
              }
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
          const uint32_t top = sp + (uint32_t)nhit - 1u;
          uint32_t next = REF_EMPTY;
#pragma unroll
          for (int k = 0; k < 8; k++) {
            if (mask & (1u << k)) {
              if (rank[k] == 0u) next = cref[k];
              else stack_write(slot, top - rank[k], cref[k], dist[k]);
            }
          }
          sp = top;
          cur = next;
          alive = true;
        }
        S[PW_CUR][slot] = __uint_as_float(cur);
        S[PW_SP][slot] = __uint_as_float(sp | (spw & 0x80000000u));
        if (COUNT && !OCCLUDED && !alive && (spw >> 31)) wc.hits++;
      }
      route(have, alive, slot, cur);
      continue;
    }

    // ---- leaf phase ------------------------------------------------------------------------------------------------
    {
      if (COUNT) nLeafPhase++;
      const uint32_t n = min(64u, nLeaf);
      const bool have = laneId < n;
      uint32_t slot = 0;
      if (have) slot = qLeaf[nLeaf - 1u - laneId];
      nLeaf -= n;
      bool alive = false;
      uint32_t cur = REF_EMPTY;
      if (have) {
        if (COUNT) wc.leaves++;
        RayState r;
        r.ox = S[PW_OX][slot]; r.oy = S[PW_OY][slot]; r.oz = S[PW_OZ][slot]; r.tnear = S[PW_TNEAR][slot];
        r.dx = S[PW_DX][slot]; r.dy = S[PW_DY][slot]; r.dz = S[PW_DZ][slot]; r.tfar = S[PW_TFAR][slot];
        r.hit = 0u;
        cur = __float_as_uint(S[PW_CUR][slot]);
        const uint32_t spw = __float_as_uint(S[PW_SP][slot]);
        uint32_t sp = spw & 0x7FFFFFFFu, hitFlag = spw & 0x80000000u;
        const uint32_t rayIdx = __float_as_uint(S[PW_IDX][slot]);
        char* rp = (char*)P.rays + (size_t)rayIdx * P.stride;
        const bool done = Leaf::template intersect<OCCLUDED, COUNT>(P, cur, r, wc, rayIdx);
        if (OCCLUDED) {
          if (done) { // any hit found: the ray is occluded (bvh_intersector1.cpp:198-201)
            ((float*)rp)[8] = -RT_INF;
            if (COUNT) wc.hits++;
            sp = 0;
          }
        } else if (r.hit) { // a closer hit: it goes to the ray record at once (a later, closer one overwrites it)
          store_hit<VEC>(rp, r, P.instID);
          S[PW_TFAR][slot] = r.tfar;
          S[PW_TRAVFAR][slot] = r.tfar; // tray.tfar = ray.tfar (bvh_intersector1.cpp:117)
          hitFlag = 0x80000000u;
        }
        alive = pop_next(slot, sp, cur, r.tfar);
        S[PW_CUR][slot] = __uint_as_float(cur);
        S[PW_SP][slot] = __uint_as_float(sp | hitFlag);
        if (COUNT && !OCCLUDED && !alive && hitFlag) wc.hits++;
      }
      route(have, alive, slot, cur);
    }
  }

  if (COUNT) {
    unsigned long long v[7] = {wc.rays, wc.nodes, wc.leaves, wc.prims, wc.inner, wc.hits, wc.spills};
#pragma unroll
    for (int i = 0; i < 7; i++)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
    const uint32_t waveIdx = blockIdx.x * POOL_WAVES + wave;
    if (laneId == 0u && waveIdx < WAVE_LOG_CAPACITY) {
      WaveRecord rec;
      rec.start = rtBegin;
      rec.end = __builtin_amdgcn_s_memrealtime();
      rec.iterations = nIter; rec.leafPhases = nLeafPhase; rec.laneIters = laneIters;
      rec.cyclesFetch = rec.cyclesNode = rec.cyclesLeaf = rec.cyclesPop = 0;
      rec.cyclesTotal = (unsigned long long)__builtin_readcyclecounter() - tBegin;
      rec.rays = v[0]; rec.nodes = v[1]; rec.leaves = v[2]; rec.prims = v[3]; rec.inner = v[4]; rec.hits = v[5]; rec.spills = v[6];
      rec.lastGrab = rtLastGrab ? rtLastGrab : rtBegin;
      rec.maxRaySteps = 0;
      rec.valid = 1ull;
      P.counters[waveIdx] = rec;
    }
  }
}

template <typename Leaf, bool ROBUST, bool OCCLUDED, bool COUNT>
inline hipError_t launch_vec_pool(const LaunchParams& p, hipStream_t stream)
{
  const bool vec = (p.stride % 16 == 0) && (((uintptr_t)p.rays) % 16 == 0);
  static int occVec = 0, occGen = 0;
  int& occ = vec ? occVec : occGen;
  if (occ == 0) {
    hipError_t e = vec ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, trace_pool_kernel<Leaf, ROBUST, OCCLUDED, COUNT, true>, POOL_BLOCK, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, trace_pool_kernel<Leaf, ROBUST, OCCLUDED, COUNT, false>, POOL_BLOCK, 0);
    if (e != hipSuccess || occ <= 0) occ = 1;
  }
  // the pool's LDS footprint decides the resident set; the host sized the overflow area for p.gridBlocks workgroups
  uint32_t blocks = (uint32_t)occ * p.numCUs;
  if (blocks > p.gridBlocks) blocks = p.gridBlocks;
  const uint32_t need = (p.count + 63u) / 64u / POOL_WAVES + 1u; // no more waves than 64-ray groups
  if (blocks > need) blocks = need;
  if (vec) hipLaunchKernelGGL((trace_pool_kernel<Leaf, ROBUST, OCCLUDED, COUNT, true>), dim3(blocks), dim3(POOL_BLOCK), 0, stream, p);
  else hipLaunchKernelGGL((trace_pool_kernel<Leaf, ROBUST, OCCLUDED, COUNT, false>), dim3(blocks), dim3(POOL_BLOCK), 0, stream, p);
  return hipGetLastError();
}

template <typename Leaf, bool ROBUST>
inline hipError_t launch_leaf_pool(const LaunchParams& p, hipStream_t stream)
{
  const bool cnt = p.counters != nullptr;
  if (p.occluded) return cnt ? launch_vec_pool<Leaf, ROBUST, true, true>(p, stream) : launch_vec_pool<Leaf, ROBUST, true, false>(p, stream);
  return cnt ? launch_vec_pool<Leaf, ROBUST, false, true>(p, stream) : launch_vec_pool<Leaf, ROBUST, false, false>(p, stream);
}

} // namespace dev
} // namespace rtamd
