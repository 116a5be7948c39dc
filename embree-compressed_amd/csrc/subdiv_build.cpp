#include "subdiv_build.h"

#include "bvh8_builder.h"
#include "cbvh_encode.h"
#include "subdiv_tess.h"

#include <atomic>
#include <chrono>
#include <functional>
#include <thread>

namespace rtamd {

// Host-side parallel loop for the commit-time builders: contiguous chunks handed out dynamically.
unsigned host_threads(const Device* dev)
{
  unsigned n = dev->numThreads > 0 ? (unsigned)dev->numThreads : std::thread::hardware_concurrency();
  return n ? std::min(n, 64u) : 1u;
}
void parallel_for_range(size_t count, unsigned threads, const std::function<void(size_t, size_t)>& body)
{
  if (count == 0) return;
  if (threads <= 1 || count < 2) { body(0, count); return; }
  const size_t chunk = std::max<size_t>(1, count / (threads * 8));
  std::atomic<size_t> next{0};
  std::exception_ptr err;
  std::mutex errMutex;
  auto work = [&]() {
    for (;;) {
      const size_t b = next.fetch_add(chunk);
      if (b >= count) return;
      try {
        body(b, std::min(count, b + chunk));
      } catch (...) {
        std::lock_guard<std::mutex> g(errMutex);
        if (!err) err = std::current_exception();
        next.store(count);
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
  if (err) std::rethrow_exception(err);
}

namespace {

// eager path: every 3x3-vertex cell of every patch grid becomes one GridCell leaf
// (reference: BVHNSubdivPatch1EagerBuilderSAH::createEager bvh_builder_subdiv.cpp:74-93 -> GridSOA, grid_soa.cpp:23-82,112-157)
void build_eager(Scene* s, const std::vector<PatchGrid>& grids, Accel& A)
{
  A.kind = ACCEL_GRIDSOA;
  A.blobStride = sizeof(GridCell);
  std::vector<BuildPrim> bp;
  size_t ncells = 0;
  for (const PatchGrid& pg : grids) {
    if (pg.n < 2) RT_THROW(RTC_ERROR_INVALID_OPERATION, "eager subdivision accel needs subdivision level >= 1");
    ncells += (size_t)(pg.n / 2) * (pg.n / 2);
  }
  if (ncells >= 0x7FFFFFFFull) RT_THROW(RTC_ERROR_OUT_OF_MEMORY, "too many grid cells");
  A.blobs.resize(ncells * sizeof(GridCell));
  bp.reserve(ncells);
  GridCell* cells = (GridCell*)A.blobs.data();
  size_t c = 0;
  for (const PatchGrid& pg : grids) {
    const unsigned w = pg.n + 1;
    const float fn = (float)pg.n;
    for (unsigned cy = 0; cy + 2 <= pg.n; cy += 2)
      for (unsigned cx = 0; cx + 2 <= pg.n; cx += 2) {
        GridCell& g = cells[c];
        BuildPrim p;
        for (unsigned r = 0; r < 3; r++)
          for (unsigned q = 0; q < 3; q++) {
            const size_t k = (size_t)(cy + r) * w + (cx + q);
            g.px[r * 3 + q] = pg.x[k]; g.py[r * 3 + q] = pg.y[k]; g.pz[r * 3 + q] = pg.z[k];
            // UV encoding, grid_soa.cpp:48-52: (int) clamp(u * (0x10000/8), 0, 0xFFFF), v in the high half
            const float u = pg.u0 + (float)(cx + q) / fn, v = pg.v0 + (float)(cy + r) / fn;
            const int iu = (int)fminf(fmaxf(u * (0x10000 / 8.0f), 0.0f), (float)0xFFFF);
            const int iv = (int)fminf(fmaxf(v * (0x10000 / 8.0f), 0.0f), (float)0xFFFF);
            g.uv[r * 3 + q] = ((uint32_t)iv << 16) | (uint32_t)iu;
            p.box.extend(V3(pg.x[k], pg.y[k], pg.z[k]));
          }
        g.geomID = pg.geomID;
        g.primID = pg.primID;
        g.pad[0] = g.pad[1] = 0;
        p.id = (uint32_t)c;
        bp.push_back(p);
        s->bounds.extend(p.box);
        c++;
      }
  }
  // the builder permutes `bp`; leaves address cells by their original index, so no reordering of `cells` is needed
  auto makeLeaf = [&](const BuildPrim* prims, size_t begin, size_t end) -> uint32_t { return REF_LEAF | prims[begin].id; };
  BuildSettings cfg; // one primitive per leaf, bvh_builder_subdiv.cpp:845-851
  cfg.blockSize = 1; cfg.minLeaf = 1; cfg.maxLeaf = 1;
  cfg.threads = host_threads(s->device);
  BuildResult r = build_bvh8(bp, cfg, makeLeaf);
  A.nodes = std::move(r.nodes);
  A.root = r.root;
  A.maxDepth = r.maxDepth;
  A.leafCount = r.leafCount;
  A.blobOffsets.assign(1, (uint32_t)ncells); // [0] = number of blobs (stats)
}

// fork path: every (2^C+1)^2 sub-grid becomes one cBVH blob
// (reference: BVHNSubdivPatch1OrientedBuilderSAH::createOriented bvh_builder_subdiv.cpp:708-733)
void build_cbvh(Scene* s, const std::vector<PatchGrid>& grids, Accel& A, CbvhMode mode)
{
  const unsigned C = s->compressionLevel;
  if (C < 1 || C > 5) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "compression level must be in 1..5 (local stack of compressed.h:510-512)");
  A.kind = mode == CBVH_BOX ? ACCEL_CBVH_BOX : (mode == CBVH_LEAF ? ACCEL_CBVH_LEAF : (mode == CBVH_GRID ? ACCEL_CBVH_GRID : ACCEL_CBVH_FULL));
  const unsigned sub = 1u << C;
  const size_t stride = cbvh_blob_bytes(C, mode);
  A.blobStride = (uint32_t)stride;
  size_t nblobs = 0;
  for (const PatchGrid& pg : grids) {
    if (pg.n < sub) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "compression level exceeds subdivision level");
    nblobs += (size_t)(pg.n / sub) * (pg.n / sub);
  }
  if (nblobs >= 0x7FFFFFFFull || nblobs * stride > ((size_t)1 << 40)) RT_THROW(RTC_ERROR_OUT_OF_MEMORY, "too many cBVH leaves");
  A.blobs.resize(nblobs * stride);
  std::vector<BuildPrim> bp(nblobs);
  // the blobs are independent of each other: encode them on all host threads (rtcCommitScene is internally parallel in
  // the reference too, scene.cpp:727-786); blob index = position in the (grid, y, x) order, as before
  std::vector<size_t> firstBlob(grids.size() + 1, 0);
  for (size_t g = 0; g < grids.size(); g++) firstBlob[g + 1] = firstBlob[g] + (size_t)(grids[g].n / sub) * (grids[g].n / sub);
  parallel_for_range(grids.size(), host_threads(s->device), [&](size_t g0, size_t g1) {
    for (size_t g = g0; g < g1; g++) {
      const PatchGrid& pg = grids[g];
      size_t b = firstBlob[g];
      for (unsigned y = 0; y < pg.n; y += sub)
        for (unsigned x = 0; x < pg.n; x += sub) {
          Box3 bounds;
          cbvh_encode(pg, x, x + sub, y, y + sub, C, mode, A.blobs.data() + b * stride, bounds);
          bp[b].box = bounds;
          bp[b].id = (uint32_t)b;
          b++;
        }
    }
  });
  for (size_t b = 0; b < nblobs; b++) s->bounds.extend(bp[b].box);
  auto makeLeaf = [&](const BuildPrim* prims, size_t begin, size_t end) -> uint32_t { return REF_LEAF | prims[begin].id; };
  BuildSettings cfg;
  cfg.blockSize = 1; cfg.minLeaf = 1; cfg.maxLeaf = 1;
  cfg.threads = host_threads(s->device);
  BuildResult r = build_bvh8(bp, cfg, makeLeaf);
  A.nodes = std::move(r.nodes);
  A.root = r.root;
  A.maxDepth = r.maxDepth;
  A.leafCount = r.leafCount;
  A.blobOffsets.assign(1, (uint32_t)nblobs);
}

} // namespace

void build_subdiv_accel(Scene* s)
{
  Accel& A = s->subdivAccel;
  A.clear();
  A.robust = 1; // all subdivision intersectors traverse robustly (bvh_intersector1_bvh4.cpp:60-78)
  bool any = false;
  for (Geometry* g : s->geometries)
    if (g && g->enabled && g->type == RTC_GEOMETRY_TYPE_SUBDIVISION) any = true;
  if (!any) return;

  // accel selection, scene.cpp:491-513
  const std::string& name = s->device->subdiv_accel;
  int mode;
  if (name == "default" || name == "bvh4.grid.eager" || name == "bvh4.subdivpatch1eager") mode = -1;
  else if (name == "bvh4.compressed.box") mode = CBVH_BOX;
  else if (name == "bvh4.compressed.leaf") mode = CBVH_LEAF;
  else if (name == "bvh4.compressed.grid") mode = CBVH_GRID;
  else if (name == "bvh4.compressed.full") mode = CBVH_FULL; // scene.cpp:510
  else RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown subdiv accel " + name);

  std::vector<PatchGrid> grids;
  const auto tBuild0 = std::chrono::steady_clock::now();
  for (unsigned gid = 0; gid < s->geometries.size(); gid++) {
    Geometry* g = s->geometries[gid];
    if (!g || !g->enabled || g->type != RTC_GEOMETRY_TYPE_SUBDIVISION) continue;
    if (g->timeSteps != 1) RT_THROW(RTC_ERROR_INVALID_OPERATION, "motion blur geometry is not supported by the device path");
    tessellate_subdiv(g, gid, s->subdivisionLevel, grids, host_threads(s->device));
  }
  if (grids.empty()) return;
  const auto tBuild1 = std::chrono::steady_clock::now();
  if (mode < 0) build_eager(s, grids, A);
  else build_cbvh(s, grids, A, (CbvhMode)mode);
  if (s->device->verbose >= 2)
    fprintf(stderr, "embree3-amd: subdivision commit: tessellation %.2f s, leaf encoding + BVH8 %.2f s (%zu patch grids)\n",
            std::chrono::duration<double>(tBuild1 - tBuild0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild1).count(), grids.size());
  s->debugGrids.clear();
  if (s->device->keepGrids) {
    for (const PatchGrid& pg : grids) {
      const uint32_t hdr[3] = {pg.geomID, pg.primID, pg.n};
      const uint8_t* h = (const uint8_t*)hdr;
      s->debugGrids.insert(s->debugGrids.end(), h, h + 12);
      for (const std::vector<float>* a : {&pg.x, &pg.y, &pg.z}) {
        const uint8_t* p = (const uint8_t*)a->data();
        s->debugGrids.insert(s->debugGrids.end(), p, p + a->size() * 4);
      }
    }
  }
}

} // namespace rtamd
