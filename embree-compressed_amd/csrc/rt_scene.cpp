// Geometry / Scene objects, accel selection at commit, upload to HBM.
#include <algorithm>

#include "bvh8_builder.h"
#include "rt_objects.h"
#include "rt_trace.h"
#include "subdiv_build.h"

namespace rtamd {

// ---- Geometry -------------------------------------------------------------------------------------------
Geometry::Geometry(Device* d, RTCGeometryType t) : device(d), type(t) { device->retain(); }

Geometry::~Geometry()
{
  for (auto& kv : views) kv.second.clear();
  device->release();
}

BufferView* Geometry::view(RTCBufferType t, unsigned slot)
{
  auto it = views.find({(int)t, slot});
  return it == views.end() ? nullptr : &it->second;
}

const BufferView* Geometry::view(RTCBufferType t, unsigned slot) const
{
  auto it = views.find({(int)t, slot});
  return it == views.end() ? nullptr : &it->second;
}

static size_t format_bytes(RTCFormat f)
{
  const unsigned fam = ((unsigned)f) >> 12, n = ((unsigned)f) & 0xfff;
  if (f == RTC_FORMAT_UNDEFINED) return 0;
  if (((unsigned)f & 0xf00) != 0) { // matrices 0x9RCc
    const unsigned r = (n >> 4) & 0xf, c = n & 0xf;
    return 4 * r * c;
  }
  static const size_t scalar[10] = {0, 1, 1, 2, 2, 4, 4, 8, 8, 4};
  return fam < 10 ? scalar[fam] * n : 0;
}

void Geometry::bind(RTCBufferType t, unsigned slot, RTCFormat f, Buffer* b, size_t off, size_t stride, size_t count)
{
  // argument checks follow rtcSetGeometryBuffer (rtcore.cpp:1236-1290) and TriangleMesh::setBuffer
  // (scene_triangle_mesh.cpp): 4-byte aligned offset/stride, matching formats per slot type.
  if ((off & 3) || (stride & 3)) RT_THROW(RTC_ERROR_INVALID_OPERATION, "buffer offset and stride must be 4-byte aligned");
  if (b && off + (count ? (count - 1) * stride + format_bytes(f) : 0) > b->bytes && !b->shared)
    RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "buffer range out of bounds");
  switch (type) {
  case RTC_GEOMETRY_TYPE_TRIANGLE:
    if (t == RTC_BUFFER_TYPE_VERTEX) {
      if (f != RTC_FORMAT_FLOAT3) RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid vertex buffer format");
    } else if (t == RTC_BUFFER_TYPE_INDEX) {
      if (slot != 0) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer slot");
      if (f != RTC_FORMAT_UINT3) RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid index buffer format");
    } else if (t != RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE)
      RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown buffer type");
    break;
  case RTC_GEOMETRY_TYPE_SUBDIVISION:
    if (t == RTC_BUFFER_TYPE_VERTEX && f != RTC_FORMAT_FLOAT3) RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid vertex buffer format");
    if ((t == RTC_BUFFER_TYPE_INDEX || t == RTC_BUFFER_TYPE_FACE) && f != RTC_FORMAT_UINT)
      RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid index/face buffer format");
    break;
  default: break;
  }
  if (t == RTC_BUFFER_TYPE_VERTEX && slot >= timeSteps) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid vertex buffer slot");
  views[{(int)t, slot}].set(b, f, off, stride, count);
  committed = false;
}

size_t Geometry::numTriangles() const
{
  const BufferView* v = view(RTC_BUFFER_TYPE_INDEX, 0);
  return v && v->valid() ? v->count : 0;
}

size_t Geometry::numVertices() const
{
  const BufferView* v = view(RTC_BUFFER_TYPE_VERTEX, 0);
  return v && v->valid() ? v->count : 0;
}

void Geometry::triangle(size_t i, unsigned idx[3]) const
{
  const unsigned* p = (const unsigned*)view(RTC_BUFFER_TYPE_INDEX, 0)->at(i);
  idx[0] = p[0]; idx[1] = p[1]; idx[2] = p[2];
}

V3 Geometry::vertex(size_t i) const
{
  const float* p = (const float*)view(RTC_BUFFER_TYPE_VERTEX, 0)->at(i);
  return V3(p[0], p[1], p[2]);
}

// TriangleMesh::valid (scene_triangle_mesh.h): indices in range and finite vertices
bool Geometry::validTriangle(size_t i) const
{
  unsigned idx[3];
  triangle(i, idx);
  const size_t nv = numVertices();
  if (idx[0] >= nv || idx[1] >= nv || idx[2] >= nv) return false;
  for (int k = 0; k < 3; k++) {
    V3 p = vertex(idx[k]);
    if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) return false;
    if (fabsf(p.x) > 1.844e18f || fabsf(p.y) > 1.844e18f || fabsf(p.z) > 1.844e18f) return false; // FLT_LARGE
  }
  return true;
}

// ---- Accel -------------------------------------------------------------------------------------------------
AccelDesc Accel::desc(size_t shard) const
{
  AccelDesc d;
  const DevCopy c = shard < dev.size() ? dev[shard] : DevCopy();
  d.nodes = (const QNode8*)c.dNodes;
  d.prims = (const TriRecord*)c.dPrims;
  d.blobs = (const uint8_t*)c.dBlobs;
  d.blobOffsets = (const uint32_t*)c.dBlobOffsets;
  d.root = root;
  d.kind = kind;
  d.robust = robust;
  d.blobStride = blobStride;
  return d;
}

size_t Accel::deviceBytes() const
{
  return nodes.size() * sizeof(QNode8) + prims.size() * sizeof(TriRecord) + blobs.size() + blobOffsets.size() * 4;
}

void Accel::freeDevice()
{
  for (size_t i = 0; i < dev.size(); i++) {
    DevCopy& c = dev[i];
    if (!(c.dNodes || c.dPrims || c.dBlobs || c.dBlobOffsets)) continue;
    hipSetDevice(devOrdinals[i]);
    if (c.dNodes) hipFree(c.dNodes);
    if (c.dPrims) hipFree(c.dPrims);
    if (c.dBlobs) hipFree(c.dBlobs);
    if (c.dBlobOffsets) hipFree(c.dBlobOffsets);
  }
  dev.clear();
  devOrdinals.clear();
}

void Accel::clear()
{
  nodes.clear();
  prims.clear();
  blobs.clear();
  blobOffsets.clear();
  root = REF_EMPTY;
  kind = ACCEL_NONE;
  maxDepth = 0;
  leafCount = 0;
}

static void* upload_array(hipStream_t stream, const void* src, size_t bytes)
{
  if (bytes == 0) return nullptr;
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, bytes + 64)); // slack: kernels may over-read one record at the array end
  HIP_CHECK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, stream));
  return d;
}

// The accel is replicated: one copy per shard of the device, uploaded on the shard's own stream.
void Accel::upload(Device* device)
{
  freeDevice();
  if (device->gpu < 0) return; // host-only device: keep the host mirror for inspection
  device->useDevice();
  dev.resize(device->shards.size());
  devOrdinals.resize(device->shards.size());
  for (size_t i = 0; i < device->shards.size(); i++) {
    Device::GpuShard& sh = *device->shards[i];
    sh.use();
    devOrdinals[i] = sh.ordinal;
    DevCopy& c = dev[i];
    c.dNodes = upload_array(sh.stream, nodes.data(), nodes.size() * sizeof(QNode8));
    c.dPrims = upload_array(sh.stream, prims.data(), prims.size() * sizeof(TriRecord));
    c.dBlobs = upload_array(sh.stream, blobs.data(), blobs.size());
    c.dBlobOffsets = upload_array(sh.stream, blobOffsets.data(), blobOffsets.size() * 4);
  }
  for (auto& sh : device->shards) {
    sh->use();
    HIP_CHECK(hipStreamSynchronize(sh->stream));
  }
  device->useDevice();
}

// ---- Scene ---------------------------------------------------------------------------------------------------
Scene::Scene(Device* d) : device(d) { device->retain(); }

Scene::~Scene()
{
  service_quiesce(device); // freeing device memory synchronises the device: do not wait for the resident service kernel's idle exit
  triAccel.freeDevice();
  subdivAccel.freeDevice();
  if (device->gpu >= 0) hipSetDevice(device->gpu);
  for (Geometry* g : geometries)
    if (g) g->release();
  device->release();
}

unsigned Scene::attach(Geometry* g)
{
  unsigned id = 0;
  while (id < geometries.size() && geometries[id]) id++; // lowest free ID, like the reference's IDPool (scene.cpp:575-590)
  attachByID(g, id);
  return id;
}

void Scene::attachByID(Geometry* g, unsigned id)
{
  if (id == RTC_INVALID_GEOMETRY_ID) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid geometry ID");
  if (id >= geometries.size()) geometries.resize(id + 1, nullptr);
  if (geometries[id]) RT_THROW(RTC_ERROR_INVALID_OPERATION, "geometry ID already in use");
  g->retain();
  geometries[id] = g;
  modified = true;
}

void Scene::detach(unsigned id)
{
  if (id >= geometries.size() || !geometries[id]) RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid geometry");
  geometries[id]->release();
  geometries[id] = nullptr;
  modified = true;
}

Geometry* Scene::get(unsigned id) const { return id < geometries.size() ? geometries[id] : nullptr; }

static void build_triangle_accel(Scene* s)
{
  Device* dev = s->device;
  Accel& A = s->triAccel;
  A.clear();

  // accel selection: scene.cpp:130-211.  Everything is served by the one BVH8 layout; the name only picks
  // the leaf arithmetic (Triangle4v/Pluecker/robust vs Triangle4/Moeller/fast).
  const std::string& name = dev->tri_accel;
  bool pluecker;
  if (name == "default") pluecker = s->isRobust();
  else if (name == "bvh8.triangle4v" || name == "bvh4.triangle4v") pluecker = true;
  else if (name == "bvh8.triangle4" || name == "bvh4.triangle4" || name == "qbvh8.triangle4") pluecker = false;
  else RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown triangle acceleration structure " + name);
  A.kind = pluecker ? ACCEL_TRI_PLUECKER : ACCEL_TRI_MOELLER;
  A.robust = pluecker ? 1 : 0;

  struct Src { unsigned geomID, primID; };
  std::vector<Src> src;
  std::vector<BuildPrim> bp;
  for (unsigned gid = 0; gid < s->geometries.size(); gid++) {
    Geometry* g = s->geometries[gid];
    if (!g || !g->enabled || g->type != RTC_GEOMETRY_TYPE_TRIANGLE) continue;
    if (g->timeSteps != 1) RT_THROW(RTC_ERROR_INVALID_OPERATION, "motion blur geometry is not supported by the device path");
    const size_t nt = g->numTriangles();
    for (size_t i = 0; i < nt; i++) {
      if (!g->validTriangle(i)) continue;
      unsigned idx[3];
      g->triangle(i, idx);
      BuildPrim p;
      p.box.extend(g->vertex(idx[0]));
      p.box.extend(g->vertex(idx[1]));
      p.box.extend(g->vertex(idx[2]));
      p.id = (uint32_t)src.size();
      src.push_back({gid, (unsigned)i});
      bp.push_back(p);
    }
  }
  if (bp.empty()) { A.kind = ACCEL_NONE; return; }
  if (bp.size() >= ((size_t)1 << TRI_START_BITS)) RT_THROW(RTC_ERROR_INVALID_OPERATION, "too many triangles for the 26-bit leaf reference");

  A.prims.reserve(bp.size());
  auto makeLeaf = [&](const BuildPrim* prims, size_t begin, size_t end) -> uint32_t {
    const uint32_t first = (uint32_t)A.prims.size();
    for (size_t i = begin; i < end; i++) {
      const Src& sr = src[prims[i].id];
      Geometry* g = s->geometries[sr.geomID];
      unsigned idx[3];
      g->triangle(sr.primID, idx);
      const V3 v0 = g->vertex(idx[0]), v1 = g->vertex(idx[1]), v2 = g->vertex(idx[2]);
      TriRecord t;
      memset(&t, 0, sizeof(t));
      t.ax = v0.x; t.ay = v0.y; t.az = v0.z;
      if (pluecker) {
        t.bx = v1.x; t.by = v1.y; t.bz = v1.z;
        t.cx = v2.x; t.cy = v2.y; t.cz = v2.z;
      } else { // TriangleM ctor: e1 = v0-v1, e2 = v2-v0 (triangle.h:52-53)
        t.bx = v0.x - v1.x; t.by = v0.y - v1.y; t.bz = v0.z - v1.z;
        t.cx = v2.x - v0.x; t.cy = v2.y - v0.y; t.cz = v2.z - v0.z;
      }
      t.geomID = sr.geomID;
      t.primID = sr.primID;
      A.prims.push_back(t);
    }
    return make_tri_leaf(first, (uint32_t)(end - begin));
  };
  BuildSettings cfg; // block 4, min leaf 4, max leaf 28 (bvh_builder_sah.cpp:651-658)
  cfg.threads = host_threads(s->device);
  BuildResult r = build_bvh8(bp, cfg, makeLeaf);
  A.nodes = std::move(r.nodes);
  A.root = r.root;
  A.maxDepth = r.maxDepth;
  A.leafCount = r.leafCount;
  for (const BuildPrim& p : bp) s->bounds.extend(p.box);
}

void Scene::commit()
{
  std::lock_guard<std::mutex> g(buildMutex);
  service_quiesce(device); // the upload allocates device memory (see Scene::~Scene)
  for (Geometry* geo : geometries) {
    if (!geo || !geo->enabled) continue;
    switch (geo->type) {
    case RTC_GEOMETRY_TYPE_TRIANGLE:
    case RTC_GEOMETRY_TYPE_SUBDIVISION: break;
    default: // scene.cpp:25-30: geometry types compiled out raise INVALID_OPERATION
      RT_THROW(RTC_ERROR_INVALID_OPERATION, "geometry type not supported by the MI355X traversal path");
    }
  }
  triIntersectFilter = triOccludedFilter = subdivFilter = false;
  for (Geometry* geo : geometries) {
    if (!geo || !geo->enabled) continue;
    if (geo->type == RTC_GEOMETRY_TYPE_TRIANGLE) {
      triIntersectFilter |= geo->intersectFilter != nullptr;
      triOccludedFilter |= geo->occludedFilter != nullptr;
    } else
      subdivFilter |= geo->intersectFilter != nullptr || geo->occludedFilter != nullptr;
  }
  if (progressFn && !progressFn(progressUser, 0.0)) RT_THROW(RTC_ERROR_CANCELLED, "progress monitor forced termination");
  bounds = Box3();
  build_triangle_accel(this);
  build_subdiv_accel(this);
  triAccel.upload(device);
  subdivAccel.upload(device);
  if (progressFn) progressFn(progressUser, 1.0);
  if (device->verbose >= 2) {
    fprintf(stderr, "embree3-amd: tri accel kind %u: %zu nodes (%zu B), %zu tris, depth %u; subdiv accel kind %u: %zu nodes, %zu blobs (%zu B)\n",
            triAccel.kind, triAccel.nodes.size(), triAccel.nodes.size() * sizeof(QNode8), triAccel.prims.size(), triAccel.maxDepth,
            subdivAccel.kind, subdivAccel.nodes.size(), subdivAccel.blobOffsets.size(), subdivAccel.blobs.size());
  }
  modified = false;
}

} // namespace rtamd
