// extern "C" embree3 entry points: handle casts, error funnel, dispatch.
// Replaces kernels/common/rtcore.cpp of the reference (entry points :34-1475, RTC_CATCH_* kernels/common/rtcore.h:40-67).
#include <new>

#include "../../include/embree3/rtcore_amd.h"
#include "rt_objects.h"
#include "rt_trace.h"
#include "cbvh_encode.h"
#include "subdiv_tess.h"

using namespace rtamd;

#define API extern "C" __attribute__((visibility("default")))

namespace {

std::mutex g_apiMutex; // rtcNewDevice / Retain / Release / GetProperty are serialised (rtcore.cpp:32,38)

void report(Device* dev, RTCError code, const char* msg)
{
  if (dev) dev->setError(code, msg);
  else if (thread_error() == RTC_ERROR_NONE) thread_error() = code; // device.cpp:262-263
}

#define CATCH_BEGIN try {
#define CATCH_END(dev)                                                                  \
  }                                                                                     \
  catch (const rtc_error& e) { report((dev), e.code, e.msg.c_str()); }                  \
  catch (const std::bad_alloc&) { report((dev), RTC_ERROR_OUT_OF_MEMORY, "out of memory"); } \
  catch (const std::exception& e) { report((dev), RTC_ERROR_UNKNOWN, e.what()); }       \
  catch (...) { report((dev), RTC_ERROR_UNKNOWN, "unknown exception caught"); }

#define VERIFY(h) \
  if ((h) == nullptr) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument: " #h " is null")

inline Device* D(RTCDevice h) { return (Device*)h; }
inline Scene* S(RTCScene h) { return (Scene*)h; }
inline Geometry* G(RTCGeometry h) { return (Geometry*)h; }
inline Buffer* B(RTCBuffer h) { return (Buffer*)h; }
inline Device* devOf(RTCScene h) { return h ? S(h)->device : nullptr; }
inline Device* devOf(RTCGeometry h) { return h ? G(h)->device : nullptr; }

[[noreturn]] void unsupported(const char* what) { RT_THROW(RTC_ERROR_INVALID_OPERATION, std::string(what) + " not supported"); }

} // namespace

// ---- device ----------------------------------------------------------------------------------------------
API RTCDevice rtcNewDevice(const char* config)
{
  std::lock_guard<std::mutex> g(g_apiMutex);
  CATCH_BEGIN
  return (RTCDevice) new Device(config);
  CATCH_END(nullptr)
  return nullptr;
}

API void rtcRetainDevice(RTCDevice h)
{
  std::lock_guard<std::mutex> g(g_apiMutex);
  CATCH_BEGIN VERIFY(h); D(h)->retain(); CATCH_END(nullptr)
}

API void rtcReleaseDevice(RTCDevice h)
{
  std::lock_guard<std::mutex> g(g_apiMutex);
  CATCH_BEGIN VERIFY(h); D(h)->release(); CATCH_END(nullptr)
}

API ssize_t rtcGetDeviceProperty(RTCDevice h, enum RTCDeviceProperty prop)
{
  std::lock_guard<std::mutex> g(g_apiMutex);
  CATCH_BEGIN
  VERIFY(h);
  switch (prop) { // device.cpp:300-371
  case RTC_DEVICE_PROPERTY_VERSION: return RTC_VERSION;
  case RTC_DEVICE_PROPERTY_VERSION_MAJOR: return RTC_VERSION_MAJOR;
  case RTC_DEVICE_PROPERTY_VERSION_MINOR: return RTC_VERSION_MINOR;
  case RTC_DEVICE_PROPERTY_VERSION_PATCH: return RTC_VERSION_PATCH;
  case RTC_DEVICE_PROPERTY_NATIVE_RAY4_SUPPORTED:
  case RTC_DEVICE_PROPERTY_NATIVE_RAY8_SUPPORTED:
  case RTC_DEVICE_PROPERTY_NATIVE_RAY16_SUPPORTED: return 0; // accepted and traced as single rays, no native packet kernels
  case RTC_DEVICE_PROPERTY_RAY_STREAM_SUPPORTED: return 1;
  case RTC_DEVICE_PROPERTY_RAY_MASK_SUPPORTED: return 0;          // EMBREE_RAY_MASK default OFF (CMakeLists.txt:114)
  case RTC_DEVICE_PROPERTY_BACKFACE_CULLING_ENABLED: return 0;    // CMakeLists.txt:115
  case RTC_DEVICE_PROPERTY_FILTER_FUNCTION_SUPPORTED: return 1;   // host callbacks, two-phase (rt_trace.cpp, triangle geometry)
  case RTC_DEVICE_PROPERTY_IGNORE_INVALID_RAYS_ENABLED: return 0; // CMakeLists.txt:117
  case RTC_DEVICE_PROPERTY_TRIANGLE_GEOMETRY_SUPPORTED: return 1;
  case RTC_DEVICE_PROPERTY_QUAD_GEOMETRY_SUPPORTED: return 0;
  case RTC_DEVICE_PROPERTY_SUBDIVISION_GEOMETRY_SUPPORTED: return 1;
  case RTC_DEVICE_PROPERTY_CURVE_GEOMETRY_SUPPORTED: return 0;
  case RTC_DEVICE_PROPERTY_USER_GEOMETRY_SUPPORTED: return 0;
  case RTC_DEVICE_PROPERTY_TASKING_SYSTEM: return 0; // "internal"
  case RTC_DEVICE_PROPERTY_JOIN_COMMIT_SUPPORTED: return 1;
  case (RTCDeviceProperty)RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES: return (ssize_t)D(h)->statLaunches.load();
  case (RTCDeviceProperty)RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS: return (ssize_t)D(h)->statCombinedCalls.load();
  case (RTCDeviceProperty)RTCAMD_DEVICE_PROPERTY_COMBINED_BATCHES: return (ssize_t)D(h)->statCombinedBatches.load();
  case (RTCDeviceProperty)RTCAMD_DEVICE_PROPERTY_SERVICE_CALLS: return (ssize_t)D(h)->statServiceCalls.load();
  default: RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown readable property");
  }
  CATCH_END(D(h))
  return 0;
}

API enum RTCError rtcGetDeviceError(RTCDevice h)
{
  if (h == nullptr) { // error of a failed rtcNewDevice is stored per thread
    RTCError e = thread_error();
    thread_error() = RTC_ERROR_NONE;
    return e;
  }
  return D(h)->takeError();
}

API void rtcSetDeviceErrorFunction(RTCDevice h, RTCErrorFunction fn, void* user)
{
  CATCH_BEGIN VERIFY(h); D(h)->errorFn = fn; D(h)->errorFnUser = user; CATCH_END(D(h))
}

API void rtcSetDeviceMemoryMonitorFunction(RTCDevice h, RTCMemoryMonitorFunction fn, void* user)
{
  CATCH_BEGIN VERIFY(h); D(h)->memFn = fn; D(h)->memFnUser = user; CATCH_END(D(h))
}

// ---- buffers -----------------------------------------------------------------------------------------------
API RTCBuffer rtcNewBuffer(RTCDevice h, size_t bytes)
{
  CATCH_BEGIN VERIFY(h); return (RTCBuffer) new Buffer(D(h), bytes, nullptr); CATCH_END(D(h))
  return nullptr;
}

API RTCBuffer rtcNewSharedBuffer(RTCDevice h, void* ptr, size_t bytes)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(ptr);
  return (RTCBuffer) new Buffer(D(h), bytes, ptr);
  CATCH_END(D(h))
  return nullptr;
}

API void* rtcGetBufferData(RTCBuffer h)
{
  CATCH_BEGIN VERIFY(h); return B(h)->ptr; CATCH_END(h ? B(h)->device : nullptr)
  return nullptr;
}

API void rtcRetainBuffer(RTCBuffer h) { CATCH_BEGIN VERIFY(h); B(h)->retain(); CATCH_END(h ? B(h)->device : nullptr) }
API void rtcReleaseBuffer(RTCBuffer h) { CATCH_BEGIN VERIFY(h); B(h)->release(); CATCH_END(nullptr) }

// ---- geometry ------------------------------------------------------------------------------------------------
API RTCGeometry rtcNewGeometry(RTCDevice h, enum RTCGeometryType type)
{
  CATCH_BEGIN
  VERIFY(h);
  switch (type) {
  case RTC_GEOMETRY_TYPE_TRIANGLE:
  case RTC_GEOMETRY_TYPE_SUBDIVISION: return (RTCGeometry) new Geometry(D(h), type);
  case RTC_GEOMETRY_TYPE_QUAD: unsupported("RTC_GEOMETRY_TYPE_QUAD");
  case RTC_GEOMETRY_TYPE_USER: unsupported("RTC_GEOMETRY_TYPE_USER");
  case RTC_GEOMETRY_TYPE_INSTANCE: unsupported("RTC_GEOMETRY_TYPE_INSTANCE");
  case RTC_GEOMETRY_TYPE_FLAT_LINEAR_CURVE:
  case RTC_GEOMETRY_TYPE_ROUND_BEZIER_CURVE:
  case RTC_GEOMETRY_TYPE_FLAT_BEZIER_CURVE:
  case RTC_GEOMETRY_TYPE_ROUND_BSPLINE_CURVE:
  case RTC_GEOMETRY_TYPE_FLAT_BSPLINE_CURVE: unsupported("RTC_GEOMETRY_TYPE_*_CURVE");
  default: RT_THROW(RTC_ERROR_UNKNOWN, "invalid geometry type");
  }
  CATCH_END(D(h))
  return nullptr;
}

API void rtcRetainGeometry(RTCGeometry h) { CATCH_BEGIN VERIFY(h); G(h)->retain(); CATCH_END(devOf(h)) }
API void rtcReleaseGeometry(RTCGeometry h) { CATCH_BEGIN VERIFY(h); G(h)->release(); CATCH_END(nullptr) }
API void rtcCommitGeometry(RTCGeometry h)
{
  CATCH_BEGIN
  VERIFY(h);
  G(h)->committed = true;
  std::lock_guard<std::mutex> g(G(h)->interpMutex);
  G(h)->interpCache.reset(); // buffers may have changed: rtcInterpolate refines them again on next use
  CATCH_END(devOf(h))
}
API void rtcEnableGeometry(RTCGeometry h) { CATCH_BEGIN VERIFY(h); G(h)->enabled = true; CATCH_END(devOf(h)) }
API void rtcDisableGeometry(RTCGeometry h) { CATCH_BEGIN VERIFY(h); G(h)->enabled = false; CATCH_END(devOf(h)) }

API void rtcSetGeometryTimeStepCount(RTCGeometry h, unsigned int n)
{
  CATCH_BEGIN
  VERIFY(h);
  if (n == 0 || n > RTC_MAX_TIME_STEP_COUNT) RT_THROW(RTC_ERROR_INVALID_OPERATION, "number of time steps out of range");
  G(h)->timeSteps = n;
  CATCH_END(devOf(h))
}

API void rtcSetGeometryVertexAttributeCount(RTCGeometry h, unsigned int n)
{
  CATCH_BEGIN VERIFY(h); G(h)->vertexAttribCount = n; CATCH_END(devOf(h))
}

API void rtcSetGeometryMask(RTCGeometry h, unsigned int mask) { CATCH_BEGIN VERIFY(h); G(h)->mask = mask; CATCH_END(devOf(h)) }

API void rtcSetGeometryBuildQuality(RTCGeometry h, enum RTCBuildQuality q)
{
  CATCH_BEGIN
  VERIFY(h);
  if (q != RTC_BUILD_QUALITY_LOW && q != RTC_BUILD_QUALITY_MEDIUM && q != RTC_BUILD_QUALITY_HIGH && q != RTC_BUILD_QUALITY_REFIT)
    RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid build quality");
  G(h)->quality = q;
  CATCH_END(devOf(h))
}

API void rtcSetGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned int slot, enum RTCFormat format, RTCBuffer buffer,
                              size_t byteOffset, size_t byteStride, size_t itemCount)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(buffer);
  if (itemCount > 0xFFFFFFFFu) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "buffer too large");
  G(h)->bind(type, slot, format, B(buffer), byteOffset, byteStride, itemCount);
  CATCH_END(devOf(h))
}

API void rtcSetSharedGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned int slot, enum RTCFormat format, const void* ptr,
                                    size_t byteOffset, size_t byteStride, size_t itemCount)
{
  CATCH_BEGIN
  VERIFY(h);
  if (itemCount > 0xFFFFFFFFu) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "buffer too large");
  Buffer* b = new Buffer(G(h)->device, itemCount * byteStride, (void*)(ptr ? ptr : (const void*)""));
  try {
    G(h)->bind(type, slot, format, b, byteOffset, byteStride, itemCount);
  } catch (...) {
    b->release();
    throw;
  }
  b->release();
  CATCH_END(devOf(h))
}

API void* rtcSetNewGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned int slot, enum RTCFormat format, size_t byteStride,
                                  size_t itemCount)
{
  CATCH_BEGIN
  VERIFY(h);
  if (itemCount > 0xFFFFFFFFu) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "buffer too large");
  size_t bytes = itemCount * byteStride;
  if (type == RTC_BUFFER_TYPE_VERTEX || type == RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE) bytes = (bytes + 15) & ~(size_t)15; // rtcore.cpp:1308-1310
  Buffer* b = new Buffer(G(h)->device, bytes, nullptr);
  try {
    G(h)->bind(type, slot, format, b, 0, byteStride, itemCount);
  } catch (...) {
    b->release();
    throw;
  }
  void* p = b->ptr;
  b->release();
  return p;
  CATCH_END(devOf(h))
  return nullptr;
}

API void* rtcGetGeometryBufferData(RTCGeometry h, enum RTCBufferType type, unsigned int slot)
{
  CATCH_BEGIN
  VERIFY(h);
  BufferView* v = G(h)->view(type, slot);
  if (!v || !v->valid()) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown buffer");
  return v->buf->ptr + v->offset;
  CATCH_END(devOf(h))
  return nullptr;
}

API void rtcUpdateGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned int slot)
{
  CATCH_BEGIN
  VERIFY(h);
  BufferView* v = G(h)->view(type, slot);
  if (v) v->modified = true;
  G(h)->committed = false;
  CATCH_END(devOf(h))
}

API void rtcSetGeometryIntersectFilterFunction(RTCGeometry h, RTCFilterFunctionN f)
{
  CATCH_BEGIN VERIFY(h); G(h)->intersectFilter = f; CATCH_END(devOf(h))
}
API void rtcSetGeometryOccludedFilterFunction(RTCGeometry h, RTCFilterFunctionN f)
{
  CATCH_BEGIN VERIFY(h); G(h)->occludedFilter = f; CATCH_END(devOf(h))
}
API void rtcSetGeometryUserData(RTCGeometry h, void* p) { CATCH_BEGIN VERIFY(h); G(h)->userPtr = p; CATCH_END(devOf(h)) }
API void* rtcGetGeometryUserData(RTCGeometry h)
{
  CATCH_BEGIN VERIFY(h); return G(h)->userPtr; CATCH_END(devOf(h))
  return nullptr;
}

API void rtcSetGeometryUserPrimitiveCount(RTCGeometry h, unsigned int) { CATCH_BEGIN VERIFY(h); unsupported("user geometry"); CATCH_END(devOf(h)) }
API void rtcSetGeometryBoundsFunction(RTCGeometry h, RTCBoundsFunction, void*) { CATCH_BEGIN VERIFY(h); unsupported("user geometry"); CATCH_END(devOf(h)) }
API void rtcSetGeometryIntersectFunction(RTCGeometry h, RTCIntersectFunctionN) { CATCH_BEGIN VERIFY(h); unsupported("user geometry"); CATCH_END(devOf(h)) }
API void rtcSetGeometryOccludedFunction(RTCGeometry h, RTCOccludedFunctionN) { CATCH_BEGIN VERIFY(h); unsupported("user geometry"); CATCH_END(devOf(h)) }
API void rtcFilterIntersection(const struct RTCIntersectFunctionNArguments*, const struct RTCFilterFunctionNArguments*)
{
  CATCH_BEGIN unsupported("rtcFilterIntersection"); CATCH_END(nullptr)
}
API void rtcFilterOcclusion(const struct RTCOccludedFunctionNArguments*, const struct RTCFilterFunctionNArguments*)
{
  CATCH_BEGIN unsupported("rtcFilterOcclusion"); CATCH_END(nullptr)
}
API void rtcSetGeometryInstancedScene(RTCGeometry h, RTCScene) { CATCH_BEGIN VERIFY(h); unsupported("instancing"); CATCH_END(devOf(h)) }
API void rtcSetGeometryTransform(RTCGeometry h, unsigned int, enum RTCFormat, const void*) { CATCH_BEGIN VERIFY(h); unsupported("instancing"); CATCH_END(devOf(h)) }
API void rtcGetGeometryTransform(RTCGeometry h, float, enum RTCFormat, void*) { CATCH_BEGIN VERIFY(h); unsupported("instancing"); CATCH_END(devOf(h)) }

API void rtcSetGeometryTessellationRate(RTCGeometry h, float rate) { CATCH_BEGIN VERIFY(h); G(h)->tessellationRate = rate; CATCH_END(devOf(h)) }

API void rtcSetGeometryTopologyCount(RTCGeometry h, unsigned int n)
{
  CATCH_BEGIN
  VERIFY(h);
  if (n == 0) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "at least one topology has to exist");
  G(h)->topologyCount = n;
  G(h)->subdivMode.resize(n, RTC_SUBDIVISION_MODE_SMOOTH_BOUNDARY);
  CATCH_END(devOf(h))
}

API void rtcSetGeometrySubdivisionMode(RTCGeometry h, unsigned int topologyID, enum RTCSubdivisionMode mode)
{
  CATCH_BEGIN
  VERIFY(h);
  if (topologyID >= G(h)->subdivMode.size()) RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid topology ID");
  G(h)->subdivMode[topologyID] = mode;
  CATCH_END(devOf(h))
}

API void rtcSetGeometryVertexAttributeTopology(RTCGeometry h, unsigned int vertexAttributeID, unsigned int topologyID)
{
  CATCH_BEGIN
  VERIFY(h);
  if (topologyID >= G(h)->topologyCount) RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid topology ID");
  if (vertexAttributeID >= 65536) RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid vertex attribute ID");
  if (G(h)->attribTopology.size() <= vertexAttributeID) G(h)->attribTopology.resize(vertexAttributeID + 1, 0u);
  G(h)->attribTopology[vertexAttributeID] = topologyID;
  G(h)->committed = false;
  CATCH_END(devOf(h))
}

API void rtcSetGeometryDisplacementFunction(RTCGeometry h, RTCDisplacementFunctionN f)
{
  CATCH_BEGIN VERIFY(h); G(h)->displacement = f; CATCH_END(devOf(h))
}

API void rtcInterpolate(const struct RTCInterpolateArguments* args)
{
  CATCH_BEGIN
  VERIFY(args);
  VERIFY(args->geometry);
  Geometry* g = G(args->geometry);
  if (g->type == RTC_GEOMETRY_TYPE_TRIANGLE) interpolate_triangles(g, args);
  else if (g->type == RTC_GEOMETRY_TYPE_SUBDIVISION) interpolate_subdiv(g, args);
  else unsupported("rtcInterpolate on this geometry type");
  CATCH_END(args && args->geometry ? devOf(args->geometry) : nullptr)
}

API void rtcInterpolateN(const struct RTCInterpolateNArguments* args)
{
  CATCH_BEGIN
  VERIFY(args);
  VERIFY(args->geometry);
  // Geometry::interpolateN, kernels/common/geometry.cpp:187-262: N independent evaluations, results stored SoA [value][i]
  if (args->valueCount > 256) RT_THROW(RTC_ERROR_INVALID_OPERATION, "maximally 256 floating point values can be interpolated per vertex");
  const int* valid = (const int*)args->valid;
  float P[256], dPdu[256], dPdv[256], ddPdudu[256], ddPdvdv[256], ddPdudv[256];
  for (unsigned i = 0; i < args->N; i++) {
    if (valid && !valid[i]) continue;
    RTCInterpolateArguments a;
    a.geometry = args->geometry;
    a.primID = args->primIDs[i];
    a.u = args->u[i];
    a.v = args->v[i];
    a.bufferType = args->bufferType;
    a.bufferSlot = args->bufferSlot;
    a.P = args->P ? P : nullptr;
    a.dPdu = args->dPdu ? dPdu : nullptr;
    a.dPdv = args->dPdu ? dPdv : nullptr;
    a.ddPdudu = args->ddPdudu ? ddPdudu : nullptr;
    a.ddPdvdv = args->ddPdudu ? ddPdvdv : nullptr;
    a.ddPdudv = args->ddPdudu ? ddPdudv : nullptr;
    a.valueCount = args->valueCount;
    rtcInterpolate(&a);
    for (unsigned j = 0; j < args->valueCount; j++) {
      if (args->P) args->P[(size_t)j * args->N + i] = P[j];
      if (args->dPdu) { args->dPdu[(size_t)j * args->N + i] = dPdu[j]; args->dPdv[(size_t)j * args->N + i] = dPdv[j]; }
      if (args->ddPdudu) {
        args->ddPdudu[(size_t)j * args->N + i] = ddPdudu[j];
        args->ddPdvdv[(size_t)j * args->N + i] = ddPdvdv[j];
        args->ddPdudv[(size_t)j * args->N + i] = ddPdudv[j];
      }
    }
  }
  CATCH_END(args && args->geometry ? devOf(args->geometry) : nullptr)
}

// ---- scene -------------------------------------------------------------------------------------------------------
API RTCScene rtcNewScene(RTCDevice h)
{
  CATCH_BEGIN VERIFY(h); return (RTCScene) new Scene(D(h)); CATCH_END(D(h))
  return nullptr;
}
API void rtcRetainScene(RTCScene h) { CATCH_BEGIN VERIFY(h); S(h)->retain(); CATCH_END(devOf(h)) }
API void rtcReleaseScene(RTCScene h) { CATCH_BEGIN VERIFY(h); S(h)->release(); CATCH_END(nullptr) }

API unsigned int rtcAttachGeometry(RTCScene h, RTCGeometry g)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(g);
  if (S(h)->device != G(g)->device) RT_THROW(RTC_ERROR_INVALID_OPERATION, "inputs are from different devices");
  return S(h)->attach(G(g));
  CATCH_END(devOf(h))
  return RTC_INVALID_GEOMETRY_ID;
}

API void rtcAttachGeometryByID(RTCScene h, RTCGeometry g, unsigned int id)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(g);
  if (S(h)->device != G(g)->device) RT_THROW(RTC_ERROR_INVALID_OPERATION, "inputs are from different devices");
  S(h)->attachByID(G(g), id);
  CATCH_END(devOf(h))
}

API void rtcDetachGeometry(RTCScene h, unsigned int id) { CATCH_BEGIN VERIFY(h); S(h)->detach(id); CATCH_END(devOf(h)) }

API RTCGeometry rtcGetGeometry(RTCScene h, unsigned int id)
{
  CATCH_BEGIN
  VERIFY(h);
  Geometry* g = S(h)->get(id);
  if (!g) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid geometry ID");
  return (RTCGeometry)g;
  CATCH_END(devOf(h))
  return nullptr;
}

API void rtcSetSceneLevels(RTCScene h, unsigned subdivisionLevel, unsigned compressionLevel)
{
  CATCH_BEGIN
  VERIFY(h);
  S(h)->subdivisionLevel = subdivisionLevel;
  S(h)->compressionLevel = compressionLevel;
  S(h)->modified = true;
  CATCH_END(devOf(h))
}

API void rtcCommitScene(RTCScene h) { CATCH_BEGIN VERIFY(h); S(h)->commit(); CATCH_END(devOf(h)) }
API void rtcJoinCommitScene(RTCScene h) { CATCH_BEGIN VERIFY(h); S(h)->commit(); CATCH_END(devOf(h)) }

API void rtcSetSceneProgressMonitorFunction(RTCScene h, RTCProgressMonitorFunction fn, void* user)
{
  CATCH_BEGIN VERIFY(h); S(h)->progressFn = fn; S(h)->progressUser = user; CATCH_END(devOf(h))
}

API void rtcSetSceneBuildQuality(RTCScene h, enum RTCBuildQuality q)
{
  CATCH_BEGIN
  VERIFY(h);
  if (q != RTC_BUILD_QUALITY_LOW && q != RTC_BUILD_QUALITY_MEDIUM && q != RTC_BUILD_QUALITY_HIGH)
    RT_THROW(RTC_ERROR_INVALID_OPERATION, "invalid build quality"); // rtcore.cpp:230-237
  S(h)->quality = q;
  CATCH_END(devOf(h))
}

API void rtcSetSceneFlags(RTCScene h, enum RTCSceneFlags f)
{
  CATCH_BEGIN VERIFY(h); S(h)->flags = f; S(h)->modified = true; CATCH_END(devOf(h))
}

API enum RTCSceneFlags rtcGetSceneFlags(RTCScene h)
{
  CATCH_BEGIN VERIFY(h); return S(h)->flags; CATCH_END(devOf(h))
  return RTC_SCENE_FLAG_NONE;
}

API void rtcGetSceneBounds(RTCScene h, struct RTCBounds* o)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(o);
  if (S(h)->modified) RT_THROW(RTC_ERROR_INVALID_OPERATION, "scene got not committed");
  const Box3& b = S(h)->bounds;
  o->lower_x = b.lo.x; o->lower_y = b.lo.y; o->lower_z = b.lo.z; o->align0 = 0;
  o->upper_x = b.hi.x; o->upper_y = b.hi.y; o->upper_z = b.hi.z; o->align1 = 0;
  CATCH_END(devOf(h))
}

API void rtcGetSceneLinearBounds(RTCScene h, struct RTCLinearBounds* o)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(o);
  rtcGetSceneBounds(h, &o->bounds0);
  o->bounds1 = o->bounds0;
  CATCH_END(devOf(h))
}

// ---- hot path --------------------------------------------------------------------------------------------------------
API void rtcIntersect1(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRayHit* rayhit)
{
  CATCH_BEGIN
  VERIFY(h);
  trace_call(S(h), rayhit, 1, sizeof(RTCRayHit), false, ctx);
  CATCH_END(devOf(h))
}

API void rtcIntersect1M(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRayHit* rayhit, unsigned int M, size_t byteStride)
{
  CATCH_BEGIN
  VERIFY(h);
  trace_call(S(h), rayhit, M, byteStride, false, ctx);
  CATCH_END(devOf(h))
}

API void rtcOccluded1(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRay* ray)
{
  CATCH_BEGIN
  VERIFY(h);
  trace_call(S(h), ray, 1, sizeof(RTCRay), true, ctx);
  CATCH_END(devOf(h))
}

API void rtcOccluded1M(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRay* ray, unsigned int M, size_t byteStride)
{
  CATCH_BEGIN
  VERIFY(h);
  trace_call(S(h), ray, M, byteStride, true, ctx);
  CATCH_END(devOf(h))
}

API void rtcIntersect1Mp(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRayHit** rayhit, unsigned int M)
{
  CATCH_BEGIN
  VERIFY(h);
  trace_pointers(S(h), (void**)rayhit, M, false, ctx);
  CATCH_END(devOf(h))
}

API void rtcOccluded1Mp(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRay** ray, unsigned int M)
{
  CATCH_BEGIN
  VERIFY(h);
  trace_pointers(S(h), (void**)ray, M, true, ctx);
  CATCH_END(devOf(h))
}

// ---- packets, packet streams, SoA streams -----------------------------------------------------------------------------
// rtcIntersect4/8/16 (rtcore.cpp:306-401), rtcIntersectNM (:450-492), rtcIntersectNp (:494-539) and their occluded
// twins: the device has one kernel family - single rays in AoS records - so the active rays of a packet / stream are
// gathered into one AoS batch, traced like rtcIntersect1M, and tfar (+ the hit) scattered back.  Inactive lanes
// (valid[i] == 0) are not touched.  Results are those of N independent rtcIntersect1 calls, which is what the reference
// guarantees for packets as well.
template <typename Get, typename Put>
static void trace_gathered(Scene* s, RTCIntersectContext* ctx, bool occluded, size_t total, Get get, Put put)
{
  std::vector<RTCRayHit> aos;
  std::vector<size_t> idx;
  aos.reserve(total);
  idx.reserve(total);
  for (size_t i = 0; i < total; i++) {
    RTCRayHit r;
    if (get(i, r)) {
      aos.push_back(r);
      idx.push_back(i);
    }
  }
  if (aos.empty()) return;
  if (aos.size() > 0xFFFFFFFFull) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "too many rays in one call");
  trace_call(s, aos.data(), (uint32_t)aos.size(), sizeof(RTCRayHit), occluded, ctx); // any-hit reads the first 48 bytes of each record
  for (size_t k = 0; k < aos.size(); k++) put(idx[k], aos[k]);
}

static void trace_packets(RTCScene h, RTCIntersectContext* ctx, const int* valid, void* packets, unsigned N, unsigned M, size_t byteStride,
                          bool occluded)
{
  VERIFY(h);
  VERIFY(packets);
  auto ray_of = [&](size_t i) { return (RTCRayN*)((char*)packets + (i / N) * byteStride); };
  trace_gathered(
      S(h), ctx, occluded, (size_t)N * M,
      [&](size_t i, RTCRayHit& r) {
        const unsigned l = (unsigned)(i % N);
        if (valid && valid[l] == 0) return false;
        RTCRayN* rn = ray_of(i);
        r.ray = rtcGetRayFromRayN(rn, N, l);
        if (occluded) {
          memset(&r.hit, 0, sizeof(r.hit));
          r.hit.geomID = r.hit.primID = r.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
        } else
          r.hit = rtcGetHitFromHitN(RTCRayHitN_HitN((RTCRayHitN*)rn, N), N, l);
        return true;
      },
      [&](size_t i, const RTCRayHit& r) {
        const unsigned l = (unsigned)(i % N);
        RTCRayN* rn = ray_of(i);
        RTCRayN_tfar(rn, N, l) = r.ray.tfar;
        if (!occluded) rtcCopyHitToHitN(RTCRayHitN_HitN((RTCRayHitN*)rn, N), &r.hit, N, l);
      });
}

#define PACKET_TRACE(W)                                                                                                        \
  API void rtcIntersect##W(const int* valid, RTCScene scene, struct RTCIntersectContext* ctx, struct RTCRayHit##W* rayhit)      \
  {                                                                                                                            \
    CATCH_BEGIN trace_packets(scene, ctx, valid, rayhit, W, 1, 0, false); CATCH_END(devOf(scene))                              \
  }                                                                                                                            \
  API void rtcOccluded##W(const int* valid, RTCScene scene, struct RTCIntersectContext* ctx, struct RTCRay##W* ray)             \
  {                                                                                                                            \
    CATCH_BEGIN trace_packets(scene, ctx, valid, ray, W, 1, 0, true); CATCH_END(devOf(scene))                                  \
  }
PACKET_TRACE(4)
PACKET_TRACE(8)
PACKET_TRACE(16)

API void rtcIntersectNM(RTCScene scene, struct RTCIntersectContext* ctx, struct RTCRayHitN* rayhit, unsigned int N, unsigned int M, size_t byteStride)
{
  CATCH_BEGIN
  if (N == 0 || M == 0) return;
  trace_packets(scene, ctx, nullptr, rayhit, N, M, byteStride, false);
  CATCH_END(devOf(scene))
}
API void rtcOccludedNM(RTCScene scene, struct RTCIntersectContext* ctx, struct RTCRayN* ray, unsigned int N, unsigned int M, size_t byteStride)
{
  CATCH_BEGIN
  if (N == 0 || M == 0) return;
  trace_packets(scene, ctx, nullptr, ray, N, M, byteStride, true);
  CATCH_END(devOf(scene))
}

static void trace_np(RTCScene h, RTCIntersectContext* ctx, const RTCRayNp& ray, const RTCHitNp* hit, unsigned N)
{
  VERIFY(h);
  const bool occluded = hit == nullptr;
  trace_gathered(
      S(h), ctx, occluded, N,
      [&](size_t i, RTCRayHit& r) {
        r.ray.org_x = ray.org_x[i]; r.ray.org_y = ray.org_y[i]; r.ray.org_z = ray.org_z[i]; r.ray.tnear = ray.tnear[i];
        r.ray.dir_x = ray.dir_x[i]; r.ray.dir_y = ray.dir_y[i]; r.ray.dir_z = ray.dir_z[i]; r.ray.time = ray.time ? ray.time[i] : 0.f;
        r.ray.tfar = ray.tfar[i];
        r.ray.mask = ray.mask ? ray.mask[i] : 0xFFFFFFFFu; r.ray.id = ray.id ? ray.id[i] : 0u; r.ray.flags = ray.flags ? ray.flags[i] : 0u;
        memset(&r.hit, 0, sizeof(r.hit));
        r.hit.geomID = r.hit.primID = r.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
        if (hit) {
          r.hit.Ng_x = hit->Ng_x[i]; r.hit.Ng_y = hit->Ng_y[i]; r.hit.Ng_z = hit->Ng_z[i]; r.hit.u = hit->u[i]; r.hit.v = hit->v[i];
          r.hit.primID = hit->primID[i]; r.hit.geomID = hit->geomID[i];
          if (hit->instID[0]) r.hit.instID[0] = hit->instID[0][i];
        }
        return true;
      },
      [&](size_t i, const RTCRayHit& r) {
        ray.tfar[i] = r.ray.tfar;
        if (hit) {
          hit->Ng_x[i] = r.hit.Ng_x; hit->Ng_y[i] = r.hit.Ng_y; hit->Ng_z[i] = r.hit.Ng_z; hit->u[i] = r.hit.u; hit->v[i] = r.hit.v;
          hit->primID[i] = r.hit.primID; hit->geomID[i] = r.hit.geomID;
          if (hit->instID[0]) hit->instID[0][i] = r.hit.instID[0];
        }
      });
}
API void rtcIntersectNp(RTCScene scene, struct RTCIntersectContext* ctx, const struct RTCRayHitNp* rayhit, unsigned int N)
{
  CATCH_BEGIN
  VERIFY(rayhit);
  if (N) trace_np(scene, ctx, rayhit->ray, &rayhit->hit, N);
  CATCH_END(devOf(scene))
}
API void rtcOccludedNp(RTCScene scene, struct RTCIntersectContext* ctx, const struct RTCRayNp* ray, unsigned int N)
{
  CATCH_BEGIN
  VERIFY(ray);
  if (N) trace_np(scene, ctx, *ray, nullptr, N);
  CATCH_END(devOf(scene))
}

// ---- BVH builder API: declared for link compatibility ------------------------------------------------------------------
API RTCBVH rtcNewBVH(RTCDevice h)
{
  CATCH_BEGIN unsupported("rtcNewBVH"); CATCH_END(D(h))
  return nullptr;
}
API void* rtcBuildBVH(const struct RTCBuildArguments*)
{
  CATCH_BEGIN unsupported("rtcBuildBVH"); CATCH_END(nullptr)
  return nullptr;
}
API void* rtcThreadLocalAlloc(RTCThreadLocalAllocator, size_t, size_t)
{
  CATCH_BEGIN unsupported("rtcThreadLocalAlloc"); CATCH_END(nullptr)
  return nullptr;
}
API void rtcRetainBVH(RTCBVH) { CATCH_BEGIN unsupported("rtcRetainBVH"); CATCH_END(nullptr) }
API void rtcReleaseBVH(RTCBVH) { CATCH_BEGIN unsupported("rtcReleaseBVH"); CATCH_END(nullptr) }

// ---- MI355X extensions (include/embree3/rtcore_amd.h) -------------------------------------------------------------------
API void* rtcamdGetDeviceStream(RTCDevice h)
{
  CATCH_BEGIN
  VERIFY(h);
  D(h)->useDevice();
  Device::GpuShard& sh = D(h)->primary();
  std::lock_guard<std::mutex> seq(sh.seqMutex); // launches read the stream under this mutex (rt_trace.cpp launch_on)
  return (void*)sh.stream;
  CATCH_END(D(h))
  return nullptr;
}

API void rtcamdSetDeviceStream(RTCDevice h, void* stream)
{
  CATCH_BEGIN
  VERIFY(h);
  Device* d = D(h);
  d->useDevice();
  // the stream of the FIRST shard: device-resident batches are traced on the GPU they live on, and a caller that manages
  // streams itself works with one GPU per RTCDevice (one process per GPU)
  // The stream is one mutable value per device: a launch reads it under the shard's seqMutex, so switching it is serialised with
  // launches here; two threads that want DIFFERENT streams at the same time still have to order their
  // {rtcamdSetDeviceStream, rtcIntersect1M} pairs themselves (or use one RTCDevice per thread).
  Device::GpuShard& sh = d->primary();
  hipStream_t old = nullptr;
  {
    std::lock_guard<std::mutex> seq(sh.seqMutex);
    if (sh.stream && sh.ownsStream) old = sh.stream;
    sh.stream = (hipStream_t)stream;
    sh.ownsStream = false;
  }
  if (old) { // the library's own stream: drained and destroyed outside the lock, nobody can pick it up any more
    HIP_CHECK(hipStreamSynchronize(old));
    HIP_CHECK(hipStreamDestroy(old));
  }
  CATCH_END(D(h))
}

API void rtcamdSynchronizeDevice(RTCDevice h)
{
  CATCH_BEGIN
  VERIFY(h);
  D(h)->synchronize();
  CATCH_END(D(h))
}

API int rtcamdGetDeviceOrdinal(RTCDevice h) { return h ? D(h)->gpu : -1; }

API void rtcamdGetSceneStats(RTCScene h, struct RTCAMDSceneStats* st)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(st);
  if (st->byteSize != sizeof(RTCAMDSceneStats)) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "RTCAMDSceneStats::byteSize mismatch");
  if (S(h)->modified) RT_THROW(RTC_ERROR_INVALID_OPERATION, "scene got not committed");
  const Accel& A = S(h)->subdivAccel.kind != ACCEL_NONE ? S(h)->subdivAccel : S(h)->triAccel;
  st->accelKind = A.kind;
  st->branching = 8;
  st->nodeCount = A.nodes.size();
  st->nodeBytes = sizeof(QNode8);
  const bool tri = A.kind == ACCEL_TRI_PLUECKER || A.kind == ACCEL_TRI_MOELLER;
  st->primCount = tri ? A.prims.size() : (A.blobStride ? A.blobs.size() / A.blobStride : 0);
  st->primBytes = tri ? sizeof(TriRecord) : A.blobStride;
  st->leafCount = A.leafCount;
  st->totalBytes = S(h)->triAccel.deviceBytes() + S(h)->subdivAccel.deviceBytes();
  st->maxDepth = A.maxDepth;
  st->reserved = 0;
  CATCH_END(devOf(h))
}

API void rtcamdIntersect1MCounted(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRayHit* rayhit, unsigned int M, size_t byteStride,
                                  struct RTCAMDTraceCounters* counters)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(counters);
  static_assert(sizeof(RTCAMDTraceCounters) == sizeof(TraceCounters), "counter structs must match");
  trace_batch(S(h), rayhit, M, byteStride, false, ctx, (TraceCounters*)counters);
  CATCH_END(devOf(h))
}

API void rtcamdOccluded1MCounted(RTCScene h, struct RTCIntersectContext* ctx, struct RTCRay* ray, unsigned int M, size_t byteStride,
                                 struct RTCAMDTraceCounters* counters)
{
  CATCH_BEGIN
  VERIFY(h);
  VERIFY(counters);
  trace_batch(S(h), ray, M, byteStride, true, ctx, (TraceCounters*)counters);
  CATCH_END(devOf(h))
}

API const void* rtcamdGetAccelData(RTCScene h, unsigned int kind, size_t* byteSize)
{
  CATCH_BEGIN
  VERIFY(h);
  if (S(h)->modified) RT_THROW(RTC_ERROR_INVALID_OPERATION, "scene got not committed");
  const Accel& A = S(h)->subdivAccel.kind != ACCEL_NONE ? S(h)->subdivAccel : S(h)->triAccel;
  const void* p = nullptr;
  size_t n = 0;
  switch (kind) {
  case 0: p = A.nodes.data(); n = A.nodes.size() * sizeof(QNode8); break;
  case 1: p = A.prims.data(); n = A.prims.size() * sizeof(TriRecord); break;
  case 2: p = A.blobs.data(); n = A.blobs.size(); break;
  case 3: p = A.blobOffsets.data(); n = A.blobOffsets.size() * 4; break;
  case 4: p = S(h)->debugGrids.data(); n = S(h)->debugGrids.size(); break;
  default: RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown accel data kind");
  }
  if (byteSize) *byteSize = n;
  return p;
  CATCH_END(devOf(h))
  if (byteSize) *byteSize = 0;
  return nullptr;
}

// development aid: the wave log of the last launch on the first shard (instrumented twin: WaveRecord[]; -DTRACE_TIMELINE builds with
// RTAMD_TIMELINE=1: 8 words per wavefront of the plain kernel, trace.h), after synchronising the device
API size_t rtcamdDebugReadWaveLog(RTCDevice hdevice, void* out, size_t bytes)
{
  Device* dev = (Device*)hdevice;
  if (!dev || dev->gpu < 0 || !out) return 0;
  Device::GpuShard& sh = dev->primary();
  sh.use();
  const size_t n = std::min(bytes, 2 * (size_t)WAVE_LOG_CAPACITY * sizeof(WaveRecord));
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, sh.countersDev, n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

// test hook: while `hold` is non-zero the call combiner's leader waits before it collects the pending calls, so that a test can make
// calls from several threads meet in ONE combined batch deterministically (tests/test_gpu_small_calls.py)
API void rtcamdDebugHoldCombiner(RTCDevice hdevice, int hold)
{
  Device* dev = (Device*)hdevice;
  if (dev) dev->combHold.store(hold != 0, std::memory_order_release);
}

API void rtcamdDebugCbvhLeafCodec(const float* box, const float* v, float extent, unsigned char* bytesOut, float* extentEstimate)
{
  cbvh_debug_leaf_codec(box, v, extent, bytesOut, extentEstimate);
}

API unsigned long long rtcamdDebugHostPoolSelfTest(RTCDevice h, unsigned int threads, unsigned int cycles, unsigned int jobs, unsigned int parts)
{
  CATCH_BEGIN
  VERIFY(h);
  Device* dev = D(h);
  std::lock_guard<std::mutex> lock(dev->launchMutex); // the pool serves one caller at a time (host-pointer batches hold this mutex too)
  if (threads > 1 && dev->hostPool.threads.size() < threads - 1) dev->hostPool.start(threads - 1);
  unsigned long long good = 0;
  std::vector<std::atomic<unsigned>> hits(parts);
  for (unsigned c = 0; c < cycles; c++) {
    dev->hostPool.begin();
    for (unsigned j = 0; j < jobs; j++) {
      for (auto& x : hits) x.store(0u);
      const unsigned salt = c * 7919u + j;
      dev->hostPool.run(parts, [&](size_t i) { hits[i].fetch_add(1u + ((unsigned)i ^ salt) * 2u); });
      for (unsigned i = 0; i < parts; i++) good += hits[i].load() == 1u + (i ^ salt) * 2u ? 1u : 0u;
    }
    dev->hostPool.end();
  }
  return good;
  CATCH_END(D(h))
  return 0;
}

API unsigned int rtcamdGetAccelRoot(RTCScene h)
{
  CATCH_BEGIN
  VERIFY(h);
  const Accel& A = S(h)->subdivAccel.kind != ACCEL_NONE ? S(h)->subdivAccel : S(h)->triAccel;
  return A.root;
  CATCH_END(devOf(h))
  return REF_EMPTY;
}
