/*
 * embree3 C API contract for the MI355X-native traversal library (libembree3.so built from
 * embree-compressed_amd/csrc).
 *
 * This single header declares the complete drop-in boundary.  Every enum value, struct layout and
 * function signature is ABI-compatible with the reference's public headers so that an application
 * (or the reference's tutorials) compiled against either set of headers links against this library
 * unchanged.  The reference interface each block replaces is cited as file:line relative to the
 * reference tree.  The per-topic headers rtcore_{common,device,buffer,geometry,scene,ray,builder,
 * version}.h exist only as forwarders to this file.
 *
 * Entry points that the reference routes to x86 packet kernels or to geometry types outside the
 * traversal hot path (rtcIntersect4/8/16, NM, Np, curves, user geometry, instances, rtcBuildBVH) are
 * declared and exported; calling them raises RTC_ERROR_INVALID_OPERATION on the device, which is what
 * the reference does for features compiled out (kernels/common/rtcore.cpp:429,680).
 */
#ifndef EMBREE3_AMD_RTCORE_H
#define EMBREE3_AMD_RTCORE_H

#include <stddef.h>
#include <stdbool.h>
#include <sys/types.h>

/* ---- version (reference: include/embree3/rtcore_version.h:17-21) ------------------------------ */
#define RTC_VERSION_MAJOR 3
#define RTC_VERSION_MINOR 0
#define RTC_VERSION_PATCH 0
#define RTC_VERSION 30000
#define RTC_VERSION_STRING "3.0.0"

/* ---- decoration macros (reference: rtcore_common.h:36-66) -------------------------------------- */
#ifndef RTC_API
#define RTC_API
#endif
#define RTC_ALIGN(...) __attribute__((aligned(__VA_ARGS__)))
#ifndef RTC_DEPRECATED
#define RTC_DEPRECATED __attribute__((deprecated))
#endif
#define RTC_FORCEINLINE inline __attribute__((always_inline))

#define RTC_INVALID_GEOMETRY_ID ((unsigned int)-1)
#define RTC_MAX_TIME_STEP_COUNT 129
#define RTC_MAX_INSTANCE_LEVEL_COUNT 1

#if defined(__cplusplus)
extern "C" {
#endif

/* ---- formats (reference: rtcore_common.h:78-167) ------------------------------------------------ */
/* Scalar families: base value is the 1-component format, +1/+2/+3 are the 2/3/4-component ones. */
#define RTC_FMT_FAMILY4(NAME, BASE) \
  RTC_FORMAT_##NAME = (BASE), RTC_FORMAT_##NAME##2, RTC_FORMAT_##NAME##3, RTC_FORMAT_##NAME##4
/* Matrix formats encode 0x9<major><rows><cols> with major 1 = row-major, 2 = column-major. */
#define RTC_FMT_MATRIX(R, C) \
  RTC_FORMAT_FLOAT##R##X##C##_ROW_MAJOR = 0x9100 | ((R) << 4) | (C), \
  RTC_FORMAT_FLOAT##R##X##C##_COLUMN_MAJOR = 0x9200 | ((R) << 4) | (C)

enum RTCFormat
{
  RTC_FORMAT_UNDEFINED = 0,
  RTC_FMT_FAMILY4(UCHAR, 0x1001),
  RTC_FMT_FAMILY4(CHAR, 0x2001),
  RTC_FMT_FAMILY4(USHORT, 0x3001),
  RTC_FMT_FAMILY4(SHORT, 0x4001),
  RTC_FMT_FAMILY4(UINT, 0x5001),
  RTC_FMT_FAMILY4(INT, 0x6001),
  RTC_FMT_FAMILY4(ULLONG, 0x7001),
  RTC_FMT_FAMILY4(LLONG, 0x8001),
  RTC_FMT_FAMILY4(FLOAT, 0x9001),
  RTC_FORMAT_FLOAT5, RTC_FORMAT_FLOAT6, RTC_FORMAT_FLOAT7, RTC_FORMAT_FLOAT8,
  RTC_FORMAT_FLOAT9, RTC_FORMAT_FLOAT10, RTC_FORMAT_FLOAT11, RTC_FORMAT_FLOAT12,
  RTC_FORMAT_FLOAT13, RTC_FORMAT_FLOAT14, RTC_FORMAT_FLOAT15, RTC_FORMAT_FLOAT16,
  RTC_FMT_MATRIX(2, 2), RTC_FMT_MATRIX(2, 3), RTC_FMT_MATRIX(2, 4),
  RTC_FMT_MATRIX(3, 2), RTC_FMT_MATRIX(3, 3), RTC_FMT_MATRIX(3, 4),
  RTC_FMT_MATRIX(4, 2), RTC_FMT_MATRIX(4, 3), RTC_FMT_MATRIX(4, 4)
};

/* ---- small enums ---------------------------------------------------------------------------------- */
enum RTCBuildQuality /* rtcore_common.h:170-176 */
{
  RTC_BUILD_QUALITY_LOW = 0, RTC_BUILD_QUALITY_MEDIUM = 1, RTC_BUILD_QUALITY_HIGH = 2, RTC_BUILD_QUALITY_REFIT = 3
};

enum RTCIntersectContextFlags /* rtcore_common.h:193-198 */
{
  RTC_INTERSECT_CONTEXT_FLAG_NONE = 0,
  RTC_INTERSECT_CONTEXT_FLAG_INCOHERENT = (0 << 0),
  RTC_INTERSECT_CONTEXT_FLAG_COHERENT = (1 << 0)
};

enum RTCError /* rtcore_device.h:68-77 */
{
  RTC_ERROR_NONE = 0, RTC_ERROR_UNKNOWN = 1, RTC_ERROR_INVALID_ARGUMENT = 2, RTC_ERROR_INVALID_OPERATION = 3,
  RTC_ERROR_OUT_OF_MEMORY = 4, RTC_ERROR_UNSUPPORTED_CPU = 5, RTC_ERROR_CANCELLED = 6
};

enum RTCDeviceProperty /* rtcore_device.h:40-63 */
{
  RTC_DEVICE_PROPERTY_VERSION = 0, RTC_DEVICE_PROPERTY_VERSION_MAJOR = 1,
  RTC_DEVICE_PROPERTY_VERSION_MINOR = 2, RTC_DEVICE_PROPERTY_VERSION_PATCH = 3,
  RTC_DEVICE_PROPERTY_NATIVE_RAY4_SUPPORTED = 32, RTC_DEVICE_PROPERTY_NATIVE_RAY8_SUPPORTED = 33,
  RTC_DEVICE_PROPERTY_NATIVE_RAY16_SUPPORTED = 34, RTC_DEVICE_PROPERTY_RAY_STREAM_SUPPORTED = 35,
  RTC_DEVICE_PROPERTY_RAY_MASK_SUPPORTED = 64, RTC_DEVICE_PROPERTY_BACKFACE_CULLING_ENABLED = 65,
  RTC_DEVICE_PROPERTY_FILTER_FUNCTION_SUPPORTED = 66, RTC_DEVICE_PROPERTY_IGNORE_INVALID_RAYS_ENABLED = 67,
  RTC_DEVICE_PROPERTY_TRIANGLE_GEOMETRY_SUPPORTED = 96, RTC_DEVICE_PROPERTY_QUAD_GEOMETRY_SUPPORTED = 97,
  RTC_DEVICE_PROPERTY_SUBDIVISION_GEOMETRY_SUPPORTED = 98, RTC_DEVICE_PROPERTY_CURVE_GEOMETRY_SUPPORTED = 99,
  RTC_DEVICE_PROPERTY_USER_GEOMETRY_SUPPORTED = 100,
  RTC_DEVICE_PROPERTY_TASKING_SYSTEM = 128, RTC_DEVICE_PROPERTY_JOIN_COMMIT_SUPPORTED = 129
};

enum RTCBufferType /* rtcore_buffer.h:27-42 */
{
  RTC_BUFFER_TYPE_INDEX = 0, RTC_BUFFER_TYPE_VERTEX = 1, RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE = 2,
  RTC_BUFFER_TYPE_FACE = 16, RTC_BUFFER_TYPE_LEVEL = 17,
  RTC_BUFFER_TYPE_EDGE_CREASE_INDEX = 18, RTC_BUFFER_TYPE_EDGE_CREASE_WEIGHT = 19,
  RTC_BUFFER_TYPE_VERTEX_CREASE_INDEX = 20, RTC_BUFFER_TYPE_VERTEX_CREASE_WEIGHT = 21,
  RTC_BUFFER_TYPE_HOLE = 22, RTC_BUFFER_TYPE_FLAGS = 32
};

enum RTCGeometryType /* rtcore_geometry.h:33-48 */
{
  RTC_GEOMETRY_TYPE_TRIANGLE = 0, RTC_GEOMETRY_TYPE_QUAD = 1, RTC_GEOMETRY_TYPE_SUBDIVISION = 8,
  RTC_GEOMETRY_TYPE_FLAT_LINEAR_CURVE = 17,
  RTC_GEOMETRY_TYPE_ROUND_BEZIER_CURVE = 24, RTC_GEOMETRY_TYPE_FLAT_BEZIER_CURVE = 25,
  RTC_GEOMETRY_TYPE_ROUND_BSPLINE_CURVE = 32, RTC_GEOMETRY_TYPE_FLAT_BSPLINE_CURVE = 33,
  RTC_GEOMETRY_TYPE_USER = 120, RTC_GEOMETRY_TYPE_INSTANCE = 121
};

enum RTCSubdivisionMode /* rtcore_geometry.h:51-58 */
{
  RTC_SUBDIVISION_MODE_NO_BOUNDARY = 0, RTC_SUBDIVISION_MODE_SMOOTH_BOUNDARY = 1,
  RTC_SUBDIVISION_MODE_PIN_CORNERS = 2, RTC_SUBDIVISION_MODE_PIN_BOUNDARY = 3, RTC_SUBDIVISION_MODE_PIN_ALL = 4
};

enum RTCCurveFlags /* rtcore_geometry.h:61-65 */
{
  RTC_CURVE_FLAG_NEIGHBOR_LEFT = (1 << 0), RTC_CURVE_FLAG_NEIGHBOR_RIGHT = (1 << 1)
};

enum RTCSceneFlags /* rtcore_scene.h:34-41 */
{
  RTC_SCENE_FLAG_NONE = 0, RTC_SCENE_FLAG_DYNAMIC = (1 << 0), RTC_SCENE_FLAG_COMPACT = (1 << 1),
  RTC_SCENE_FLAG_ROBUST = (1 << 2), RTC_SCENE_FLAG_CONTEXT_FILTER_FUNCTION = (1 << 3)
};

enum RTCBuildFlags /* rtcore_builder.h:61-65 */
{
  RTC_BUILD_FLAG_NONE = 0, RTC_BUILD_FLAG_DYNAMIC = (1 << 0)
};

/* ---- opaque handles -------------------------------------------------------------------------------- */
typedef struct RTCDeviceTy* RTCDevice;     /* rtcore_device.h:26 */
typedef struct RTCBufferTy* RTCBuffer;     /* rtcore_buffer.h:45 */
typedef struct RTCSceneTy* RTCScene;       /* rtcore_geometry.h:27 */
typedef struct RTCGeometryTy* RTCGeometry; /* rtcore_geometry.h:30 */
typedef struct RTCBVHTy* RTCBVH;           /* rtcore_builder.h:26 */
typedef struct RTCThreadLocalAllocatorTy* RTCThreadLocalAllocator;

/* ---- bounds (rtcore_common.h:179-190) ---------------------------------------------------------------- */
struct RTC_ALIGN(16) RTCBounds
{
  float lower_x, lower_y, lower_z, align0;
  float upper_x, upper_y, upper_z, align1;
};

struct RTC_ALIGN(16) RTCLinearBounds
{
  struct RTCBounds bounds0;
  struct RTCBounds bounds1;
};

/* ---- ray / hit records (rtcore_ray.h:26-64).  The single-ray record is the unit of every stream the
 * device kernels consume: 48-byte ray followed by a 32-byte hit, 80 bytes, 16-byte aligned. ------------- */
struct RTC_ALIGN(16) RTCRay
{
  float org_x, org_y, org_z; /* origin */
  float tnear;               /* start of the ray segment */
  float dir_x, dir_y, dir_z; /* direction (not required to be normalised) */
  float time;                /* motion-blur time, unused by this library */
  float tfar;                /* end of the segment; overwritten with the hit distance, -inf when occluded */
  unsigned int mask;
  unsigned int id;
  unsigned int flags;
};

struct RTCHit
{
  float Ng_x, Ng_y, Ng_z; /* unnormalised geometry normal */
  float u, v;             /* barycentric / patch coordinates */
  unsigned int primID;
  unsigned int geomID;    /* RTC_INVALID_GEOMETRY_ID while nothing has been hit */
  unsigned int instID[RTC_MAX_INSTANCE_LEVEL_COUNT];
};

struct RTCRayHit
{
  struct RTCRay ray;
  struct RTCHit hit;
};

/* Fixed-width SoA packets (rtcore_ray.h:67-186).  Only their layout matters here; the entry points
 * taking them are not on the device path. */
#define RTC_DECLARE_PACKET(W, ALIGNMENT)                                                        \
  struct RTC_ALIGN(ALIGNMENT) RTCRay##W                                                         \
  {                                                                                             \
    float org_x[W], org_y[W], org_z[W], tnear[W];                                               \
    float dir_x[W], dir_y[W], dir_z[W], time[W];                                                \
    float tfar[W];                                                                              \
    unsigned int mask[W], id[W], flags[W];                                                      \
  };                                                                                            \
  struct RTC_ALIGN(ALIGNMENT) RTCHit##W                                                         \
  {                                                                                             \
    float Ng_x[W], Ng_y[W], Ng_z[W];                                                            \
    float u[W], v[W];                                                                           \
    unsigned int primID[W], geomID[W], instID[RTC_MAX_INSTANCE_LEVEL_COUNT][W];                 \
  };                                                                                            \
  struct RTCRayHit##W                                                                           \
  {                                                                                             \
    struct RTCRay##W ray;                                                                       \
    struct RTCHit##W hit;                                                                       \
  }
RTC_DECLARE_PACKET(4, 16);
RTC_DECLARE_PACKET(8, 32);
RTC_DECLARE_PACKET(16, 64);

/* Pointer-SoA streams (rtcore_ray.h:190-228). */
struct RTCRayNp
{
  float *org_x, *org_y, *org_z, *tnear;
  float *dir_x, *dir_y, *dir_z, *time;
  float* tfar;
  unsigned int *mask, *id, *flags;
};

struct RTCHitNp
{
  float *Ng_x, *Ng_y, *Ng_z;
  float *u, *v;
  unsigned int *primID, *geomID;
  unsigned int* instID[RTC_MAX_INSTANCE_LEVEL_COUNT];
};

struct RTCRayHitNp
{
  struct RTCRayNp ray;
  struct RTCHitNp hit;
};

struct RTCRayN;
struct RTCHitN;
struct RTCRayHitN;

/* ---- callbacks and their argument blocks --------------------------------------------------------------- */
struct RTCIntersectContext;

struct RTCFilterFunctionNArguments /* rtcore_common.h:201-209 */
{
  int* valid;
  void* geometryUserPtr;
  const struct RTCIntersectContext* context;
  struct RTCRayN* ray;
  struct RTCHitN* hit;
  unsigned int N;
};
typedef void (*RTCFilterFunctionN)(const struct RTCFilterFunctionNArguments* args);

struct RTCIntersectContext /* rtcore_common.h:215-220 */
{
  enum RTCIntersectContextFlags flags;
  RTCFilterFunctionN filter;
  unsigned int instID[RTC_MAX_INSTANCE_LEVEL_COUNT];
};

RTC_FORCEINLINE void rtcInitIntersectContext(struct RTCIntersectContext* context) /* rtcore_common.h:223-228 */
{
  context->flags = RTC_INTERSECT_CONTEXT_FLAG_INCOHERENT;
  context->filter = NULL;
  context->instID[0] = RTC_INVALID_GEOMETRY_ID;
}

typedef void (*RTCErrorFunction)(void* userPtr, enum RTCError code, const char* str);  /* rtcore_device.h:83 */
typedef bool (*RTCMemoryMonitorFunction)(void* ptr, ssize_t bytes, bool post);         /* rtcore_device.h:89 */
typedef bool (*RTCProgressMonitorFunction)(void* ptr, double n);                       /* rtcore_scene.h:74 */

struct RTCBoundsFunctionArguments /* rtcore_geometry.h:68-74 */
{
  void* geometryUserPtr;
  unsigned int primID;
  unsigned int timeStep;
  struct RTCBounds* bounds_o;
};
typedef void (*RTCBoundsFunction)(const struct RTCBoundsFunctionArguments* args);

struct RTCIntersectFunctionNArguments /* rtcore_geometry.h:80-88 */
{
  int* valid;
  void* geometryUserPtr;
  unsigned int primID;
  struct RTCIntersectContext* context;
  struct RTCRayHitN* rayhit;
  unsigned int N;
};
typedef void (*RTCIntersectFunctionN)(const struct RTCIntersectFunctionNArguments* args);

struct RTCOccludedFunctionNArguments /* rtcore_geometry.h:94-102 */
{
  int* valid;
  void* geometryUserPtr;
  unsigned int primID;
  struct RTCIntersectContext* context;
  struct RTCRayN* ray;
  unsigned int N;
};
typedef void (*RTCOccludedFunctionN)(const struct RTCOccludedFunctionNArguments* args);

/* Displacement callback: invoked on the host while rtcCommitScene tessellates a subdivision geometry
 * (rtcore_geometry.h:108-126; protocol kernels/subdiv/subdivpatch1base_eval.cpp:132-149). */
struct RTCDisplacementFunctionNArguments
{
  void* geometryUserPtr;
  RTCGeometry geometry;
  unsigned int primID;
  unsigned int timeStep;
  const float* u;
  const float* v;
  const float* Ng_x;
  const float* Ng_y;
  const float* Ng_z;
  float* P_x;
  float* P_y;
  float* P_z;
  unsigned int N;
};
typedef void (*RTCDisplacementFunctionN)(const struct RTCDisplacementFunctionNArguments* args);

struct RTCInterpolateArguments /* rtcore_geometry.h:234-249 */
{
  RTCGeometry geometry;
  unsigned int primID;
  float u;
  float v;
  enum RTCBufferType bufferType;
  unsigned int bufferSlot;
  float* P;
  float* dPdu;
  float* dPdv;
  float* ddPdudu;
  float* ddPdvdv;
  float* ddPdudv;
  unsigned int valueCount;
};

struct RTCInterpolateNArguments /* rtcore_geometry.h:318-335 */
{
  RTCGeometry geometry;
  const void* valid;
  const unsigned int* primIDs;
  const float* u;
  const float* v;
  unsigned int N;
  enum RTCBufferType bufferType;
  unsigned int bufferSlot;
  float* P;
  float* dPdu;
  float* dPdv;
  float* ddPdudu;
  float* ddPdvdv;
  float* ddPdudv;
  unsigned int valueCount;
};

/* ---- device (rtcore_device.h:29-92; kernels/common/rtcore.cpp:34-130) ---------------------------------- */
/* config: "key=value,..." — keys understood: gpu (HIP device ordinal), tri_accel, subdiv_accel, accel,
 * verbose, threads, isa, max_isa, set_affinity, start_threads, benchmark, hugepages (the x86-only ones are
 * accepted and ignored; kernels/common/state.cpp:241-430). */
RTC_API RTCDevice rtcNewDevice(const char* config);
RTC_API void rtcRetainDevice(RTCDevice device);
RTC_API void rtcReleaseDevice(RTCDevice device);
RTC_API ssize_t rtcGetDeviceProperty(RTCDevice device, enum RTCDeviceProperty prop);
RTC_API enum RTCError rtcGetDeviceError(RTCDevice device); /* returns and clears */
RTC_API void rtcSetDeviceErrorFunction(RTCDevice device, RTCErrorFunction error, void* userPtr);
RTC_API void rtcSetDeviceMemoryMonitorFunction(RTCDevice device, RTCMemoryMonitorFunction memoryMonitor, void* userPtr);

/* ---- buffers (rtcore_buffer.h:48-60; rtcore.cpp:132-190) ------------------------------------------------ */
RTC_API RTCBuffer rtcNewBuffer(RTCDevice device, size_t byteSize);
RTC_API RTCBuffer rtcNewSharedBuffer(RTCDevice device, void* ptr, size_t byteSize);
RTC_API void* rtcGetBufferData(RTCBuffer buffer);
RTC_API void rtcRetainBuffer(RTCBuffer buffer);
RTC_API void rtcReleaseBuffer(RTCBuffer buffer);

/* ---- geometry (rtcore_geometry.h:131-231; rtcore.cpp:1060-1466) ------------------------------------------- */
RTC_API RTCGeometry rtcNewGeometry(RTCDevice device, enum RTCGeometryType type);
RTC_API void rtcRetainGeometry(RTCGeometry geometry);
RTC_API void rtcReleaseGeometry(RTCGeometry geometry);
RTC_API void rtcCommitGeometry(RTCGeometry geometry);
RTC_API void rtcEnableGeometry(RTCGeometry geometry);
RTC_API void rtcDisableGeometry(RTCGeometry geometry);
RTC_API void rtcSetGeometryTimeStepCount(RTCGeometry geometry, unsigned int timeStepCount);
RTC_API void rtcSetGeometryVertexAttributeCount(RTCGeometry geometry, unsigned int vertexAttributeCount);
RTC_API void rtcSetGeometryMask(RTCGeometry geometry, unsigned int mask);
RTC_API void rtcSetGeometryBuildQuality(RTCGeometry geometry, enum RTCBuildQuality quality);
RTC_API void rtcSetGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot, enum RTCFormat format,
                                  RTCBuffer buffer, size_t byteOffset, size_t byteStride, size_t itemCount);
RTC_API void rtcSetSharedGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot, enum RTCFormat format,
                                        const void* ptr, size_t byteOffset, size_t byteStride, size_t itemCount);
RTC_API void* rtcSetNewGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot, enum RTCFormat format,
                                      size_t byteStride, size_t itemCount);
RTC_API void* rtcGetGeometryBufferData(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot);
RTC_API void rtcUpdateGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot);
RTC_API void rtcSetGeometryIntersectFilterFunction(RTCGeometry geometry, RTCFilterFunctionN filter);
RTC_API void rtcSetGeometryOccludedFilterFunction(RTCGeometry geometry, RTCFilterFunctionN filter);
RTC_API void rtcSetGeometryUserData(RTCGeometry geometry, void* ptr);
RTC_API void* rtcGetGeometryUserData(RTCGeometry geometry);
RTC_API void rtcSetGeometryUserPrimitiveCount(RTCGeometry geometry, unsigned int userPrimitiveCount);
RTC_API void rtcSetGeometryBoundsFunction(RTCGeometry geometry, RTCBoundsFunction bounds, void* userPtr);
RTC_API void rtcSetGeometryIntersectFunction(RTCGeometry geometry, RTCIntersectFunctionN intersect);
RTC_API void rtcSetGeometryOccludedFunction(RTCGeometry geometry, RTCOccludedFunctionN occluded);
RTC_API void rtcFilterIntersection(const struct RTCIntersectFunctionNArguments* args, const struct RTCFilterFunctionNArguments* filterArgs);
RTC_API void rtcFilterOcclusion(const struct RTCOccludedFunctionNArguments* args, const struct RTCFilterFunctionNArguments* filterArgs);
RTC_API void rtcSetGeometryInstancedScene(RTCGeometry geometry, RTCScene scene);
RTC_API void rtcSetGeometryTransform(RTCGeometry geometry, unsigned int timeStep, enum RTCFormat format, const void* xfm);
RTC_API void rtcGetGeometryTransform(RTCGeometry geometry, float time, enum RTCFormat format, void* xfm);
RTC_API void rtcSetGeometryTessellationRate(RTCGeometry geometry, float tessellationRate);
RTC_API void rtcSetGeometryTopologyCount(RTCGeometry geometry, unsigned int topologyCount);
RTC_API void rtcSetGeometrySubdivisionMode(RTCGeometry geometry, unsigned int topologyID, enum RTCSubdivisionMode mode);
RTC_API void rtcSetGeometryVertexAttributeTopology(RTCGeometry geometry, unsigned int vertexAttributeID, unsigned int topologyID);
RTC_API void rtcSetGeometryDisplacementFunction(RTCGeometry geometry, RTCDisplacementFunctionN displacement);
RTC_API void rtcInterpolate(const struct RTCInterpolateArguments* args);
RTC_API void rtcInterpolateN(const struct RTCInterpolateNArguments* args);

/* Convenience wrappers with 0, 1 and 2 derivative orders (rtcore_geometry.h:255-315). */
RTC_FORCEINLINE void rtcInterpolate2(RTCGeometry geometry, unsigned int primID, float u, float v, enum RTCBufferType bufferType,
                                     unsigned int bufferSlot, float* P, float* dPdu, float* dPdv, float* ddPdudu, float* ddPdvdv,
                                     float* ddPdudv, unsigned int valueCount)
{
  struct RTCInterpolateArguments a;
  a.geometry = geometry; a.primID = primID; a.u = u; a.v = v;
  a.bufferType = bufferType; a.bufferSlot = bufferSlot;
  a.P = P; a.dPdu = dPdu; a.dPdv = dPdv; a.ddPdudu = ddPdudu; a.ddPdvdv = ddPdvdv; a.ddPdudv = ddPdudv;
  a.valueCount = valueCount;
  rtcInterpolate(&a);
}
RTC_FORCEINLINE void rtcInterpolate1(RTCGeometry geometry, unsigned int primID, float u, float v, enum RTCBufferType bufferType,
                                     unsigned int bufferSlot, float* P, float* dPdu, float* dPdv, unsigned int valueCount)
{
  rtcInterpolate2(geometry, primID, u, v, bufferType, bufferSlot, P, dPdu, dPdv, NULL, NULL, NULL, valueCount);
}
RTC_FORCEINLINE void rtcInterpolate0(RTCGeometry geometry, unsigned int primID, float u, float v, enum RTCBufferType bufferType,
                                     unsigned int bufferSlot, float* P, unsigned int valueCount)
{
  rtcInterpolate2(geometry, primID, u, v, bufferType, bufferSlot, P, NULL, NULL, NULL, NULL, NULL, valueCount);
}

/* ---- scene (rtcore_scene.h:44-139; rtcore.cpp:192-286,1468-1475) --------------------------------------------- */
RTC_API RTCScene rtcNewScene(RTCDevice device);
RTC_API void rtcRetainScene(RTCScene scene);
RTC_API void rtcReleaseScene(RTCScene scene);
RTC_API unsigned int rtcAttachGeometry(RTCScene scene, RTCGeometry geometry);
RTC_API void rtcAttachGeometryByID(RTCScene scene, RTCGeometry geometry, unsigned int geomID);
RTC_API void rtcDetachGeometry(RTCScene scene, unsigned int geomID);
RTC_API RTCGeometry rtcGetGeometry(RTCScene scene, unsigned int geomID);
/* Fork addition: fixed tessellation level L ((2^L+1)^2 vertices per patch) and cBVH depth C
 * (rtcore_scene.h:64-65, rtcore.cpp:1468-1475).  Must be called before rtcCommitScene. */
RTC_API void rtcSetSceneLevels(RTCScene scene, unsigned subdivisionLevel, unsigned compressionLevel);
RTC_API void rtcCommitScene(RTCScene scene);
RTC_API void rtcJoinCommitScene(RTCScene scene);
RTC_API void rtcSetSceneProgressMonitorFunction(RTCScene scene, RTCProgressMonitorFunction progress, void* ptr);
RTC_API void rtcSetSceneBuildQuality(RTCScene scene, enum RTCBuildQuality quality);
RTC_API void rtcSetSceneFlags(RTCScene scene, enum RTCSceneFlags flags);
RTC_API enum RTCSceneFlags rtcGetSceneFlags(RTCScene scene);
RTC_API void rtcGetSceneBounds(RTCScene scene, struct RTCBounds* bounds_o);
RTC_API void rtcGetSceneLinearBounds(RTCScene scene, struct RTCLinearBounds* bounds_o);

/* ---- the hot path (rtcore_scene.h:92-139; rtcore.cpp:287-304,403-432,541-555,656-683) -------------------------- */
/* rayhit / ray may point to host memory (staged through the device's pinned buffers) or to HIP device
 * memory (traced in place, stream-ordered on the device's stream; see rtcore_amd.h).  Semantics equal M
 * independent single-ray calls (kernels/bvh/bvh_intersector_stream_filters.cpp:148-164): rays with
 * tnear > tfar are skipped, a miss leaves the hit record untouched, occluded rays get tfar = -inf. */
RTC_API void rtcIntersect1(RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHit* rayhit);
RTC_API void rtcIntersect1M(RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHit* rayhit, unsigned int M, size_t byteStride);
RTC_API void rtcOccluded1(RTCScene scene, struct RTCIntersectContext* context, struct RTCRay* ray);
RTC_API void rtcOccluded1M(RTCScene scene, struct RTCIntersectContext* context, struct RTCRay* ray, unsigned int M, size_t byteStride);
/* pointer streams are gathered into one device batch */
RTC_API void rtcIntersect1Mp(RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHit** rayhit, unsigned int M);
RTC_API void rtcOccluded1Mp(RTCScene scene, struct RTCIntersectContext* context, struct RTCRay** ray, unsigned int M);
/* packets / packet streams / SoA streams: the active rays are gathered into one single-ray batch */
RTC_API void rtcIntersect4(const int* valid, RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHit4* rayhit);
RTC_API void rtcIntersect8(const int* valid, RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHit8* rayhit);
RTC_API void rtcIntersect16(const int* valid, RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHit16* rayhit);
RTC_API void rtcIntersectNM(RTCScene scene, struct RTCIntersectContext* context, struct RTCRayHitN* rayhit, unsigned int N, unsigned int M, size_t byteStride);
RTC_API void rtcIntersectNp(RTCScene scene, struct RTCIntersectContext* context, const struct RTCRayHitNp* rayhit, unsigned int N);
RTC_API void rtcOccluded4(const int* valid, RTCScene scene, struct RTCIntersectContext* context, struct RTCRay4* ray);
RTC_API void rtcOccluded8(const int* valid, RTCScene scene, struct RTCIntersectContext* context, struct RTCRay8* ray);
RTC_API void rtcOccluded16(const int* valid, RTCScene scene, struct RTCIntersectContext* context, struct RTCRay16* ray);
RTC_API void rtcOccludedNM(RTCScene scene, struct RTCIntersectContext* context, struct RTCRayN* ray, unsigned int N, unsigned int M, size_t byteStride);
RTC_API void rtcOccludedNp(RTCScene scene, struct RTCIntersectContext* context, const struct RTCRayNp* ray, unsigned int N);

/* ---- BVH builder API (rtcore_builder.h:28-133): declared for link compatibility, unsupported ------------------- */
struct RTC_ALIGN(32) RTCBuildPrimitive
{
  float lower_x, lower_y, lower_z;
  unsigned int geomID;
  float upper_x, upper_y, upper_z;
  unsigned int primID;
};
typedef void* (*RTCCreateNodeFunction)(RTCThreadLocalAllocator allocator, unsigned int childCount, void* userPtr);
typedef void (*RTCSetNodeChildrenFunction)(void* nodePtr, void** children, unsigned int childCount, void* userPtr);
typedef void (*RTCSetNodeBoundsFunction)(void* nodePtr, const struct RTCBounds** bounds, unsigned int childCount, void* userPtr);
typedef void* (*RTCCreateLeafFunction)(RTCThreadLocalAllocator allocator, const struct RTCBuildPrimitive* primitives, size_t primitiveCount, void* userPtr);
typedef void (*RTCSplitPrimitiveFunction)(const struct RTCBuildPrimitive* primitive, unsigned int dimension, float position,
                                          struct RTCBounds* leftBounds, struct RTCBounds* rightBounds, void* userPtr);
struct RTCBuildArguments
{
  size_t byteSize;
  enum RTCBuildQuality buildQuality;
  enum RTCBuildFlags buildFlags;
  unsigned int maxBranchingFactor;
  unsigned int maxDepth;
  unsigned int sahBlockSize;
  unsigned int minLeafSize;
  unsigned int maxLeafSize;
  float traversalCost;
  float intersectionCost;
  RTCBVH bvh;
  struct RTCBuildPrimitive* primitives;
  size_t primitiveCount;
  size_t primitiveArrayCapacity;
  RTCCreateNodeFunction createNode;
  RTCSetNodeChildrenFunction setNodeChildren;
  RTCSetNodeBoundsFunction setNodeBounds;
  RTCCreateLeafFunction createLeaf;
  RTCSplitPrimitiveFunction splitPrimitive;
  RTCProgressMonitorFunction buildProgress;
  void* userPtr;
};
RTC_API RTCBVH rtcNewBVH(RTCDevice device);
RTC_API void* rtcBuildBVH(const struct RTCBuildArguments* args);
RTC_API void* rtcThreadLocalAlloc(RTCThreadLocalAllocator allocator, size_t bytes, size_t align);
RTC_API void rtcRetainBVH(RTCBVH bvh);
RTC_API void rtcReleaseBVH(RTCBVH bvh);

#if defined(__cplusplus)
} /* extern "C" */

inline RTCSceneFlags operator|(RTCSceneFlags a, RTCSceneFlags b) /* rtcore_scene.h:144-147 */
{
  return (RTCSceneFlags)((size_t)a | (size_t)b);
}

/* SoA accessors for RTCRayN / RTCHitN blobs of run-time width N (rtcore_ray.h:241-270): component c of
 * ray i lives at float index c*N+i; the hit block starts after the 12 ray components. */
#define RTC_SOA_FIELD(TYPE, BLOB, NAME, C) \
  RTC_FORCEINLINE TYPE& BLOB##_##NAME(BLOB* p, unsigned int N, unsigned int i) { return ((TYPE*)p)[(C) * N + i]; }
RTC_SOA_FIELD(float, RTCRayN, org_x, 0)
RTC_SOA_FIELD(float, RTCRayN, org_y, 1)
RTC_SOA_FIELD(float, RTCRayN, org_z, 2)
RTC_SOA_FIELD(float, RTCRayN, tnear, 3)
RTC_SOA_FIELD(float, RTCRayN, dir_x, 4)
RTC_SOA_FIELD(float, RTCRayN, dir_y, 5)
RTC_SOA_FIELD(float, RTCRayN, dir_z, 6)
RTC_SOA_FIELD(float, RTCRayN, time, 7)
RTC_SOA_FIELD(float, RTCRayN, tfar, 8)
RTC_SOA_FIELD(unsigned int, RTCRayN, mask, 9)
RTC_SOA_FIELD(unsigned int, RTCRayN, id, 10)
RTC_SOA_FIELD(unsigned int, RTCRayN, flags, 11)
#undef RTC_SOA_FIELD
#define RTC_SOA_HITFIELD(TYPE, NAME, C) \
  RTC_FORCEINLINE TYPE& RTCHitN_##NAME(const RTCHitN* p, unsigned int N, unsigned int i) { return ((TYPE*)p)[(C) * N + i]; }
RTC_SOA_HITFIELD(float, Ng_x, 0)
RTC_SOA_HITFIELD(float, Ng_y, 1)
RTC_SOA_HITFIELD(float, Ng_z, 2)
RTC_SOA_HITFIELD(float, u, 3)
RTC_SOA_HITFIELD(float, v, 4)
RTC_SOA_HITFIELD(unsigned int, primID, 5)
RTC_SOA_HITFIELD(unsigned int, geomID, 6)
#undef RTC_SOA_HITFIELD
RTC_FORCEINLINE unsigned int& RTCHitN_instID(const RTCHitN* p, unsigned int N, unsigned int i, unsigned int l)
{
  return ((unsigned int*)p)[7 * N + i + N * l];
}
RTC_FORCEINLINE RTCRayN* RTCRayHitN_RayN(RTCRayHitN* rh, unsigned int N) { (void)N; return (RTCRayN*)rh; }
RTC_FORCEINLINE RTCHitN* RTCRayHitN_HitN(RTCRayHitN* rh, unsigned int N) { return (RTCHitN*)&((float*)rh)[12 * N]; }

/* compile-time-width helper packets (rtcore_ray.h:273-313) */
template <int N> struct RTCRayNt
{
  float org_x[N], org_y[N], org_z[N], tnear[N];
  float dir_x[N], dir_y[N], dir_z[N], time[N];
  float tfar[N];
  unsigned int mask[N], id[N], flags[N];
};
template <int N> struct RTCHitNt
{
  float Ng_x[N], Ng_y[N], Ng_z[N];
  float u[N], v[N];
  unsigned int primID[N], geomID[N], instID[RTC_MAX_INSTANCE_LEVEL_COUNT][N];
};
template <int N> struct RTCRayHitNt
{
  RTCRayNt<N> ray;
  RTCHitNt<N> hit;
};

/* AoS <-> SoA copies (rtcore_ray.h:316-392) */
RTC_FORCEINLINE RTCRay rtcGetRayFromRayN(RTCRayN* r, unsigned int N, unsigned int i)
{
  RTCRay o;
  o.org_x = RTCRayN_org_x(r, N, i); o.org_y = RTCRayN_org_y(r, N, i); o.org_z = RTCRayN_org_z(r, N, i);
  o.tnear = RTCRayN_tnear(r, N, i);
  o.dir_x = RTCRayN_dir_x(r, N, i); o.dir_y = RTCRayN_dir_y(r, N, i); o.dir_z = RTCRayN_dir_z(r, N, i);
  o.time = RTCRayN_time(r, N, i); o.tfar = RTCRayN_tfar(r, N, i);
  o.mask = RTCRayN_mask(r, N, i); o.id = RTCRayN_id(r, N, i); o.flags = RTCRayN_flags(r, N, i);
  return o;
}
RTC_FORCEINLINE RTCHit rtcGetHitFromHitN(RTCHitN* h, unsigned int N, unsigned int i)
{
  RTCHit o;
  o.Ng_x = RTCHitN_Ng_x(h, N, i); o.Ng_y = RTCHitN_Ng_y(h, N, i); o.Ng_z = RTCHitN_Ng_z(h, N, i);
  o.u = RTCHitN_u(h, N, i); o.v = RTCHitN_v(h, N, i);
  o.primID = RTCHitN_primID(h, N, i); o.geomID = RTCHitN_geomID(h, N, i);
  for (unsigned int l = 0; l < RTC_MAX_INSTANCE_LEVEL_COUNT; l++) o.instID[l] = RTCHitN_instID(h, N, i, l);
  return o;
}
RTC_FORCEINLINE void rtcCopyHitToHitN(RTCHitN* h, const RTCHit* s, unsigned int N, unsigned int i)
{
  RTCHitN_Ng_x(h, N, i) = s->Ng_x; RTCHitN_Ng_y(h, N, i) = s->Ng_y; RTCHitN_Ng_z(h, N, i) = s->Ng_z;
  RTCHitN_u(h, N, i) = s->u; RTCHitN_v(h, N, i) = s->v;
  RTCHitN_primID(h, N, i) = s->primID; RTCHitN_geomID(h, N, i) = s->geomID;
  for (unsigned int l = 0; l < RTC_MAX_INSTANCE_LEVEL_COUNT; l++) RTCHitN_instID(h, N, i, l) = s->instID[l];
}
RTC_FORCEINLINE RTCRayHit rtcGetRayHitFromRayHitN(RTCRayHitN* rh, unsigned int N, unsigned int i)
{
  RTCRayHit o;
  o.ray = rtcGetRayFromRayN(RTCRayHitN_RayN(rh, N), N, i);
  o.hit = rtcGetHitFromHitN(RTCRayHitN_HitN(rh, N), N, i);
  return o;
}
#endif /* __cplusplus */

#endif /* EMBREE3_AMD_RTCORE_H */
