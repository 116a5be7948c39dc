/*
 * Row f2: what an UNCHANGED embree application sees when its harness threads call
 * rtcIntersect1 one ray at a time.  T threads trace N rays each against a small triangle scene; the library combines
 * the calls that are pending together into shared launches (include/embree3/rtcore_amd.h, RTCAMDDeviceProperty).
 * Prints calls/s, launches, and checks every hit against the closed-form answer.
 *   gcc -std=c99 -D_POSIX_C_SOURCE=200112L -O2 -Iinclude examples/small_calls_mt.c -Lembree-compressed_amd/lib \
 *       -lembree3 -lm -lpthread -Wl,-rpath,$PWD/embree-compressed_amd/lib -o /tmp/small_calls_mt && /tmp/small_calls_mt 64 2000
 */
#include <embree3/rtcore.h>
#include <embree3/rtcore_amd.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static RTCScene g_scene;
static int g_rays_per_thread = 1000;

typedef struct { int id; int bad; } Job;

static void* worker(void* p)
{
  Job* job = (Job*)p;
  struct RTCIntersectContext ctx;
  rtcInitIntersectContext(&ctx);
  unsigned s = 12345u + 977u * (unsigned)job->id;
  for (int i = 0; i < g_rays_per_thread; i++) {
    /* rays from z=-1 towards a point of the square [0,1]^2 x {0}; the scene is that square (two triangles) */
    s = s * 1664525u + 1013904223u; const float x = (float)(s >> 8) / 16777216.f * 1.5f - 0.25f;
    s = s * 1664525u + 1013904223u; const float y = (float)(s >> 8) / 16777216.f * 1.5f - 0.25f;
    struct RTCRayHit rh;
    memset(&rh, 0, sizeof(rh));
    rh.ray.org_x = x; rh.ray.org_y = y; rh.ray.org_z = -1.f;
    rh.ray.dir_z = 1.f;
    rh.ray.tfar = INFINITY; rh.ray.mask = 0xFFFFFFFFu;
    rh.hit.geomID = rh.hit.primID = rh.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
    rtcIntersect1(g_scene, &ctx, &rh);
    const int inside = x > 0.001f && x < 0.999f && y > 0.001f && y < 0.999f;
    const int outside = x < -0.001f || x > 1.001f || y < -0.001f || y > 1.001f;
    const int hit = rh.hit.geomID != RTC_INVALID_GEOMETRY_ID;
    if ((inside && (!hit || fabsf(rh.ray.tfar - 1.f) > 1e-5f)) || (outside && hit)) job->bad++;
  }
  return NULL;
}

int main(int argc, char** argv)
{
  const int T = argc > 1 ? atoi(argv[1]) : 64;
  g_rays_per_thread = argc > 2 ? atoi(argv[2]) : 1000;
  RTCDevice dev = rtcNewDevice(argc > 3 ? argv[3] : "");
  if (!dev) { fprintf(stderr, "rtcNewDevice failed: %d\n", (int)rtcGetDeviceError(NULL)); return 2; }
  g_scene = rtcNewScene(dev);
  RTCGeometry mesh = rtcNewGeometry(dev, RTC_GEOMETRY_TYPE_TRIANGLE);
  float* v = (float*)rtcSetNewGeometryBuffer(mesh, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, 12, 4);
  const float quad[12] = {0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0};
  memcpy(v, quad, sizeof(quad));
  unsigned* t = (unsigned*)rtcSetNewGeometryBuffer(mesh, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, 12, 2);
  t[0] = 0; t[1] = 1; t[2] = 2; t[3] = 0; t[4] = 2; t[5] = 3;
  rtcCommitGeometry(mesh);
  rtcAttachGeometry(g_scene, mesh);
  rtcReleaseGeometry(mesh);
  rtcCommitScene(g_scene);

  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)T);
  Job* jobs = (Job*)calloc((size_t)T, sizeof(Job));
  const long l0 = (long)rtcGetDeviceProperty(dev, (enum RTCDeviceProperty)RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES);
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < T; i++) { jobs[i].id = i; pthread_create(&th[i], NULL, worker, &jobs[i]); }
  int bad = 0;
  for (int i = 0; i < T; i++) { pthread_join(th[i], NULL); bad += jobs[i].bad; }
  clock_gettime(CLOCK_MONOTONIC, &b);
  const double sec = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
  const long launches = (long)rtcGetDeviceProperty(dev, (enum RTCDeviceProperty)RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES) - l0;
  const long calls = (long)T * g_rays_per_thread;
  printf("small_calls_mt: %d threads x %d rtcIntersect1 calls: %.0f calls/s, %ld launches (%.1f calls per launch), %d mismatches\n", T,
         g_rays_per_thread, (double)calls / sec, launches, launches ? (double)calls / (double)launches : 0.0, bad);
  free(th); free(jobs);
  rtcReleaseScene(g_scene);
  rtcReleaseDevice(dev);
  return bad ? 1 : 0;
}
