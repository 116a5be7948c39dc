// ORACLE / TEST INFRASTRUCTURE ONLY - boundary proof against the reference's own callers.
//
// Stands in for the tutorials' windowing framework (tutorials/common/tutorial/tutorial.cpp: GLFW window, command line,
// scene loading) so that a tutorial's UNCHANGED device code - tutorials/<name>/<name>_device.cpp plus
// tutorials/common/tutorial/tutorial_device.cpp, compiled from the reference tree where they lie, against THIS repository's
// include/embree3 (oracle/Makefile builds a symlink farm so that their relative #include "../../../include/embree3/rtcore.h"
// lands on our header) - can be linked against embree-compressed_amd/lib/libembree3.so and render one frame.
// It defines the framework's globals the device code refers to (tutorial.cpp:45-74) and starts the reference's internal
// task scheduler, which the reference's libembree3.so would have created in rtcNewDevice (kernels/common/state.cpp,
// exported through kernels/export.linux.map:3): a tutorial built against this library links common/tasking + common/sys itself.
//
//   tut_<name> <cfg> <width> <height> <out.raw> fromx fromy fromz tox toy toz [fov]
#include "tutorials/common/tutorial/tutorial_device.h"
#include "common/tasking/taskscheduler.h"
#include "common/sys/sysinfo.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace embree {
extern "C" {
void device_init(char* cfg);
void device_render(int* pixels, const unsigned int width, const unsigned int height, const float time, const ISPCCamera& camera);
void device_cleanup();
// framework state referenced by the device code
float g_debug = 0.0f;
Mode g_mode = MODE_NORMAL;
ISPCScene* g_ispc_scene = nullptr;
float scale = 1.0f / 1000000.0f;
bool g_changed = false;
int64_t get_tsc() { return read_tsc(); }
unsigned int g_numThreads = 0;
RTCIntersectContextFlags g_iflags_coherent = RTC_INTERSECT_CONTEXT_FLAG_COHERENT;
RTCIntersectContextFlags g_iflags_incoherent = RTC_INTERSECT_CONTEXT_FLAG_INCOHERENT;
RayStats* g_stats = nullptr;
unsigned g_subdivisionLevel = 5;
unsigned g_compressionLevel = 2;
bool g_adjustedIncoherentBench = false;
bool g_adjustedCoherentBench = false;
bool g_scroll_cams = false;
unsigned g_curr_cam = 0;
unsigned g_num_cams = 0;
void tutorial_error_handler(void* userPtr, RTCError code, const char* str)
{
  if (code == RTC_ERROR_NONE) return;
  fprintf(stderr, "embree error %d: %s\n", (int)code, str ? str : "");
  exit(3);
}
}
} // namespace embree

using namespace embree;

int main(int argc, char** argv)
{
  if (argc < 11) {
    fprintf(stderr, "usage: %s cfg width height out.raw fromx fromy fromz tox toy toz [fov] [subdivLevel compressionLevel]\n", argv[0]);
    return 2;
  }
  const unsigned w = (unsigned)atoi(argv[2]), h = (unsigned)atoi(argv[3]);
  Camera camera;
  camera.from = Vec3fa((float)atof(argv[5]), (float)atof(argv[6]), (float)atof(argv[7]));
  camera.to = Vec3fa((float)atof(argv[8]), (float)atof(argv[9]), (float)atof(argv[10]));
  if (argc > 11) camera.fov = (float)atof(argv[11]);
  if (argc > 13) { g_subdivisionLevel = (unsigned)atoi(argv[12]); g_compressionLevel = (unsigned)atoi(argv[13]); }
  TaskScheduler::create(getNumberOfLogicalThreads(), false, true); // what rtcNewDevice does inside the reference's library (state.cpp / device.cpp)
  g_stats = (RayStats*)alignedMalloc(TaskScheduler::threadCount() * sizeof(RayStats), 64);
  for (size_t i = 0; i < TaskScheduler::threadCount(); i++) g_stats[i].numRays = 0;
  device_init(argv[1]);
  std::vector<int> px((size_t)w * h, 0);
  device_render(px.data(), w, h, 0.0f, camera.getISPCCamera(w, h));
  FILE* f = fopen(argv[4], "wb");
  if (!f) return 4;
  fwrite(px.data(), 4, px.size(), f);
  fclose(f);
  long long rays = 0;
  for (size_t i = 0; i < TaskScheduler::threadCount(); i++) rays += g_stats[i].numRays;
  printf("%s: %ux%u frame rendered by the tutorial's own device code, %lld rays, %zu host threads\n", argv[0], w, h, rays, TaskScheduler::threadCount());
  device_cleanup();
  alignedFree(g_stats);
  TaskScheduler::destroy();
  return 0;
}
