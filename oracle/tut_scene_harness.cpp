// ORACLE / TEST INFRASTRUCTURE ONLY - boundary proof against the reference's scene-loading tutorials (viewer_stream, pathtracer).
//
// Stands in for SceneLoadingTutorialApplication::main (tutorials/common/tutorial/tutorial.cpp:1087-1155: command line, GLFW window):
// loads a Wavefront .obj with the reference's own loader in subdivision mode (scenegraph/obj_loader.cpp, as `-i bomberman.obj` with
// `--compress.leaf` does), flattens it, builds the ISPCScene with the reference's own scene_device.cpp, sets the fork's levels like
// set_scene (tutorial.cpp:726-734) and renders ONE frame with the tutorial's UNCHANGED device code (tutorials/<name>/<name>_device.cpp).
// Everything of the reference is compiled where it lies, against THIS repository's include/embree3 (oracle/Makefile, symlink farm),
// and linked against embree-compressed_amd/lib/libembree3.so.  The harness only defines the framework globals the device code
// refers to (tutorial.cpp:45-74, pathtracer.cpp:21-24) and starts the reference's internal task scheduler.
//
//   tut_<name> <cfg> <width> <height> <out.raw> <scene.obj> fromx fromy fromz tox toy toz fov subdivLevel compressionLevel [spp [threads]]
// With spp given (pathtracer) an ambient and a directional light are added the way `--ambientlight 0.15 0.15 0.15 --directionallight -1 -1 -1 0.6 0.6 0.6`
// does (tutorial.cpp:413-430); threads = host threads of the tutorial's task scheduler (default: all).
#include "tutorials/common/tutorial/tutorial_device.h"
#include "tutorials/common/tutorial/scene_device.h"
#include "tutorials/common/tutorial/camera.h"
#include "tutorials/common/scenegraph/obj_loader.h"
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
int g_spp = 1;
bool g_accumulate = false;
void tutorial_error_handler(void* userPtr, RTCError code, const char* str)
{
  if (code == RTC_ERROR_NONE) return;
  fprintf(stderr, "embree error %d: %s\n", (int)code, str ? str : "");
  exit(3);
}
void progressStart() {}
bool progressMonitor(void* ptr, const double n) { return true; }
void progressEnd() {}
}
} // namespace embree

using namespace embree;

int main(int argc, char** argv)
{
  if (argc < 15) {
    fprintf(stderr, "usage: %s cfg width height out.raw scene.obj fromx fromy fromz tox toy toz fov subdivLevel compressionLevel [spp [threads]]\n", argv[0]);
    return 2;
  }
  const unsigned w = (unsigned)atoi(argv[2]), h = (unsigned)atoi(argv[3]);
  Camera camera;
  camera.from = Vec3fa((float)atof(argv[6]), (float)atof(argv[7]), (float)atof(argv[8]));
  camera.to = Vec3fa((float)atof(argv[9]), (float)atof(argv[10]), (float)atof(argv[11]));
  camera.fov = (float)atof(argv[12]);
  g_subdivisionLevel = (unsigned)atoi(argv[13]);
  g_compressionLevel = (unsigned)atoi(argv[14]);
  if (argc > 15) g_spp = atoi(argv[15]);
  const size_t threads = argc > 16 && atoi(argv[16]) > 0 ? (size_t)atoi(argv[16]) : getNumberOfLogicalThreads();
  TaskScheduler::create(threads, false, true); // what rtcNewDevice does inside the reference's library (state.cpp / device.cpp)
  g_stats = (RayStats*)alignedMalloc(TaskScheduler::threadCount() * sizeof(RayStats), 64);
  for (size_t i = 0; i < TaskScheduler::threadCount(); i++) g_stats[i].numRays = 0;
  try {
    // SceneLoadingTutorialApplication::main: load (subdivision mode), flatten, set_scene
    Ref<SceneGraph::GroupNode> scene = new SceneGraph::GroupNode;
    scene->add(loadOBJ(FileName(argv[5]), true));
    if (argc > 15) { // tutorial.cpp:413-430
      scene->add(new SceneGraph::LightNode(new SceneGraph::AmbientLight(Vec3fa(0.15f, 0.15f, 0.15f))));
      scene->add(new SceneGraph::LightNode(new SceneGraph::DirectionalLight(Vec3fa(-1.f, -1.f, -1.f), Vec3fa(0.6f, 0.6f, 0.6f))));
    }
    TutorialScene obj_scene;
    obj_scene.add(SceneGraph::flatten(scene, SceneGraph::INSTANCING_NONE));
    scene = nullptr;
    ISPCScene* ispc_scene = new ISPCScene(&obj_scene);
    g_ispc_scene = ispc_scene;
    g_compressionLevel = min(g_compressionLevel, g_subdivisionLevel); // tutorial.cpp:731-733
    g_ispc_scene->subdivisionLevel = g_subdivisionLevel;
    g_ispc_scene->compressionLevel = g_compressionLevel;
    device_init(argv[1]);
    std::vector<int> px((size_t)w * h, 0);
    device_render(px.data(), w, h, 0.0f, camera.getISPCCamera(w, h));
    FILE* f = fopen(argv[4], "wb");
    if (!f) return 4;
    fwrite(px.data(), 4, px.size(), f);
    fclose(f);
    long long rays = 0;
    for (size_t i = 0; i < TaskScheduler::threadCount(); i++) rays += g_stats[i].numRays;
    printf("%s: %ux%u frame rendered by the tutorial's own device code, %u geometries, %u materials, levels %u/%u, %lld rays, %zu host threads\n", argv[0], w, h,
           g_ispc_scene->numGeometries, g_ispc_scene->numMaterials, g_subdivisionLevel, g_compressionLevel, rays, TaskScheduler::threadCount());
    device_cleanup();
    delete ispc_scene;
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  alignedFree(g_stats);
  TaskScheduler::destroy();
  return 0;
}
