// ORACLE / TEST INFRASTRUCTURE ONLY.
//
// oracle/_ref/libref_tutorial.so: the displacement shader of the reference's displacement_geometry tutorial (BASELINE config 3)
// as a callback the tests can hand to rtcSetGeometryDisplacementFunction.  The Perlin noise itself is the REFERENCE's
// tutorials/common/tutorial/noise.cpp, compiled where it lies (oracle/Makefile); only the two small functions below are
// restated here, from tutorials/displacement_geometry/displacement_geometry_device.cpp:89-127:
//   displacement(P) = sum over freq = 1, 2, 4, .. < 40 of 1.4 * |noise(freq * P)|^2 / freq     (:89-97)
//   displacementFunction: P += displacement(P) * Ng for each of the N points                     (:111-127)
#include "tutorials/common/tutorial/noise.h"
#include "include/embree3/rtcore.h" // this repository's header (through the include farm)

using namespace embree;

static float tutorial_displacement(const Vec3fa& P)
{
  float dN = 0.0f;
  for (float freq = 1.0f; freq < 40.0f; freq *= 2) {
    float n = embree::abs(noise(freq * P)); // math.h:86 of the reference: fabsf
    dN += 1.4f * n * n / freq;
  }
  return dN;
}

extern "C" void ref_tutorial_displacementFunction(const struct RTCDisplacementFunctionNArguments* args)
{
  const float* nx = args->Ng_x;
  const float* ny = args->Ng_y;
  const float* nz = args->Ng_z;
  float* px = args->P_x;
  float* py = args->P_y;
  float* pz = args->P_z;
  for (unsigned int i = 0; i < args->N; i++) {
    const Vec3fa P = Vec3fa(px[i], py[i], pz[i]);
    const Vec3fa Ng = Vec3fa(nx[i], ny[i], nz[i]);
    const Vec3fa dP = tutorial_displacement(P) * Ng;
    px[i] += dP.x; py[i] += dP.y; pz[i] += dP.z;
  }
}

extern "C" float ref_tutorial_displacement(float x, float y, float z) { return tutorial_displacement(Vec3fa(x, y, z)); }
