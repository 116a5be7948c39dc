/*
 * BASELINE config 0 ("plumbing"): the scene of the reference's triangle_geometry tutorial — a 12-triangle cube and a
 * 2-triangle ground plane, default scene flags — driven through the plain C API exactly like the tutorial's device
 * code does (tutorials/triangle_geometry/triangle_geometry_device.cpp:27-165): rtcSetNewGeometryBuffer for vertices
 * and indices, a shared vertex-attribute buffer, one rtcIntersect1 + one rtcOccluded1 (shadow ray) per pixel.
 * The results are checked against the closed-form ray/box and ray/plane answers, and the same rays are pushed
 * through rtcIntersect1M as one stream, which must give identical records.
 *
 * C99 on purpose: it proves that include/embree3/rtcore.h is a C header.
 *   gcc -std=c99 -Iinclude examples/triangle_geometry_min.c -Lembree-compressed_amd/lib -lembree3 -lm \
 *       -Wl,-rpath,$PWD/embree-compressed_amd/lib -o /tmp/triangle_geometry_min
 */
#include <embree3/rtcore.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } Vertex;
typedef struct { unsigned v0, v1, v2; } Triangle;

static int g_errors = 0;
static void error_handler(void* user, enum RTCError code, const char* str)
{
  (void)user;
  if (code == RTC_ERROR_NONE) return;
  fprintf(stderr, "embree error %d: %s\n", (int)code, str ? str : "");
  g_errors++;
}

static float vertex_colors[8][4];

static unsigned add_cube(RTCDevice dev, RTCScene scene)
{
  RTCGeometry mesh = rtcNewGeometry(dev, RTC_GEOMETRY_TYPE_TRIANGLE);
  Vertex* v = (Vertex*)rtcSetNewGeometryBuffer(mesh, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, sizeof(Vertex), 8);
  for (int i = 0; i < 8; i++) {
    v[i].x = (i & 4) ? 1.f : -1.f;
    v[i].y = (i & 2) ? 1.f : -1.f;
    v[i].z = (i & 1) ? 1.f : -1.f;
  }
  static const unsigned idx[12][3] = {{0, 1, 2}, {1, 3, 2}, {4, 6, 5}, {5, 6, 7}, {0, 4, 1}, {1, 4, 5},
                                      {2, 3, 6}, {3, 7, 6}, {0, 2, 4}, {2, 6, 4}, {1, 5, 3}, {3, 5, 7}};
  Triangle* t = (Triangle*)rtcSetNewGeometryBuffer(mesh, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, sizeof(Triangle), 12);
  for (int i = 0; i < 12; i++) { t[i].v0 = idx[i][0]; t[i].v1 = idx[i][1]; t[i].v2 = idx[i][2]; }
  rtcSetGeometryVertexAttributeCount(mesh, 1);
  rtcSetSharedGeometryBuffer(mesh, RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, 0, RTC_FORMAT_FLOAT3, vertex_colors, 0, 16, 8);
  rtcCommitGeometry(mesh);
  unsigned id = rtcAttachGeometry(scene, mesh);
  rtcReleaseGeometry(mesh);
  return id;
}

static unsigned add_ground_plane(RTCDevice dev, RTCScene scene)
{
  RTCGeometry mesh = rtcNewGeometry(dev, RTC_GEOMETRY_TYPE_TRIANGLE);
  Vertex* v = (Vertex*)rtcSetNewGeometryBuffer(mesh, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, sizeof(Vertex), 4);
  v[0].x = -10; v[0].y = -2; v[0].z = -10;
  v[1].x = -10; v[1].y = -2; v[1].z = +10;
  v[2].x = +10; v[2].y = -2; v[2].z = -10;
  v[3].x = +10; v[3].y = -2; v[3].z = +10;
  Triangle* t = (Triangle*)rtcSetNewGeometryBuffer(mesh, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, sizeof(Triangle), 2);
  t[0].v0 = 0; t[0].v1 = 1; t[0].v2 = 2;
  t[1].v0 = 1; t[1].v1 = 3; t[1].v2 = 2;
  rtcCommitGeometry(mesh);
  unsigned id = rtcAttachGeometry(scene, mesh);
  rtcReleaseGeometry(mesh);
  return id;
}

/* closed-form nearest hit with the unit cube [-1,1]^3 (geom 0) and the ground quad y=-2, |x|,|z|<=10 (geom 1) */
static int closed_form(const float o[3], const float d[3], double* t_out)
{
  double tn = 0.0, tf = 1e30;
  int hit = 1;
  for (int k = 0; k < 3; k++) {
    if (fabs(d[k]) < 1e-12) { if (o[k] < -1 || o[k] > 1) hit = 0; continue; }
    double a = (-1.0 - o[k]) / d[k], b = (1.0 - o[k]) / d[k];
    if (a > b) { double s = a; a = b; b = s; }
    if (a > tn) tn = a;
    if (b < tf) tf = b;
  }
  double best = 1e30;
  int geom = -1;
  if (hit && tn <= tf && tn > 0) { best = tn; geom = 0; }
  if (d[1] < 0) {
    const double t = (-2.0 - o[1]) / d[1];
    const double x = o[0] + t * d[0], z = o[2] + t * d[2];
    if (t > 0 && fabs(x) <= 10 && fabs(z) <= 10 && t < best) { best = t; geom = 1; }
  }
  *t_out = best;
  return geom;
}

int main(int argc, char** argv)
{
  const char* cfg = argc > 1 ? argv[1] : "";
  RTCDevice dev = rtcNewDevice(cfg);
  if (!dev) { fprintf(stderr, "rtcNewDevice failed: error %d\n", (int)rtcGetDeviceError(NULL)); return 2; }
  rtcSetDeviceErrorFunction(dev, error_handler, NULL);
  RTCScene scene = rtcNewScene(dev);
  const unsigned cube = add_cube(dev, scene), plane = add_ground_plane(dev, scene);
  rtcCommitScene(scene);
  if (cube != 0 || plane != 1) { fprintf(stderr, "unexpected geometry ids %u %u\n", cube, plane); return 1; }

  /* camera of the tutorial: from (1.5,1.5,-1.5) looking at the origin, fov 90 (triangle_geometry.cpp:27-28) */
  const int W = 64, H = 48;
  const float from[3] = {1.5f, 1.5f, -1.5f};
  float Z[3] = {-from[0], -from[1], -from[2]}, up[3] = {0, 1, 0}, U[3], V[3];
  float l = sqrtf(Z[0] * Z[0] + Z[1] * Z[1] + Z[2] * Z[2]);
  for (int k = 0; k < 3; k++) Z[k] /= l;
  U[0] = up[1] * Z[2] - up[2] * Z[1]; U[1] = up[2] * Z[0] - up[0] * Z[2]; U[2] = up[0] * Z[1] - up[1] * Z[0];
  l = sqrtf(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
  for (int k = 0; k < 3; k++) U[k] /= l;
  V[0] = Z[1] * U[2] - Z[2] * U[1]; V[1] = Z[2] * U[0] - Z[0] * U[2]; V[2] = Z[0] * U[1] - Z[1] * U[0];

  struct RTCRayHit* stream = NULL;
  if (posix_memalign((void**)&stream, 16, sizeof(struct RTCRayHit) * (size_t)(W * H))) return 2;
  struct RTCIntersectContext ctx;
  rtcInitIntersectContext(&ctx);
  int bad = 0, nhit = 0, nshadow = 0;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      struct RTCRayHit rh;
      memset(&rh, 0, sizeof(rh));
      float d[3];
      for (int k = 0; k < 3; k++) d[k] = ((float)x - 0.5f * W) * -U[k] + (0.5f * H - (float)y) * V[k] + 0.5f * H * Z[k];
      l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      rh.ray.org_x = from[0]; rh.ray.org_y = from[1]; rh.ray.org_z = from[2];
      rh.ray.dir_x = d[0] / l; rh.ray.dir_y = d[1] / l; rh.ray.dir_z = d[2] / l;
      rh.ray.tnear = 0.f; rh.ray.tfar = INFINITY; rh.ray.mask = 0xFFFFFFFFu;
      rh.hit.geomID = rh.hit.primID = rh.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
      stream[y * W + x] = rh;
      rtcIntersect1(scene, &ctx, &rh);
      const float o[3] = {from[0], from[1], from[2]}, dd[3] = {rh.ray.dir_x, rh.ray.dir_y, rh.ray.dir_z};
      double t;
      const int geom = closed_form(o, dd, &t);
      if (geom < 0) { if (rh.hit.geomID != RTC_INVALID_GEOMETRY_ID) bad++; continue; }
      nhit++;
      if (rh.hit.geomID != (unsigned)geom || fabs(rh.ray.tfar - t) > 1e-4 * t) { bad++; continue; }
      if (geom == 0 && rh.hit.primID >= 12) bad++;
      if (geom == 1 && rh.hit.primID >= 2) bad++;
      /* shadow ray towards the light, as renderPixelStandard does */
      struct RTCRay sh;
      memset(&sh, 0, sizeof(sh));
      const float ld = 1.f / sqrtf(3.f);
      sh.org_x = o[0] + rh.ray.tfar * dd[0]; sh.org_y = o[1] + rh.ray.tfar * dd[1]; sh.org_z = o[2] + rh.ray.tfar * dd[2];
      sh.dir_x = ld; sh.dir_y = ld; sh.dir_z = ld;
      sh.tnear = 0.001f; sh.tfar = INFINITY; sh.mask = 0xFFFFFFFFu;
      rtcOccluded1(scene, &ctx, &sh);
      const float so[3] = {sh.org_x, sh.org_y, sh.org_z}, sd[3] = {ld, ld, ld};
      double ts;
      const int sg = closed_form(so, sd, &ts);
      const int occluded = sh.tfar < 0.f;
      /* points on the cube's own surface are tnear-limited: only count clear-cut cases */
      if (geom == 1) { if (occluded != (sg == 0 && ts > 0.002)) bad++; }
      nshadow += occluded;
    }
  /* the same rays as one stream */
  rtcIntersect1M(scene, &ctx, stream, (unsigned)(W * H), sizeof(struct RTCRayHit));
  int nstream = 0;
  for (int i = 0; i < W * H; i++) nstream += stream[i].hit.geomID != RTC_INVALID_GEOMETRY_ID;
  if (nstream != nhit) { fprintf(stderr, "stream hits %d != single-ray hits %d\n", nstream, nhit); bad++; }
  /* a packet of four of the same rays (lane 2 masked out) must reproduce the single-ray answers */
  {
    struct RTCRayHit4 p4;
    int valid[4] = {-1, -1, 0, -1};
    const int pick[4] = {0, W * H / 2 + W / 2, 5, W * H - 1};
    for (int l = 0; l < 4; l++) {
      const struct RTCRayHit* s = &stream[pick[l]]; /* already traced: its hit must be found again at the same distance */
      p4.ray.org_x[l] = s->ray.org_x; p4.ray.org_y[l] = s->ray.org_y; p4.ray.org_z[l] = s->ray.org_z; p4.ray.tnear[l] = 0.f;
      p4.ray.dir_x[l] = s->ray.dir_x; p4.ray.dir_y[l] = s->ray.dir_y; p4.ray.dir_z[l] = s->ray.dir_z; p4.ray.time[l] = 0.f;
      p4.ray.tfar[l] = INFINITY; p4.ray.mask[l] = 0xFFFFFFFFu; p4.ray.id[l] = (unsigned)l; p4.ray.flags[l] = 0;
      p4.hit.geomID[l] = p4.hit.primID[l] = p4.hit.instID[0][l] = RTC_INVALID_GEOMETRY_ID;
    }
    rtcIntersect4(valid, scene, &ctx, &p4);
    for (int l = 0; l < 4; l++) {
      const struct RTCRayHit* s = &stream[pick[l]];
      if (l == 2) { if (p4.hit.geomID[l] != RTC_INVALID_GEOMETRY_ID || p4.ray.tfar[l] != INFINITY) { fprintf(stderr, "masked lane touched\n"); bad++; } }
      else if (p4.hit.geomID[l] != s->hit.geomID || p4.hit.primID[l] != s->hit.primID || (s->hit.geomID != RTC_INVALID_GEOMETRY_ID && p4.ray.tfar[l] != s->ray.tfar)) {
        fprintf(stderr, "packet lane %d differs from the stream\n", l); bad++;
      }
    }
  }
  /* error convention: an entry point that is not provided reports through the device error state */
  rtcNewBVH(dev);
  if (g_errors != 1 || rtcGetDeviceError(dev) != RTC_ERROR_INVALID_OPERATION) { fprintf(stderr, "error convention broken\n"); bad++; }
  free(stream);
  rtcReleaseScene(scene);
  rtcReleaseDevice(dev);
  printf("triangle_geometry_min: %d pixels, %d hits, %d shadowed, %d mismatches\n", W * H, nhit, nshadow, bad);
  return bad ? 1 : 0;
}
