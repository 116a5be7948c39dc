/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's single-ray traversal hot path, used solely as the checker in
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in embree-compressed_amd/
 * includes, links or executes this code.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - triangle path: pinned by the reference's known-answer test TriangleHitTest (tutorials/verify/verify.cpp:
 *     2118-2205, restated in tests/test_oracle_kat.py), by the reference outputs recorded in SURVEY.md section 8d
 *     (227 188 hits / sum primID 10 389 122 on the 1 M-ray bomberman set), and by oracle/_ref (the reference's own
 *     common/math + common/simd headers compiled in place) for the arithmetic primitives.
 *   - cBVH / GridSOA paths: "parity unpinned" beyond the hit-count anchors of SURVEY.md section 6.
 */
#ifndef EMBREE_ORACLE_H
#define EMBREE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* mode 0: BVH8 + Triangle4v, robust traversal, Pluecker test   (bvh_intersector1_bvh8.cpp:29)
 * mode 1: BVH8 + Triangle4,  fast traversal,   Moeller test     (bvh_intersector1_bvh8.cpp:27)
 * verts: nverts x 3 floats; tris: ntris x 3 vertex indices; geomIDs / primIDs: per triangle (NULL -> 0 / index). */
orc_scene* orc_scene_new_triangles(const float* verts, size_t nverts, const uint32_t* tris, const uint32_t* geomIDs,
                                   const uint32_t* primIDs, size_t ntris, int mode);
void orc_scene_free(orc_scene* s);

/* Subdivision scenes over leaf records exported by the product (rtcamdGetAccelData kind 2; formats in
 * embree-compressed_amd/csrc/accel.h).  mode 2: eager grid cells (GridCell, stride 160); mode 3/4/5/6: fork cBVH blobs
 * (box / leaf / grid / full) of `stride` bytes with `levels` = compression level C.  The oracle builds its own BVH over them. */
orc_scene* orc_scene_new_subdiv(const void* blobs, size_t stride, size_t count, int mode, unsigned levels);
/* Same, but traversing the product's own outer BVH8 (rtcamdGetAccelData kind 0 + rtcamdGetAccelRoot): required for the
 * order-dependent fork modes box / leaf, see subdiv_oracle.inc. */
orc_scene* orc_scene_new_subdiv_qbvh(const void* blobs, size_t stride, size_t count, int mode, unsigned levels, const void* qnodes,
                                     size_t nqnodes, uint32_t rootRef);

/* rayhit: embree3 RTCRayHit layout (80 bytes); ray: RTCRay layout (48 bytes). */
void orc_intersect1(const orc_scene* s, void* rayhit, uint32_t instID);
void orc_occluded1(const orc_scene* s, void* ray);

/* Stream versions with the semantics of rtcIntersect1M / rtcOccluded1M (rays with tnear > tfar are skipped).
 * nthreads > 1 splits the range into blocks of 1024 rays over that many pthreads (verify.cpp:3850-3857 style). */
void orc_intersect1M(const orc_scene* s, void* rayhits, uint32_t M, size_t byteStride, uint32_t instID, int nthreads);
void orc_occluded1M(const orc_scene* s, void* rays, uint32_t M, size_t byteStride, int nthreads);

/* Work counters of the last single-threaded orc_intersect1M call (nodes, leaves, blocks), like EMBREE_STAT_COUNTERS. */
void orc_get_counters(unsigned long long out[3]);

/* Arithmetic primitives, exported so that tests can compare them with oracle/_ref. */
float orc_rcp(float x);
float orc_rsqrt(float x);                        /* math.h:86-97 */
float orc_dot3fa(const float a[3], const float b[3]); /* Vec3fa dot = dpps 0x7F, vec3fa.h:289-291 */
float orc_length3(const float a[3]);             /* vec3fa.h:310 */
void orc_normalize3(const float a[3], float o[3]); /* vec3fa.h:311 */
/* Arithmetic of the fork's cBVH helpers (compressed.h, compressed_help.h, compressed_leaf.h).
 * 0 (default) = the REFERENCE's: rcp / rcp_safe = rcpss + Newton step, getDelta = rcp(16), normalize / length on Vec3fa =
 *               dpps + rsqrtss + Newton step / sqrtss - each pinned bit for bit to oracle/_ref/libref_prims.so;
 * 1           = the PRODUCT's device arithmetic (IEEE division, exact 1/16, fma-chain squared length): regression tests that
 *               want the HIP kernels bit for bit select this.  Not thread-safe: set it before tracing. */
void orc_set_fork_arith(int mode);
int orc_get_fork_arith(void);
void orc_set_fork_trace(int on); /* debugging aid: print the cell tests of the following (single-threaded) calls */
void orc_cross(const float a[3], const float b[3], float out[3]);
float orc_dot(const float a[3], const float b[3]);
void orc_stable_triangle_normal(const float a[3], const float b[3], const float c[3], float out[3]);
/* one Triangle4v block vs one ray: returns lane (0..3) of the accepted hit or -1; out = t,u,v,Ngx,Ngy,Ngz */
int orc_pluecker_block(const float v0[12], const float v1[12], const float v2[12], const float org[3], const float dir[3],
                       float tnear, float tfar, float out[6]);
int orc_moeller_block(const float v0[12], const float v1[12], const float v2[12], const float org[3], const float dir[3],
                      float tnear, float tfar, float out[6]);

/* drand48-compatible 48-bit LCG ray generator of BASELINE.md section 3 (viewer_device.cpp:367-392). */
void orc_make_random_rays(void* rayhits, uint32_t M, size_t byteStride, const float lo[3], const float hi[3], uint64_t seed,
                          int doubleEval);

#ifdef __cplusplus
}
#endif
#endif
