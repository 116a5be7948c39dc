/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see embree_oracle.h).
 *
 * CPU restatement, in plain C, of the reference's AVX2 single-ray path for triangle scenes.  SIMD lanes of
 * the reference are modelled as small arrays processed lane by lane with the same per-lane arithmetic:
 * fused multiply-adds only where the AVX2 build has them (madd/msub in common/math/vec3.h:193-212,
 * common/simd/vfloat4_sse2.h:361-364), rcp = rcpss + one Newton step (common/math/math.h:60-75).
 * Build with -ffp-contract=off so the compiler adds no further contractions.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference tree).
 */
#include "embree_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif

#define ORC_INF (__builtin_inff())
#define ORC_EMPTY 0xFFFFFFFFu
#define ORC_LEAF 0x80000000u

/* ------------------------------------------------------------------------------------------------------ */
/* arithmetic primitives                                                                                    */
/* ------------------------------------------------------------------------------------------------------ */
static inline float madd(float a, float b, float c) { return fmaf(a, b, c); }   /* a*b+c fused */
static inline float msub(float a, float b, float c) { return fmaf(a, b, -c); }  /* a*b-c fused */
static inline float nmadd(float a, float b, float c) { return fmaf(-a, b, c); } /* -a*b+c fused */

/* common/math/math.h:60-75 (AVX2 branch): r = rcpss(x); r*(2 - r*x) with fnmadd */
float orc_rcp(float x)
{
#if defined(__SSE__)
  const float r = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x)));
  return r * nmadd(r, x, 2.0f);
#else
  return 1.0f / x;
#endif
}

/* common/math/math.h:86-97: r = rsqrtss(x); 1.5*r + ((x*-0.5)*r)*(r*r), separate multiplies and one add (intrinsics, never fused) */
float orc_rsqrt(float x)
{
#if defined(__SSE__)
  const float r = _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(x)));
  volatile float a = 1.5f * r, b = ((x * -0.5f) * r) * (r * r);
  return a + b;
#else
  return 1.0f / sqrtf(x);
#endif
}

/* Vec3fa dot on an SSE4.1+ target = dpps with mask 0x7F (common/math/vec3fa.h:289-291): (x*x' + y*y') + (z*z' + 0), no fusion */
float orc_dot3fa(const float a[3], const float b[3])
{
  volatile float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
  volatile float s01 = p0 + p1, s23 = p2 + 0.0f;
  return s01 + s23;
}
float orc_length3(const float a[3]) { return sqrtf(orc_dot3fa(a, a)); }                       /* vec3fa.h:310 */
void orc_normalize3(const float a[3], float o[3])                                            /* vec3fa.h:311 */
{
  const float r = orc_rsqrt(orc_dot3fa(a, a));
  o[0] = a[0] * r; o[1] = a[1] * r; o[2] = a[2] * r;
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float xorf(float a, uint32_t sign) { return u2f(f2u(a) ^ sign); }

/* common/math/vec3.h:193 */
float orc_dot(const float a[3], const float b[3]) { return madd(a[0], b[0], madd(a[1], b[1], a[2] * b[2])); }
/* common/math/vec3.h:198 */
void orc_cross(const float a[3], const float b[3], float o[3])
{
  o[0] = msub(a[1], b[2], a[2] * b[1]);
  o[1] = msub(a[2], b[0], a[0] * b[2]);
  o[2] = msub(a[0], b[1], a[1] * b[0]);
}
/* common/math/vec3.h:200-212 */
void orc_stable_triangle_normal(const float a[3], const float b[3], const float c[3], float o[3])
{
  const float ab_x = a[2] * b[1], ab_y = a[0] * b[2], ab_z = a[1] * b[0];
  const float bc_x = b[2] * c[1], bc_y = b[0] * c[2], bc_z = b[1] * c[0];
  const float cab[3] = {msub(a[1], b[2], ab_x), msub(a[2], b[0], ab_y), msub(a[0], b[1], ab_z)};
  const float cbc[3] = {msub(b[1], c[2], bc_x), msub(b[2], c[0], bc_y), msub(b[0], c[1], bc_z)};
  o[0] = fabsf(ab_x) < fabsf(bc_x) ? cab[0] : cbc[0];
  o[1] = fabsf(ab_y) < fabsf(bc_y) ? cab[1] : cbc[1];
  o[2] = fabsf(ab_z) < fabsf(bc_z) ? cab[2] : cbc[2];
}

/* ------------------------------------------------------------------------------------------------------ */
/* data structures: BVH8 AlignedNode (kernels/bvh/bvh.h:433-594) and Triangle4v/Triangle4 blocks           */
/* (kernels/geometry/trianglev.h:156-161, triangle.h:26-)                                                   */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct
{
  uint32_t child[8];
  float lower_x[8], upper_x[8], lower_y[8], upper_y[8], lower_z[8], upper_z[8];
} onode;

typedef struct
{
  float a[3][4]; /* v0.x[4], v0.y[4], v0.z[4] */
  float b[3][4]; /* mode 0: v1 ; mode 1: e1 = v0-v1 */
  float c[3][4]; /* mode 0: v2 ; mode 1: e2 = v2-v0 */
  int32_t geomID[4], primID[4];
} oblock;

struct orc_scene
{
  int mode;
  onode* nodes;
  size_t nnodes, capnodes;
  oblock* blocks;
  size_t nblocks, capblocks;
  uint32_t root;
  /* subdivision scenes (oracle_subdiv part below): mode 2 = eager grid cells, 3/4/5 = cBVH box/leaf/grid */
  uint8_t* blobs;
  size_t blobStride;
  unsigned levels;
};
static void subdiv_leaf_intersect(const struct orc_scene* s, uint32_t blob, void* rayhit, uint32_t instID);
static int subdiv_leaf_occluded(const struct orc_scene* s, uint32_t blob, void* rayhit);

/* leaf reference: ORC_LEAF | nblocks<<27 | first block (bvh.h:1472-1482 keeps the block count in the low 4 bits) */
static inline uint32_t mk_leaf(uint32_t first, uint32_t n) { return ORC_LEAF | (n << 27) | first; }

typedef struct { float lo[3], hi[3]; uint32_t tri; } obprim;

static void box_empty(float lo[3], float hi[3]) { for (int k = 0; k < 3; k++) { lo[k] = ORC_INF; hi[k] = -ORC_INF; } }
static float box_area(const float lo[3], const float hi[3])
{
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * (dy + dz) + dy * dz;
}

typedef struct { float lo[3], hi[3]; size_t begin, end; int left, right; } obin;
typedef struct
{
  obprim* prims;
  obin* bins; size_t nbins, capbins;
  const float* verts; const uint32_t* tris; const uint32_t* geomIDs; const uint32_t* primIDs;
  orc_scene* s;
} obuild;

static int g_axis;
static int cmp_centroid(const void* pa, const void* pb)
{
  const obprim* a = (const obprim*)pa; const obprim* b = (const obprim*)pb;
  const float ca = a->lo[g_axis] + a->hi[g_axis], cb = b->lo[g_axis] + b->hi[g_axis];
  if (ca < cb) return -1; if (ca > cb) return 1;
  return a->tri < b->tri ? -1 : (a->tri > b->tri ? 1 : 0);
}

/* The oracle's own builder: object-median splits on the widest centroid axis down to <= 8 triangles (two
 * blocks, so multi-block leaves are exercised).  Deliberately different from the product's binned SAH:
 * closest hits must not depend on tree shape. */
static int bin_build(obuild* B, size_t begin, size_t end)
{
  if (B->nbins == B->capbins) { B->capbins = B->capbins ? 2 * B->capbins : 1024; B->bins = (obin*)realloc(B->bins, B->capbins * sizeof(obin)); }
  const int me = (int)B->nbins++;
  float lo[3], hi[3], clo[3], chi[3];
  box_empty(lo, hi); box_empty(clo, chi);
  for (size_t i = begin; i < end; i++)
    for (int k = 0; k < 3; k++) {
      lo[k] = fminf(lo[k], B->prims[i].lo[k]); hi[k] = fmaxf(hi[k], B->prims[i].hi[k]);
      const float c = B->prims[i].lo[k] + B->prims[i].hi[k];
      clo[k] = fminf(clo[k], c); chi[k] = fmaxf(chi[k], c);
    }
  memcpy(B->bins[me].lo, lo, 12); memcpy(B->bins[me].hi, hi, 12);
  B->bins[me].begin = begin; B->bins[me].end = end; B->bins[me].left = B->bins[me].right = -1;
  if (end - begin <= (B->s->mode >= 2 ? 1u : 8u)) return me; /* subdiv builders: one primitive per leaf (bvh_builder_subdiv.cpp:845-851) */
  int axis = 0;
  if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
  if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
  g_axis = axis;
  qsort(B->prims + begin, end - begin, sizeof(obprim), cmp_centroid);
  const size_t mid = begin + (end - begin) / 2;
  const int l = bin_build(B, begin, mid);
  const int r = bin_build(B, mid, end);
  B->bins[me].left = l; B->bins[me].right = r;
  return me;
}

static uint32_t emit_leaf(obuild* B, size_t begin, size_t end)
{
  orc_scene* s = B->s;
  const size_t n = end - begin, nb = (n + 3) / 4;
  if (s->nblocks + nb > s->capblocks) { s->capblocks = 2 * s->capblocks + nb + 64; s->blocks = (oblock*)realloc(s->blocks, s->capblocks * sizeof(oblock)); }
  const uint32_t first = (uint32_t)s->nblocks;
  for (size_t b = 0; b < nb; b++) {
    /* TriangleMv::fill, trianglev.h:104-126: unused lanes are zero vertices with geomID = primID = -1 */
    oblock* blk = &s->blocks[s->nblocks++];
    memset(blk, 0, sizeof(*blk));
    for (int l = 0; l < 4; l++) { blk->geomID[l] = -1; blk->primID[l] = -1; }
    for (int l = 0; l < 4 && begin + 4 * b + l < end; l++) {
      const uint32_t t = B->prims[begin + 4 * b + l].tri;
      const float* p0 = B->verts + 3 * (size_t)B->tris[3 * t + 0];
      const float* p1 = B->verts + 3 * (size_t)B->tris[3 * t + 1];
      const float* p2 = B->verts + 3 * (size_t)B->tris[3 * t + 2];
      for (int k = 0; k < 3; k++) {
        blk->a[k][l] = p0[k];
        if (s->mode == 0) { blk->b[k][l] = p1[k]; blk->c[k][l] = p2[k]; }
        else { blk->b[k][l] = p0[k] - p1[k]; blk->c[k][l] = p2[k] - p0[k]; } /* TriangleM ctor, triangle.h:52-53 */
      }
      blk->geomID[l] = (int32_t)(B->geomIDs ? B->geomIDs[t] : 0);
      blk->primID[l] = (int32_t)(B->primIDs ? B->primIDs[t] : t);
    }
  }
  return mk_leaf(first, (uint32_t)nb);
}

static uint32_t emit_node(obuild* B, int bn)
{
  const obin* b = &B->bins[bn];
  if (b->left < 0) return B->s->mode >= 2 ? mk_leaf(B->prims[b->begin].tri, 1) : emit_leaf(B, b->begin, b->end);
  int kids[8], nk = 0;
  kids[nk++] = b->left; kids[nk++] = b->right;
  while (nk < 8) { /* open the largest inner child */
    int best = -1; float bestA = -1.f;
    for (int i = 0; i < nk; i++) {
      if (B->bins[kids[i]].left < 0) continue;
      const float a = box_area(B->bins[kids[i]].lo, B->bins[kids[i]].hi);
      if (a > bestA) { bestA = a; best = i; }
    }
    if (best < 0) break;
    const int open = kids[best];
    kids[best] = B->bins[open].left; kids[nk++] = B->bins[open].right;
  }
  orc_scene* s = B->s;
  if (s->nnodes == s->capnodes) { s->capnodes = s->capnodes ? 2 * s->capnodes : 256; s->nodes = (onode*)realloc(s->nodes, s->capnodes * sizeof(onode)); }
  const uint32_t me = (uint32_t)s->nnodes++;
  uint32_t refs[8];
  for (int i = 0; i < nk; i++) refs[i] = emit_node(B, kids[i]);
  onode* n = &s->nodes[me]; /* after recursion: the array may have moved */
  for (int i = 0; i < 8; i++) { /* AlignedNode::clear, bvh.h:447-453: empty = (+inf,-inf), child = emptyNode */
    n->child[i] = ORC_EMPTY;
    n->lower_x[i] = n->lower_y[i] = n->lower_z[i] = ORC_INF;
    n->upper_x[i] = n->upper_y[i] = n->upper_z[i] = -ORC_INF;
  }
  for (int i = 0; i < nk; i++) {
    const obin* c = &B->bins[kids[i]];
    n->child[i] = refs[i];
    n->lower_x[i] = c->lo[0]; n->upper_x[i] = c->hi[0];
    n->lower_y[i] = c->lo[1]; n->upper_y[i] = c->hi[1];
    n->lower_z[i] = c->lo[2]; n->upper_z[i] = c->hi[2];
  }
  return me;
}

orc_scene* orc_scene_new_triangles(const float* verts, size_t nverts, const uint32_t* tris, const uint32_t* geomIDs,
                                   const uint32_t* primIDs, size_t ntris, int mode)
{
  orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
  s->mode = mode;
  s->root = ORC_EMPTY;
  obuild B; memset(&B, 0, sizeof(B));
  B.verts = verts; B.tris = tris; B.geomIDs = geomIDs; B.primIDs = primIDs; B.s = s;
  B.prims = (obprim*)malloc((ntris ? ntris : 1) * sizeof(obprim));
  size_t n = 0;
  for (size_t t = 0; t < ntris; t++) {
    int ok = 1;
    for (int k = 0; k < 3; k++) if (tris[3 * t + k] >= nverts) ok = 0;
    if (!ok) continue;
    obprim* p = &B.prims[n];
    box_empty(p->lo, p->hi);
    for (int k = 0; k < 3 && ok; k++) {
      const float* v = verts + 3 * (size_t)tris[3 * t + k];
      for (int a = 0; a < 3; a++) {
        if (!isfinite(v[a])) ok = 0;
        p->lo[a] = fminf(p->lo[a], v[a]); p->hi[a] = fmaxf(p->hi[a], v[a]);
      }
    }
    if (!ok) continue; /* TriangleMesh::valid: invalid triangles are not built */
    p->tri = (uint32_t)t;
    n++;
  }
  if (n) {
    const int root = bin_build(&B, 0, n);
    s->root = emit_node(&B, root);
  }
  free(B.prims); free(B.bins);
  return s;
}

void orc_scene_free(orc_scene* s)
{
  if (!s) return;
  free(s->nodes); free(s->blocks); free(s->blobs); free(s);
}

/* ------------------------------------------------------------------------------------------------------ */
/* ray state                                                                                                */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct
{
  float org[3], tnear, dir[3], time, tfar; uint32_t mask, id, flags; /* RTCRay, rtcore_ray.h:26-42 */
  float Ng[3], u, v; uint32_t primID, geomID, instID;                /* RTCHit, rtcore_ray.h:45-57 */
} orayhit;

typedef struct
{
  float org[3], dir[3];
  float rdir_near[3], rdir_far[3]; /* robust */
  float rdir[3], org_rdir[3];      /* fast */
  int nearIdx[3];                  /* 0: lower is the near plane, 1: upper */
  float tnear, tfar;
} travray;

/* vec3fa.h:163-165 */
static inline float zero_fix(float a) { return fabsf(a) < 1e-18f ? 1e-18f : a; }

/* TravRayBase<N,Nx,true>, node_intersector1.h:113-129 / TravRayBase<N,Nx,false>, :38-57 (AVX2) */
static void travray_init(travray* t, const orayhit* r, int robust)
{
  for (int k = 0; k < 3; k++) { t->org[k] = r->org[k]; t->dir[k] = r->dir[k]; }
  if (robust) {
    const float ulp3 = 1.0f + 3.0f * 1.1920929e-7f;
    for (int k = 0; k < 3; k++) {
      t->rdir_near[k] = 1.0f / zero_fix(r->dir[k]);
      t->rdir_far[k] = t->rdir_near[k] * ulp3;
      t->nearIdx[k] = t->rdir_near[k] >= 0.0f ? 0 : 1;
    }
  } else {
    for (int k = 0; k < 3; k++) {
      t->rdir[k] = orc_rcp(zero_fix(r->dir[k])); /* rcp_safe, vec3fa.h:166-168 */
      t->org_rdir[k] = r->org[k] * t->rdir[k];
      t->nearIdx[k] = t->rdir[k] >= 0.0f ? 0 : 1;
    }
  }
  t->tnear = fmaxf(r->tnear, 0.0f); /* bvh_intersector1.cpp:67 */
  t->tfar = fmaxf(r->tfar, 0.0f);
}

/* x86 maxps/minps return the second operand when the compare is false or unordered */
static inline float sse_max(float a, float b) { return a > b ? a : b; }
static inline float sse_min(float a, float b) { return a < b ? a : b; }

/* intersectNodeRobust, node_intersector1.h:334-349 ; intersectNode<8,8> fast, :249-286 */
static unsigned node_test(const onode* n, const travray* t, int robust, float tNear[8])
{
  const float* px[2] = {n->lower_x, n->upper_x};
  const float* py[2] = {n->lower_y, n->upper_y};
  const float* pz[2] = {n->lower_z, n->upper_z};
  unsigned mask = 0;
  for (int i = 0; i < 8; i++) {
    float nx, ny, nz, fx, fy, fz;
    if (robust) {
      nx = (px[t->nearIdx[0]][i] - t->org[0]) * t->rdir_near[0];
      ny = (py[t->nearIdx[1]][i] - t->org[1]) * t->rdir_near[1];
      nz = (pz[t->nearIdx[2]][i] - t->org[2]) * t->rdir_near[2];
      fx = (px[1 - t->nearIdx[0]][i] - t->org[0]) * t->rdir_far[0];
      fy = (py[1 - t->nearIdx[1]][i] - t->org[1]) * t->rdir_far[1];
      fz = (pz[1 - t->nearIdx[2]][i] - t->org[2]) * t->rdir_far[2];
    } else {
      nx = msub(px[t->nearIdx[0]][i], t->rdir[0], t->org_rdir[0]);
      ny = msub(py[t->nearIdx[1]][i], t->rdir[1], t->org_rdir[1]);
      nz = msub(pz[t->nearIdx[2]][i], t->rdir[2], t->org_rdir[2]);
      fx = msub(px[1 - t->nearIdx[0]][i], t->rdir[0], t->org_rdir[0]);
      fy = msub(py[1 - t->nearIdx[1]][i], t->rdir[1], t->org_rdir[1]);
      fz = msub(pz[1 - t->nearIdx[2]][i], t->rdir[2], t->org_rdir[2]);
    }
    const float tn = sse_max(sse_max(nx, ny), sse_max(nz, t->tnear)); /* max(a,b,c,d) = max(max(a,b),max(c,d)), vfloat8_avx.h */
    const float tf = sse_min(sse_min(fx, fy), sse_min(fz, t->tfar));
    tNear[i] = tn;
    if (tn <= tf) mask |= 1u << i;
  }
  return mask;
}

/* ------------------------------------------------------------------------------------------------------ */
/* leaf blocks                                                                                              */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct { float t[4], u[4], v[4], Ng[3][4]; unsigned valid; } blockhit;

/* PlueckerIntersector1<4>::intersect + PlueckerHitM::finalize, triangle_intersector_pluecker.h:79-132, :41-51 */
static unsigned pluecker4(const oblock* b, const float org[3], const float dir[3], float tnear, float tfar, blockhit* h)
{
  unsigned valid = 0;
  for (int l = 0; l < 4; l++) {
    float v0[3], v1[3], v2[3], e0[3], e1[3], e2[3], s[3], c[3], Ng[3];
    for (int k = 0; k < 3; k++) { v0[k] = b->a[k][l] - org[k]; v1[k] = b->b[k][l] - org[k]; v2[k] = b->c[k][l] - org[k]; }
    for (int k = 0; k < 3; k++) { e0[k] = v2[k] - v0[k]; e1[k] = v0[k] - v1[k]; e2[k] = v1[k] - v2[k]; }
    for (int k = 0; k < 3; k++) s[k] = v2[k] + v0[k];
    orc_cross(e0, s, c); const float U = orc_dot(c, dir);
    for (int k = 0; k < 3; k++) s[k] = v0[k] + v1[k];
    orc_cross(e1, s, c); const float V = orc_dot(c, dir);
    for (int k = 0; k < 3; k++) s[k] = v1[k] + v2[k];
    orc_cross(e2, s, c); const float W = orc_dot(c, dir);
    const float minUVW = sse_min(sse_min(U, V), W), maxUVW = sse_max(sse_max(U, V), W);
    if (!((minUVW >= 0.0f) || (maxUVW <= 0.0f))) continue;
    orc_stable_triangle_normal(e0, e1, e2, Ng);
    const float dn = orc_dot(Ng, dir);
    const float den = dn + dn;
    const float absDen = fabsf(den);
    const uint32_t sgn = f2u(den) & 0x80000000u;
    const float tn = orc_dot(v0, Ng);
    const float T = tn + tn;
    const float Ts = xorf(T, sgn);
    if (!(absDen * tnear < Ts)) continue;
    if (!(Ts <= absDen * tfar)) continue;
    if (!(den != 0.0f)) continue;
    valid |= 1u << l;
    /* finalize */
    const float rcpDen = orc_rcp(den);
    h->t[l] = T * rcpDen;
    const float UVW = U + V + W;
    const float rcpUVW = fabsf(UVW) < 1e-18f ? 0.0f : orc_rcp(UVW);
    h->u[l] = U * rcpUVW;
    h->v[l] = V * rcpUVW;
    for (int k = 0; k < 3; k++) h->Ng[k][l] = Ng[k];
  }
  h->valid = valid;
  return valid;
}

/* MoellerTrumboreIntersector1<4>::intersect(Edge) + finalize, triangle_intersector_moeller.h:75-123, :42-48 */
static unsigned moeller4(const oblock* b, const float org[3], const float dir[3], float tnear, float tfar, blockhit* h)
{
  unsigned valid = 0;
  for (int l = 0; l < 4; l++) {
    float v0[3], e1[3], e2[3], Ng[3], C[3], R[3];
    for (int k = 0; k < 3; k++) { v0[k] = b->a[k][l]; e1[k] = b->b[k][l]; e2[k] = b->c[k][l]; }
    orc_cross(e2, e1, Ng);
    for (int k = 0; k < 3; k++) C[k] = v0[k] - org[k];
    orc_cross(C, dir, R);
    const float den = orc_dot(Ng, dir);
    const float absDen = fabsf(den);
    const uint32_t sgn = f2u(den) & 0x80000000u;
    const float U = xorf(orc_dot(R, e2), sgn);
    const float V = xorf(orc_dot(R, e1), sgn);
    if (!((den != 0.0f) && (U >= 0.0f) && (V >= 0.0f) && (U + V <= absDen))) continue;
    const float T = xorf(orc_dot(Ng, C), sgn);
    if (!((absDen * tnear < T) && (T <= absDen * tfar))) continue;
    valid |= 1u << l;
    const float r = orc_rcp(absDen);
    h->t[l] = T * r; h->u[l] = U * r; h->v[l] = V * r;
    for (int k = 0; k < 3; k++) h->Ng[k][l] = Ng[k];
  }
  h->valid = valid;
  return valid;
}

/* select_min, vfloat4_sse2.h:654-659: lowest lane whose value equals the minimum over the valid lanes */
static int select_min4(unsigned valid, const float t[4])
{
  float m = ORC_INF;
  for (int l = 0; l < 4; l++) if (valid & (1u << l)) m = sse_min(t[l], m);
  for (int l = 0; l < 4; l++) if ((valid & (1u << l)) && t[l] == m) return l;
  for (int l = 0; l < 4; l++) if (valid & (1u << l)) return l;
  return -1;
}

int orc_pluecker_block(const float v0[12], const float v1[12], const float v2[12], const float org[3], const float dir[3],
                       float tnear, float tfar, float out[6])
{
  oblock b; memset(&b, 0, sizeof(b));
  memcpy(b.a, v0, 48); memcpy(b.b, v1, 48); memcpy(b.c, v2, 48);
  blockhit h;
  if (!pluecker4(&b, org, dir, tnear, tfar, &h)) return -1;
  const int l = select_min4(h.valid, h.t);
  out[0] = h.t[l]; out[1] = h.u[l]; out[2] = h.v[l]; out[3] = h.Ng[0][l]; out[4] = h.Ng[1][l]; out[5] = h.Ng[2][l];
  return l;
}

int orc_moeller_block(const float v0[12], const float v1[12], const float v2[12], const float org[3], const float dir[3],
                      float tnear, float tfar, float out[6])
{
  oblock b; memset(&b, 0, sizeof(b));
  memcpy(b.a, v0, 48);
  for (int k = 0; k < 3; k++) for (int l = 0; l < 4; l++) { b.b[k][l] = v0[4 * k + l] - v1[4 * k + l]; b.c[k][l] = v2[4 * k + l] - v0[4 * k + l]; }
  blockhit h;
  if (!moeller4(&b, org, dir, tnear, tfar, &h)) return -1;
  const int l = select_min4(h.valid, h.t);
  out[0] = h.t[l]; out[1] = h.u[l]; out[2] = h.v[l]; out[3] = h.Ng[0][l]; out[4] = h.Ng[1][l]; out[5] = h.Ng[2][l];
  return l;
}

/* ------------------------------------------------------------------------------------------------------ */
/* traversal                                                                                                */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct { uint32_t ptr, dist; } sitem;
#define ORC_STACK 564 /* 1+(N-1)*maxDepth+3 with N=8, maxDepth=80 (bvh_intersector1.h:39, bvh.h:135-137) */

static __thread unsigned long long g_cnt[3];
void orc_get_counters(unsigned long long out[3]) { out[0] = g_cnt[0]; out[1] = g_cnt[1]; out[2] = g_cnt[2]; }

static inline void xchg(sitem* a, sitem* b) { sitem t = *a; *a = *b; *b = t; }
/* stack_item.h:39-61 */
static inline void sort3(sitem* s1, sitem* s2, sitem* s3)
{
  if (s2->dist < s1->dist) xchg(s2, s1);
  if (s3->dist < s2->dist) xchg(s3, s2);
  if (s2->dist < s1->dist) xchg(s2, s1);
}
static inline void sort4(sitem* s1, sitem* s2, sitem* s3, sitem* s4)
{
  if (s2->dist < s1->dist) xchg(s2, s1);
  if (s4->dist < s3->dist) xchg(s4, s3);
  if (s3->dist < s1->dist) xchg(s3, s1);
  if (s4->dist < s2->dist) xchg(s4, s2);
  if (s3->dist < s2->dist) xchg(s3, s2);
}
/* stack_item.h:64-80 */
static inline void sortN(sitem* begin, sitem* end)
{
  for (sitem* i = begin + 1; i != end; ++i) {
    const sitem item = *i;
    sitem* j = i;
    while (j != begin && (j - 1)->dist < item.dist) { *j = *(j - 1); --j; }
    *j = item;
  }
}
static inline int bscf(unsigned* m) { const int i = __builtin_ctz(*m); *m &= *m - 1; return i; }

/* BVHNNodeTraverser1Hit<8,...>::traverseClosestHit, bvh_traverser1.h:549-635 (non-AVX512 path) */
static void traverse_closest(const onode* n, unsigned mask, const float tNear[8], uint32_t* cur, sitem** sp)
{
  int r = bscf(&mask);
  *cur = n->child[r];
  if (mask == 0) return;
  const uint32_t c0 = *cur; const uint32_t d0 = f2u(tNear[r]);
  r = bscf(&mask);
  const uint32_t c1 = n->child[r]; const uint32_t d1 = f2u(tNear[r]);
  if (mask == 0) {
    if (d0 < d1) { (*sp)->ptr = c1; (*sp)->dist = d1; (*sp)++; *cur = c0; return; }
    else         { (*sp)->ptr = c0; (*sp)->dist = d0; (*sp)++; *cur = c1; return; }
  }
  (*sp)->ptr = c0; (*sp)->dist = d0; (*sp)++;
  (*sp)->ptr = c1; (*sp)->dist = d1; (*sp)++;
  r = bscf(&mask);
  (*sp)->ptr = n->child[r]; (*sp)->dist = f2u(tNear[r]); (*sp)++;
  if (mask == 0) { sort3(*sp - 1, *sp - 2, *sp - 3); *cur = (*sp)[-1].ptr; (*sp)--; return; }
  r = bscf(&mask);
  (*sp)->ptr = n->child[r]; (*sp)->dist = f2u(tNear[r]); (*sp)++;
  if (mask == 0) { sort4(*sp - 1, *sp - 2, *sp - 3, *sp - 4); *cur = (*sp)[-1].ptr; (*sp)--; return; }
  sitem* first = *sp - 4;
  while (1) {
    r = bscf(&mask);
    (*sp)->ptr = n->child[r]; (*sp)->dist = f2u(tNear[r]); (*sp)++;
    if (mask == 0) break;
  }
  sortN(first, *sp);
  *cur = (*sp)[-1].ptr; (*sp)--;
}

/* traverseAnyHit, bvh_traverser1.h:638-666 */
static void traverse_any(const onode* n, unsigned mask, uint32_t* cur, uint32_t** sp)
{
  int r = bscf(&mask);
  *cur = n->child[r];
  if (mask == 0) return;
  *(*sp)++ = *cur;
  for (;;) {
    r = bscf(&mask);
    *cur = n->child[r];
    if (mask == 0) return;
    *(*sp)++ = *cur;
  }
}

static int g_fork_trace; /* defined in subdiv_oracle.inc (debugging aid) */
/* BVHNIntersector1<8,BVH_AN1,robust,ArrayIntersector1<...>>::intersect, bvh_intersector1.cpp:40-126 */
static void intersect1(const orc_scene* s, orayhit* ray, uint32_t instID)
{
  if (s->root == ORC_EMPTY) return;
  const int robust = s->mode != 1; /* Triangle4v and every subdivision intersector traverse robustly */
  sitem stack[ORC_STACK];
  sitem* sp = stack + 1;
  stack[0].ptr = s->root; stack[0].dist = f2u(-ORC_INF);
  travray tr; travray_init(&tr, ray, robust);
  while (1) {
  pop:
    if (sp == stack) break;
    sp--;
    uint32_t cur = sp->ptr;
    if (g_fork_trace) printf("  CPU pop sp %d ref %08x dist %08x tfar %a\n", (int)(sp - stack), cur, sp->dist, ray->tfar);
    if (u2f(sp->dist) > ray->tfar) continue; /* :86 */
    while (1) {
      if (cur & ORC_LEAF) break;
      float tNear[8];
      g_cnt[0]++;
      const onode* n = &s->nodes[cur];
      const unsigned mask = node_test(n, &tr, robust, tNear);
      if (g_fork_trace)
        printf("  CPU node %u sp %d mask %02x dist %08x %08x %08x %08x %08x %08x %08x %08x refs %08x %08x %08x %08x %08x %08x %08x %08x tfar %a\n", cur, (int)(sp - stack), mask,
               f2u(tNear[0]), f2u(tNear[1]), f2u(tNear[2]), f2u(tNear[3]), f2u(tNear[4]), f2u(tNear[5]), f2u(tNear[6]), f2u(tNear[7]),
               n->child[0], n->child[1], n->child[2], n->child[3], n->child[4], n->child[5], n->child[6], n->child[7], ray->tfar);
      if (mask == 0) goto pop;
      traverse_closest(n, mask, tNear, &cur, &sp);
    }
    g_cnt[1]++;
    if (s->mode >= 2) { /* subdivision leaf: a grid cell or a cBVH blob */
      g_cnt[2]++;
      subdiv_leaf_intersect(s, cur & 0x07FFFFFFu, ray, instID);
      tr.tfar = ray->tfar;
      continue;
    }
    const uint32_t first = cur & 0x07FFFFFFu, num = (cur >> 27) & 15u;
    for (uint32_t i = 0; i < num; i++) { /* ArrayIntersector1::intersect, intersector_iterators.h:32-36 */
      const oblock* b = &s->blocks[first + i];
      blockhit h;
      g_cnt[2]++;
      const unsigned valid = robust ? pluecker4(b, ray->org, ray->dir, ray->tnear, ray->tfar, &h)
                                    : moeller4(b, ray->org, ray->dir, ray->tnear, ray->tfar, &h);
      if (!valid) continue;
      /* Intersect1EpilogM, intersector_epilog.h:240-305 (no ray mask, no filter) */
      const int l = select_min4(valid, h.t);
      ray->tfar = h.t[l];
      ray->Ng[0] = h.Ng[0][l]; ray->Ng[1] = h.Ng[1][l]; ray->Ng[2] = h.Ng[2][l];
      ray->u = h.u[l]; ray->v = h.v[l];
      ray->primID = (uint32_t)b->primID[l];
      ray->geomID = (uint32_t)b->geomID[l];
      ray->instID = instID;
    }
    tr.tfar = ray->tfar; /* :117 */
  }
}

/* BVHNIntersector1::occluded, bvh_intersector1.cpp:128-209 ; ray = first 48 bytes of orayhit */
static void occluded1(const orc_scene* s, orayhit* ray)
{
  if (ray->tfar < 0.0f) return; /* :132-134 */
  if (s->root == ORC_EMPTY) return;
  const int robust = s->mode != 1;
  uint32_t stack[ORC_STACK];
  uint32_t* sp = stack + 1;
  stack[0] = s->root;
  travray tr; travray_init(&tr, ray, robust);
  while (1) {
  pop:
    if (sp == stack) break;
    sp--;
    uint32_t cur = *sp;
    while (1) {
      if (cur & ORC_LEAF) break;
      float tNear[8];
      const onode* n = &s->nodes[cur];
      const unsigned mask = node_test(n, &tr, robust, tNear);
      if (mask == 0) goto pop;
      traverse_any(n, mask, &cur, &sp);
    }
    if (s->mode >= 2) {
      if (subdiv_leaf_occluded(s, cur & 0x07FFFFFFu, ray)) { ray->tfar = -ORC_INF; return; }
      continue;
    }
    const uint32_t first = cur & 0x07FFFFFFu, num = (cur >> 27) & 15u;
    for (uint32_t i = 0; i < num; i++) {
      const oblock* b = &s->blocks[first + i];
      blockhit h;
      const unsigned valid = robust ? pluecker4(b, ray->org, ray->dir, ray->tnear, ray->tfar, &h)
                                    : moeller4(b, ray->org, ray->dir, ray->tnear, ray->tfar, &h);
      if (valid) { ray->tfar = -ORC_INF; return; } /* Occluded1EpilogM -> true; :198-201 */
    }
  }
}

void orc_intersect1(const orc_scene* s, void* rayhit, uint32_t instID)
{
  orayhit r; memcpy(&r, rayhit, 80);
  intersect1(s, &r, instID);
  memcpy(rayhit, &r, 80);
}

void orc_occluded1(const orc_scene* s, void* ray)
{
  orayhit r; memcpy(&r, ray, 48);
  occluded1(s, &r);
  memcpy((char*)ray + 32, &r.tfar, 4);
}

typedef struct { const orc_scene* s; char* base; uint32_t M; size_t stride; uint32_t instID; int occl; int tid, nthreads; } ojob;

static void run_range(const ojob* j, uint32_t b, uint32_t e)
{
  for (uint32_t i = b; i < e; i++) {
    char* p = j->base + (size_t)i * j->stride;
    orayhit r;
    memcpy(&r, p, j->occl ? 48 : 80);
    if (!(r.tnear <= r.tfar)) continue; /* stream front-end, bvh_intersector_stream_filters.cpp:156 / rtcore.cpp:417 */
    if (j->occl) { occluded1(j->s, &r); memcpy(p + 32, &r.tfar, 4); }
    else {
      intersect1(j->s, &r, j->instID);
      /* setHitByOffset, kernels/common/ray.h:817-853: stored only when geomID is valid */
      if (r.geomID != 0xFFFFFFFFu) { memcpy(p + 32, &r.tfar, 4); memcpy(p + 48, &r.Ng[0], 32); }
    }
  }
}

static void* worker(void* arg)
{
  const ojob* j = (const ojob*)arg;
  const uint32_t nblk = (j->M + 1023) / 1024;
  for (uint32_t b = (uint32_t)j->tid; b < nblk; b += (uint32_t)j->nthreads) {
    const uint32_t lo = b * 1024, hi = lo + 1024 < j->M ? lo + 1024 : j->M;
    run_range(j, lo, hi);
  }
  return NULL;
}

static void run_stream(const orc_scene* s, void* base, uint32_t M, size_t stride, uint32_t instID, int occl, int nthreads)
{
  if (nthreads <= 1) {
    ojob j = {s, (char*)base, M, stride, instID, occl, 0, 1};
    g_cnt[0] = g_cnt[1] = g_cnt[2] = 0;
    run_range(&j, 0, M);
    return;
  }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  ojob* jobs = (ojob*)malloc(sizeof(ojob) * (size_t)nthreads);
  for (int t = 0; t < nthreads; t++) {
    ojob j = {s, (char*)base, M, stride, instID, occl, t, nthreads};
    jobs[t] = j;
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
}

void orc_intersect1M(const orc_scene* s, void* rayhits, uint32_t M, size_t byteStride, uint32_t instID, int nthreads)
{
  run_stream(s, rayhits, M, byteStride, instID, 0, nthreads);
}

void orc_occluded1M(const orc_scene* s, void* rays, uint32_t M, size_t byteStride, int nthreads)
{
  run_stream(s, rays, M, byteStride, 0, 1, nthreads);
}

/* ------------------------------------------------------------------------------------------------------ */
/* ray generator: makeRandomRay (tutorials/viewer/viewer_device.cpp:367-392) on a libc-independent drand48    */
/* ------------------------------------------------------------------------------------------------------ */
static inline double lcg48(uint64_t* x)
{
  *x = (*x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL;
  return (double)*x * (1.0 / 281474976710656.0); /* 2^-48 */
}

void orc_make_random_rays(void* rayhits, uint32_t M, size_t byteStride, const float lo[3], const float hi[3], uint64_t seed,
                          int doubleEval)
{
  uint64_t x = ((seed & 0xFFFFFFFFULL) << 16) | 0x330EULL; /* srand48(seed) */
  float diam[3];
  for (int k = 0; k < 3; k++) diam[k] = hi[k] - lo[k];
  for (uint32_t i = 0; i < M; i++) {
    float p[2][3];
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 3; k++) {
        const double u = lcg48(&x);
        if (doubleEval) p[j][k] = (float)(u * ((double)hi[k] - (double)lo[k]) + (double)lo[k]); /* SURVEY.md 8d probe variant */
        else { const float X = (float)u; p[j][k] = X * diam[k] + lo[k]; }                       /* viewer: fp32 */
      }
    orayhit r; memset(&r, 0, sizeof(r));
    float d[3];
    for (int k = 0; k < 3; k++) { r.org[k] = p[0][k]; d[k] = p[1][k] - p[0][k]; }
    const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; k++) r.dir[k] = d[k] / len;
    r.tnear = 0.0f; r.time = 0.0f; r.tfar = ORC_INF; r.mask = 0xFFFFFFFFu; r.id = i; r.flags = 0;
    r.primID = r.geomID = r.instID = 0xFFFFFFFFu;
    memcpy((char*)rayhits + (size_t)i * byteStride, &r, 80);
  }
}

/* subdivision leaves (eager grid cells, fork cBVH blobs) */
#include "subdiv_oracle.inc"
