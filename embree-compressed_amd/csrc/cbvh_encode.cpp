#include "cbvh_encode.h"

#include <algorithm>
#include <cfloat>

namespace rtamd {

const float CBVH_TABLE_BORDER[8] = {0.000f, 0.005f, 0.010f, 0.050f, 0.100f, 0.200f, 0.400f, 0.600f}; // compressed_node.h:31-38
const float CBVH_TABLE_MID[8] = {0.00f, 0.40f, 0.48f, 0.49f, 0.50f, 0.51f, 0.52f, 0.60f};            // :22-29
const float CBVH_TABLE_Z[4] = {0.0f, 0.25f, 0.5f, 0.75f};                                            // LookupTable<uni,2>, :67-76

namespace {

struct F3
{
  float x, y, z;
};
inline F3 f3(float a, float b, float c) { return F3{a, b, c}; }
inline F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float dot3(F3 a, F3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); } // vec3.h:193
inline F3 cross3(F3 a, F3 b)                                                         // vec3.h:198
{
  return f3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
inline F3 normalize3(F3 a)
{
  const float r = 1.0f / sqrtf(dot3(a, a)); // reference: rsqrtps + Newton step (vec3fa.h:153-161)
  return f3(a.x * r, a.y * r, a.z * r);
}

// LinearSpace3f with column vectors vx,vy,vz (common/math/linearspace3.h)
struct Lin3
{
  F3 vx, vy, vz;
};
inline F3 xfm(const Lin3& s, F3 p) // xfmPoint / xfmVector, linearspace3.h:168-169
{
  return f3(fmaf(p.x, s.vx.x, fmaf(p.y, s.vy.x, p.z * s.vz.x)), fmaf(p.x, s.vx.y, fmaf(p.y, s.vy.y, p.z * s.vz.y)),
            fmaf(p.x, s.vx.z, fmaf(p.y, s.vy.z, p.z * s.vz.z)));
}
inline Lin3 inverse(const Lin3& m) // adjoint()/det(), linearspace3.h:57-63
{
  const F3 c0 = cross3(m.vy, m.vz), c1 = cross3(m.vz, m.vx), c2 = cross3(m.vx, m.vy);
  const float det = dot3(m.vx, c0);
  // adjoint = LinearSpace3(c0,c1,c2).transposed(): columns (c0.x,c1.x,c2.x), (c0.y,c1.y,c2.y), (c0.z,c1.z,c2.z)
  Lin3 r;
  r.vx = f3(c0.x / det, c1.x / det, c2.x / det);
  r.vy = f3(c0.y / det, c1.y / det, c2.y / det);
  r.vz = f3(c0.z / det, c1.z / det, c2.z / det);
  return r;
}

struct M3
{
  float m[9]; // row-major
};
inline M3 mul(const M3& a, const M3& b)
{
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[3 * i + j] = (a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j]) + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
inline M3 inverse(const M3& a) // cofactor expansion (what Eigen's fixed-size 3x3 inverse() does)
{
  const float* m = a.m;
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = (m[0] * c00 + m[1] * c01) + m[2] * c02;
  const float id = 1.0f / det;
  M3 r;
  r.m[0] = c00 * id; r.m[1] = (m[2] * m[7] - m[1] * m[8]) * id; r.m[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  r.m[3] = c01 * id; r.m[4] = (m[0] * m[8] - m[2] * m[6]) * id; r.m[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r.m[6] = c02 * id; r.m[7] = (m[1] * m[6] - m[0] * m[7]) * id; r.m[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return r;
}
inline F3 project(F3 a, const M3& p) // compressed_help.h:86-90: homography on (x,y), z kept
{
  const float x = (p.m[0] * a.x + p.m[1] * a.y) + p.m[2];
  const float y = (p.m[3] * a.x + p.m[4] * a.y) + p.m[5];
  const float w = (p.m[6] * a.x + p.m[7] * a.y) + p.m[8];
  return f3(x / w, y / w, a.z);
}

// 8x8 dense solve in double with complete pivoting (stands in for Eigen::FullPivLU, compressed_help.h:77)
bool solve8(double A[8][8], double b[8], double x[8])
{
  int colperm[8];
  for (int i = 0; i < 8; i++) colperm[i] = i;
  for (int k = 0; k < 8; k++) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int i = k; i < 8; i++)
      for (int j = k; j < 8; j++)
        if (fabs(A[i][j]) > best) { best = fabs(A[i][j]); pr = i; pc = j; }
    if (!(best > 0.0)) { // rank deficient: remaining unknowns are 0 (what a truncated LU solve yields)
      for (int i = k; i < 8; i++) b[i] = 0.0;
      for (int kk = k - 1; kk >= 0; kk--) { /* fallthrough to back substitution below */ }
      double y[8];
      for (int i = 0; i < 8; i++) y[i] = 0.0;
      for (int i = k - 1; i >= 0; i--) {
        double s = b[i];
        for (int j = i + 1; j < k; j++) s -= A[i][j] * y[j];
        y[i] = s / A[i][i];
      }
      for (int i = 0; i < 8; i++) x[colperm[i]] = y[i];
      return false;
    }
    if (pr != k) { for (int j = 0; j < 8; j++) std::swap(A[k][j], A[pr][j]); std::swap(b[k], b[pr]); }
    if (pc != k) { for (int i = 0; i < 8; i++) std::swap(A[i][k], A[i][pc]); std::swap(colperm[k], colperm[pc]); }
    for (int i = k + 1; i < 8; i++) {
      const double f = A[i][k] / A[k][k];
      if (f == 0.0) continue;
      for (int j = k; j < 8; j++) A[i][j] -= f * A[k][j];
      b[i] -= f * b[k];
    }
  }
  double y[8];
  for (int i = 7; i >= 0; i--) {
    double s = b[i];
    for (int j = i + 1; j < 8; j++) s -= A[i][j] * y[j];
    y[i] = s / A[i][i];
  }
  for (int i = 0; i < 8; i++) x[colperm[i]] = y[i];
  return true;
}

// ComputeLinearEstimate, compressed_help.h:54-84: homography mapping 4 source points onto 4 target points
M3 linear_estimate(const float src[4][2], const float tgt[4][2])
{
  double A[8][8], b[8], x[8];
  for (int i = 0; i < 4; i++) {
    const float p0 = tgt[i][0], p1 = tgt[i][1], q0 = src[i][0], q1 = src[i][1];
    const double r0[8] = {q0, q1, 1.f, 0.f, 0.f, 0.f, (double)(-q0 * p0), (double)(-q1 * p0)};
    const double r1[8] = {0.f, 0.f, 0.f, q0, q1, 1.f, (double)(-q0 * p1), (double)(-q1 * p1)};
    for (int j = 0; j < 8; j++) { A[i][j] = r0[j]; A[4 + i][j] = r1[j]; }
    b[i] = tgt[i][0];
    b[4 + i] = tgt[i][1];
  }
  solve8(A, b, x);
  M3 m;
  for (int i = 0; i < 8; i++) m.m[i] = (float)x[i];
  m.m[8] = 1.0f;
  return m;
}

inline uint8_t lookup_idx(const float* table, int n, float val) // BaseLookupTable::lookUpIdx, compressed_node.h:47-55
{
  uint8_t ret = 0;
  for (int i = 0; i < n; i++) {
    if (table[i] <= val) ret = (uint8_t)i;
    else break;
  }
  return ret;
}

inline void box_empty(Box3f& b)
{
  for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
}
inline void box_extend(Box3f& b, F3 p)
{
  b.lo[0] = fminf(b.lo[0], p.x); b.lo[1] = fminf(b.lo[1], p.y); b.lo[2] = fminf(b.lo[2], p.z);
  b.hi[0] = fmaxf(b.hi[0], p.x); b.hi[1] = fmaxf(b.hi[1], p.y); b.hi[2] = fmaxf(b.hi[2], p.z);
}
inline void box_extend(Box3f& b, const Box3f& o)
{
  for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], o.lo[k]); b.hi[k] = fmaxf(b.hi[k], o.hi[k]); }
}

// baseLeaf::refitTriangle, compressed_leaf.h:117-169: z of the plane through p1,p2,p3 above the 2-D point p
float refit_triangle(float px, float py, F3 p1, F3 p2, F3 p3)
{
  float tmp = p1.x, tt = p1.y, tmp2 = p1.z;
  const float a = tmp - p2.x, f = tmp - p3.x, j = tmp - px;
  const float b = tt - p2.y, e = tt - p3.y, k = tt - py;
  const float c = tmp2 - p2.z;
  tt = tmp2 - p3.z;
  const float l = tmp2 - 0.f;
  const float g = 0.f, h = 0.f, gamma = 1.f;
  tmp = e * gamma - h * tt;
  tmp2 = g * tt - f * gamma;
  float M = a * tmp;
  tmp = f * h - e * g;
  M += b * tmp2;
  tmp2 = a * k - j * b;
  M += c * tmp;
  tmp = j * c - a * l;
  tt *= tmp2;
  M = 1.f / M;
  tmp2 = b * l - k * c;
  tt += e * tmp;
  tt += f * tmp2;
  tt *= -M;
  return tt;
}

void corner_heights(const Box3f& pb, F3 v1, F3 v2, F3 v3, F3 v4, float z[4])
{
  z[0] = refit_triangle(pb.lo[0], pb.lo[1], v1, v2, v3); // lowerLeft
  z[1] = refit_triangle(pb.hi[0], pb.lo[1], v1, v2, v4); // lowerRight
  z[2] = refit_triangle(pb.lo[0], pb.hi[1], v1, v3, v4); // upperLeft
  z[3] = refit_triangle(pb.hi[0], pb.hi[1], v2, v3, v4); // upperRight
}

// quantTris::estimateExtent, compressed_leaf.h:223-251
float estimate_extent(const Box3f& pb, F3 v1, F3 v2, F3 v3, F3 v4)
{
  float z[4];
  corner_heights(pb, v1, v2, v3, v4, z);
  float t[4];
  for (int i = 0; i < 4; i++) t[i] = fmaxf(fmaxf(z[i] - pb.hi[2], 0.f), fabsf(fminf(z[i] - pb.lo[2], 0.f)));
  const double zF = pb.hi[2] - pb.lo[2];
  if (zF == 0.0) return 0.f;
  return (float)(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])) / zF);
}

// quantTris::setZ + leafStorage<4>::setZ, compressed_leaf.h:193-221, :40-47 -> bytes z12, z34
void set_leaf_z(const Box3f& pb, F3 v1, F3 v2, F3 v3, F3 v4, float extent, uint8_t out[2])
{
  float z[4];
  corner_heights(pb, v1, v2, v3, v4, z);
  const double zF = pb.hi[2] - pb.lo[2];
  uint8_t q[4] = {0, 0, 0, 0};
  if (zF != 0.0) {
    const float zf = (float)((1.0f + 2.0f * extent) * zF);
    const float rcpF = 16.f / zf;
    for (int i = 0; i < 4; i++) {
      const float zi = (float)(z[i] - (pb.lo[2] - extent * zF));
      q[i] = (uint8_t)fmaxf(0.f, fminf(15.f, zi * rcpF));
    }
  }
  out[0] = (uint8_t)((q[0] << 4) | (q[1] & 0x0f));
  out[1] = (uint8_t)((q[2] << 4) | (q[3] & 0x0f));
}

} // namespace

// test hook (rtcamdDebugCbvhLeafCodec): the leaf quantiser on caller-supplied inputs
void cbvh_debug_leaf_codec(const float box[6], const float v[12], float extent, uint8_t bytesOut[2], float* extentEstimate)
{
  Box3f pb;
  for (int k = 0; k < 3; k++) { pb.lo[k] = box[k]; pb.hi[k] = box[3 + k]; }
  F3 p[4];
  for (int i = 0; i < 4; i++) { p[i].x = v[3 * i]; p[i].y = v[3 * i + 1]; p[i].z = v[3 * i + 2]; }
  set_leaf_z(pb, p[0], p[1], p[2], p[3], extent, bytesOut);
  if (extentEstimate) *extentEstimate = estimate_extent(pb, p[0], p[1], p[2], p[3]);
}

// Node::setAABB, compressed_node.h:406-448
void cbvh_encode_node(const Box3f& P, const Box3f c[4], CbvhNode& out)
{
  double xF = 1.0 / (P.hi[0] - P.lo[0]);
  double yF = 1.0 / (P.hi[1] - P.lo[1]);
  double zF = 1.0 / (P.hi[2] - P.lo[2]);
  if (!std::isfinite(xF)) xF = FLT_MIN;
  if (!std::isfinite(yF)) yF = FLT_MIN;
  if (!std::isfinite(zF)) zF = FLT_MIN;
  const float x1 = fminf(c[0].lo[0], c[2].lo[0]), x2 = fminf(c[1].lo[0], c[3].lo[0]);
  const float x3 = fmaxf(c[0].hi[0], c[2].hi[0]), x4 = fmaxf(c[1].hi[0], c[3].hi[0]);
  const float y1 = fminf(c[0].lo[1], c[1].lo[1]), y2 = fminf(c[2].lo[1], c[3].lo[1]);
  const float y3 = fmaxf(c[0].hi[1], c[1].hi[1]), y4 = fmaxf(c[2].hi[1], c[3].hi[1]);
  const float z1 = fminf(fminf(c[0].lo[2], c[1].lo[2]), fminf(c[2].lo[2], c[3].lo[2]));
  const float z2 = fmaxf(fmaxf(c[0].hi[2], c[1].hi[2]), fmaxf(c[2].hi[2], c[3].hi[2]));
  const uint8_t X1 = lookup_idx(CBVH_TABLE_BORDER, 8, (float)((x1 - P.lo[0]) * xF));
  const uint8_t X2 = lookup_idx(CBVH_TABLE_MID, 8, (float)((x2 - P.lo[0]) * xF));
  const uint8_t X3 = lookup_idx(CBVH_TABLE_MID, 8, (float)((P.hi[0] - x3) * xF));
  const uint8_t X4 = lookup_idx(CBVH_TABLE_BORDER, 8, (float)((P.hi[0] - x4) * xF));
  const uint8_t Y1 = lookup_idx(CBVH_TABLE_BORDER, 8, (float)((y1 - P.lo[1]) * yF));
  const uint8_t Y2 = lookup_idx(CBVH_TABLE_MID, 8, (float)((y2 - P.lo[1]) * yF));
  const uint8_t Y3 = lookup_idx(CBVH_TABLE_MID, 8, (float)((P.hi[1] - y3) * yF));
  const uint8_t Y4 = lookup_idx(CBVH_TABLE_BORDER, 8, (float)((P.hi[1] - y4) * yF));
  const uint8_t minZ = lookup_idx(CBVH_TABLE_Z, 4, (float)((z1 - P.lo[2]) * zF));
  const uint8_t maxZ = lookup_idx(CBVH_TABLE_Z, 4, (float)((P.hi[2] - z2) * zF));
  out.xz = (uint8_t)((X1 << 5) | (X2 << 2) | minZ); // NodeStorage<com,3,3,2>, compressed_node.h:261-295
  out.x = (uint8_t)((X3 << 5) | (X4 << 2));
  out.yz = (uint8_t)((Y1 << 5) | (Y2 << 2) | maxZ);
  out.y = (uint8_t)((Y3 << 5) | (Y4 << 2));
}

// Node::getAABB, compressed_node.h:450-486 (== getNode :488-510 lane `loc`): mul then add, no fused op
Box3f cbvh_decode_child(const CbvhNode& n, const Box3f& P, int loc)
{
  const float dim[3] = {P.hi[0] - P.lo[0], P.hi[1] - P.lo[1], P.hi[2] - P.lo[2]};
  const int X1 = n.xz >> 5, X2 = (n.xz >> 2) & 7, minZ = n.xz & 3, X3 = n.x >> 5, X4 = (n.x >> 2) & 7;
  const int Y1 = n.yz >> 5, Y2 = (n.yz >> 2) & 7, maxZ = n.yz & 3, Y3 = n.y >> 5, Y4 = (n.y >> 2) & 7;
  float mn[3], mx[3];
  mn[2] = CBVH_TABLE_Z[minZ];
  mx[2] = 1.f - CBVH_TABLE_Z[maxZ];
  mn[0] = (loc & 1) ? CBVH_TABLE_MID[X2] : CBVH_TABLE_BORDER[X1];
  mx[0] = (loc & 1) ? 1.f - CBVH_TABLE_BORDER[X4] : 1.f - CBVH_TABLE_MID[X3];
  mn[1] = (loc & 2) ? CBVH_TABLE_MID[Y2] : CBVH_TABLE_BORDER[Y1];
  mx[1] = (loc & 2) ? 1.f - CBVH_TABLE_BORDER[Y4] : 1.f - CBVH_TABLE_MID[Y3];
  Box3f r;
  for (int k = 0; k < 3; k++) {
    r.lo[k] = mn[k] * dim[k] + P.lo[k];
    r.hi[k] = mx[k] * dim[k] + P.lo[k];
  }
  return r;
}

size_t cbvh_blob_bytes(unsigned C, CbvhMode mode) { return cbvh_stride(C, (uint32_t)mode); } // section offsets: accel.h

// CompressedBVH::CompressedBVH, compressed.h:49-337
void cbvh_encode(const PatchGrid& pg, unsigned x0, unsigned x1, unsigned y0, unsigned y1, unsigned C, CbvhMode mode, uint8_t* blob,
                 Box3& boundsOut)
{
  const unsigned width = x1 - x0 + 1, height = y1 - y0 + 1, gw = pg.n + 1;
  const bool use_leaf = mode == CBVH_LEAF, use_grid = mode == CBVH_GRID;
  const size_t cells = (size_t)(width - 1) * (height - 1);
  const size_t elems = (((size_t)1 << (2 * C)) - 1) / 3;
  memset(blob, 0, cbvh_blob_bytes(C, mode));
  CbvhHeader* H = (CbvhHeader*)blob;
  CbvhMid* Hm = (CbvhMid*)(blob + CBVH_HEADER_BYTES);
  CbvhTail* Ht = (CbvhTail*)(blob + cbvh_tail_offset(C, (uint32_t)mode));
  CbvhNode* nodes = (CbvhNode*)(blob + CBVH_NODES_OFFSET);
  uint8_t* leaves = blob + cbvh_payload_offset(C, (uint32_t)mode);

  std::vector<F3> v((size_t)width * height);
  for (unsigned y = 0; y < height; y++)
    for (unsigned x = 0; x < width; x++) {
      const size_t k = (size_t)(y0 + y) * gw + (x0 + x);
      v[(size_t)y * width + x] = f3(pg.x[k], pg.y[k], pg.z[k]);
    }
  const unsigned i00 = 0, i10 = width - 1, i01 = width * (height - 1), i11 = width * height - 1;

  Hm->geomID = pg.geomID;
  Hm->primID = pg.primID;
  const float fn = (float)pg.n;
  Hm->uv0x = pg.u0 + (float)x0 / fn; Hm->uv0y = pg.v0 + (float)y0 / fn;           // grid_u/grid_v of the first vertex
  Hm->uv1x = (pg.u0 + (float)x1 / fn) - Hm->uv0x; Hm->uv1y = (pg.v0 + (float)y1 / fn) - Hm->uv0y; // :85-86
  H->rcp_edges = 1.f / (float)(1u << C);                         // :88-89
  Hm->elems = (uint32_t)elems;
  Hm->grid_width = width;
  H->levels = C;

  // frame from the (un-displaced, in leaf mode) corner vertices, :91-126
  F3 f00, f10, f01, f11;
  if (!use_leaf) { f00 = v[i00]; f10 = v[i10]; f01 = v[i01]; f11 = v[i11]; }
  else {
    auto base = [&](unsigned x, unsigned y) { const size_t k = (size_t)(y0 + y) * gw + (x0 + x); return f3(pg.px(k, true), pg.py(k, true), pg.pz(k, true)); };
    f00 = base(0, 0); f10 = base(width - 1, 0); f01 = base(0, height - 1); f11 = base(width - 1, height - 1);
  }
  Lin3 world;
  world.vx = normalize3(sub(add(sub(f10, f00), f11), f01));
  world.vy = normalize3(sub(add(sub(f01, f00), f11), f10));
  world.vz = normalize3(cross3(world.vx, world.vy));
  const Lin3 space = inverse(world);

  const F3 l00 = xfm(space, v[i00]), l10 = xfm(space, v[i10]), l01 = xfm(space, v[i01]), l11 = xfm(space, v[i11]);
  float source[4][2] = {{l00.x, l00.y}, {l10.x, l10.y}, {l01.x, l01.y}, {l11.x, l11.y}};
  const float target[4][2] = {{-1.f, -1.f}, {1.f, -1.f}, {-1.f, 1.f}, {1.f, 1.f}};

  // proper alignment check, :147-166
  bool patchOK = !use_grid;
  Box3f lBox;
  box_empty(lBox);
  for (unsigned y = 0; y + 1 < height; y++)
    for (unsigned x = 0; x + 1 < width; x++) {
      const F3 a = xfm(space, v[y * width + x]), b = xfm(space, v[y * width + x + 1]);
      const F3 c = xfm(space, v[(y + 1) * width + x]), d = xfm(space, v[(y + 1) * width + x + 1]);
      box_extend(lBox, a); box_extend(lBox, b); box_extend(lBox, c); box_extend(lBox, d);
      if (a.x > b.x || c.x > d.x || a.y > c.y || b.y > d.y) patchOK = false;
    }
  Box3f pBox;
  box_empty(pBox);
  M3 proj;
  memset(&proj, 0, sizeof(proj));
  if (patchOK) { // :171-187
    proj = linear_estimate(source, target);
    for (const F3& p0 : v) {
      const F3 p = project(xfm(space, p0), proj);
      if (!std::isfinite(p.x) || !std::isfinite(p.y) || p.x < -1.5f || p.x > 1.5f || p.y < -1.5f || p.y > 1.5f) { patchOK = false; break; }
      box_extend(pBox, p);
    }
  }
  if (patchOK) { // :189-199
    const float s2[4][2] = {{pBox.lo[0], pBox.lo[1]}, {pBox.hi[0], pBox.lo[1]}, {pBox.lo[0], pBox.hi[1]}, {pBox.hi[0], pBox.hi[1]}};
    proj = mul(linear_estimate(s2, target), proj);
  } else { // :200-210
    const float s2[4][2] = {{lBox.lo[0], lBox.lo[1]}, {lBox.hi[0], lBox.lo[1]}, {lBox.lo[0], lBox.hi[1]}, {lBox.hi[0], lBox.hi[1]}};
    proj = linear_estimate(s2, target);
  }
  const M3 iproj = inverse(proj);

  // projected vertices (the reference recomputes them at each use)
  std::vector<F3> pv(v.size());
  for (size_t i = 0; i < v.size(); i++) pv[i] = project(xfm(space, v[i]), proj);

  // Morton-ordered leaf boxes, merged bottom-up, :381-405
  std::vector<std::vector<Box3f>> hier(C + 1);
  hier[C].resize(cells);
  for (size_t i = 0; i < cells; i++) {
    const unsigned x = cbvh_morton_x((uint32_t)i), y = cbvh_morton_y((uint32_t)i);
    Box3f b;
    box_empty(b);
    box_extend(b, pv[y * width + x]); box_extend(b, pv[y * width + x + 1]);
    box_extend(b, pv[(y + 1) * width + x]); box_extend(b, pv[(y + 1) * width + x + 1]);
    hier[C][i] = b;
  }
  for (int l = (int)C - 1; l >= 0; l--) {
    hier[l].resize(hier[l + 1].size() / 4);
    for (size_t k = 0; k < hier[l].size(); k++) {
      Box3f a = hier[l + 1][4 * k], b = hier[l + 1][4 * k + 2];
      box_extend(a, hier[l + 1][4 * k + 1]); // merge(merge(a,b),merge(c,d))
      box_extend(b, hier[l + 1][4 * k + 3]);
      box_extend(a, b);
      hier[l][k] = a;
    }
  }

  // top-down encode against re-decoded parents, world bounds from the re-decoded leaf boxes, :225-273
  Box3f projBox;
  box_empty(projBox);
  boundsOut = Box3();
  unsigned curr = 0;
  for (unsigned l = 0; l < C; l++)
    for (size_t k = 0; k < hier[l].size(); k++) {
      if (mode == CBVH_FULL) { // Node<flavor::ref,...>::setAABB (compressed_node.h:669-685): the child boxes as they are
        float* rec = (float*)(blob + CBVH_NODES_OFFSET) + (size_t)curr * (CBVH_FULL_NODE_BYTES / 4);
        for (int m = 0; m < 4; m++) {
          const Box3f& cbx = hier[l + 1][4 * k + m];
          rec[0 + m] = cbx.lo[0]; rec[4 + m] = cbx.hi[0];
          rec[8 + m] = cbx.lo[1]; rec[12 + m] = cbx.hi[1];
          rec[16 + m] = cbx.lo[2]; rec[20 + m] = cbx.hi[2];
        }
      } else
        cbvh_encode_node(hier[l][k], &hier[l + 1][4 * k], nodes[curr]);
      for (int m = 0; m < 4; m++) {
        const Box3f cb = mode == CBVH_FULL ? hier[l + 1][4 * k + m] : cbvh_decode_child(nodes[curr], hier[l][k], m); // getAABB: identity / decode
        hier[l + 1][4 * k + m] = cb;
        if (l == C - 1) {
          Box3f tb;
          box_empty(tb);
          box_extend(tb, project(f3(cb.lo[0], cb.lo[1], cb.lo[2]), iproj));
          box_extend(tb, project(f3(cb.hi[0], cb.lo[1], cb.lo[2]), iproj));
          box_extend(tb, project(f3(cb.lo[0], cb.hi[1], cb.lo[2]), iproj));
          box_extend(tb, project(f3(cb.hi[0], cb.hi[1], cb.lo[2]), iproj));
          box_extend(tb, project(f3(cb.lo[0], cb.lo[1], cb.hi[2]), iproj));
          box_extend(tb, project(f3(cb.hi[0], cb.lo[1], cb.hi[2]), iproj));
          box_extend(tb, project(f3(cb.lo[0], cb.hi[1], cb.hi[2]), iproj));
          box_extend(tb, project(f3(cb.hi[0], cb.hi[1], cb.hi[2]), iproj));
          box_extend(projBox, cb);
          // the reference's corner list: (lx,ly,uz) is absent and (ux,uy,uz) appears twice, :260-267
          const F3 cs[8] = {f3(tb.lo[0], tb.lo[1], tb.lo[2]), f3(tb.hi[0], tb.lo[1], tb.lo[2]), f3(tb.lo[0], tb.hi[1], tb.lo[2]),
                            f3(tb.hi[0], tb.hi[1], tb.lo[2]), f3(tb.hi[0], tb.lo[1], tb.hi[2]), f3(tb.lo[0], tb.hi[1], tb.hi[2]),
                            f3(tb.hi[0], tb.hi[1], tb.hi[2]), f3(tb.hi[0], tb.hi[1], tb.hi[2])};
          for (const F3& cpt : cs) {
            const F3 w = xfm(world, cpt);
            boundsOut.extend(V3(w.x, w.y, w.z));
          }
        }
      }
      curr++;
    }

  // frustum box, :277-292
  const F3 p00 = project(f3(projBox.lo[0], projBox.lo[1], projBox.lo[2]), iproj);
  const F3 p10 = project(f3(projBox.hi[0], projBox.lo[1], projBox.lo[2]), iproj);
  const F3 p01 = project(f3(projBox.lo[0], projBox.hi[1], projBox.hi[2]), iproj);
  const F3 p11 = project(f3(projBox.hi[0], projBox.hi[1], projBox.hi[2]), iproj);
  H->box[0] = projBox.lo[2]; H->box[1] = projBox.hi[2];
  H->box[2] = p00.x; H->box[3] = p00.y; H->box[4] = p10.x; H->box[5] = p10.y;
  H->box[6] = p01.x; H->box[7] = p01.y; H->box[8] = p11.x; H->box[9] = p11.y;

  // pizza-box heights, :296-327
  H->extent = 0.f;
  if (use_leaf) {
    float extent = 0.f;
    for (size_t i = 0; i < cells; i++) {
      const unsigned x = cbvh_morton_x((uint32_t)i), y = cbvh_morton_y((uint32_t)i);
      extent = fmaxf(extent, estimate_extent(hier[C][i], pv[y * width + x], pv[y * width + x + 1], pv[(y + 1) * width + x], pv[(y + 1) * width + x + 1]));
    }
    extent = fminf(extent, 1.0f); // MAX_EXTENT
    H->extent = extent;
    for (size_t i = 0; i < cells; i++) {
      const unsigned x = cbvh_morton_x((uint32_t)i), y = cbvh_morton_y((uint32_t)i);
      set_leaf_z(hier[C][i], pv[y * width + x], pv[y * width + x + 1], pv[(y + 1) * width + x], pv[(y + 1) * width + x + 1], extent, leaves + 2 * i);
    }
  }
  if (use_grid) { // :329-335
    float* g = (float*)(blob + cbvh_payload_offset(C, (uint32_t)mode));
    for (size_t i = 0; i < v.size(); i++) { g[3 * i] = v[i].x; g[3 * i + 1] = v[i].y; g[3 * i + 2] = v[i].z; }
  }

  // device-side matrices: space as rows (l.k = fma chain over the row), proj / iproj row-major
  H->space[0] = space.vx.x; H->space[1] = space.vy.x; H->space[2] = space.vz.x;
  H->space[3] = space.vx.y; H->space[4] = space.vy.y; H->space[5] = space.vz.y;
  H->space[6] = space.vx.z; H->space[7] = space.vy.z; H->space[8] = space.vz.z;
  memcpy(H->proj, proj.m, 36);
  memcpy(Ht->iproj, iproj.m, 36);
  Ht->wlo[0] = boundsOut.lo.x; Ht->wlo[1] = boundsOut.lo.y; Ht->wlo[2] = boundsOut.lo.z;
  Ht->whi[0] = boundsOut.hi.x; Ht->whi[1] = boundsOut.hi.y; Ht->whi[2] = boundsOut.hi.z;
  if (mode != CBVH_FULL) memcpy(&H->rootWord, nodes, 4); // line 0 carries the root's word: a visit rejected by the frustum test reads nothing else
}

} // namespace rtamd
