#include "subdiv_tess.h"
#include "subdiv_build.h"
#include <map>
#include <memory>
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>

namespace rtamd {

namespace {

struct D3
{
  double x = 0, y = 0, z = 0;
  D3() {}
  D3(double a, double b, double c) : x(a), y(b), z(c) {}
};
inline D3 operator+(const D3& a, const D3& b) { return D3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline D3 operator-(const D3& a, const D3& b) { return D3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline D3 operator*(const D3& a, double s) { return D3(a.x * s, a.y * s, a.z * s); }
inline D3& operator+=(D3& a, const D3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline D3 cross(const D3& a, const D3& b) { return D3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline double dot(const D3& a, const D3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// accumulators of one vertex over its incident faces / edges
struct Acc
{
  D3 sumF; // sum of adjacent face points
  D3 sumN; // sum of edge neighbours
  D3 sumB; // sum of neighbours along boundary edges
  uint32_t nf = 0, ne = 0, nb = 0;
  // semi-sharp creases on interior edges (crease weight > 0): count, sum of the far endpoints, the first two weights
  D3 sumC;
  uint32_t nc = 0;
  float cw[2] = {0.f, 0.f};
};

struct BorderEdge
{
  uint32_t idx; // running index among the unique face-border edges of this level
  uint32_t count;
  uint32_t a, b;
  D3 sumFp;
  float crease = 0.f; // semi-sharp weight of an interior edge (0: smooth)
};

struct Level
{
  unsigned n = 1;                           // cells per face side
  std::vector<D3> P;                        // vertex positions
  std::vector<std::vector<uint32_t>> grid;  // per face: (n+1)^2 vertex ids, row-major [j][i]
  std::vector<uint8_t> pinned;              // per vertex: position is fixed under refinement
  std::vector<uint8_t> bpin;                // per vertex: lies on a pinned (linear) boundary
  // creases (rtcore_geometry.h crease buffers; rules of CatmullClark1RingT::subdivide, catmullclark_ring.h:213-315):
  // weight per creased interior edge (key = edge_key of its endpoints at THIS level, weights > 0 only) and per vertex
  // (empty = no vertex creases).  Mesh boundaries are not stored here: they are creases of infinite weight by rule.
  std::unordered_map<uint64_t, float> crease;
  std::vector<float> vcrease;
  float vertex_crease(uint32_t v) const { return v < vcrease.size() ? vcrease[v] : 0.f; }
};

inline uint64_t edge_key(uint32_t a, uint32_t b) { return a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a; }

struct Refiner
{
  const Level& in;
  RTCSubdivisionMode mode;
  bool first;
  std::unordered_map<uint64_t, BorderEdge> border;
  std::vector<Acc> acc;
  std::vector<std::vector<D3>> fp; // per face: n*n face points

  Refiner(const Level& l, RTCSubdivisionMode m, bool firstLevel) : in(l), mode(m), first(firstLevel) {}

  // face points, per-vertex sums, unique border edges
  void accumulate()
  {
    const unsigned n = in.n, w = n + 1;
    const size_t nf = in.grid.size();
    acc.assign(in.P.size(), Acc());
    fp.resize(nf);
    border.reserve(nf * 4 * n * 2);
    for (size_t f = 0; f < nf; f++) {
      const std::vector<uint32_t>& g = in.grid[f];
      std::vector<D3>& F = fp[f];
      F.resize((size_t)n * n);
      for (unsigned j = 0; j < n; j++)
        for (unsigned i = 0; i < n; i++) {
          const uint32_t a = g[j * w + i], b = g[j * w + i + 1], c = g[(j + 1) * w + i + 1], d = g[(j + 1) * w + i];
          const D3 p = (in.P[a] + in.P[b] + in.P[c] + in.P[d]) * 0.25;
          F[(size_t)j * n + i] = p;
          acc[a].sumF += p; acc[a].nf++;
          acc[b].sumF += p; acc[b].nf++;
          acc[c].sumF += p; acc[c].nf++;
          acc[d].sumF += p; acc[d].nf++;
        }
      // edges strictly inside the face grid are unique to this face
      for (unsigned j = 1; j < n; j++)
        for (unsigned i = 0; i < n; i++) link(g[j * w + i], g[j * w + i + 1]);
      for (unsigned j = 0; j < n; j++)
        for (unsigned i = 1; i < n; i++) link(g[j * w + i], g[(j + 1) * w + i]);
      // edges on the face border are shared with the neighbouring face (or are mesh boundary)
      for (unsigned k = 0; k < n; k++) {
        add_border(g[k], g[k + 1], F[k]);                                              // j = 0
        add_border(g[n * w + k], g[n * w + k + 1], F[(size_t)(n - 1) * n + k]);        // j = n
        add_border(g[k * w], g[(k + 1) * w], F[(size_t)k * n]);                        // i = 0
        add_border(g[k * w + n], g[(k + 1) * w + n], F[(size_t)k * n + n - 1]);        // i = n
      }
    }
    for (auto& kv : border) {
      BorderEdge& e = kv.second;
      link(e.a, e.b);
      if (e.count != 2) { // mesh boundary (count 1) or non-manifold edge
        acc[e.a].sumB += in.P[e.b]; acc[e.a].nb++;
        acc[e.b].sumB += in.P[e.a]; acc[e.b].nb++;
      } else if (!in.crease.empty()) {
        auto it = in.crease.find(kv.first);
        if (it != in.crease.end() && it->second > 0.f) {
          e.crease = it->second;
          for (int side = 0; side < 2; side++) {
            Acc& a = acc[side ? e.b : e.a];
            a.sumC += in.P[side ? e.a : e.b];
            if (a.nc < 2) a.cw[a.nc] = e.crease;
            a.nc++;
          }
        }
      }
    }
  }

  static D3 lerp3(const D3& a, const D3& b, double t) { return a * (1.0 - t) + b * t; }
  bool creased(uint32_t v) const { return acc[v].nc != 0 || in.vertex_crease(v) > 0.f; }

  void link(uint32_t a, uint32_t b)
  {
    acc[a].sumN += in.P[b]; acc[a].ne++;
    acc[b].sumN += in.P[a]; acc[b].ne++;
  }

  void add_border(uint32_t a, uint32_t b, const D3& facePoint)
  {
    auto it = border.find(edge_key(a, b));
    if (it == border.end()) {
      BorderEdge e;
      e.idx = (uint32_t)border.size();
      e.count = 1;
      e.a = a; e.b = b;
      e.sumFp = facePoint;
      border.emplace(edge_key(a, b), e);
    } else {
      it->second.count++;
      it->second.sumFp += facePoint;
    }
  }

  // boundary interpolation rules (rtcore_geometry.h:51-58; semantics of kernels/common/scene_subdiv_mesh.cpp
  // and the crease handling of kernels/subdiv/catmullclark_ring.h)
  void classify(std::vector<uint8_t>& pinned, std::vector<uint8_t>& bpin) const
  {
    if (!first) return;
    for (size_t v = 0; v < acc.size(); v++) {
      const Acc& a = acc[v];
      if (a.nb == 0) {
        if (mode == RTC_SUBDIVISION_MODE_PIN_ALL) pinned[v] = 1;
        continue;
      }
      if (a.nb != 2) pinned[v] = 1; // non-manifold boundary: keep
      switch (mode) {
      case RTC_SUBDIVISION_MODE_PIN_CORNERS: if (a.nf == 1) pinned[v] = 1; break;
      case RTC_SUBDIVISION_MODE_PIN_BOUNDARY: pinned[v] = 1; bpin[v] = 1; break;
      case RTC_SUBDIVISION_MODE_PIN_ALL: pinned[v] = 1; bpin[v] = 1; break;
      default: break;
      }
    }
  }

  D3 vertex_point(uint32_t v, const std::vector<uint8_t>& pinned) const
  {
    const Acc& a = acc[v];
    const D3& p = in.P[v];
    if (pinned[v] || a.ne == 0) return p;
    D3 smooth;
    if (a.nb == 0) { // smooth interior vertex of valence n
      const double n = (double)a.ne;
      smooth = a.sumF * (1.0 / (n * n)) + a.sumN * (1.0 / (n * n)) + p * ((n - 2.0) / n);
    } else
      smooth = p * 0.75 + a.sumB * 0.125; // smooth boundary: cubic B-spline curve rule (= crease rule with the two border edges)
    if (!creased(v)) return smooth;
    // vertex crease, then edge creases; border edges count as creases of infinite weight (catmullclark_ring.h:276-314)
    const float vw = in.vertex_crease(v);
    if (vw > 0.f) return vw >= 1.f ? p : lerp3(smooth, p, vw);
    const uint32_t sharp = a.nb + a.nc;
    if (sharp <= 1) return smooth; // dart
    if (sharp > 2) return p;       // corner
    const D3 crease = p * 0.75 + a.sumC * 0.125; // nb == 0, nc == 2
    const double blend = 0.5 * ((double)a.cw[0] + (double)a.cw[1]);
    return blend >= 1.0 ? crease : lerp3(smooth, crease, blend);
  }

  D3 limit_point(uint32_t v, const std::vector<uint8_t>& pinned) const
  {
    const Acc& a = acc[v];
    const D3& p = in.P[v];
    if (pinned[v] || a.ne == 0) return p;
    D3 smooth;
    if (a.nb == 0) {
      const double n = (double)a.ne;
      const D3 sumD = a.sumF * 4.0 - p * n - a.sumN * 2.0; // diagonal neighbours, from F = (V+N_i+N_i+1+D_i)/4
      smooth = (p * (n * n) + a.sumN * 4.0 + sumD) * (1.0 / (n * (n + 5.0)));
    } else
      smooth = (p * 4.0 + a.sumB) * (1.0 / 6.0);
    if (!creased(v)) return smooth;
    // Creases that are still alive at the tessellation level: weights >= 1 are treated as sharp for good, fractional
    // rests blend the sharp and the smooth limit.  (The reference subdivides on adaptively until every weight has
    // decayed, patch.h:40-41; exact here whenever the weights are integers <= the level or much larger than it.)
    const float vw = in.vertex_crease(v);
    if (vw > 0.f) return vw >= 1.f ? p : lerp3(smooth, p, vw);
    const uint32_t sharp = a.nb + a.nc;
    if (sharp <= 1) return smooth;
    if (sharp > 2) return p;
    const D3 crease = (p * 4.0 + a.sumC) * (1.0 / 6.0);
    const double blend = std::min(1.0, 0.5 * ((double)a.cw[0] + (double)a.cw[1]));
    return lerp3(smooth, crease, blend);
  }

  void refine(Level& out)
  {
    const unsigned n = in.n, w = n + 1, n2 = 2 * n, w2 = n2 + 1;
    const size_t nf = in.grid.size();
    const size_t nV = in.P.size(), nBE = border.size();
    const size_t perFace = (size_t)2 * n * (n - 1) + (size_t)n * n;
    if (nV + nBE + nf * perFace > 0xFFFFFFF0ull) RT_THROW(RTC_ERROR_OUT_OF_MEMORY, "subdivision level too high for 32-bit vertex ids");
    out.n = n2;
    out.P.resize(nV + nBE + nf * perFace);
    out.pinned = in.pinned;
    out.bpin = in.bpin;
    classify(out.pinned, out.bpin);
    out.pinned.resize(out.P.size(), 0);
    out.bpin.resize(out.P.size(), 0);
    for (uint32_t v = 0; v < nV; v++) out.P[v] = vertex_point(v, out.pinned);
    const bool pinAll = mode == RTC_SUBDIVISION_MODE_PIN_ALL;
    out.crease.clear();
    out.vcrease.clear();
    if (!in.vcrease.empty()) {
      out.vcrease.assign(out.P.size(), 0.f);
      for (size_t v = 0; v < in.vcrease.size() && v < nV; v++) out.vcrease[v] = std::max(in.vcrease[v] - 1.f, 0.f);
    }
    // weight of the half of a creased edge next to endpoint v (catmullclark_ring.h:244,299-302): Chaikin's rule where
    // exactly two creases meet at a vertex without vertex crease, otherwise weight - 1
    auto child_weight = [&](uint32_t v, float wgt) -> float {
      const Acc& a = acc[v];
      if (in.vertex_crease(v) <= 0.f && a.nb == 0 && a.nc == 2) {
        const float other = a.cw[0] + a.cw[1] - wgt;
        return std::max(0.25f * (3.0f * wgt + other) - 1.0f, 0.0f);
      }
      return std::max(wgt - 1.0f, 0.0f);
    };
    for (auto& kv : border) {
      const BorderEdge& e = kv.second;
      const size_t id = nV + e.idx;
      if (e.count == 2 && !pinAll) {
        const D3 smoothE = (in.P[e.a] + in.P[e.b] + e.sumFp) * 0.25;
        if (e.crease <= 0.f) out.P[id] = smoothE;
        else {
          const D3 mid = (in.P[e.a] + in.P[e.b]) * 0.5;
          out.P[id] = e.crease >= 1.f ? mid : lerp3(smoothE, mid, e.crease);
          const float wa = child_weight(e.a, e.crease), wb = child_weight(e.b, e.crease);
          if (wa > 0.f) out.crease[edge_key(e.a, (uint32_t)id)] = wa;
          if (wb > 0.f) out.crease[edge_key(e.b, (uint32_t)id)] = wb;
        }
      } else {
        out.P[id] = (in.P[e.a] + in.P[e.b]) * 0.5;
        if (out.bpin[e.a] && out.bpin[e.b]) { out.pinned[id] = 1; out.bpin[id] = 1; } // linear boundary stays linear
      }
      if (pinAll) out.pinned[id] = 1;
    }
    out.grid.resize(nf);
    for (size_t f = 0; f < nf; f++) {
      const std::vector<uint32_t>& g = in.grid[f];
      const std::vector<D3>& F = fp[f];
      const size_t base = nV + nBE + f * perFace;
      const size_t baseV = base + (size_t)n * (n - 1), baseF = base + (size_t)2 * n * (n - 1);
      for (unsigned j = 1; j < n; j++)
        for (unsigned i = 0; i < n; i++) {
          const D3 ab = in.P[g[j * w + i]] + in.P[g[j * w + i + 1]];
          const size_t id = base + (size_t)(j - 1) * n + i;
          out.P[id] = pinAll ? ab * 0.5 : (ab + F[(size_t)(j - 1) * n + i] + F[(size_t)j * n + i]) * 0.25;
          if (pinAll) out.pinned[id] = 1;
        }
      for (unsigned j = 0; j < n; j++)
        for (unsigned i = 1; i < n; i++) {
          const D3 ab = in.P[g[j * w + i]] + in.P[g[(j + 1) * w + i]];
          const size_t id = baseV + (size_t)j * (n - 1) + (i - 1);
          out.P[id] = pinAll ? ab * 0.5 : (ab + F[(size_t)j * n + i - 1] + F[(size_t)j * n + i]) * 0.25;
          if (pinAll) out.pinned[id] = 1;
        }
      for (size_t k = 0; k < (size_t)n * n; k++) {
        out.P[baseF + k] = F[k];
        if (pinAll) out.pinned[baseF + k] = 1;
      }

      std::vector<uint32_t>& G = out.grid[f];
      G.resize((size_t)w2 * w2);
      for (unsigned J = 0; J <= n2; J++)
        for (unsigned I = 0; I <= n2; I++) {
          const unsigned i = I >> 1, j = J >> 1;
          uint32_t id;
          if (!(I & 1) && !(J & 1)) id = g[j * w + i];
          else if ((I & 1) && (J & 1)) id = (uint32_t)(baseF + (size_t)j * n + i);
          else if (I & 1) { // horizontal edge (i,j)-(i+1,j)
            if (j == 0 || j == n) id = (uint32_t)(nV + border.find(edge_key(g[j * w + i], g[j * w + i + 1]))->second.idx);
            else id = (uint32_t)(base + (size_t)(j - 1) * n + i);
          } else { // vertical edge (i,j)-(i,j+1)
            if (i == 0 || i == n) id = (uint32_t)(nV + border.find(edge_key(g[j * w + i], g[(j + 1) * w + i]))->second.idx);
            else id = (uint32_t)(baseV + (size_t)j * (n - 1) + (i - 1));
          }
          G[(size_t)J * w2 + I] = id;
        }
    }
  }
};

// limit normals at the vertices of the final level from ordered one-rings (only needed for displacement)
struct NormalEval
{
  const Level& lv;
  const Refiner& R;
  struct Corner { uint32_t next, diag, prev; };
  std::vector<uint32_t> start; // CSR over vertices
  std::vector<Corner> corners;

  NormalEval(const Level& l, const Refiner& r) : lv(l), R(r)
  {
    const unsigned n = lv.n, w = n + 1;
    std::vector<uint32_t> cnt(lv.P.size() + 1, 0);
    for (auto& g : lv.grid)
      for (unsigned j = 0; j < n; j++)
        for (unsigned i = 0; i < n; i++) {
          cnt[g[j * w + i] + 1]++; cnt[g[j * w + i + 1] + 1]++; cnt[g[(j + 1) * w + i + 1] + 1]++; cnt[g[(j + 1) * w + i] + 1]++;
        }
    for (size_t v = 0; v < lv.P.size(); v++) cnt[v + 1] += cnt[v];
    start = cnt;
    corners.resize(start.back());
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (auto& g : lv.grid)
      for (unsigned j = 0; j < n; j++)
        for (unsigned i = 0; i < n; i++) {
          const uint32_t q[4] = {g[j * w + i], g[j * w + i + 1], g[(j + 1) * w + i + 1], g[(j + 1) * w + i]};
          for (int k = 0; k < 4; k++) corners[fill[q[k]]++] = Corner{q[(k + 1) & 3], q[(k + 2) & 3], q[(k + 3) & 3]};
        }
  }

  D3 normal(uint32_t v) const
  {
    const uint32_t b = start[v], e = start[v + 1];
    const uint32_t nq = e - b;
    if (nq == 0) return D3(0, 0, 0);
    // order the incident quads counter-clockwise: Q_{i+1}.next == Q_i.prev
    uint32_t order[32];
    if (nq > 32) return D3(0, 0, 0);
    uint32_t first = b;
    for (uint32_t k = b; k < e; k++) { // boundary: start at the quad whose `next` is nobody's `prev`
      bool hasPred = false;
      for (uint32_t m = b; m < e; m++) if (m != k && corners[m].prev == corners[k].next) hasPred = true;
      if (!hasPred) { first = k; break; }
    }
    order[0] = first;
    for (uint32_t i = 1; i < nq; i++) {
      uint32_t nxt = order[i - 1];
      for (uint32_t m = b; m < e; m++) if (corners[m].next == corners[order[i - 1]].prev && m != order[i - 1]) { nxt = m; break; }
      order[i] = nxt;
    }
    const Acc& a = R.acc[v];
    const D3& P = lv.P[v];
    D3 ta, tb;
    if (a.nb == 0 && a.ne == nq && nq >= 3 && !R.creased(v)) {
      // limit tangents of an interior vertex of valence n (Halstead et al. 1993): cos / sin combinations of the ring
      const double n = (double)nq, two_pi_n = 2.0 * M_PI / n;
      const double An = 1.0 + cos(two_pi_n) + cos(M_PI / n) * sqrt(2.0 * (9.0 + cos(two_pi_n)));
      for (uint32_t i = 0; i < nq; i++) {
        const Corner& c = corners[order[i]];
        const double c0 = cos(two_pi_n * i), c1 = cos(two_pi_n * (i + 1)), s0 = sin(two_pi_n * i), s1 = sin(two_pi_n * (i + 1));
        ta += lv.P[c.next] * (An * c0) + lv.P[c.diag] * (c0 + c1);
        tb += lv.P[c.next] * (An * s0) + lv.P[c.diag] * (s0 + s1);
      }
    } else if (a.nb == 2 && nq == 2 && !R.creased(v)) {
      // regular boundary vertex: mirror the interior row across the boundary (phantom vertices of the cubic
      // B-spline end condition) and apply the regular stencil
      const Corner& q0 = corners[order[0]];
      const Corner& q1 = corners[order[1]];
      const D3 B0 = lv.P[q0.next], D0 = lv.P[q0.diag], E = lv.P[q0.prev], D1 = lv.P[q1.diag], B1 = lv.P[q1.prev];
      const D3 Em = P * 2.0 - E, D0m = B0 * 2.0 - D0, D1m = B1 * 2.0 - D1;
      // ring (ccw): e0=B0 f0=D0 e1=E f1=D1 e2=B1 f2=D1m e3=Em f3=D0m
      ta = (B0 - B1) * 4.0 + (D0 - D1 - D1m + D0m);
      tb = (E - Em) * 4.0 + (D0 + D1 - D1m - D0m);
    } else {
      // corners and irregular boundary vertices: area-weighted average of the incident quad normals
      D3 nsum;
      for (uint32_t i = 0; i < nq; i++) {
        const Corner& c = corners[order[i]];
        nsum += cross(lv.P[c.next] - P, lv.P[c.prev] - P);
      }
      return nsum;
    }
    return cross(ta, tb);
  }
};

} // namespace

// What a face of the control mesh became in the base level of the refiner.
struct FaceMap
{
  unsigned primID;   // face number in the geometry
  unsigned corners;  // 4: the face is one grid; N != 4: the face was split into N sub-patches (see below)
  unsigned sub;      // sub-patch number (0 for quads)
};

// One generic Catmull-Clark step on a polygon mesh (faces of any arity >= 3): every N-gon becomes N quads,
// sub-quad k = (vertex point of corner k, edge point of edge k -> k+1, face point, edge point of edge k-1 -> k), the
// vertex order of the reference's sub-patches (GeneralCatmullClarkPatch::subdivide, catmullclark_patch.h:442-483:
// ring[0] = corner, ring[1] = next edge, ring[2] = centre, ring[3] = previous edge).  Same rules as Refiner::refine.
static void refine_polygons(const std::vector<D3>& P, const std::vector<std::vector<uint32_t>>& faces, RTCSubdivisionMode mode,
                            const std::unordered_map<uint64_t, float>& crease, const std::vector<float>& vcrease, Level& out)
{
  auto vertex_crease = [&](uint32_t v) { return v < vcrease.size() ? vcrease[v] : 0.f; };
  auto lerp3 = [](const D3& a, const D3& b, double t) { return a * (1.0 - t) + b * t; };
  const size_t nV = P.size(), nF = faces.size();
  std::vector<Acc> acc(nV);
  std::vector<D3> fp(nF);
  struct Edge { uint32_t idx, count, a, b; D3 sumFp; float crease; };
  std::unordered_map<uint64_t, Edge> edges;
  edges.reserve(nF * 4);
  for (size_t f = 0; f < nF; f++) {
    const std::vector<uint32_t>& q = faces[f];
    D3 c;
    for (uint32_t v : q) c += P[v];
    fp[f] = c * (1.0 / (double)q.size());
    for (size_t k = 0; k < q.size(); k++) {
      acc[q[k]].sumF += fp[f];
      acc[q[k]].nf++;
      const uint32_t a = q[k], b = q[(k + 1) % q.size()];
      auto it = edges.find(edge_key(a, b));
      if (it == edges.end()) edges.emplace(edge_key(a, b), Edge{(uint32_t)edges.size(), 1u, a, b, fp[f], 0.f});
      else { it->second.count++; it->second.sumFp += fp[f]; }
    }
  }
  for (auto& kv : edges) {
    Edge& e = kv.second;
    acc[e.a].sumN += P[e.b]; acc[e.a].ne++;
    acc[e.b].sumN += P[e.a]; acc[e.b].ne++;
    if (e.count != 2) {
      acc[e.a].sumB += P[e.b]; acc[e.a].nb++;
      acc[e.b].sumB += P[e.a]; acc[e.b].nb++;
    } else {
      auto it = crease.find(kv.first);
      if (it != crease.end() && it->second > 0.f) {
        e.crease = it->second;
        for (int side = 0; side < 2; side++) {
          Acc& a = acc[side ? e.b : e.a];
          a.sumC += P[side ? e.a : e.b];
          if (a.nc < 2) a.cw[a.nc] = e.crease;
          a.nc++;
        }
      }
    }
  }
  // boundary classification exactly as Refiner::classify on the first level
  const size_t nE = edges.size();
  out.n = 1;
  out.P.resize(nV + nE + nF);
  out.pinned.assign(out.P.size(), 0);
  out.bpin.assign(out.P.size(), 0);
  for (size_t v = 0; v < nV; v++) {
    const Acc& a = acc[v];
    if (a.nb == 0) {
      if (mode == RTC_SUBDIVISION_MODE_PIN_ALL) out.pinned[v] = 1;
      continue;
    }
    if (a.nb != 2) out.pinned[v] = 1;
    switch (mode) {
    case RTC_SUBDIVISION_MODE_PIN_CORNERS: if (a.nf == 1) out.pinned[v] = 1; break;
    case RTC_SUBDIVISION_MODE_PIN_BOUNDARY: out.pinned[v] = 1; out.bpin[v] = 1; break;
    case RTC_SUBDIVISION_MODE_PIN_ALL: out.pinned[v] = 1; out.bpin[v] = 1; break;
    default: break;
    }
  }
  const bool pinAll = mode == RTC_SUBDIVISION_MODE_PIN_ALL;
  if (!vcrease.empty()) {
    out.vcrease.assign(out.P.size(), 0.f);
    for (size_t v = 0; v < vcrease.size() && v < nV; v++) out.vcrease[v] = std::max(vcrease[v] - 1.f, 0.f);
  }
  for (size_t v = 0; v < nV; v++) { // same rules as Refiner::vertex_point
    const Acc& a = acc[v];
    if (out.pinned[v] || a.ne == 0) { out.P[v] = P[v]; continue; }
    D3 smooth;
    if (a.nb == 0) {
      const double n = (double)a.ne;
      smooth = a.sumF * (1.0 / (n * n)) + a.sumN * (1.0 / (n * n)) + P[v] * ((n - 2.0) / n);
    } else
      smooth = P[v] * 0.75 + a.sumB * 0.125;
    const float vw = vertex_crease((uint32_t)v);
    const uint32_t sharp = a.nb + a.nc;
    if (a.nc == 0 && vw <= 0.f) out.P[v] = smooth;
    else if (vw > 0.f) out.P[v] = vw >= 1.f ? P[v] : lerp3(smooth, P[v], vw);
    else if (sharp <= 1) out.P[v] = smooth;
    else if (sharp > 2) out.P[v] = P[v];
    else {
      const D3 cr = P[v] * 0.75 + a.sumC * 0.125;
      const double blend = 0.5 * ((double)a.cw[0] + (double)a.cw[1]);
      out.P[v] = blend >= 1.0 ? cr : lerp3(smooth, cr, blend);
    }
  }
  auto child_weight = [&](uint32_t v, float wgt) -> float {
    const Acc& a = acc[v];
    if (vertex_crease(v) <= 0.f && a.nb == 0 && a.nc == 2) return std::max(0.25f * (3.0f * wgt + (a.cw[0] + a.cw[1] - wgt)) - 1.0f, 0.0f);
    return std::max(wgt - 1.0f, 0.0f);
  };
  for (auto& kv : edges) {
    const Edge& e = kv.second;
    const size_t id = nV + e.idx;
    if (e.count == 2 && !pinAll) {
      const D3 smoothE = (P[e.a] + P[e.b] + e.sumFp) * 0.25;
      if (e.crease <= 0.f) out.P[id] = smoothE;
      else {
        const D3 mid = (P[e.a] + P[e.b]) * 0.5;
        out.P[id] = e.crease >= 1.f ? mid : lerp3(smoothE, mid, e.crease);
        const float wa = child_weight(e.a, e.crease), wb = child_weight(e.b, e.crease);
        if (wa > 0.f) out.crease[edge_key(e.a, (uint32_t)id)] = wa;
        if (wb > 0.f) out.crease[edge_key(e.b, (uint32_t)id)] = wb;
      }
    } else {
      out.P[id] = (P[e.a] + P[e.b]) * 0.5;
      if (out.bpin[e.a] && out.bpin[e.b]) { out.pinned[id] = 1; out.bpin[id] = 1; }
    }
    if (pinAll) out.pinned[id] = 1;
  }
  for (size_t f = 0; f < nF; f++) {
    out.P[nV + nE + f] = fp[f];
    if (pinAll) out.pinned[nV + nE + f] = 1;
  }
  out.grid.clear();
  for (size_t f = 0; f < nF; f++) {
    const std::vector<uint32_t>& q = faces[f];
    const size_t N = q.size();
    const uint32_t F = (uint32_t)(nV + nE + f);
    for (size_t k = 0; k < N; k++) {
      const uint32_t eNext = (uint32_t)(nV + edges.find(edge_key(q[k], q[(k + 1) % N]))->second.idx);
      const uint32_t ePrev = (uint32_t)(nV + edges.find(edge_key(q[(k + N - 1) % N], q[k]))->second.idx);
      out.grid.push_back({q[k], eNext, ePrev, F}); // row-major 2x2: (0,0) corner, (1,0) next edge, (0,1) previous edge, (1,1) centre
    }
  }
}

// Base level of a subdivision geometry for the refiner.  All faces quads: level 0 = the control mesh, one 2x2 grid
// per valid face (`firstStep` = true: the boundary classification still has to run).  Otherwise: the mesh after ONE
// generic Catmull-Clark step, where every N-gon (quads included) is N sub-quads (`firstStep` = false, `faceMap` has
// one entry per sub-quad).  A face is valid if it is not a hole, its indices are in range and its vertices are finite
// (SubdivMesh::valid).
// `values` / `topo`: refine per-vertex data other than the positions, connected by index buffer slot `topo` (face-varying
// vertex attributes, rtcSetGeometryVertexAttributeTopology); the faces and their validity always come from topology 0.
static RTCSubdivisionMode build_base_level(const Geometry* geom, Level& cur, std::vector<FaceMap>& faceMap, bool& firstStep,
                                           const std::vector<D3>* values = nullptr, unsigned topo = 0)
{
  const BufferView* vb = geom->view(RTC_BUFFER_TYPE_VERTEX, 0);
  const BufferView* ib = geom->view(RTC_BUFFER_TYPE_INDEX, 0);
  const BufferView* fb = geom->view(RTC_BUFFER_TYPE_FACE, 0);
  if (!vb || !vb->valid() || !ib || !ib->valid() || !fb || !fb->valid())
    RT_THROW(RTC_ERROR_INVALID_OPERATION, "subdivision geometry needs vertex, index and face buffers");
  const BufferView* ibt = topo ? geom->view(RTC_BUFFER_TYPE_INDEX, topo) : ib;
  if (topo && (!ibt || !ibt->valid() || ibt->count != ib->count || !values))
    RT_THROW(RTC_ERROR_INVALID_OPERATION, "topology without a matching index buffer");
  // crease buffers (SubdivMesh::edgeCreaseMap / vertexCreaseMap, scene_subdiv_mesh.cpp:150-190): pairs of vertex indices
  // with a weight each, vertex indices with a weight each
  std::unordered_map<uint64_t, float> crease;
  std::vector<float> vcrease;
  {
    const BufferView* ei = geom->view(RTC_BUFFER_TYPE_EDGE_CREASE_INDEX, 0);
    const BufferView* ew = geom->view(RTC_BUFFER_TYPE_EDGE_CREASE_WEIGHT, 0);
    if (ei && ei->valid() && ei->count) {
      if (!ew || !ew->valid() || ew->count < ei->count) RT_THROW(RTC_ERROR_INVALID_OPERATION, "edge crease index buffer without matching weight buffer");
      for (size_t i = 0; i < ei->count; i++) {
        const unsigned* e = (const unsigned*)ei->at(i);
        const float wgt = *(const float*)ew->at(i);
        if (e[0] != e[1] && e[0] < vb->count && e[1] < vb->count && wgt > 0.f) crease[edge_key(e[0], e[1])] = wgt;
      }
    }
    const BufferView* vi = geom->view(RTC_BUFFER_TYPE_VERTEX_CREASE_INDEX, 0);
    const BufferView* vw = geom->view(RTC_BUFFER_TYPE_VERTEX_CREASE_WEIGHT, 0);
    if (vi && vi->valid() && vi->count) {
      if (!vw || !vw->valid() || vw->count < vi->count) RT_THROW(RTC_ERROR_INVALID_OPERATION, "vertex crease index buffer without matching weight buffer");
      vcrease.assign(vb->count, 0.f);
      for (size_t i = 0; i < vi->count; i++) {
        const unsigned v = *(const unsigned*)vi->at(i);
        if (v < vb->count) vcrease[v] = std::max(0.f, *(const float*)vw->at(i));
      }
    }
  }
  const RTCSubdivisionMode mode = topo < geom->subdivMode.size() ? geom->subdivMode[topo] : RTC_SUBDIVISION_MODE_SMOOTH_BOUNDARY;
  if (topo) { crease.clear(); vcrease.clear(); } // crease buffers address the vertices of the first topology

  std::vector<D3> P(vb->count);
  for (size_t i = 0; i < vb->count; i++) {
    const float* p = (const float*)vb->at(i);
    P[i] = D3(p[0], p[1], p[2]);
  }
  std::vector<uint8_t> hole(fb->count, 0);
  if (const BufferView* hb = geom->view(RTC_BUFFER_TYPE_HOLE, 0))
    if (hb->valid())
      for (size_t i = 0; i < hb->count; i++) {
        const unsigned f = *(const unsigned*)hb->at(i);
        if (f < hole.size()) hole[f] = 1;
      }

  std::vector<std::vector<uint32_t>> faces;
  std::vector<unsigned> facePrim;
  bool allQuads = true;
  size_t cursor = 0;
  for (size_t f = 0; f < fb->count; f++) {
    const unsigned nv = *(const unsigned*)fb->at(f);
    if (cursor + nv > ib->count) RT_THROW(RTC_ERROR_INVALID_OPERATION, "face buffer overruns the index buffer");
    if (nv < 3 || nv > 16) RT_THROW(RTC_ERROR_INVALID_OPERATION, "subdivision faces need 3..16 vertices (MAX_PATCH_VALENCE, catmullclark_coefficients.h:23)");
    std::vector<uint32_t> q(nv);
    bool ok = !hole[f];
    for (unsigned k = 0; k < nv; k++) {
      q[k] = *(const unsigned*)ib->at(cursor + k);
      if (q[k] >= P.size()) ok = false;
      else if (!(std::isfinite(P[q[k]].x) && std::isfinite(P[q[k]].y) && std::isfinite(P[q[k]].z))) ok = false;
    }
    if (ok && topo) // same face, other connectivity
      for (unsigned k = 0; k < nv; k++) {
        q[k] = *(const unsigned*)ibt->at(cursor + k);
        if (q[k] >= values->size()) ok = false;
      }
    cursor += nv;
    if (!ok) continue;
    if (nv != 4) allQuads = false;
    faces.push_back(std::move(q));
    facePrim.push_back((unsigned)f);
  }
  faceMap.clear();
  firstStep = allQuads;
  if (values) { // same faces and face validity, other per-vertex data (vertex attributes)
    if (topo) P.resize(values->size());
    for (size_t i = 0; i < P.size(); i++) P[i] = i < values->size() ? (*values)[i] : D3();
  }
  if (allQuads) {
    cur.n = 1;
    cur.P = std::move(P);
    cur.pinned.assign(cur.P.size(), 0);
    cur.bpin.assign(cur.P.size(), 0);
    cur.crease = crease;
    cur.vcrease = vcrease;
    cur.grid.clear();
    for (size_t f = 0; f < faces.size(); f++) {
      const std::vector<uint32_t>& q = faces[f];
      cur.grid.push_back({q[0], q[1], q[3], q[2]}); // row-major 2x2: (0,0)=v0 (1,0)=v1 (0,1)=v3 (1,1)=v2
      faceMap.push_back(FaceMap{facePrim[f], 4u, 0u});
    }
  } else {
    refine_polygons(P, faces, mode, crease, vcrease, cur);
    for (size_t f = 0; f < faces.size(); f++)
      for (unsigned k = 0; k < faces[f].size(); k++) faceMap.push_back(FaceMap{facePrim[f], (unsigned)faces[f].size(), k});
  }
  return mode;
}

// Vertex ids of the (n+1)^2 grid of one output patch.  Pure-quad meshes and sub-patches of non-quad faces: the face's
// own grid.  A quad face of a mesh that also has non-quad faces exists as four sub-quads (k = corner number) of n
// cells each; its grid at n cells per side takes every other point of them:
//   k=0: (u,v) = (s/2, t/2)   k=1: (1 - t/2, s/2)   k=2: (1 - s/2, 1 - t/2)   k=3: (t/2, 1 - s/2)
static void patch_vertex_ids(const Level& lv, const std::vector<FaceMap>& faceMap, size_t f, std::vector<uint32_t>& ids)
{
  const unsigned n = lv.n, w = n + 1;
  if (!(faceMap[f].corners == 4 && lv.grid.size() == faceMap.size() && f + 3 < faceMap.size() && faceMap[f + 3].primID == faceMap[f].primID &&
        faceMap[f + 1].sub == 1)) {
    ids = lv.grid[f];
    return;
  }
  ids.resize((size_t)w * w);
  for (unsigned J = 0; J <= n; J++)
    for (unsigned I = 0; I <= n; I++) {
      const bool lowU = 2 * I <= n, lowV = 2 * J <= n;
      unsigned k, si, ti; // sub-quad, local indices along its s and t
      if (lowU && lowV) { k = 0; si = 2 * I; ti = 2 * J; }
      else if (!lowU && lowV) { k = 1; si = 2 * J; ti = 2 * (n - I); }
      else if (!lowU && !lowV) { k = 2; si = 2 * (n - I); ti = 2 * (n - J); }
      else { k = 3; si = 2 * (n - J); ti = 2 * I; }
      ids[(size_t)J * w + I] = lv.grid[f + k][(size_t)ti * w + si];
    }
}

// L refinement rounds of `cur` + limit stencils; emits the patch grids of the faces marked in `emit` (nullptr: all faces) in
// face order.  `cur` is consumed.
static void refine_and_emit(const Geometry* geom, unsigned geomID, unsigned L, Level& cur, const std::vector<FaceMap>& faceMap, RTCSubdivisionMode mode,
                            bool first, const std::vector<uint8_t>* emit, std::vector<PatchGrid>& out)
{
  const bool mixed = !first; // faces of other arity than 4 exist: the base level already is one Catmull-Clark step deep
  for (unsigned l = 0; l < L; l++) {
    Refiner r(cur, mode, first);
    r.accumulate();
    Level next;
    r.refine(next);
    cur = std::move(next);
    first = false;
  }
  Refiner fin(cur, mode, first);
  fin.accumulate();
  std::vector<uint8_t> pinned = cur.pinned, bpin = cur.bpin;
  fin.classify(pinned, bpin);

  const unsigned n = cur.n, w = n + 1;
  const bool displ = geom->displacement != nullptr;
  std::unique_ptr<NormalEval> ne;
  if (displ) ne.reset(new NormalEval(cur, fin));

  // limit positions are shared between faces: evaluate each vertex once (only the vertices of emitted faces when a subset is asked for)
  std::vector<D3> limit(cur.P.size());
  if (!emit)
    for (uint32_t v = 0; v < cur.P.size(); v++) limit[v] = fin.limit_point(v, pinned);
  else {
    std::vector<uint8_t> need(cur.P.size(), 0);
    for (size_t f = 0; f < cur.grid.size(); f++)
      if ((*emit)[f])
        for (uint32_t v : cur.grid[f]) need[v] = 1;
    for (uint32_t v = 0; v < cur.P.size(); v++)
      if (need[v]) limit[v] = fin.limit_point(v, pinned);
  }

  const size_t N = (size_t)w * w;
  const bool noBoundary = mode == RTC_SUBDIVISION_MODE_NO_BOUNDARY;
  std::vector<float> gu(N), gv(N), nx, ny, nz;
  std::vector<uint32_t> g;
  for (size_t f = 0; f < cur.grid.size(); f++) {
    const FaceMap& fm = faceMap[f];
    if (emit && !(*emit)[f]) continue;
    if (mixed && fm.corners == 4 && fm.sub != 0) continue; // sub-quads 1..3 of a quad face are consumed with sub-quad 0
    patch_vertex_ids(cur, faceMap, f, g);
    if (noBoundary) { // RTC_SUBDIVISION_MODE_NO_BOUNDARY: patches touching the border are not rendered
      bool touches = false;
      for (size_t k = 0; k < N && !touches; k++) touches = fin.acc[g[k]].nb != 0;
      if (touches) continue;
    }
    out.emplace_back();
    PatchGrid& pg = out.back();
    pg.geomID = geomID;
    pg.primID = fm.primID;
    pg.n = n;
    if (fm.corners != 4) { // sub-patch number in the integer part of uv, patch_eval_grid.h:241-254
      pg.u0 = 2.0f * (float)(fm.sub & 3) + 0.5f;
      pg.v0 = 2.0f * (float)((fm.sub >> 2) & 3) + 0.5f;
    }
    pg.x.resize(N); pg.y.resize(N); pg.z.resize(N);
    for (size_t k = 0; k < N; k++) {
      const D3& p = limit[g[k]];
      pg.x[k] = (float)p.x; pg.y[k] = (float)p.y; pg.z[k] = (float)p.z;
    }
    // development aid (encoder sensitivity): RTAMD_DEBUG_GRID_JITTER=<seed> moves every grid coordinate by -1, 0 or +1 ulp,
    // the same way for equal positions (a hash of the position's bits), i.e. by less than the difference between two correct fp32 evaluations
    // of the limit surface - what the reference's float B-spline evaluation and this double-precision refinement differ by
    static const char* jit = getenv("RTAMD_DEBUG_GRID_JITTER");
    if (jit && atoi(jit) != 0) {
      const uint32_t seed = (uint32_t)atoi(jit);
      for (size_t k = 0; k < N; k++) {
        uint32_t b[3];
        memcpy(&b[0], &pg.x[k], 4); memcpy(&b[1], &pg.y[k], 4); memcpy(&b[2], &pg.z[k], 4);
        uint32_t h = seed * 0x9E3779B9u ^ b[0];
        h = (h ^ (h >> 15)) * 0x2C1B3C6Du ^ b[1];
        h = (h ^ (h >> 12)) * 0x297A2D39u ^ b[2];
        h ^= h >> 15;
        float* c[3] = {&pg.x[k], &pg.y[k], &pg.z[k]};
        for (int a = 0; a < 3; a++) {
          const uint32_t r = (h >> (8 * a)) % 3u;
          if (r == 1u) *c[a] = nextafterf(*c[a], INFINITY);
          else if (r == 2u) *c[a] = nextafterf(*c[a], -INFINITY);
        }
      }
    }
    if (displ) {
      pg.bx = pg.x; pg.by = pg.y; pg.bz = pg.z;
      nx.resize(N); ny.resize(N); nz.resize(N);
      for (unsigned j = 0; j <= n; j++)
        for (unsigned i = 0; i <= n; i++) {
          gu[(size_t)j * w + i] = pg.u0 + (float)i / (float)n; // gridUVTessellator: (x0+i) * rcp(n), exact for n = 2^L
          gv[(size_t)j * w + i] = pg.v0 + (float)j / (float)n;
        }
      for (size_t k = 0; k < N; k++) {
        D3 nn = ne->normal(g[k]);
        const double len2 = dot(nn, nn);
        if (len2 > 0.0) nn = nn * (1.0 / sqrt(len2)); // normalize_safe
        nx[k] = (float)nn.x; ny[k] = (float)nn.y; nz[k] = (float)nn.z;
      }
      // displacement callback protocol: subdivpatch1base_eval.cpp:139-156
      RTCDisplacementFunctionNArguments args;
      args.geometryUserPtr = geom->userPtr;
      args.geometry = (RTCGeometry)geom;
      args.primID = pg.primID;
      args.timeStep = 0;
      args.u = gu.data(); args.v = gv.data();
      args.Ng_x = nx.data(); args.Ng_y = ny.data(); args.Ng_z = nz.data();
      args.P_x = pg.x.data(); args.P_y = pg.y.data(); args.P_z = pg.z.data();
      args.N = (unsigned)N;
      geom->displacement(&args);
    }
  }
}

// Tessellation in CHUNKS of faces, on all host threads (round 3).  Catmull-Clark refinement is local: the level-l points of a face
// depend on the face's one-ring only.  A chunk = a few neighbouring faces of the base level plus their one-ring (every face that
// shares a vertex with a chunk face); that sub-mesh is refined L times on its own and only the chunk faces' grids are kept.  The cut
// turns the outer border of the halo into an artificial mesh boundary; what is computed wrongly because of it moves inwards by one
// ring of the CURRENT level per round: after round k the wrong points keep a distance of at least 2^-(k-1) base cells from the chunk
// faces, i.e. two cells of the final level, while the limit (and normal) stencils of the chunk faces' border vertices reach one cell
// out.  The grids are therefore the ones the whole-mesh refinement produces (same formulas on the same one-rings; only the order in
// which a vertex's neighbours are summed can differ, in double precision, before the rounding to float).
// Why: the whole-mesh refinement is one thread walking gigabytes (bomberman L6 2 s, L7 25 s, L8 125 s); chunks stay in cache and run
// in parallel (rtcCommitScene is internally parallel in the reference too, scene.cpp:727-786).  The displacement callback is then
// called concurrently from the builder threads, as the reference does (subdivpatch1base_eval.cpp:139-156 under parallel_for).
static void tessellate_chunked(const Geometry* geom, unsigned geomID, unsigned L, const Level& base, const std::vector<FaceMap>& faceMap,
                               RTCSubdivisionMode mode, bool first, unsigned threads, size_t chunkFaces, std::vector<PatchGrid>& out)
{
  const size_t nF = base.grid.size(), nV = base.P.size();
  // faces of one control face stay together (a quad of a mixed mesh is re-assembled from its four sub-quads): group = run of equal primID
  std::vector<uint32_t> groupOf(nF), groupStart;
  for (size_t f = 0; f < nF; f++) {
    if (f == 0 || faceMap[f].primID != faceMap[f - 1].primID) groupStart.push_back((uint32_t)f);
    groupOf[f] = (uint32_t)groupStart.size() - 1;
  }
  groupStart.push_back((uint32_t)nF);
  const size_t nG = groupStart.size() - 1;
  // vertex -> faces (CSR)
  std::vector<uint32_t> vStart(nV + 1, 0), vFaces;
  for (const auto& g : base.grid)
    for (uint32_t v : g) vStart[v + 1]++;
  for (size_t v = 0; v < nV; v++) vStart[v + 1] += vStart[v];
  vFaces.resize(vStart[nV]);
  {
    std::vector<uint32_t> fill(vStart.begin(), vStart.end() - 1);
    for (size_t f = 0; f < nF; f++)
      for (uint32_t v : base.grid[f]) vFaces[fill[v]++] = (uint32_t)f;
  }
  // chunks: breadth-first over vertex-adjacent groups, so that a chunk is a compact neighbourhood (small halo)
  std::vector<std::vector<uint32_t>> chunks; // groups of each chunk
  {
    std::vector<uint8_t> taken(nG, 0);
    for (size_t g0 = 0; g0 < nG; g0++) {
      if (taken[g0]) continue;
      std::vector<uint32_t> c{(uint32_t)g0};
      taken[g0] = 1;
      size_t faces = groupStart[g0 + 1] - groupStart[g0];
      for (size_t head = 0; head < c.size() && faces < chunkFaces; head++)
        for (uint32_t f = groupStart[c[head]]; f < groupStart[c[head] + 1] && faces < chunkFaces; f++)
          for (uint32_t v : base.grid[f])
            for (uint32_t k = vStart[v]; k < vStart[v + 1] && faces < chunkFaces; k++) {
              const uint32_t g = groupOf[vFaces[k]];
              if (taken[g]) continue;
              taken[g] = 1;
              c.push_back(g);
              faces += groupStart[g + 1] - groupStart[g];
            }
      std::sort(c.begin(), c.end());
      chunks.push_back(std::move(c));
    }
  }
  std::vector<std::vector<PatchGrid>> results(chunks.size());
  parallel_for_range(chunks.size(), threads, [&](size_t c0, size_t c1) {
    std::vector<uint32_t> toLocal(nV, 0xFFFFFFFFu); // per worker call; chunks are few thousand vertices, the reset below is per chunk
    for (size_t c = c0; c < c1; c++) {
      // faces of the chunk, then every face that shares a vertex with one of them
      std::vector<uint32_t> own, local;
      for (uint32_t g : chunks[c])
        for (uint32_t f = groupStart[g]; f < groupStart[g + 1]; f++) own.push_back(f);
      local = own;
      for (uint32_t f : own)
        for (uint32_t v : base.grid[f])
          for (uint32_t k = vStart[v]; k < vStart[v + 1]; k++) local.push_back(vFaces[k]);
      std::sort(local.begin(), local.end());
      local.erase(std::unique(local.begin(), local.end()), local.end());
      // halo faces drag in the rest of their control face's group only when it matters: patch_vertex_ids looks at f .. f+3 of EMITTED faces
      std::vector<uint32_t> verts;
      for (uint32_t f : local)
        for (uint32_t v : base.grid[f]) verts.push_back(v);
      std::sort(verts.begin(), verts.end());
      verts.erase(std::unique(verts.begin(), verts.end()), verts.end());
      for (size_t i = 0; i < verts.size(); i++) toLocal[verts[i]] = (uint32_t)i;
      Level lv;
      lv.n = base.n;
      lv.P.resize(verts.size());
      lv.pinned.resize(verts.size());
      lv.bpin.resize(verts.size());
      if (!base.vcrease.empty()) lv.vcrease.assign(verts.size(), 0.f);
      for (size_t i = 0; i < verts.size(); i++) {
        const uint32_t v = verts[i];
        lv.P[i] = base.P[v];
        lv.pinned[i] = base.pinned[v];
        lv.bpin[i] = base.bpin[v];
        if (!base.vcrease.empty() && v < base.vcrease.size()) lv.vcrease[i] = base.vcrease[v];
      }
      std::vector<FaceMap> fmLocal;
      std::vector<uint8_t> emit;
      lv.grid.reserve(local.size());
      size_t o = 0;
      for (uint32_t f : local) {
        std::vector<uint32_t> g = base.grid[f];
        if (!base.crease.empty()) { // the 2x2 grid of a base face is (0,0) (1,0) (0,1) (1,1): its border edges
          static const int E[4][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}};
          for (const auto& e : E) {
            auto it = base.crease.find(edge_key(g[e[0]], g[e[1]]));
            if (it != base.crease.end()) lv.crease[edge_key(toLocal[g[e[0]]], toLocal[g[e[1]]])] = it->second;
          }
        }
        for (uint32_t& v : g) v = toLocal[v];
        lv.grid.push_back(std::move(g));
        fmLocal.push_back(faceMap[f]);
        while (o < own.size() && own[o] < f) o++;
        emit.push_back(o < own.size() && own[o] == f ? 1 : 0);
      }
      for (uint32_t v : verts) toLocal[v] = 0xFFFFFFFFu;
      refine_and_emit(geom, geomID, L, lv, fmLocal, mode, first, &emit, results[c]);
    }
  });
  // back into face order (chunks hold ascending groups, but interleave with each other)
  struct Slot { uint32_t primID, sub; PatchGrid* pg; };
  std::vector<Slot> slots;
  for (auto& r : results)
    for (PatchGrid& pg : r) slots.push_back(Slot{pg.primID, 0u, &pg});
  // sub-patches of one control face were emitted in order inside their chunk; a stable sort by primID restores the global order
  std::stable_sort(slots.begin(), slots.end(), [](const Slot& a, const Slot& b) { return a.primID < b.primID; });
  out.reserve(out.size() + slots.size());
  for (const Slot& sl : slots) out.push_back(std::move(*sl.pg));
}

void tessellate_subdiv(const Geometry* geom, unsigned geomID, unsigned L, std::vector<PatchGrid>& out, unsigned threads)
{
  if (L > 10) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "subdivision level too high");
  Level cur;
  std::vector<FaceMap> faceMap;
  bool first = true;
  const RTCSubdivisionMode mode = build_base_level(geom, cur, faceMap, first);
  if (cur.grid.empty()) return;
  // chunked (parallel, cache-resident) from ~1 M final cells on; RTAMD_TESS_CHUNK=0 forces the whole-mesh refinement, =N sets the
  // faces per chunk
  static const char* env = getenv("RTAMD_TESS_CHUNK");
  const size_t envChunk = env ? (size_t)atol(env) : (size_t)-1;
  const double cells = (double)cur.grid.size() * (double)((size_t)1 << (2 * L));
  const bool chunked = env ? envChunk != 0 : (cells >= 1.0e6 && cur.grid.size() >= 8);
  if (!chunked) {
    out.reserve(out.size() + cur.grid.size());
    refine_and_emit(geom, geomID, L, cur, faceMap, mode, first, nullptr, out);
    return;
  }
  // faces per chunk: enough work per chunk to amortise the halo (16 faces + ~24 halo faces), fewer when a level is large enough that the
  // chunk's working set (~120 B per final vertex and face of the sub-mesh) would leave the caches
  const size_t chunkFaces = env && envChunk ? envChunk : (L >= 8 ? 4 : (L >= 7 ? 8 : 16));
  tessellate_chunked(geom, geomID, L, cur, faceMap, mode, first, std::max(1u, threads), chunkFaces, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// rtcInterpolate
// ---------------------------------------------------------------------------------------------------------------------
// Triangle meshes: the reference's arithmetic (scene_triangle_mesh.cpp:214-270).
void interpolate_triangles(const Geometry* geom, const RTCInterpolateArguments* args)
{
  const BufferView* src = geom->view(args->bufferType, args->bufferSlot);
  if (!src || !src->valid()) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "rtcInterpolate: buffer slot is not bound");
  if (args->primID >= geom->numTriangles()) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "rtcInterpolate: invalid primID");
  unsigned idx[3];
  geom->triangle(args->primID, idx);
  const float u = args->u, v = args->v, w = 1.0f - u - v;
  for (unsigned i = 0; i < args->valueCount; i++) {
    const float p0 = ((const float*)src->at(idx[0]))[i], p1 = ((const float*)src->at(idx[1]))[i], p2 = ((const float*)src->at(idx[2]))[i];
    if (args->P) args->P[i] = fmaf(w, p0, fmaf(u, p1, v * p2));
    if (args->dPdu) { args->dPdu[i] = p1 - p0; args->dPdv[i] = p2 - p0; }
    if (args->ddPdudu) { args->ddPdudu[i] = 0.f; args->ddPdvdv[i] = 0.f; args->ddPdudv[i] = 0.f; }
  }
}

// Subdivision meshes: Catmull-Clark limit surface of any vertex / vertex-attribute buffer at (primID, u, v).
// The reference evaluates through patch classification, bicubic B-spline patches, feature-adaptive subdivision and
// Gregory patches (scene_subdiv_mesh.cpp:757-864, patch_eval.h).  Here the buffer is refined K = 3 times with the
// tessellator's refiner (per three channels); a sub-face all of whose corners are regular interior vertices is a uniform
// bicubic B-spline patch over its 4x4 neighbourhood - the exact limit surface with exact derivatives.  The remaining
// sub-faces (touching an extraordinary vertex, a crease or the boundary: 1/64 of a face per such corner) are evaluated
// FEATURE-ADAPTIVELY like the reference's FeatureAdaptiveEval (feature_adaptive_eval.h:130-140, patch.h:40-42): the
// sub-face and its one-ring are cut out, refined once more, the quarter that contains (u,v) and ITS one-ring are cut out
// again, and so on until that quarter is a regular B-spline patch or the reference's maximum depth of 10 levels below
// the base face is reached (PATCH_MAX_EVAL_DEPTH_IRREGULAR / _CREASE); only there - a cell of 2^-10 of the face's side,
// where the reference fills in a Gregory patch (PATCH_USE_GREGORY == 1: "fill") - the limit points of the cell's
// corners are interpolated bilinearly.  (One-rings suffice: every point of a child cell's one-ring is a vertex, edge or
// face point whose stencil lies inside the parent cell's one-ring.)
struct InterpChannels
{
  Level lvl;              // level-K values of three channels
  std::vector<D3> limit;  // their limit points
};
struct SubdivInterpCache
{
  unsigned n = 8; // sub-faces per base-face side (2^K)
  RTCSubdivisionMode mode = RTC_SUBDIVISION_MODE_SMOOTH_BOUNDARY; // boundary rule of the geometry (the refiner's PIN_ALL edge rule needs it)
  bool mixed = false;             // the mesh has faces that are not quads: base faces are the sub-quads of one generic step
  std::vector<int> primToFace;    // primID -> first base face (-1: invalid face)
  std::vector<unsigned> primCorners; // primID -> number of corners
  std::vector<std::vector<uint32_t>> grid;   // level-K vertex ids per face
  std::vector<uint32_t> vqStart, vqList;     // vertex -> incident sub-faces (id = face*n*n + j*n + i)
  std::vector<uint8_t> irregular;            // vertex is on a boundary, pinned, or has valence != 4
  std::map<std::pair<int, unsigned>, std::vector<InterpChannels>> buffers; // (type, slot) -> channel triples
  std::mutex mutex;
};
static const unsigned INTERP_LEVELS = 3;
static const unsigned INTERP_MAX_DEPTH = 10; // PATCH_MAX_EVAL_DEPTH_IRREGULAR / _CREASE (patch.h:40-41)

static void refine_to_interp_level(Level& cur, RTCSubdivisionMode mode, bool first, std::vector<D3>* limit, std::vector<uint8_t>* irregular)
{
  for (unsigned l = 0; l < INTERP_LEVELS; l++) {
    Refiner r(cur, mode, first);
    r.accumulate();
    Level next;
    r.refine(next);
    cur = std::move(next);
    first = false;
  }
  Refiner fin(cur, mode, first);
  fin.accumulate();
  std::vector<uint8_t> pinned = cur.pinned, bpin = cur.bpin;
  fin.classify(pinned, bpin);
  if (limit) {
    limit->resize(cur.P.size());
    for (uint32_t v = 0; v < cur.P.size(); v++) (*limit)[v] = fin.limit_point(v, pinned);
  }
  if (irregular) {
    irregular->resize(cur.P.size());
    for (size_t v = 0; v < cur.P.size(); v++) (*irregular)[v] = (fin.acc[v].nb != 0 || pinned[v] || fin.acc[v].nf != 4 || fin.acc[v].ne != 4 || fin.creased((uint32_t)v)) ? 1 : 0;
  }
}

static void cubic_bspline(double s, double B[4], double dB[4], double ddB[4])
{
  const double s2 = s * s, s3 = s2 * s, r = 1.0 - s;
  B[0] = r * r * r / 6.0; B[1] = (3.0 * s3 - 6.0 * s2 + 4.0) / 6.0; B[2] = (-3.0 * s3 + 3.0 * s2 + 3.0 * s + 1.0) / 6.0; B[3] = s3 / 6.0;
  dB[0] = -0.5 * r * r; dB[1] = 1.5 * s2 - 2.0 * s; dB[2] = -1.5 * s2 + s + 0.5; dB[3] = 0.5 * s2;
  ddB[0] = r; ddB[1] = 3.0 * s - 2.0; ddB[2] = -3.0 * s + 1.0; ddB[3] = s;
}

// ---- local feature-adaptive evaluation (see the comment above SubdivInterpCache) ---------------------------------------
struct MiniCell { uint32_t v[4]; }; // corners (0,0), (1,0), (1,1), (0,1) of a cell, ids of the level it was cut from

// a level of single-quad faces (n = 1) made of `cells` (face k = cells[k]) with the per-vertex / per-edge state of `src`
static void mini_from_cells(const Level& src, const std::vector<MiniCell>& cells, Level& out)
{
  out = Level();
  out.n = 1;
  std::unordered_map<uint32_t, uint32_t> id;
  auto map_vertex = [&](uint32_t v) -> uint32_t {
    auto it = id.find(v);
    if (it != id.end()) return it->second;
    const uint32_t k = (uint32_t)id.size();
    id.emplace(v, k);
    out.P.push_back(src.P[v]);
    out.pinned.push_back(v < src.pinned.size() ? src.pinned[v] : 0);
    out.bpin.push_back(v < src.bpin.size() ? src.bpin[v] : 0);
    return k;
  };
  out.grid.reserve(cells.size());
  for (const MiniCell& c : cells) {
    const uint32_t a = map_vertex(c.v[0]), b = map_vertex(c.v[1]), cc = map_vertex(c.v[2]), d = map_vertex(c.v[3]);
    out.grid.push_back({a, b, d, cc}); // row-major [j][i]
  }
  if (!src.vcrease.empty()) {
    out.vcrease.assign(out.P.size(), 0.f);
    for (auto& kv : id) out.vcrease[kv.second] = src.vertex_crease(kv.first);
  }
  if (!src.crease.empty())
    for (const MiniCell& c : cells)
      for (int k = 0; k < 4; k++) {
        const uint32_t a = c.v[k], b = c.v[(k + 1) & 3];
        auto it = src.crease.find(edge_key(a, b));
        if (it != src.crease.end() && it->second > 0.f) out.crease[edge_key(id[a], id[b])] = it->second;
      }
}

struct PatchValue { D3 P, Pu, Pv, Puu, Pvv, Puv; }; // derivatives w.r.t. the cell's own (s, t) in [0,1]^2

static PatchValue eval_bspline16(const std::vector<D3>& P, const uint32_t C[4][4], double s, double t)
{
  double Bu[4], dBu[4], ddBu[4], Bv[4], dBv[4], ddBv[4];
  cubic_bspline(s, Bu, dBu, ddBu);
  cubic_bspline(t, Bv, dBv, ddBv);
  PatchValue o;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      const D3& cp = P[C[r][c]];
      o.P += cp * (Bv[r] * Bu[c]);
      o.Pu += cp * (Bv[r] * dBu[c]);
      o.Pv += cp * (dBv[r] * Bu[c]);
      o.Puu += cp * (Bv[r] * ddBu[c]);
      o.Pvv += cp * (ddBv[r] * Bu[c]);
      o.Puv += cp * (dBv[r] * dBu[c]);
    }
  return o;
}

static PatchValue eval_bilinear(const D3& L00, const D3& L10, const D3& L11, const D3& L01, double s, double t)
{
  PatchValue o;
  o.P = L00 * ((1 - s) * (1 - t)) + L10 * (s * (1 - t)) + L01 * ((1 - s) * t) + L11 * (s * t);
  o.Pu = (L10 - L00) * (1 - t) + (L11 - L01) * t;
  o.Pv = (L01 - L00) * (1 - s) + (L11 - L10) * s;
  o.Puv = (L11 - L10) - (L01 - L00);
  return o;
}

// the 4x4 control net around face `self` of a level of single-quad faces, if all four corners are regular interior vertices
static bool mini_regular_net(const Level& lv, const Refiner& R, uint32_t self, uint32_t C[4][4])
{
  auto corners = [&](uint32_t f, uint32_t out[4]) { const std::vector<uint32_t>& g = lv.grid[f]; out[0] = g[0]; out[1] = g[1]; out[2] = g[3]; out[3] = g[2]; };
  uint32_t q[4];
  corners(self, q);
  for (int k = 0; k < 4; k++) {
    const Acc& a = R.acc[q[k]];
    if (a.nb != 0 || a.nf != 4 || a.ne != 4 || lv.pinned[q[k]] || R.creased(q[k])) return false;
  }
  const uint32_t nf = (uint32_t)lv.grid.size();
  auto across = [&](uint32_t a, uint32_t b, uint32_t& oa, uint32_t& ob) -> int64_t { // the other face with edge (a,b); oa / ob: its vertices next to a / b
    for (uint32_t f = 0; f < nf; f++) {
      if (f == self) continue;
      uint32_t v[4];
      corners(f, v);
      int ia = -1, ib = -1;
      for (int m = 0; m < 4; m++) { if (v[m] == a) ia = m; if (v[m] == b) ib = m; }
      if (ia < 0 || ib < 0) continue;
      if ((ia + 1) % 4 == ib) { oa = v[(ia + 3) % 4]; ob = v[(ib + 1) % 4]; }
      else if ((ib + 1) % 4 == ia) { oa = v[(ia + 1) % 4]; ob = v[(ib + 3) % 4]; }
      else continue;
      return f;
    }
    return -1;
  };
  auto diagonal = [&](uint32_t a, int64_t n1, int64_t n2, uint32_t& od) -> bool { // the fourth face around a: its vertex opposite a
    for (uint32_t f = 0; f < nf; f++) {
      if (f == self || (int64_t)f == n1 || (int64_t)f == n2) continue;
      uint32_t v[4];
      corners(f, v);
      for (int m = 0; m < 4; m++)
        if (v[m] == a) { od = v[(m + 2) % 4]; return true; }
    }
    return false;
  };
  const uint32_t c00 = q[0], c10 = q[1], c11 = q[2], c01 = q[3];
  C[1][1] = c00; C[1][2] = c10; C[2][2] = c11; C[2][1] = c01;
  const int64_t qb = across(c00, c10, C[0][1], C[0][2]);
  const int64_t qr = across(c10, c11, C[1][3], C[2][3]);
  const int64_t qt = across(c11, c01, C[3][2], C[3][1]);
  const int64_t ql = across(c01, c00, C[2][0], C[1][0]);
  return qb >= 0 && qr >= 0 && qt >= 0 && ql >= 0 && diagonal(c00, qb, ql, C[0][0]) && diagonal(c10, qb, qr, C[0][3]) &&
         diagonal(c11, qr, qt, C[3][3]) && diagonal(c01, qt, ql, C[3][0]);
}

// Evaluates the cell ring[0] of level `lvK` (ring = the cell and every other cell that shares a vertex with it) at (s, t).
// `depthOut` = number of refinements below lvK at which the value was taken (the derivatives are w.r.t. that cell's own
// parameters: the caller scales them by 2^depth).
static PatchValue adaptive_eval(const Level& lvK, RTCSubdivisionMode mode, const std::vector<MiniCell>& ring, double s, double t,
                                unsigned maxDepth, unsigned& depthOut)
{
  Level mini;
  mini_from_cells(lvK, ring, mini);
  for (unsigned d = 0;; d++) {
    Refiner R(mini, mode, false);
    R.accumulate();
    uint32_t C[4][4];
    depthOut = d;
    if (mini_regular_net(mini, R, 0u, C)) return eval_bspline16(mini.P, C, s, t);
    const std::vector<uint32_t>& g = mini.grid[0];
    if (d >= maxDepth)
      return eval_bilinear(R.limit_point(g[0], mini.pinned), R.limit_point(g[1], mini.pinned), R.limit_point(g[3], mini.pinned),
                           R.limit_point(g[2], mini.pinned), s, t);
    Level next;
    R.refine(next);
    const unsigned ci = s >= 0.5 ? 1u : 0u, cj = t >= 0.5 ? 1u : 0u;
    s = std::min(1.0, std::max(0.0, 2.0 * s - ci));
    t = std::min(1.0, std::max(0.0, 2.0 * t - cj));
    auto cell_of = [&](size_t f, unsigned i, unsigned j) {
      const std::vector<uint32_t>& G = next.grid[f]; // 3 x 3
      MiniCell c;
      c.v[0] = G[j * 3 + i]; c.v[1] = G[j * 3 + i + 1]; c.v[2] = G[(j + 1) * 3 + i + 1]; c.v[3] = G[(j + 1) * 3 + i];
      return c;
    };
    std::vector<MiniCell> cells;
    const MiniCell child = cell_of(0, ci, cj);
    cells.push_back(child);
    for (size_t f = 0; f < next.grid.size(); f++)
      for (unsigned j = 0; j < 2; j++)
        for (unsigned i = 0; i < 2; i++) {
          if (f == 0 && i == ci && j == cj) continue;
          const MiniCell c = cell_of(f, i, j);
          bool touches = false;
          for (int a = 0; a < 4 && !touches; a++)
            for (int b = 0; b < 4; b++)
              if (c.v[a] == child.v[b]) { touches = true; break; }
          if (touches) cells.push_back(c);
        }
    Level cut;
    mini_from_cells(next, cells, cut);
    mini = std::move(cut);
  }
}

void interpolate_subdiv(Geometry* geom, const RTCInterpolateArguments* args)
{
  if (args->bufferType != RTC_BUFFER_TYPE_VERTEX && args->bufferType != RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE)
    RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "rtcInterpolate: invalid buffer type");
  const BufferView* src = geom->view(args->bufferType, args->bufferSlot);
  if (!src || !src->valid()) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "rtcInterpolate: buffer slot is not bound");
  if (args->valueCount > 256) RT_THROW(RTC_ERROR_INVALID_OPERATION, "maximally 256 floating point values can be interpolated per vertex");

  unsigned topo = 0;
  if (args->bufferType == RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE && args->bufferSlot < geom->attribTopology.size()) topo = geom->attribTopology[args->bufferSlot];
  typedef std::map<unsigned, std::shared_ptr<SubdivInterpCache>> CacheMap; // one per topology
  std::shared_ptr<SubdivInterpCache> cache;
  {
    std::lock_guard<std::mutex> g(geom->interpMutex);
    if (!geom->interpCache) geom->interpCache = std::make_shared<CacheMap>();
    CacheMap& caches = *std::static_pointer_cast<CacheMap>(geom->interpCache);
    cache = caches[topo];
    if (!cache) { // connectivity part, once per commit of the geometry and topology
      cache = std::make_shared<SubdivInterpCache>();
      Level cur;
      std::vector<FaceMap> faceMap;
      bool first = true;
      std::vector<D3> dummy;
      if (topo) { // only the connectivity matters here: any per-vertex values of the right count
        const BufferView* any = geom->view(args->bufferType, args->bufferSlot);
        dummy.assign(any->count, D3());
      }
      const RTCSubdivisionMode mode = build_base_level(geom, cur, faceMap, first, topo ? &dummy : nullptr, topo);
      cache->mode = mode;
      cache->mixed = !first;
      const BufferView* fb = geom->view(RTC_BUFFER_TYPE_FACE, 0);
      cache->primToFace.assign(fb->count, -1);
      cache->primCorners.assign(fb->count, 0);
      for (size_t f = faceMap.size(); f-- > 0;) { // descending: the entry that stays is the first base face of the prim
        cache->primToFace[faceMap[f].primID] = (int)f;
        cache->primCorners[faceMap[f].primID] = faceMap[f].corners;
      }
      refine_to_interp_level(cur, mode, first, nullptr, &cache->irregular);
      cache->n = cur.n;
      cache->grid = cur.grid;
      const unsigned n = cur.n, w = n + 1;
      std::vector<uint32_t> deg(cur.P.size() + 1, 0);
      for (size_t f = 0; f < cur.grid.size(); f++)
        for (unsigned j = 0; j < n; j++)
          for (unsigned i = 0; i < n; i++)
            for (uint32_t v : {cur.grid[f][j * w + i], cur.grid[f][j * w + i + 1], cur.grid[f][(j + 1) * w + i + 1], cur.grid[f][(j + 1) * w + i]}) deg[v + 1]++;
      for (size_t v = 0; v < cur.P.size(); v++) deg[v + 1] += deg[v];
      cache->vqStart = deg;
      cache->vqList.resize(deg.back());
      std::vector<uint32_t> fill(deg.begin(), deg.end() - 1);
      for (size_t f = 0; f < cur.grid.size(); f++)
        for (unsigned j = 0; j < n; j++)
          for (unsigned i = 0; i < n; i++) {
            const uint32_t q = (uint32_t)(f * n * n + (size_t)j * n + i);
            for (uint32_t v : {cur.grid[f][j * w + i], cur.grid[f][j * w + i + 1], cur.grid[f][(j + 1) * w + i + 1], cur.grid[f][(j + 1) * w + i]}) cache->vqList[fill[v]++] = q;
          }
      caches[topo] = cache;
    }
  }
  const unsigned groups = (args->valueCount + 2) / 3;
  const std::vector<InterpChannels>* chans = nullptr;
  {
    std::lock_guard<std::mutex> g(cache->mutex);
    std::vector<InterpChannels>& c = cache->buffers[std::make_pair((int)args->bufferType, args->bufferSlot)];
    if (c.size() < groups) { // refine the channels not seen before
      const size_t floatsPerVertex = src->stride / sizeof(float);
      for (unsigned gi = (unsigned)c.size(); gi < groups; gi++) {
        // the topology comes from the vertex buffer (face validity), the values from three channels of `src`
        std::vector<D3> values(src->count);
        for (size_t v = 0; v < src->count; v++) {
          const float* p = (const float*)src->at(v);
          double val[3] = {0.0, 0.0, 0.0};
          for (unsigned k = 0; k < 3; k++)
            if (3 * gi + k < args->valueCount && 3 * gi + k < floatsPerVertex) val[k] = p[3 * gi + k];
          values[v] = D3(val[0], val[1], val[2]);
        }
        InterpChannels ch;
        std::vector<FaceMap> faceMap;
        bool first = true;
        const RTCSubdivisionMode mode = build_base_level(geom, ch.lvl, faceMap, first, &values, topo);
        refine_to_interp_level(ch.lvl, mode, first, &ch.limit, nullptr);
        c.push_back(std::move(ch));
      }
    }
    chans = &c;
  }

  if (args->primID >= cache->primToFace.size() || cache->primToFace[args->primID] < 0)
    RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "rtcInterpolate: invalid primID");
  // base face and local parameters (ls, lt) in [0,1]^2; d(ls,lt)/d(u,v) = [[jsu, jsv], [jtu, jtv]]
  size_t f = (size_t)cache->primToFace[args->primID];
  double ls = args->u, lt = args->v, jsu = 1.0, jsv = 0.0, jtu = 0.0, jtv = 1.0;
  if (cache->mixed) {
    if (cache->primCorners[args->primID] != 4) { // PatchEval::eval_general, patch_eval.h:71-77
      const double hu = 0.5 * args->u, hv = 0.5 * args->v;
      const unsigned l = (unsigned)std::max(0.0, floor(hu)), h = (unsigned)std::max(0.0, floor(hv));
      const unsigned sub = 4 * h + l;
      if (sub >= cache->primCorners[args->primID]) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "rtcInterpolate: uv does not address a sub-patch of this face");
      f += sub;
      ls = std::min(1.0, std::max(0.0, 2.0 * (hu - floor(hu)) - 0.5));
      lt = std::min(1.0, std::max(0.0, 2.0 * (hv - floor(hv)) - 0.5));
    } else { // quad face held as four corner sub-quads (see patch_vertex_ids)
      const double u = args->u, v = args->v;
      const bool lowU = u <= 0.5, lowV = v <= 0.5;
      if (lowU && lowV) { ls = 2 * u; lt = 2 * v; jsu = 2; jtv = 2; }
      else if (!lowU && lowV) { f += 1; ls = 2 * v; lt = 2 * (1 - u); jsu = 0; jsv = 2; jtu = -2; jtv = 0; }
      else if (!lowU && !lowV) { f += 2; ls = 2 * (1 - u); lt = 2 * (1 - v); jsu = -2; jtv = -2; }
      else { f += 3; ls = 2 * (1 - v); lt = 2 * u; jsu = 0; jsv = -2; jtu = 2; jtv = 0; }
    }
  }
  const unsigned n = cache->n, w = n + 1;
  const std::vector<uint32_t>& g = cache->grid[f];
  const double x = ls * n, y = lt * n;
  const unsigned i = (unsigned)std::min<double>(std::max(0.0, floor(x)), n - 1), j = (unsigned)std::min<double>(std::max(0.0, floor(y)), n - 1);
  const double s = x - i, t = y - j;
  const uint32_t q = (uint32_t)(f * n * n + (size_t)j * n + i);
  const uint32_t c00 = g[j * w + i], c10 = g[j * w + i + 1], c11 = g[(j + 1) * w + i + 1], c01 = g[(j + 1) * w + i];

  auto quad_verts = [&](uint32_t qq, uint32_t out[4]) {
    const size_t ff = qq / (n * n), r = qq % (n * n), jj = r / n, ii = r % n;
    const std::vector<uint32_t>& gg = cache->grid[ff];
    out[0] = gg[jj * w + ii]; out[1] = gg[jj * w + ii + 1]; out[2] = gg[(jj + 1) * w + ii + 1]; out[3] = gg[(jj + 1) * w + ii];
  };
  // the sub-face other than `self` that contains both a and b; oa / ob = its vertices next to a / b
  auto across = [&](uint32_t self, uint32_t a, uint32_t b, uint32_t& oa, uint32_t& ob) -> int64_t {
    for (uint32_t k = cache->vqStart[a]; k < cache->vqStart[a + 1]; k++) {
      const uint32_t qq = cache->vqList[k];
      if (qq == self) continue;
      uint32_t vv[4];
      quad_verts(qq, vv);
      int ia = -1, ib = -1;
      for (int m = 0; m < 4; m++) { if (vv[m] == a) ia = m; if (vv[m] == b) ib = m; }
      if (ia < 0 || ib < 0) continue;
      if ((ia + 1) % 4 == ib) { oa = vv[(ia + 3) % 4]; ob = vv[(ib + 1) % 4]; }
      else if ((ib + 1) % 4 == ia) { oa = vv[(ia + 1) % 4]; ob = vv[(ib + 3) % 4]; }
      else continue;
      return qq;
    }
    return -1;
  };
  // the fourth sub-face around the regular vertex a (not self, not n1, not n2): its vertex opposite a
  auto diagonal = [&](uint32_t a, uint32_t self, int64_t n1, int64_t n2, uint32_t& od) -> bool {
    for (uint32_t k = cache->vqStart[a]; k < cache->vqStart[a + 1]; k++) {
      const uint32_t qq = cache->vqList[k];
      if (qq == self || (int64_t)qq == n1 || (int64_t)qq == n2) continue;
      uint32_t vv[4];
      quad_verts(qq, vv);
      for (int m = 0; m < 4; m++)
        if (vv[m] == a) { od = vv[(m + 2) % 4]; return true; }
    }
    return false;
  };

  uint32_t C[4][4];
  bool regular = !cache->irregular[c00] && !cache->irregular[c10] && !cache->irregular[c11] && !cache->irregular[c01];
  if (regular) {
    C[1][1] = c00; C[1][2] = c10; C[2][2] = c11; C[2][1] = c01;
    const int64_t qb = across(q, c00, c10, C[0][1], C[0][2]);
    const int64_t qr = across(q, c10, c11, C[1][3], C[2][3]);
    const int64_t qt = across(q, c11, c01, C[3][2], C[3][1]);
    const int64_t ql = across(q, c01, c00, C[2][0], C[1][0]);
    regular = qb >= 0 && qr >= 0 && qt >= 0 && ql >= 0 && diagonal(c00, q, qb, ql, C[0][0]) && diagonal(c10, q, qb, qr, C[0][3]) &&
              diagonal(c11, q, qr, qt, C[3][3]) && diagonal(c01, q, qt, ql, C[3][0]);
  }
  // irregular cell: its one-ring (the cell first), for the feature-adaptive evaluation
  std::vector<MiniCell> ring;
  if (!regular) {
    const uint32_t own[4] = {c00, c10, c11, c01};
    MiniCell self;
    for (int k = 0; k < 4; k++) self.v[k] = own[k];
    ring.push_back(self);
    std::vector<uint32_t> seen(1, q);
    for (int k = 0; k < 4; k++)
      for (uint32_t e = cache->vqStart[own[k]]; e < cache->vqStart[own[k] + 1]; e++) {
        const uint32_t qq = cache->vqList[e];
        if (std::find(seen.begin(), seen.end(), qq) != seen.end()) continue;
        seen.push_back(qq);
        MiniCell c;
        quad_verts(qq, c.v);
        ring.push_back(c);
      }
  }
  double sc = (double)n;
  for (unsigned gi = 0; gi < groups; gi++) {
    const InterpChannels& ch = (*chans)[gi];
    PatchValue val;
    if (regular) val = eval_bspline16(ch.lvl.P, C, s, t);
    else {
      unsigned depth = 0;
      val = adaptive_eval(ch.lvl, cache->mode, ring, s, t, INTERP_MAX_DEPTH > INTERP_LEVELS ? INTERP_MAX_DEPTH - INTERP_LEVELS : 0u, depth);
      sc = (double)n * (double)(1u << depth);
    }
    const D3 &P = val.P, &Pu = val.Pu, &Pv = val.Pv, &Puu = val.Puu, &Pvv = val.Pvv, &Puv = val.Puv;
    // derivatives w.r.t. the base face's (ls, lt), then the chain rule to the caller's (u, v) (the map is affine)
    const D3 Ds = Pu * sc, Dt = Pv * sc, Dss = Puu * (sc * sc), Dtt = Pvv * (sc * sc), Dst = Puv * (sc * sc);
    const D3 Du = Ds * jsu + Dt * jtu, Dv = Ds * jsv + Dt * jtv;
    const D3 Duu = Dss * (jsu * jsu) + Dst * (2.0 * jsu * jtu) + Dtt * (jtu * jtu);
    const D3 Dvv = Dss * (jsv * jsv) + Dst * (2.0 * jsv * jtv) + Dtt * (jtv * jtv);
    const D3 Duv = Dss * (jsu * jsv) + Dst * (jsu * jtv + jsv * jtu) + Dtt * (jtu * jtv);
    const double vP[3] = {P.x, P.y, P.z}, vu[3] = {Du.x, Du.y, Du.z}, vv[3] = {Dv.x, Dv.y, Dv.z};
    const double vuu[3] = {Duu.x, Duu.y, Duu.z}, vvv[3] = {Dvv.x, Dvv.y, Dvv.z}, vuv[3] = {Duv.x, Duv.y, Duv.z};
    for (unsigned k = 0; k < 3 && 3 * gi + k < args->valueCount; k++) {
      const unsigned o = 3 * gi + k;
      if (args->P) args->P[o] = (float)vP[k];
      if (args->dPdu) { args->dPdu[o] = (float)vu[k]; args->dPdv[o] = (float)vv[k]; }
      if (args->ddPdudu) { args->ddPdudu[o] = (float)vuu[k]; args->ddPdvdv[o] = (float)vvv[k]; args->ddPdudv[o] = (float)vuv[k]; }
    }
  }
}

} // namespace rtamd
