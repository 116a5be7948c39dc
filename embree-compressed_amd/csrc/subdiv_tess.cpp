#include "subdiv_tess.h"

#include <algorithm>
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
};

struct BorderEdge
{
  uint32_t idx; // running index among the unique face-border edges of this level
  uint32_t count;
  uint32_t a, b;
  D3 sumFp;
};

struct Level
{
  unsigned n = 1;                           // cells per face side
  std::vector<D3> P;                        // vertex positions
  std::vector<std::vector<uint32_t>> grid;  // per face: (n+1)^2 vertex ids, row-major [j][i]
  std::vector<uint8_t> pinned;              // per vertex: position is fixed under refinement
  std::vector<uint8_t> bpin;                // per vertex: lies on a pinned (linear) boundary
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
      const BorderEdge& e = kv.second;
      link(e.a, e.b);
      if (e.count != 2) { // mesh boundary (count 1) or non-manifold edge
        acc[e.a].sumB += in.P[e.b]; acc[e.a].nb++;
        acc[e.b].sumB += in.P[e.a]; acc[e.b].nb++;
      }
    }
  }

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
    if (a.nb == 0) { // smooth interior vertex of valence n
      const double n = (double)a.ne;
      return a.sumF * (1.0 / (n * n)) + a.sumN * (1.0 / (n * n)) + p * ((n - 2.0) / n);
    }
    return p * 0.75 + a.sumB * 0.125; // smooth boundary: cubic B-spline curve rule
  }

  D3 limit_point(uint32_t v, const std::vector<uint8_t>& pinned) const
  {
    const Acc& a = acc[v];
    const D3& p = in.P[v];
    if (pinned[v] || a.ne == 0) return p;
    if (a.nb == 0) {
      const double n = (double)a.ne;
      const D3 sumD = a.sumF * 4.0 - p * n - a.sumN * 2.0; // diagonal neighbours, from F = (V+N_i+N_i+1+D_i)/4
      return (p * (n * n) + a.sumN * 4.0 + sumD) * (1.0 / (n * (n + 5.0)));
    }
    return (p * 4.0 + a.sumB) * (1.0 / 6.0);
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
    for (auto& kv : border) {
      const BorderEdge& e = kv.second;
      const size_t id = nV + e.idx;
      if (e.count == 2 && !pinAll) out.P[id] = (in.P[e.a] + in.P[e.b] + e.sumFp) * 0.25;
      else {
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
    if (a.nb == 0 && a.ne == nq && nq >= 3) {
      // limit tangents of an interior vertex of valence n (Halstead et al. 1993): cos / sin combinations of the ring
      const double n = (double)nq, two_pi_n = 2.0 * M_PI / n;
      const double An = 1.0 + cos(two_pi_n) + cos(M_PI / n) * sqrt(2.0 * (9.0 + cos(two_pi_n)));
      for (uint32_t i = 0; i < nq; i++) {
        const Corner& c = corners[order[i]];
        const double c0 = cos(two_pi_n * i), c1 = cos(two_pi_n * (i + 1)), s0 = sin(two_pi_n * i), s1 = sin(two_pi_n * (i + 1));
        ta += lv.P[c.next] * (An * c0) + lv.P[c.diag] * (c0 + c1);
        tb += lv.P[c.next] * (An * s0) + lv.P[c.diag] * (s0 + s1);
      }
    } else if (a.nb == 2 && nq == 2) {
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

void tessellate_subdiv(const Geometry* geom, unsigned geomID, unsigned L, std::vector<PatchGrid>& out)
{
  const BufferView* vb = geom->view(RTC_BUFFER_TYPE_VERTEX, 0);
  const BufferView* ib = geom->view(RTC_BUFFER_TYPE_INDEX, 0);
  const BufferView* fb = geom->view(RTC_BUFFER_TYPE_FACE, 0);
  if (!vb || !vb->valid() || !ib || !ib->valid() || !fb || !fb->valid())
    RT_THROW(RTC_ERROR_INVALID_OPERATION, "subdivision geometry needs vertex, index and face buffers");
  for (RTCBufferType t : {RTC_BUFFER_TYPE_EDGE_CREASE_INDEX, RTC_BUFFER_TYPE_VERTEX_CREASE_INDEX}) {
    const BufferView* c = geom->view(t, 0);
    if (c && c->valid() && c->count) RT_THROW(RTC_ERROR_INVALID_OPERATION, "crease buffers are not supported by the MI355X tessellator yet");
  }
  if (L > 10) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "subdivision level too high");
  const RTCSubdivisionMode mode = geom->subdivMode.empty() ? RTC_SUBDIVISION_MODE_SMOOTH_BOUNDARY : geom->subdivMode[0];

  Level cur;
  cur.n = 1;
  cur.P.resize(vb->count);
  for (size_t i = 0; i < vb->count; i++) {
    const float* p = (const float*)vb->at(i);
    cur.P[i] = D3(p[0], p[1], p[2]);
  }
  cur.pinned.assign(cur.P.size(), 0);
  cur.bpin.assign(cur.P.size(), 0);

  std::vector<uint8_t> hole(fb->count, 0);
  if (const BufferView* hb = geom->view(RTC_BUFFER_TYPE_HOLE, 0))
    if (hb->valid())
      for (size_t i = 0; i < hb->count; i++) {
        const unsigned f = *(const unsigned*)hb->at(i);
        if (f < hole.size()) hole[f] = 1;
      }

  // faces: quads only; a face is valid if its indices are in range and its vertices finite (SubdivMesh::valid)
  std::vector<unsigned> facePrim;
  size_t cursor = 0;
  for (size_t f = 0; f < fb->count; f++) {
    const unsigned nv = *(const unsigned*)fb->at(f);
    if (cursor + nv > ib->count) RT_THROW(RTC_ERROR_INVALID_OPERATION, "face buffer overruns the index buffer");
    if (nv != 4) RT_THROW(RTC_ERROR_INVALID_OPERATION, "only quad faces are supported by the MI355X tessellator yet");
    uint32_t q[4];
    bool ok = !hole[f];
    for (unsigned k = 0; k < 4; k++) {
      q[k] = *(const unsigned*)ib->at(cursor + k);
      if (q[k] >= cur.P.size()) ok = false;
      else {
        const D3& p = cur.P[q[k]];
        if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) ok = false;
      }
    }
    cursor += nv;
    if (!ok) continue;
    cur.grid.push_back({q[0], q[1], q[3], q[2]}); // row-major 2x2: (0,0)=v0 (1,0)=v1 (0,1)=v3 (1,1)=v2
    facePrim.push_back((unsigned)f);
  }
  if (cur.grid.empty()) return;

  bool first = true;
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

  // limit positions are shared between faces: evaluate each vertex once
  std::vector<D3> limit(cur.P.size());
  for (uint32_t v = 0; v < cur.P.size(); v++) limit[v] = fin.limit_point(v, pinned);

  const size_t N = (size_t)w * w;
  const bool noBoundary = mode == RTC_SUBDIVISION_MODE_NO_BOUNDARY;
  out.reserve(out.size() + cur.grid.size());
  std::vector<float> gu(N), gv(N), nx, ny, nz;
  for (unsigned j = 0; j <= n; j++)
    for (unsigned i = 0; i <= n; i++) {
      gu[(size_t)j * w + i] = (float)i / (float)n; // gridUVTessellator: (x0+i) * rcp(n), exact for n = 2^L
      gv[(size_t)j * w + i] = (float)j / (float)n;
    }
  for (size_t f = 0; f < cur.grid.size(); f++) {
    const std::vector<uint32_t>& g = cur.grid[f];
    if (noBoundary) { // RTC_SUBDIVISION_MODE_NO_BOUNDARY: patches touching the border are not rendered
      bool touches = false;
      for (size_t k = 0; k < N && !touches; k++) touches = fin.acc[g[k]].nb != 0;
      if (touches) continue;
    }
    out.emplace_back();
    PatchGrid& pg = out.back();
    pg.geomID = geomID;
    pg.primID = facePrim[f];
    pg.n = n;
    pg.x.resize(N); pg.y.resize(N); pg.z.resize(N);
    for (size_t k = 0; k < N; k++) {
      const D3& p = limit[g[k]];
      pg.x[k] = (float)p.x; pg.y[k] = (float)p.y; pg.z[k] = (float)p.z;
    }
    if (displ) {
      pg.bx = pg.x; pg.by = pg.y; pg.bz = pg.z;
      nx.resize(N); ny.resize(N); nz.resize(N);
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

} // namespace rtamd
