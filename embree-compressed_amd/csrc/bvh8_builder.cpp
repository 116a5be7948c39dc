#include "bvh8_builder.h"

#include <algorithm>
#include <thread>

namespace rtamd {

namespace {

struct BinNode
{
  Box3 box;
  size_t begin, end;
  int left = -1, right = -1;
  bool leaf() const { return left < 0; }
};

struct Binner
{
  static const int NBINS = 16;
  std::vector<BuildPrim>& prims;
  const BuildSettings& cfg;
  std::vector<BinNode> tree;

  int forks = 0; // levels of the recursion that may still fork a thread

  Binner(std::vector<BuildPrim>& p, const BuildSettings& c, size_t expect) : prims(p), cfg(c) { tree.reserve(expect / 2 + 16); }

  // append another builder's subtree (indices are relative to its own vector); returns the new index of its root
  int splice(const Binner& sub, int subRoot)
  {
    const int off = (int)tree.size();
    for (const BinNode& n : sub.tree) {
      tree.push_back(n);
      if (n.left >= 0) { tree.back().left += off; tree.back().right += off; }
    }
    return subRoot + off;
  }

  float blocks(size_t n) const { return float((n + cfg.blockSize - 1) / cfg.blockSize); }

  // Build the binary subtree over prims[begin,end); returns its index in `tree`.
  int build(size_t begin, size_t end)
  {
    Box3 box, cbox;
    for (size_t i = begin; i < end; i++) {
      box.extend(prims[i].box);
      V3 c = prims[i].box.center2();
      cbox.extend(c);
    }
    int me = (int)tree.size();
    tree.push_back(BinNode());
    tree[me].box = box;
    tree[me].begin = begin;
    tree[me].end = end;
    size_t n = end - begin;
    if (n <= cfg.minLeaf) return me;

    // binned SAH over the three centroid axes
    float bestCost = std::numeric_limits<float>::infinity();
    int bestAxis = -1, bestSplit = -1;
    V3 cext = cbox.size();
    for (int axis = 0; axis < 3; axis++) {
      if (!(cext[axis] > 0.f)) continue;
      Box3 bb[NBINS];
      size_t cnt[NBINS] = {0};
      float scale = float(NBINS) * (1.f - 1e-6f) / cext[axis];
      for (size_t i = begin; i < end; i++) {
        float c = prims[i].box.center2()[axis];
        int b = (int)((c - cbox.lo[axis]) * scale);
        b = std::min(std::max(b, 0), NBINS - 1);
        bb[b].extend(prims[i].box);
        cnt[b]++;
      }
      // sweep from the right, then from the left
      float rArea[NBINS];
      size_t rCnt[NBINS];
      Box3 acc;
      size_t c = 0;
      for (int b = NBINS - 1; b >= 1; b--) {
        acc.extend(bb[b]);
        c += cnt[b];
        rArea[b] = acc.half_area();
        rCnt[b] = c;
      }
      acc = Box3();
      c = 0;
      for (int b = 1; b < NBINS; b++) {
        acc.extend(bb[b - 1]);
        c += cnt[b - 1];
        if (c == 0 || rCnt[b] == 0) continue;
        float cost = acc.half_area() * blocks(c) + rArea[b] * blocks(rCnt[b]);
        if (cost < bestCost) {
          bestCost = cost;
          bestAxis = axis;
          bestSplit = b;
        }
      }
    }

    size_t mid;
    if (bestAxis < 0) {
      // all centroids coincide: leaf if allowed, otherwise split by index
      if (n <= cfg.maxLeaf) return me;
      mid = begin + n / 2;
    } else {
      float leafCost = cfg.intCost * box.half_area() * blocks(n);
      float splitCost = cfg.travCost * box.half_area() + cfg.intCost * bestCost;
      if (n <= cfg.maxLeaf && splitCost >= leafCost) return me;
      float scale = float(NBINS) * (1.f - 1e-6f) / cext[bestAxis];
      float lo = cbox.lo[bestAxis];
      auto it = std::partition(prims.begin() + begin, prims.begin() + end, [&](const BuildPrim& p) {
        int b = (int)((p.box.center2()[bestAxis] - lo) * scale);
        b = std::min(std::max(b, 0), NBINS - 1);
        return b < bestSplit;
      });
      mid = (size_t)(it - prims.begin());
      if (mid == begin || mid == end) mid = begin + n / 2; // numerical corner case
    }
    int l, r;
    if (forks > 0 && n > 65536) {
      // the two halves are disjoint ranges of `prims`: build them concurrently in builders of their own, then splice
      // (same splits as the sequential build, hence the same tree)
      Binner L(prims, cfg, mid - begin), R(prims, cfg, end - mid);
      L.forks = R.forks = forks - 1;
      int lroot = -1, rroot = -1;
      std::thread t([&]() { lroot = L.build(begin, mid); });
      rroot = R.build(mid, end);
      t.join();
      l = splice(L, lroot);
      r = splice(R, rroot);
    } else {
      l = build(begin, mid);
      r = build(mid, end);
    }
    tree[me].left = l;
    tree[me].right = r;
    return me;
  }
};

struct Collapser
{
  const std::vector<BinNode>& tree;
  const std::vector<BuildPrim>& prims;
  const MakeLeafFn& makeLeaf;
  BuildResult& out;

  uint32_t emit(int bn, uint32_t depth)
  {
    const BinNode& b = tree[bn];
    if (b.leaf()) {
      out.leafCount++;
      out.maxDepth = std::max(out.maxDepth, depth);
      return makeLeaf(prims.data(), b.begin, b.end);
    }
    // open the child with the largest surface area until 8 children or only leaves remain
    int kids[8];
    int nk = 0;
    kids[nk++] = b.left;
    kids[nk++] = b.right;
    while (nk < 8) {
      int best = -1;
      float bestArea = -1.f;
      for (int i = 0; i < nk; i++) {
        if (tree[kids[i]].leaf()) continue;
        float a = tree[kids[i]].box.half_area();
        if (a > bestArea) {
          bestArea = a;
          best = i;
        }
      }
      if (best < 0) break;
      int open = kids[best];
      kids[best] = tree[open].left;
      kids[nk++] = tree[open].right;
    }
    uint32_t me = (uint32_t)out.nodes.size();
    out.nodes.push_back(QNode8());
    Box3 boxes[8];
    uint32_t refs[8];
    for (int i = 0; i < nk; i++) {
      boxes[i] = tree[kids[i]].box;
      refs[i] = emit(kids[i], depth + 1);
    }
    QNode8 q;
    quantize_node(boxes, refs, nk, q);
    out.nodes[me] = q;
    return me;
  }
};

inline float scale_from_exp(uint8_t e)
{
  uint32_t bits = uint32_t(e) << 23;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

} // namespace

Box3 dequantize_child(const QNode8& n, int i)
{
  Box3 b;
  for (int a = 0; a < 3; a++) {
    float s = scale_from_exp(n.exp[a]);
    b.lo[a] = fmaf(float(n.q[2 * a + 0][i]), s, n.origin[a]);
    b.hi[a] = fmaf(float(n.q[2 * a + 1][i]), s, n.origin[a]);
  }
  return b;
}

void quantize_node(const Box3* boxes, const uint32_t* refs, int n, QNode8& out)
{
  memset(&out, 0, sizeof(out));
  Box3 all;
  for (int i = 0; i < n; i++) all.extend(boxes[i]);
  for (int i = 0; i < 8; i++) out.child[i] = i < n ? refs[i] : REF_EMPTY;
  for (int a = 0; a < 3; a++) {
    const float o = all.lo[a];
    out.origin[a] = o;
    const double ext = (double)all.hi[a] - (double)o;
    int e = 0; // exponent byte; 0 encodes scale 0.0 for a flat axis
    if (ext > 0.0) {
      int ex;
      frexp(ext / 255.0, &ex); // ext/255 = m * 2^ex, m in [0.5,1)  ->  2^ex >= ext/255
      e = std::min(std::max(ex + 127, 1), 254);
    }
    for (;;) {
      const float s = scale_from_exp((uint8_t)e);
      bool ok = true;
      for (int i = 0; i < n && ok; i++) {
        int qlo = 0, qhi = 0;
        if (e != 0) {
          qlo = (int)floor(((double)boxes[i].lo[a] - (double)o) / (double)s);
          qhi = (int)ceil(((double)boxes[i].hi[a] - (double)o) / (double)s);
          qlo = std::min(std::max(qlo, 0), 255);
          qhi = std::max(qhi, 0);
          // enforce conservativeness under the exact fp32 decode the kernels perform
          while (qlo > 0 && fmaf(float(qlo), s, o) > boxes[i].lo[a]) qlo--;
          while (qhi <= 255 && fmaf(float(std::min(qhi, 255)), s, o) < boxes[i].hi[a]) qhi++;
          if (qhi > 255) {
            ok = false;
            break;
          }
        }
        out.q[2 * a + 0][i] = (uint8_t)qlo;
        out.q[2 * a + 1][i] = (uint8_t)qhi;
      }
      if (ok) break;
      if (e >= 254) RT_THROW(RTC_ERROR_UNKNOWN, "bvh8 quantizer: extent not representable");
      e++;
    }
    out.exp[a] = (uint8_t)e;
    for (int i = n; i < 8; i++) { // inverted box for empty slots
      out.q[2 * a + 0][i] = 255;
      out.q[2 * a + 1][i] = 0;
    }
  }
}

BuildResult build_bvh8(std::vector<BuildPrim>& prims, const BuildSettings& settings, const MakeLeafFn& makeLeaf)
{
  BuildResult out;
  if (prims.empty()) return out;
  Binner binner(prims, settings, prims.size());
  for (unsigned t = settings.threads; t > 1; t >>= 1) binner.forks++;
  int root = binner.build(0, prims.size());
  Collapser c{binner.tree, prims, makeLeaf, out};
  out.nodes.reserve(prims.size() / 8 + 8);
  out.root = c.emit(root, 0);
  return out;
}

} // namespace rtamd
