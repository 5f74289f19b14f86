// bvh8_build.hpp — the lean traversal kernels' own acceleration structure: an 8-wide BVH with child boxes on an
// 11-bit grid (host build).
//
// The reference walks one binary 20-bin SAH tree per mesh (core/bvh.hpp:21-33, 140-184, 273-347) in a fixed order
// (cpu/ray-integrator.cpp:84-160), and two things depend on that order: which of several equal-t triangles wins,
// and the sampler dimensions its stochastic alpha tests draw (:207-211). For every other ray the closest hit /
// occlusion is a property of the triangle SET, so the lean kernels may use any conservative structure as long as
// the triangle test (:163-229) stays the reference's and the rays that could depend on the order are detected and
// handed to the general kernels, which walk the reference's tree (trace_lean_wide.inc lists the hand-over rules).
//
// This file builds that structure, per mesh and per triangle subset:
//   1. a binary tree over the reference's own padded triangle boxes (bounds.hpp:89-104), 32-bin SAH down to one
//      triangle per leaf — independent of the reference's tree;
//   2. the SAH-optimal collapse into nodes of up to 8 children and leaves of up to 3 triangles (dynamic program
//      over "subtree as a forest of at most i roots", Ylitie, Karras, Laine: Efficient Incoherent Ray Traversal
//      on GPUs Through Compressed Wide BVHs, HPG 2017, section 3.1);
//   3. children assigned to the 8 slots so that (slot XOR ray octant) ascending is an approximate front-to-back
//      order (section 3.2): the traversal needs no distance sort;
//   4. child boxes quantised CONSERVATIVELY onto a per-node grid of 2048 steps per axis (origin p, step 2^e),
//      stored as half-precision integers so that a plane's slab distance is one v_fma_mix_f32.
//
// Node = 128 bytes = one cache line: header 32 B, then 12 bytes per child slot (per axis one word: low plane |
// high plane << 16). The cooperative walk (trace_lean_coop.inc) puts one lane on each child: the eight lanes of a
// ray read the header (one address) and their own 12 bytes — the whole line, once — and a rotate by 0 or 16 bits
// puts the ray's near plane of an axis into the low half of the word (round 4's layout was per axis and per
// side, for one lane reading all eight children).
//
// Conservativeness. A child's grid box contains the child's exact box (union of the reference's padded triangle
// boxes) grown by `pad` on every side; `pad` covers the difference between the kernel's slab arithmetic
// (fma(q, 2^e * idir, (p - o) * idir)) and the reference's (b * idir + odir, unfused) for every ray whose
// object-space origin has |o|_inf <= roBound (error bound): whenever the reference's test of a box
// inside the child passes, the kernel's test of the child passes.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "bvh_build.hpp"

namespace yart_hip {

struct Wide8Node {             // 128 bytes
  float p[3];                  // grid origin
  uint8_t e[3];                // grid step = 2^(e[0] - 127) (the float's biased exponent), one for the three axes: e[1] = e[2] = e[0]
  uint8_t imask;               // slots holding an inner node
  uint32_t childBase;          // node index of the first inner child (inner children are consecutive, in slot order)
  uint32_t triBase;            // record index of the first triangle of the leaf children
  uint32_t triValid;           // bit 3 * slot + j: triangle j of the leaf in `slot`; record = triBase + popcount(bits below)
  uint32_t pad;
  uint32_t child[8][3];        // [slot][axis]: low plane | high plane << 16, grid coordinates 0..2047 as IEEE halves
  void setPlanes(int slot, int axis, uint32_t lo, uint32_t hi);
};
static_assert(sizeof(Wide8Node) == 128, "Wide8Node is one cache line");

constexpr uint32_t kWideGrid = 2047;       // largest grid coordinate (integers up to 2048 are exact in half precision)
constexpr uint32_t kWideLeafMax = 3;
constexpr uint32_t kWideStackDepth = 64;   // what the walks hold per ray (= traverse.hpp::kRefStackDepth, the reference's own depth, ray-integrator.cpp:92-93)

inline uint16_t halfOfInt(uint32_t q) {    // exact for q <= 2048
  if (q == 0u) return 0u;
  int m = 31 - __builtin_clz(q);
  const uint32_t mant = (q << (10 - m)) & 0x3ffu;
  return uint16_t(((15 + m) << 10) | mant);
}

inline void Wide8Node::setPlanes(int slot, int axis, uint32_t lo, uint32_t hi) { child[slot][axis] = uint32_t(halfOfInt(lo)) | (uint32_t(halfOfInt(hi)) << 16); }

class Bvh8Builder {
 public:
  // boxes / centroids: per triangle of the mesh (the reference's padded boxes); ids: the subset to build over.
  // Returns false (nothing built) if a box is not finite.
  bool build(const std::vector<Bounds3>& boxes, const std::vector<float>& centroids, const std::vector<uint32_t>& ids, double pad) {
    nodes.clear(); order.clear(); bn_.clear();
    boxes_ = &boxes; cent_ = &centroids; pad_ = pad;
    idx_ = ids;
    if (idx_.empty()) return false;
    for (uint32_t t : idx_)
      for (int a = 0; a < 3; a++)
        if (!std::isfinite(boxes[t].mn[a]) || !std::isfinite(boxes[t].mx[a]) || !std::isfinite(centroids[size_t(t) * 3 + a])) return false;
    buildBinary();
    collapse();
    return true;
  }
  std::vector<Wide8Node> nodes;      // node 0 = root
  std::vector<uint32_t> order;       // triangle ids in record order (Wide8Node::triBase indexes this)
  // statistics of the last build
  double sahCost = 0.0;
  // The walk (trace_lean_wide.inc, traverse_wide.hpp) pushes one stack entry per node with more than one pending inner hit: the
  // deepest stack a ray can need is the largest number of nodes with two or more inner children on a root-to-leaf path.
  uint32_t maxStack = 0;

 private:
  struct BNode { Bounds3 b; uint32_t left = 0, right = 0, first = 0, count = 0; };   // count > 0: leaf
  const std::vector<Bounds3>* boxes_ = nullptr;
  const std::vector<float>* cent_ = nullptr;
  double pad_ = 0.0;
  std::vector<uint32_t> idx_;
  std::vector<BNode> bn_;

  static double halfArea(const Bounds3& b) {
    const double x = double(b.mx[0]) - b.mn[0], y = double(b.mx[1]) - b.mn[1], z = double(b.mx[2]) - b.mn[2];
    return x * y + y * z + z * x;
  }
  Bounds3 rangeBounds(uint32_t first, uint32_t count) const {
    Bounds3 b;
    for (uint32_t k = 0; k < count; k++) {
      const Bounds3& t = (*boxes_)[idx_[first + k]];
      for (int a = 0; a < 3; a++) { b.mn[a] = std::min(b.mn[a], t.mn[a]); b.mx[a] = std::max(b.mx[a], t.mx[a]); }
    }
    return b;
  }

  // ---- 1. binary tree: 32-bin SAH, one triangle per leaf
  void buildBinary() {
    constexpr int kBins = 32;
    bn_.reserve(idx_.size() * 2);
    BNode root; root.first = 0; root.count = uint32_t(idx_.size()); root.b = rangeBounds(0, root.count);
    bn_.push_back(root);
    std::vector<uint32_t> todo{0};
    while (!todo.empty()) {
      const uint32_t ni = todo.back(); todo.pop_back();
      const uint32_t first = bn_[ni].first, count = bn_[ni].count;
      if (count <= 1) continue;
      // centroid bounds
      float cmn[3] = {kInf, kInf, kInf}, cmx[3] = {-kInf, -kInf, -kInf};
      for (uint32_t k = 0; k < count; k++)
        for (int a = 0; a < 3; a++) {
          const float c = (*cent_)[size_t(idx_[first + k]) * 3 + a];
          cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c);
        }
      double bestCost = std::numeric_limits<double>::infinity();
      int bestAxis = -1; int bestBin = 0;
      for (int a = 0; a < 3; a++) {
        const double ext = double(cmx[a]) - cmn[a];
        if (!(ext > 0.0)) continue;
        Bounds3 bb[kBins]; uint32_t bc[kBins] = {0};
        const double scale = kBins / ext;
        for (uint32_t k = 0; k < count; k++) {
          const uint32_t t = idx_[first + k];
          int bi = int((double((*cent_)[size_t(t) * 3 + a]) - cmn[a]) * scale);
          bi = bi < 0 ? 0 : bi >= kBins ? kBins - 1 : bi;
          bc[bi]++;
          const Bounds3& tb = (*boxes_)[t];
          for (int c = 0; c < 3; c++) { bb[bi].mn[c] = std::min(bb[bi].mn[c], tb.mn[c]); bb[bi].mx[c] = std::max(bb[bi].mx[c], tb.mx[c]); }
        }
        double rightArea[kBins]; uint32_t rightCount[kBins];
        Bounds3 acc; uint32_t n = 0;
        for (int i = kBins - 1; i > 0; i--) {
          if (bc[i]) for (int c = 0; c < 3; c++) { acc.mn[c] = std::min(acc.mn[c], bb[i].mn[c]); acc.mx[c] = std::max(acc.mx[c], bb[i].mx[c]); }
          n += bc[i];
          rightArea[i] = n ? halfArea(acc) : 0.0; rightCount[i] = n;
        }
        Bounds3 accL; uint32_t nl = 0;
        for (int i = 0; i < kBins - 1; i++) {
          if (bc[i]) for (int c = 0; c < 3; c++) { accL.mn[c] = std::min(accL.mn[c], bb[i].mn[c]); accL.mx[c] = std::max(accL.mx[c], bb[i].mx[c]); }
          nl += bc[i];
          if (nl == 0 || rightCount[i + 1] == 0) continue;
          const double cost = halfArea(accL) * nl + rightArea[i + 1] * rightCount[i + 1];
          if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = i; }
        }
      }
      uint32_t mid;
      if (bestAxis < 0) {
        mid = first + count / 2;                         // coincident centroids: split by index
      } else {
        const double ext = double(cmx[bestAxis]) - cmn[bestAxis], scale = kBins / ext;
        uint32_t i = first, j = first + count;
        while (i < j) {
          int bi = int((double((*cent_)[size_t(idx_[i]) * 3 + bestAxis]) - cmn[bestAxis]) * scale);
          bi = bi < 0 ? 0 : bi >= kBins ? kBins - 1 : bi;
          if (bi <= bestBin) i++; else std::swap(idx_[i], idx_[--j]);
        }
        mid = i;
        if (mid == first || mid == first + count) mid = first + count / 2;
      }
      BNode l, r;
      l.first = first; l.count = mid - first; l.b = rangeBounds(l.first, l.count);
      r.first = mid; r.count = first + count - mid; r.b = rangeBounds(r.first, r.count);
      const uint32_t li = uint32_t(bn_.size());
      bn_.push_back(l); bn_.push_back(r);
      bn_[ni].left = li; bn_[ni].right = li + 1; bn_[ni].count = 0;
      todo.push_back(li); todo.push_back(li + 1);
    }
  }

  // ---- 2. collapse (dynamic program)
  static constexpr double kCostNode = 1.0, kCostPrim = 0.3;
  enum : uint8_t { D_LEAF = 0, D_INTERNAL = 1, D_FALL = 2, D_DIST = 3 };   // D_DIST + k: distribute, k roots to the left child
  std::vector<float> cost_;          // [node][i - 1], i = 1..7
  std::vector<uint8_t> dec_;
  std::vector<uint32_t> prims_;
  std::vector<uint8_t> dist8_;       // best k of the distribution into 8 (C_internal)

  uint32_t primCount(uint32_t n) const { return prims_[n]; }

  void collapse() {
    const size_t N = bn_.size();
    cost_.assign(N * 7, 0.0f); dec_.assign(N * 7, D_LEAF); prims_.assign(N, 0); dist8_.assign(N, 0);
    // children have larger indices than their parent: bottom-up = reverse index order
    for (size_t n = N; n-- > 0;) {
      const BNode& b = bn_[n];
      const double A = halfArea(b.b);
      if (b.count > 0) {                                 // binary leaf (one triangle; more only for coincident input that could not be split)
        prims_[n] = b.count;
        for (int i = 0; i < 7; i++) { cost_[n * 7 + i] = float(A * b.count * kCostPrim); dec_[n * 7 + i] = D_LEAF; }
        continue;
      }
      const uint32_t L = b.left, R = b.right;
      prims_[n] = prims_[L] + prims_[R];
      auto C = [&](uint32_t m, int i) { return double(cost_[size_t(m) * 7 + (i - 1)]); };
      auto distribute = [&](int j, int& bestK) {
        double best = std::numeric_limits<double>::infinity(); bestK = 1;
        for (int k = 1; k < j; k++) {
          const int kr = j - k;
          if (k > 7 || kr > 7) continue;
          const double c = C(L, k) + C(R, kr);
          if (c < best) { best = c; bestK = k; }
        }
        return best;
      };
      int k8; const double cInternal = distribute(8, k8) + A * kCostNode;
      dist8_[n] = uint8_t(k8);
      const double cLeaf = prims_[n] <= kWideLeafMax ? A * prims_[n] * kCostPrim : std::numeric_limits<double>::infinity();
      if (cLeaf <= cInternal) { cost_[n * 7] = float(cLeaf); dec_[n * 7] = D_LEAF; }
      else { cost_[n * 7] = float(cInternal); dec_[n * 7] = D_INTERNAL; }
      for (int i = 2; i <= 7; i++) {
        int k; const double cd = distribute(i, k);
        const double prev = cost_[n * 7 + (i - 2)];
        if (cd < prev) { cost_[n * 7 + (i - 1)] = float(cd); dec_[n * 7 + (i - 1)] = uint8_t(D_DIST + k); }
        else { cost_[n * 7 + (i - 1)] = float(prev); dec_[n * 7 + (i - 1)] = D_FALL; }
      }
    }
    sahCost = cost_[0];
    emit();
  }

  struct Child { uint32_t bnode; bool inner; };
  void collect(uint32_t n, int i, std::vector<Child>& out) const {
    const uint8_t d = dec_[size_t(n) * 7 + (i - 1)];
    if (bn_[n].count > 0 || d == D_LEAF) { out.push_back({n, false}); return; }
    if (d == D_INTERNAL) { out.push_back({n, true}); return; }
    if (d == D_FALL) { collect(n, i - 1, out); return; }
    const int k = d - D_DIST;
    collect(bn_[n].left, k, out); collect(bn_[n].right, i - k, out);
  }
  void leafTris(uint32_t n, std::vector<uint32_t>& out) const {
    if (bn_[n].count > 0) { for (uint32_t k = 0; k < bn_[n].count; k++) out.push_back(idx_[bn_[n].first + k]); return; }
    leafTris(bn_[n].left, out); leafTris(bn_[n].right, out);
  }

  void emit() {
    struct Item { uint32_t bnode, wide, pend; };     // pend: stack entries a ray can hold when it reaches this node
    std::vector<Item> queue;
    nodes.push_back(Wide8Node{});
    queue.push_back({0, 0, 0});
    maxStack = 0;
    std::vector<Child> kids;
    for (size_t qi = 0; qi < queue.size(); qi++) {
      const Item it = queue[qi];
      kids.clear();
      const BNode& b = bn_[it.bnode];
      // (binary leaves hold exactly one triangle — buildBinary splits by index when it cannot split by position —, so the only
      // binary leaf that becomes a node is the root of a one-triangle subset)
      if (b.count > 0) kids.push_back({it.bnode, false});
      else {
        const int k8 = dist8_[it.bnode];
        collect(b.left, k8, kids); collect(b.right, 8 - k8, kids);
      }
      std::vector<std::vector<uint32_t>> tris;
      std::vector<Bounds3> cb;
      std::vector<bool> inner;
      for (const Child& c : kids) {
        inner.push_back(c.inner); cb.push_back(bn_[c.bnode].b);
        std::vector<uint32_t> t;
        if (!c.inner) leafTris(c.bnode, t);
        tris.push_back(t);
      }
      const size_t nc = cb.size();
      // ---- 3. slots: greedy assignment minimising dot(centroid - centre, dir(slot)), dir bit a set = -1 on axis a
      double centre[3] = {0, 0, 0};
      {
        Bounds3 u; for (const Bounds3& x : cb) for (int a = 0; a < 3; a++) { u.mn[a] = std::min(u.mn[a], x.mn[a]); u.mx[a] = std::max(u.mx[a], x.mx[a]); }
        for (int a = 0; a < 3; a++) centre[a] = 0.5 * (double(u.mn[a]) + u.mx[a]);
      }
      int slotOf[8]; bool slotUsed[8] = {false}; bool done[8] = {false};
      for (size_t r = 0; r < nc; r++) {
        double best = std::numeric_limits<double>::infinity(); int bc = -1, bs = -1;
        for (size_t c = 0; c < nc; c++) {
          if (done[c]) continue;
          for (int s = 0; s < 8; s++) {
            if (slotUsed[s]) continue;
            double v = 0;
            for (int a = 0; a < 3; a++) v += (0.5 * (double(cb[c].mn[a]) + cb[c].mx[a]) - centre[a]) * (((s >> a) & 1) ? -1.0 : 1.0);
            if (v < best) { best = v; bc = int(c); bs = s; }
          }
        }
        slotOf[bc] = bs; slotUsed[bs] = true; done[bc] = true;
      }
      int childAt[8]; for (int s = 0; s < 8; s++) childAt[s] = -1;
      for (size_t c = 0; c < nc; c++) childAt[slotOf[c]] = int(c);
      // ---- 4. the node: grid, planes, links
      Wide8Node w{};
      long double lo[3], hi[3];
      for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<double>::infinity(); hi[a] = -std::numeric_limits<double>::infinity(); }
      for (const Bounds3& x : cb) for (int a = 0; a < 3; a++) { lo[a] = std::min<long double>(lo[a], (long double)x.mn[a] - pad_); hi[a] = std::max<long double>(hi[a], (long double)x.mx[a] + pad_); }
      long double step[3];
      {
        // one step 2^e for the three axes: the smallest that spans the largest extent in kWideGrid steps (a kernel multiply less per
        // axis; the thin axes of a flat node get the same absolute resolution as its long ones)
        int e = -100;
        for (int a = 0; a < 3; a++) {
          float p = float(lo[a]);
          if ((long double)p > lo[a]) p = std::nextafterf(p, -kInf);
          w.p[a] = p;
          const long double ext = hi[a] - (long double)p;
          int ea = ext > 0 ? int(std::ceil(std::log2((double)(ext / kWideGrid)))) : -100;
          ea = ea < -100 ? -100 : ea;
          while (std::ldexp(1.0L, ea) * kWideGrid < ext) ea++;
          e = std::max(e, ea);
        }
        for (int a = 0; a < 3; a++) { w.e[a] = uint8_t(e + 127); step[a] = std::ldexp(1.0L, e); }
      }
      for (int s = 0; s < 8; s++)
        for (int a = 0; a < 3; a++) w.setPlanes(s, a, kWideGrid, 0);   // empty slot: inverted box
      w.childBase = uint32_t(nodes.size());
      w.triBase = uint32_t(order.size());
      uint32_t nInner = 0;
      for (size_t c = 0; c < nc; c++) nInner += inner[c] ? 1u : 0u;
      const uint32_t pendBelow = it.pend + (nInner >= 2u ? 1u : 0u);
      if (pendBelow > maxStack) maxStack = pendBelow;
      for (int s = 0; s < 8; s++) {
        const int c = childAt[s];
        if (c < 0) continue;
        for (int a = 0; a < 3; a++) {
          const long double clo = (long double)cb[c].mn[a] - pad_, chi = (long double)cb[c].mx[a] + pad_;
          long double ql = std::floor((clo - (long double)w.p[a]) / step[a]), qh = std::ceil((chi - (long double)w.p[a]) / step[a]);
          ql = ql < 0 ? 0 : ql > kWideGrid ? kWideGrid : ql; qh = qh < 0 ? 0 : qh > kWideGrid ? kWideGrid : qh;
          w.setPlanes(s, a, uint32_t(ql), uint32_t(qh));
        }
        if (inner[c]) {
          w.imask |= uint8_t(1u << s);
          queue.push_back({kids[c].bnode, uint32_t(nodes.size()), pendBelow});
          nodes.push_back(Wide8Node{});
        } else {
          for (size_t j = 0; j < tris[c].size(); j++) { w.triValid |= 1u << (3 * s + int(j)); order.push_back(tris[c][j]); }
        }
      }
      nodes[it.wide] = w;
    }
  }
};

}  // namespace yart_hip
