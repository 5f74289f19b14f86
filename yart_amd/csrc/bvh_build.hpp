// bvh_build.hpp — host-side binned-SAH BVH build producing the node array and the
// index permutation the kernels traverse.
//
// Restates reference core/bvh.hpp:41-184 (init / updateBounds / subdivide) and
// :273-347 (SahBVH::getSplit, 20 bins x 3 axes), math/bounds.hpp:38-104 and
// core/primitives.hpp:34-47 with the same float operations in the same order, so
// the node array is identical to the reference's (tests compare it byte for byte
// against oracle/_ref). The traversal order — hence equal-t tie breaks and the
// sampler dimensions consumed by alpha tests — depends on this tree, which is why
// it is rebuilt exactly rather than replaced by a "better" builder.
//
// The build is task-parallel without changing a single result: the two subtrees of a node work on disjoint
// ranges of the index array, and the reference's node numbering — children pair allocated when the parent is
// split, then every descendant of the left child before any of the right (bvh.hpp:163-183) — is restored by
// concatenating the subtrees' node lists in that order and shifting their links. Large nodes additionally
// bin their triangles on several threads (bin bounds are min / max folds and counts: order independent and
// exact). `threads = 1` is the plain recursion; tests compare both byte for byte.
#pragma once
#include <algorithm>
#include <atomic>
#include <future>
#include <limits>
#include <thread>
#include <vector>
#include "scene_types.hpp"

namespace yart_hip {

struct Bounds3 {
  float mn[3] = {kInf, kInf, kInf};
  float mx[3] = {-kInf, -kInf, -kInf};
  float area() const {                                   // bounds.hpp:38-41 (half area)
    float sx = mx[0] - mn[0], sy = mx[1] - mn[1], sz = mx[2] - mn[2];
    return sx * sy + sy * sz + sz * sx;
  }
  void join(const Bounds3& b) {                          // bounds.hpp:48-59
    // the reference starts from an empty union and folds both operands in
    Bounds3 u;
    for (int i = 0; i < 3; i++) { u.mn[i] = ymin(u.mn[i], mn[i]); u.mx[i] = ymax(u.mx[i], mx[i]); }
    for (int i = 0; i < 3; i++) { u.mn[i] = ymin(u.mn[i], b.mn[i]); u.mx[i] = ymax(u.mx[i], b.mx[i]); }
    *this = u;
  }
  void expand(const float* p) {                          // bounds.hpp:43-46
    for (int i = 0; i < 3; i++) { mn[i] = ymin(mn[i], p[i]); mx[i] = ymax(mx[i], p[i]); }
  }
};

// bounds::fromPoints (bounds.hpp:89-104): strict comparisons, then pad by 0.001
inline Bounds3 boundsFromPoints(const float* const* pts, int n) {
  Bounds3 b;
  for (int k = 0; k < n; k++)
    for (int i = 0; i < 3; i++) {
      if (pts[k][i] < b.mn[i]) b.mn[i] = pts[k][i];
      if (pts[k][i] > b.mx[i]) b.mx[i] = pts[k][i];
    }
  for (int i = 0; i < 3; i++) { b.mn[i] -= 0.001f; b.mx[i] += 0.001f; }
  return b;
}

// uint32_t(float) as x86-64 performs it (cvttss2si r64 + truncation): NaN -> 0
inline uint32_t f2u_x86(float x) {
  if (x != x) return 0u;
  if (x >= 9.2e18f || x <= -9.2e18f) return 0u;
  return uint32_t(int64_t(x));
}

class SahBvhBuilder {
 public:
  // positions: nv*3 floats; tris: nf * (i0,i1,i2[,...]) with the given stride in uint32
  void build(const float* positions, const uint32_t* tris, uint32_t triStride, uint32_t nTris) {
    pos_ = positions; tris_ = tris; stride_ = triStride; n_ = nTris;
    centroids_.resize(size_t(n_) * 3);
    for (uint32_t i = 0; i < n_; i++) {                  // primitives.hpp:41-46
      const float *v0 = vert(i, 0), *v1 = vert(i, 1), *v2 = vert(i, 2);
      for (int c = 0; c < 3; c++) centroids_[size_t(i) * 3 + c] = ((v0[c] + v1[c]) + v2[c]) / 3.0f;
    }
    triBounds_.resize(n_);                               // fromPoints of each triangle, computed once (same values every time)
    for (uint32_t i = 0; i < n_; i++) {
      const float* p[3] = {vert(i, 0), vert(i, 1), vert(i, 2)};
      triBounds_[i] = boundsFromPoints(p, 3);
    }
    indices.resize(n_);
    for (uint32_t i = 0; i < n_; i++) indices[i] = i;
    BvhNode root{};
    resetBounds(root);
    root.leftFirst = 0;
    root.span = n_;
    nodes = buildSubtree(root, /*boundsKnown=*/false);
    nodesUsed = uint32_t(nodes.size());
  }
  // number of worker threads (0 = hardware concurrency, capped at 32); 1 = the plain recursion
  void setThreads(unsigned t) {
    if (t == 0) { t = std::thread::hardware_concurrency(); if (t == 0) t = 1; }
    threads_ = t > 32 ? 32 : t;
  }

  std::vector<BvhNode> nodes;
  std::vector<uint32_t> indices;
  uint32_t nodesUsed = 1;

 private:
  const float* pos_ = nullptr;
  const uint32_t* tris_ = nullptr;
  uint32_t stride_ = 3, n_ = 0;
  unsigned threads_ = 1;
  static constexpr uint32_t kTaskSpan = 4096;   // subtrees of at most this many triangles are built by one thread
  static constexpr uint32_t kBinSpan = 16384;   // nodes of at least this many triangles bin on several threads
  std::vector<float> centroids_;
  std::vector<Bounds3> triBounds_;

  // the node list of the subtree rooted at `node` (element 0), in the reference's allocation order with links
  // relative to the list: the reference's recursion on one thread
  std::vector<BvhNode> buildSerial(const BvhNode& node, bool boundsKnown) {
    std::vector<BvhNode> out(node.span ? size_t(node.span) * 2 - 1 : 1, BvhNode{});
    for (auto& nd : out) resetBounds(nd);
    out[0] = node;
    uint32_t used = 1;
    if (!boundsKnown) updateBounds(out, 0);
    subdivide(out, used, 0);
    out.resize(used);
    return out;
  }
  // Three passes. (A) from the root down, every node of more than kTaskSpan triangles is split (binning on
  // several threads); what remains are independent subtrees over disjoint index ranges. (B) those are built by a
  // pool of threads, each with the plain recursion. (C) the node lists are concatenated in allocation order.
  struct TopNode { BvhNode node; int left = -1, right = -1, job = -1; };
  std::vector<BvhNode> buildSubtree(BvhNode root, bool boundsKnown) {
    if (threads_ <= 1 || root.span <= kTaskSpan) return buildSerial(root, boundsKnown);
    std::vector<TopNode> top;
    std::vector<int> jobs;                                  // indices into top
    {
      std::vector<BvhNode> tmp(1, root);
      if (!boundsKnown) updateBounds(tmp, 0);
      top.push_back({tmp[0]});
    }
    for (size_t t = 0; t < top.size(); t++) {               // (A) — top grows while it is walked
      if (top[t].node.span <= kTaskSpan) { top[t].job = int(jobs.size()); jobs.push_back(int(t)); continue; }
      BvhNode left{}, right{};
      if (!splitNode(top[t].node, left, right)) continue;   // stays a leaf
      top[t].left = int(top.size()); top.push_back({left});
      top[t].right = int(top.size()); top.push_back({right});
    }
    std::vector<std::vector<BvhNode>> built(jobs.size());   // (B)
    {
      std::atomic<size_t> next{0};
      auto worker = [&] { for (size_t j; (j = next.fetch_add(1)) < jobs.size();) built[j] = buildSerial(top[size_t(jobs[j])].node, true); };
      std::vector<std::future<void>> futs;
      const unsigned extra = unsigned(std::min<size_t>(threads_ - 1, jobs.size() > 0 ? jobs.size() - 1 : 0));
      for (unsigned c = 0; c < extra; c++) futs.push_back(std::async(std::launch::async, worker));
      worker();
      for (auto& f : futs) f.get();
    }
    return assemble(top, built, 0);                         // (C)
  }
  // [node, L0, R0, desc(L)..., desc(R)...]: L[k>=1] -> k + 2, R[k>=1] -> k + |L| + 1
  std::vector<BvhNode> assemble(const std::vector<TopNode>& top, std::vector<std::vector<BvhNode>>& built, int t) const {
    const TopNode& tn = top[size_t(t)];
    if (tn.job >= 0) return std::move(built[size_t(tn.job)]);
    std::vector<BvhNode> out(1, tn.node);
    if (tn.left < 0) return out;
    const std::vector<BvhNode> L = assemble(top, built, tn.left), R = assemble(top, built, tn.right);
    const uint32_t nl = uint32_t(L.size()), nr = uint32_t(R.size());
    out.reserve(size_t(1) + nl + nr);
    out[0].leftFirst = 1; out[0].span = 0;
    auto shifted = [](BvhNode nd, uint32_t by) { if (nd.span == 0) nd.leftFirst += by; return nd; };
    out.push_back(shifted(L[0], 2));
    out.push_back(shifted(R[0], nl + 1));
    for (uint32_t k = 1; k < nl; k++) out.push_back(shifted(L[k], 2));
    for (uint32_t k = 1; k < nr; k++) out.push_back(shifted(R[k], nl + 1));
    return out;
  }

  const float* vert(uint32_t tri, int k) const { return pos_ + size_t(tris_[size_t(tri) * stride_ + k]) * 3; }
  static void resetBounds(BvhNode& nd) {
    for (int i = 0; i < 3; i++) { nd.bmin[i] = kInf; nd.bmax[i] = -kInf; }
  }
  const Bounds3& triBounds(uint32_t tri) const { return triBounds_[tri]; }

  // fn(chunk, lo, hi) over [0, span) cut into contiguous chunks on up to 16 threads (the callers only fold with
  // min / max / integer addition, so the chunking does not change any result)
  template <class Fn>
  void forChunks(uint32_t span, Fn fn) {
    const unsigned parts = (threads_ > 1 && span >= kBinSpan) ? std::min(threads_, 16u) : 1u;
    if (parts == 1) { fn(0u, 0u, span); return; }
    std::vector<std::future<void>> futs;
    for (unsigned c = 1; c < parts; c++) {
      const uint32_t lo = uint32_t(uint64_t(span) * c / parts), hi = uint32_t(uint64_t(span) * (c + 1) / parts);
      futs.push_back(std::async(std::launch::async, [=, &fn] { fn(c, lo, hi); }));
    }
    fn(0u, 0u, uint32_t(uint64_t(span) / parts));
    for (auto& f : futs) f.get();
  }

  void updateBounds(std::vector<BvhNode>& nodes, uint32_t ni) {        // bvh.hpp:101-115
    BvhNode& nd = nodes[ni];
    Bounds3 part[16];
    forChunks(nd.span, [&](unsigned c, uint32_t lo, uint32_t hi) {
      Bounds3 b;
      for (uint32_t i = lo; i < hi; i++) b.join(triBounds(indices[nd.leftFirst + i]));
      part[c] = b;
    });
    Bounds3 b;
    for (int i = 0; i < 3; i++) { b.mn[i] = nd.bmin[i]; b.mx[i] = nd.bmax[i]; }
    for (const Bounds3& q : part) b.join(q);             // empty parts are the identity of the fold
    for (int i = 0; i < 3; i++) { nd.bmin[i] = b.mn[i]; nd.bmax[i] = b.mx[i]; }
  }

  bool getSplit(const BvhNode& nd, uint8_t& axis, float& splitPos) {   // bvh.hpp:273-347
    float minCost = kInf;
    Bounds3 cb;                                          // getCentroidBounds, bvh.hpp:123-134
    {
      Bounds3 part[16];
      forChunks(nd.span, [&](unsigned c, uint32_t lo, uint32_t hi) {
        Bounds3 b;
        for (uint32_t i = nd.leftFirst + lo; i < nd.leftFirst + hi; i++) b.expand(&centroids_[size_t(indices[i]) * 3]);
        part[c] = b;
      });
      for (const Bounds3& q : part) for (int i = 0; i < 3; i++) { cb.mn[i] = ymin(cb.mn[i], q.mn[i]); cb.mx[i] = ymax(cb.mx[i], q.mx[i]); }
    }
    constexpr uint32_t nBins = 20, nSplits = nBins - 1;
    for (uint8_t a = 0; a < 3; a++) {
      float bmin = cb.mn[a], bsize = cb.mx[a] - cb.mn[a];
      uint32_t count[nBins] = {0};
      Bounds3 bb[nBins];
      float scale = float(nBins) / bsize;
      struct Bins { uint32_t count[nBins]; Bounds3 bb[nBins]; };
      auto binRange = [&](uint32_t* cnt, Bounds3* bnd, uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; i++) {
          uint32_t t = indices[nd.leftFirst + i];
          const Bounds3& tb = triBounds(t);
          uint32_t b = f2u_x86(scale * (centroids_[size_t(t) * 3 + a] - bmin));
          if (nBins - 1 < b) b = nBins - 1;                // std::min(nBins - 1, b)
          cnt[b]++;
          bnd[b].join(tb);
        }
      };
      if (threads_ <= 1 || nd.span < kBinSpan) binRange(count, bb, 0, nd.span);
      else {
        std::vector<Bins> part(16);
        bool used[16] = {false};
        forChunks(nd.span, [&](unsigned c, uint32_t lo, uint32_t hi) {
          Bins& q = part[c];
          used[c] = true;
          for (uint32_t k = 0; k < nBins; k++) q.count[k] = 0;
          binRange(q.count, q.bb, lo, hi);
        });
        for (unsigned c = 0; c < 16; c++) if (used[c])
          for (uint32_t k = 0; k < nBins; k++) { count[k] += part[c].count[k]; bb[k].join(part[c].bb[k]); }
      }
      float costs[nSplits] = {0.0f};
      uint32_t countBelow = 0;
      Bounds3 below;
      for (uint32_t i = 0; i < nSplits; i++) {
        below.join(bb[i]);
        countBelow += count[i];
        costs[i] += float(countBelow) * below.area();
      }
      uint32_t countAbove = 0;
      Bounds3 above;
      for (uint32_t i = nSplits; i > 0; i--) {
        above.join(bb[i]);
        countAbove += count[i];
        costs[i - 1] += float(countAbove) * above.area();
      }
      for (uint32_t i = 0; i < nSplits; i++) {
        if (costs[i] < minCost) {
          minCost = costs[i];
          axis = a;
          splitPos = bmin + bsize * (float(i + 1) / float(nBins));
        }
      }
    }
    Bounds3 nb;
    for (int i = 0; i < 3; i++) { nb.mn[i] = nd.bmin[i]; nb.mx[i] = nd.bmax[i]; }
    float leafCost = (float(nd.span) - 0.5f) * nb.area();
    if (nd.span <= 20 && leafCost < minCost) return false;   // MAX_LEAF_SIZE, bvh.hpp:14
    return true;
  }

  // one step of bvh.hpp:140-184: choose the split, partition the node's index range, bound the two halves
  bool splitNode(const BvhNode& nd, BvhNode& left, BvhNode& right) {
    uint8_t axis = 0;
    float splitPos = 0;
    if (!getSplit(nd, axis, splitPos)) return false;
    const uint32_t first = nd.leftFirst, span = nd.span;
    int64_t i = first;
    int64_t j = i + span - 1;
    while (i <= j) {
      float c = centroids_[size_t(indices[i]) * 3 + axis];
      if (c < splitPos) i++;
      else { uint32_t t = indices[i]; indices[i] = indices[j]; indices[j] = t; j--; }
    }
    uint32_t leftCount = uint32_t(i - first);
    if (leftCount == 0 || leftCount == span) return false;
    std::vector<BvhNode> tmp(2);
    resetBounds(tmp[0]); resetBounds(tmp[1]);
    tmp[0].leftFirst = first; tmp[0].span = leftCount;
    tmp[1].leftFirst = uint32_t(i); tmp[1].span = span - leftCount;
    updateBounds(tmp, 0);
    updateBounds(tmp, 1);
    left = tmp[0]; right = tmp[1];
    return true;
  }

  void subdivide(std::vector<BvhNode>& nodes, uint32_t& used, uint32_t ni) {   // bvh.hpp:140-184
    BvhNode left{}, right{};
    if (!splitNode(nodes[ni], left, right)) return;
    uint32_t leftIdx = used++;
    uint32_t rightIdx = used++;
    nodes[leftIdx] = left;
    nodes[rightIdx] = right;
    nodes[ni].leftFirst = leftIdx;
    nodes[ni].span = 0;
    subdivide(nodes, used, leftIdx);
    subdivide(nodes, used, rightIdx);
  }
};

}  // namespace yart_hip
