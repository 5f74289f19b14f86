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
#pragma once
#include <vector>
#include <limits>
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
    indices.resize(n_);
    for (uint32_t i = 0; i < n_; i++) indices[i] = i;
    nodes.assign(n_ ? size_t(n_) * 2 - 1 : 1, BvhNode{});
    for (auto& nd : nodes) resetBounds(nd);
    nodesUsed = 1;
    nodes[0].leftFirst = 0;
    nodes[0].span = n_;
    updateBounds(0);
    subdivide(0);
    nodes.resize(nodesUsed);
  }

  std::vector<BvhNode> nodes;
  std::vector<uint32_t> indices;
  uint32_t nodesUsed = 1;

 private:
  const float* pos_ = nullptr;
  const uint32_t* tris_ = nullptr;
  uint32_t stride_ = 3, n_ = 0;
  std::vector<float> centroids_;

  const float* vert(uint32_t tri, int k) const { return pos_ + size_t(tris_[size_t(tri) * stride_ + k]) * 3; }
  static void resetBounds(BvhNode& nd) {
    for (int i = 0; i < 3; i++) { nd.bmin[i] = kInf; nd.bmax[i] = -kInf; }
  }
  Bounds3 triBounds(uint32_t tri) const {
    const float* p[3] = {vert(tri, 0), vert(tri, 1), vert(tri, 2)};
    return boundsFromPoints(p, 3);
  }

  void updateBounds(uint32_t ni) {                       // bvh.hpp:101-115
    BvhNode& nd = nodes[ni];
    Bounds3 b;
    for (int i = 0; i < 3; i++) { b.mn[i] = nd.bmin[i]; b.mx[i] = nd.bmax[i]; }
    for (uint32_t i = 0; i < nd.span; i++) b.join(triBounds(indices[nd.leftFirst + i]));
    for (int i = 0; i < 3; i++) { nd.bmin[i] = b.mn[i]; nd.bmax[i] = b.mx[i]; }
  }

  bool getSplit(uint32_t ni, uint8_t& axis, float& splitPos) const {   // bvh.hpp:273-347
    const BvhNode& nd = nodes[ni];
    float minCost = kInf;
    Bounds3 cb;                                          // getCentroidBounds, bvh.hpp:123-134
    for (uint32_t i = nd.leftFirst; i < nd.leftFirst + nd.span; i++)
      cb.expand(&centroids_[size_t(indices[i]) * 3]);
    constexpr uint32_t nBins = 20, nSplits = nBins - 1;
    for (uint8_t a = 0; a < 3; a++) {
      float bmin = cb.mn[a], bsize = cb.mx[a] - cb.mn[a];
      uint32_t count[nBins] = {0};
      Bounds3 bb[nBins];
      float scale = float(nBins) / bsize;
      for (uint32_t i = 0; i < nd.span; i++) {
        uint32_t t = indices[nd.leftFirst + i];
        Bounds3 tb = triBounds(t);
        uint32_t b = f2u_x86(scale * (centroids_[size_t(t) * 3 + a] - bmin));
        if (nBins - 1 < b) b = nBins - 1;                // std::min(nBins - 1, b)
        count[b]++;
        bb[b].join(tb);
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

  void subdivide(uint32_t ni) {                          // bvh.hpp:140-184
    uint8_t axis = 0;
    float splitPos = 0;
    if (!getSplit(ni, axis, splitPos)) return;
    const uint32_t first = nodes[ni].leftFirst, span = nodes[ni].span;
    int64_t i = first;
    int64_t j = i + span - 1;
    while (i <= j) {
      float c = centroids_[size_t(indices[i]) * 3 + axis];
      if (c < splitPos) i++;
      else { uint32_t t = indices[i]; indices[i] = indices[j]; indices[j] = t; j--; }
    }
    uint32_t leftCount = uint32_t(i - first);
    if (leftCount == 0 || leftCount == span) return;
    uint32_t leftIdx = nodesUsed++;
    uint32_t rightIdx = nodesUsed++;
    nodes[leftIdx].leftFirst = first;
    nodes[leftIdx].span = leftCount;
    nodes[rightIdx].leftFirst = uint32_t(i);
    nodes[rightIdx].span = span - leftCount;
    nodes[ni].leftFirst = leftIdx;
    nodes[ni].span = 0;
    updateBounds(leftIdx);
    updateBounds(rightIdx);
    subdivide(leftIdx);
    subdivide(rightIdx);
  }
};

}  // namespace yart_hip
