// wide_bvh.hpp — 4-wide re-layout of the reference's binary SAH tree.
//
// The binary tree itself is built exactly as the reference builds it (bvh_build.hpp, byte-identical node and index
// arrays: core/bvh.hpp:140-184, 273-347) and stays the tree of the general / retry kernels, whose walk is the
// reference's walk step for step. The lean kernels — rays that are known not to have met an alpha-tested or
// NEE-transparent candidate — do not depend on that order: a closest hit is the minimum over the triangles the ray
// reaches, whatever the order they are reached in (only two triangles hit at exactly the same t can swap, and those
// are reported by the parity tests' identical-pixel fraction). For them every inner node of the binary tree whose
// parent was collapsed away becomes one 128-byte record holding up to four children:
//
//   * children = the node's two children, the larger (by surface area) inner ones replaced by THEIR children until
//     there are four (greedy surface-area collapse); the boxes are the binary nodes' own float boxes, untouched, so a
//     child is reached by exactly the rays that pass the reference's test of that box;
//   * boxes are stored as six rows of four floats (min x | min y | min z | max x | max y | max z of the four children):
//     a ray fetches its entry planes from one of two rows per axis, chosen once per ray by the sign of its direction
//     (an address offset), instead of twelve per-lane selects per box pair;
//   * a link word per child: leaf -> first leaf record | alpha bit | span << 27 (31 = 31 or more, traverse.hpp::leafSpan),
//     inner -> index of the child's record | alpha bit; an empty slot has a box nothing hits.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "scene_types.hpp"

namespace yart_hip {

struct Wide4 {
  float lo[3][4];            // min x / y / z of the four children
  float hi[3][4];            // max x / y / z
  uint32_t link[4];
  uint32_t pad[4];
};
static_assert(sizeof(Wide4) == 128, "one wide node is one 128-byte line");
constexpr uint32_t kWideEmpty = 0xffffffffu;

// nodes: one mesh's binary tree, mesh-local indices, leftFirst carrying kLinkAlphaBit (host_scene.hpp sets it before
// this is called). Appends the mesh's wide records to `out`; record 0 of the mesh stands for the binary root.
inline void buildWide4(const BvhNode* nodes, uint32_t nNodes, std::vector<Wide4>& out) {
  const size_t base = out.size();
  auto area = [&](uint32_t n) {
    const BvhNode& b = nodes[n];
    const float dx = b.bmax[0] - b.bmin[0], dy = b.bmax[1] - b.bmin[1], dz = b.bmax[2] - b.bmin[2];
    return dx * dy + dy * dz + dz * dx;
  };
  auto emptySlot = [](Wide4& w, int k) {
    for (int a = 0; a < 3; a++) { w.lo[a][k] = std::numeric_limits<float>::infinity(); w.hi[a][k] = -std::numeric_limits<float>::infinity(); }
    w.link[k] = kWideEmpty;
  };
  // work list of (binary node, its record); records are allocated when their node is first referenced
  std::vector<uint32_t> recordOf(nNodes, kWideEmpty);
  std::vector<uint32_t> todo;
  auto recordFor = [&](uint32_t n) {
    if (recordOf[n] == kWideEmpty) { recordOf[n] = uint32_t(out.size() - base); out.push_back(Wide4{}); todo.push_back(n); }
    return recordOf[n];
  };
  (void)nNodes;
  recordFor(0);
  for (size_t t = 0; t < todo.size(); t++) {
    const uint32_t n = todo[t];
    uint32_t kids[4]; int nk = 0;
    if (nodes[n].span > 0) kids[nk++] = n;                      // a mesh whose root is a leaf: one child, the leaf itself
    else {
      const uint32_t l = nodes[n].leftFirst & kLinkIndexMask;
      kids[nk++] = l; kids[nk++] = l + 1;
      while (nk < 4) {                                          // open the inner child of largest surface area
        int best = -1; float bestA = -1.0f;
        for (int k = 0; k < nk; k++)
          if (nodes[kids[k]].span == 0 && area(kids[k]) > bestA) { bestA = area(kids[k]); best = k; }
        if (best < 0) break;
        const uint32_t c = nodes[kids[best]].leftFirst & kLinkIndexMask;
        kids[best] = c; kids[nk++] = c + 1;
      }
    }
    Wide4 w{};
    for (int k = 0; k < 4; k++) {
      if (k >= nk) { emptySlot(w, k); continue; }
      const BvhNode& b = nodes[kids[k]];
      for (int a = 0; a < 3; a++) { w.lo[a][k] = b.bmin[a]; w.hi[a][k] = b.bmax[a]; }
      const uint32_t alpha = b.leftFirst & kLinkAlphaBit;
      if (b.span > 0) w.link[k] = (b.leftFirst & kLinkIndexMask) | alpha | ((b.span < kSpanBig ? b.span : kSpanBig) << kSpanShift);
      else w.link[k] = recordFor(kids[k]) | alpha;
    }
    out[base + recordOf[n]] = w;
  }
}

}  // namespace yart_hip
