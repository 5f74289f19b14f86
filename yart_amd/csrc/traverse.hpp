// traverse.hpp — scene-graph walk, ordered BVH traversal, ray/triangle and ray/box tests.
//
// Restates reference cpu/ray-integrator.cpp:20-261 (testNode, testMesh, testBVH,
// testTriangle, testBoundingBox) and core/ray.hpp:15-25. The traversal order is
// the reference's exactly (near child first, far child pushed with its entry
// distance, pop-cull on d < hit.t, leaf triangles in index order, NEE rays stop a
// leaf at the first hit but keep walking) because two things depend on it:
// which of several equal-t triangles wins, and the sampler dimension consumed by
// stochastic alpha tests inside the traversal (ray-integrator.cpp:207-211).
//
// GPU layout decisions (vs the reference's pointer chasing):
//  * the recursive testNode becomes a linear pre-order walk with skip links;
//  * an inner visit reads both children with one 64-byte access (nodes are 32 B
//    and siblings adjacent), a leaf visit streams 48-byte LeafTri records;
//  * the 64-entry traversal stack lives in LDS, lane-interleaved (8-byte
//    entries: node index + entry distance), with a global-memory spill area for
//    the entries beyond LDS_STACK — no overflow is possible up to the
//    reference's own fixed depth of 64;
//  * only (t, u, v, slot, node) is tracked during the walk; normals, uvs,
//    tangents and the object->world chain are evaluated once for the final hit.
#pragma once
#include "bsdf.hpp"
#include "sampler.hpp"

namespace yart_hip {

constexpr uint32_t kRefStackDepth = 64;     // ray-integrator.cpp:92-93
struct RayO {                // core/ray.hpp:9-30 (object-space ray with slab precomputations)
  f3 o, d, idir, odir;
  uint32_t sx, sy, sz;
};
YART_HD RayO makeRay(f3 o, f3 d) {
  RayO r; r.o = o; r.d = d;
  r.idir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);   // "1.0 / dir": double divide narrowed == float divide
  r.odir = (-o) / d;
  r.sx = d.x < 0.0f ? 1u : 0u; r.sy = d.y < 0.0f ? 1u : 0u; r.sz = d.z < 0.0f ? 1u : 0u;
  return r;
}

// ray-integrator.cpp:231-261
YART_HD bool testBox(const RayO& r, float tIntMin, float tIntMax, const float* bmin, const float* bmax,
                     float& d) {
  float lo_x = r.sx ? bmax[0] : bmin[0], hi_x = r.sx ? bmin[0] : bmax[0];
  float lo_y = r.sy ? bmax[1] : bmin[1], hi_y = r.sy ? bmin[1] : bmax[1];
  float lo_z = r.sz ? bmax[2] : bmin[2], hi_z = r.sz ? bmin[2] : bmax[2];
  float tmin0 = lo_x * r.idir.x + r.odir.x, tmin1 = lo_y * r.idir.y + r.odir.y, tmin2 = lo_z * r.idir.z + r.odir.z;
  float tmax0 = hi_x * r.idir.x + r.odir.x, tmax1 = hi_y * r.idir.y + r.odir.y, tmax2 = hi_z * r.idir.z + r.odir.z;
  // Reference: t0 = max(tmin[i], t0) with max(m,n) = m > n ? m : n, t1 = min(tmax[i], t1)
  // likewise. t0 / t1 themselves are never NaN (they start from tMin / hit.t and a NaN
  // candidate loses the comparison), so IEEE maxNum / minNum — which also return the
  // non-NaN operand — give the same value; they compile to v_max3_f32 / v_min3_f32.
  float t0 = tIntMin, t1 = tIntMax;
  t0 = fmaxf(tmin0, t0); t0 = fmaxf(tmin1, t0); t0 = fmaxf(tmin2, t0);
  t1 = fminf(tmax0, t1); t1 = fminf(tmax1, t1); t1 = fminf(tmax2, t1);
  d = t0;
  return t1 >= t0;
}

// Both children of an inner node at once (same operations per box as testBox). (Packed fp32 for the six multiply /
// add pairs was measured: 12 VALU fewer per step, but past an occupancy step in registers and slower)
YART_HD void testBox2(const RayO& r, float tIntMin, float tIntMax, const BvhNode& c1, const BvhNode& c2,
                      bool& hit1, bool& hit2, float& d1, float& d2) {
  hit1 = testBox(r, tIntMin, tIntMax, c1.bmin, c1.bmax, d1);
  hit2 = testBox(r, tIntMin, tIntMax, c2.bmin, c2.bmax, d2);
}

// Lane-private traversal stack: entry k of this lane is lds[k * ldsStride] for
// k < ldsDepth, spill[(k - ldsDepth) * spillStride] beyond.
// The LDS part is addressed through an LDS-address-space pointer so that it compiles to
// ds_write_b64 / ds_read_b64 (a generic pointer selected against the spill pointer turns
// every pop into a flat_load with a full vmcnt+lgkmcnt wait).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) uint64_t lds_u64;
#else
typedef uint64_t lds_u64;
#endif
// Walks the lean kernels hand to the general ones are RESUMED, not restarted (trace_lean.hpp): one record per handed-over ray.
//   word 0: path slot, scene node, flags (bit 0 didHit, bit 1 meshDidHit, bits 8.. stack entries), link word of the leaf
//   word 1: the leaf's entry distance, hit.t, candidate mask      word 2 (closest-hit rays that have a hit): hit.u, hit.v, hit.tri,
//   hit.node | hit.backSide << 20                                  words 3..10: the traversal stack, two entries per word
constexpr uint32_t kResumeStack = 16;                              // deeper stacks: the ray is restarted instead
constexpr uint32_t kResumeWords = 3 + kResumeStack / 2;            // 16-byte words per record (176 B)
constexpr uint32_t kResumeFlag = 0x80000000u;                      // retry-queue word = record index | flag (plain word = slot: restart)
struct TravStack {
  lds_u64* lds; uint32_t ldsStride; uint32_t ldsDepth;
  uint64_t* spill; uint32_t spillStride;
  // resume records of this launch (null: every hand-over is a restart): written by the lean kernels, read by the general ones
  f4* rec = nullptr; uint32_t* recCursor = nullptr; uint32_t recCap = 0;
};
YART_HD void stackPush(const TravStack& s, uint32_t k, uint32_t node, float d) {
  uint32_t db = __builtin_bit_cast(uint32_t, d);
  uint64_t e = uint64_t(node) | (uint64_t(db) << 32);
  if (k < s.ldsDepth) s.lds[k * s.ldsStride] = e;
  else s.spill[(k - s.ldsDepth) * s.spillStride] = e;
}
YART_HD void stackPop(const TravStack& s, uint32_t k, uint32_t& node, float& d) {
  uint64_t e;
  if (k < s.ldsDepth) e = s.lds[k * s.ldsStride];
  else e = s.spill[(k - s.ldsDepth) * s.spillStride];
  node = uint32_t(e);
  d = __builtin_bit_cast(float, uint32_t(e >> 32));
}

YART_HD uint64_t stackPeek(const TravStack& s, uint32_t k) {
  return k < s.ldsDepth ? s.lds[k * s.ldsStride] : s.spill[(k - s.ldsDepth) * s.spillStride];
}
YART_HD void stackPoke(const TravStack& s, uint32_t k, uint64_t e) {
  if (k < s.ldsDepth) s.lds[k * s.ldsStride] = e;
  else s.spill[(k - s.ldsDepth) * s.spillStride] = e;
}

struct HitRec {              // what the walk tracks of cpu/hit.hpp
  float t;                   // in: tMax, out: closest t
  float u, v;                // barycentrics of p1, p2
  uint32_t tri;              // mesh-local triangle index (hit.idx)
  uint32_t node;             // scene node owning the hit
  uint32_t backSide;         // bit 0: back side; bits 1..: material of the hit triangle (shade-queue class key)
};

// Instrumented build (-DYART_COUNT_TRAVERSAL, libyart_hip_count.so): exact numbers of
// box tests / triangle tests / traversals for the algorithmic-bytes roofline figure
// (B_traversal = 32*N_box + 52*N_tri + 48). Never in the timed library.
#if defined(YART_COUNT_TRAVERSAL)
#define YART_COUNT(field, n) (actx.field += (n))
#else
#define YART_COUNT(field, n) ((void)0)
#endif

// Traversal variants (template parameter MODE of traverseMesh / traverseScene):
//   TRAV_FAST      no alpha / transparency code: a candidate hit on such a triangle sets
//                  AlphaCtx::deferred and abandons the ray, which the caller then traces again
//                  with the general variant (results are per ray, so a restart is exact). The
//                  side path (texture fetches + a ZSobol draw) costs ~20 VGPRs in the leaf loop.
//   TRAV_IDENTITY  every scene node's transform chain is the identity (checked at scene build):
//                  no 4x4 products, the world ray (+0) is used for every node; ~30 VGPRs.
// Together they bring the closest-hit kernel from 127 to 74 VGPRs, i.e. from 4 to 6 waves/SIMD.
enum : int { TRAV_GENERAL = 0, TRAV_FAST = 1, TRAV_IDENTITY = 2 };

struct AlphaCtx {            // state the stochastic alpha test draws from
  Sampler* sampler;
  SamplerConfig cfg;
  bool deferred = false;     // TRAV_FAST: the ray met an alpha / transparent candidate
#if defined(YART_COUNT_TRAVERSAL)
  uint32_t nBox = 0, nTri = 0, nTrav = 0, nResumed = 0;
#endif
};

// uv / normal interpolation of ray-integrator.cpp:198-213
// (tri: scene-wide index of the triangle's ShadeTri record, as the leaf records and HitRec::tri carry it)
YART_HD void interpUVN(const SceneDev& sc, uint32_t tri, float u, float v, f2& uv, f3& n) {
  const ShadeTri& st = sc.shadeTris[tri];
  const float w = 1.0f - u - v;
  const f2 t0 = mk2(st.uv[0][0], st.uv[0][1]), t1 = mk2(st.uv[1][0], st.uv[1][1]), t2 = mk2(st.uv[2][0], st.uv[2][1]);
  uv = (w * t0 + u * t1) + v * t2;
  n = (w * mk3(st.n[0][0], st.n[0][1], st.n[0][2]) + u * mk3(st.n[1][0], st.n[1][1], st.n[1][2])) +
      v * mk3(st.n[2][0], st.n[2][1], st.n[2][2]);
}

// testBVH (ray-integrator.cpp:84-160) + testTriangle (:163-229) for one mesh.
//
// Same per-ray operation order as the reference's loop, arranged for SIMT execution:
//  * "while-while": inner-node and pop steps run in a tight loop until the lane stands
//    at a leaf it has to test (d < hit.t) or has left the tree; the leaf's triangle loop
//    then runs — lanes of a wave reconverge on the (rare, expensive) leaf code instead of
//    interleaving it with other lanes' inner steps;
//  * a stack entry carries the far child's link word (leftFirst | span << 27) and entry
//    distance, so a pop needs no dependent node fetch before the children can be loaded.
// The link word of a stack entry has 5 bits for the leaf span. The reference keeps a node as a leaf of ANY span
// when its split is degenerate (bvh.hpp:159-161: coincident centroids, stacked duplicates), so 31 stands for
// "31 or more" and the leaf loop then takes the true span from the leaf's first record (LeafTri::matFlags
// bits 8..31, written at scene build for every leaf).
YART_HD uint32_t packLink(uint32_t leftFirst, uint32_t span) {
  return leftFirst | ((span < kSpanBig ? span : kSpanBig) << kSpanShift);
}
YART_HD uint32_t leafSpan(const LeafTri* leaves, uint32_t first, uint32_t span) {
  return span < kSpanBig ? span : (leaves[first].matFlags >> kLeafSpanShift);
}
template <bool NEE, int MODE = TRAV_GENERAL>
YART_HD bool traverseMesh(const SceneDev& sc, const MeshDev& mesh, uint32_t nodeIdx, const RayO& ray,
                          float tMin, HitRec& hit, f3& attenuation, const TravStack& stk,
                          AlphaCtx& actx, bool occludedBefore = false) {
  // Lean shadow kernel only (NEE && TRAV_FAST): once the ray is known to be occluded, the rest of
  // the reference's walk can only matter through alpha tests (sampler draws); subtrees without
  // alpha-tested triangles are skipped, and an alpha candidate hands the ray to the general kernel
  // as before. From the first hit on the walk keeps its interval and looks at every triangle of the
  // leaves it visits (see the leaf loop's end): a superset of the alpha candidates the reference
  // tests — never a missed hand-over, at worst a spurious one.
  constexpr bool kPrune = NEE && (MODE & TRAV_FAST);
  const BvhNode* nodes = sc.bvhNodes + mesh.nodeOffset;
  const LeafTri* leaves = sc.leafTris + mesh.leafOffset;
  uint32_t stackIdx = 0;
  bool didHit = false;
  float d;
  const BvhNode root = nodes[0];
  YART_COUNT(nBox, 1);
  if (!testBox(ray, tMin, hit.t, root.bmin, root.bmax, d)) return false;
  uint32_t leftFirst = root.leftFirst, span = root.span;
  bool alive = true;
  // "visit the current node": the reference's pop-cull, plus the pruning above
#define YART_VISIT() (d < hit.t && (!kPrune || !(occludedBefore || didHit) || (leftFirst & kLinkAlphaBit)))
  while (alive) {
    // ---- inner nodes and pops, until a leaf must be tested
    while (alive && !(span > 0 && YART_VISIT())) {
      bool pop = true;
      if (YART_VISIT()) {
        const BvhNode* pair = nodes + (leftFirst & kLinkIndexMask);   // siblings are adjacent: one 64-byte access
        const BvhNode c1 = pair[0], c2 = pair[1];
        YART_COUNT(nBox, 2);
        float d1, d2;
        bool hit1, hit2;
        testBox2(ray, tMin, hit.t, c1, c2, hit1, hit2, d1, d2);
        if (hit1 || hit2) {
          // child1 is the near one unless it was missed or (both hit and d1 > d2)
          const bool firstNear = hit1 && !(hit2 && d1 > d2);
          if (hit1 && hit2)
            stackPush(stk, stackIdx++, firstNear ? packLink(c2.leftFirst, c2.span) : packLink(c1.leftFirst, c1.span),
                      firstNear ? d2 : d1);
          d = firstNear ? d1 : d2;
          leftFirst = firstNear ? c1.leftFirst : c2.leftFirst;
          span = firstNear ? c1.span : c2.span;
          pop = false;
        }
      }
      if (pop) {
        if (stackIdx == 0) alive = false;
        else {
          uint32_t link;
          stackPop(stk, --stackIdx, link, d);
          leftFirst = link & ((1u << kSpanShift) - 1u); span = link >> kSpanShift;
        }
      }
    }
    if (!alive) break;
    // ---- leaf: triangles in index order
    const uint32_t nLeaf = leafSpan(leaves, leftFirst & kLinkIndexMask, span);
    for (uint32_t i = 0; i < nLeaf; i++) {
      const LeafTri tr = leaves[(leftFirst & kLinkIndexMask) + i];
      YART_COUNT(nTri, 1);
      const f3 p0 = mk3(tr.p0[0], tr.p0[1], tr.p0[2]);
      const f3 edge1 = mk3(tr.e1[0], tr.e1[1], tr.e1[2]);
      const f3 edge2 = mk3(tr.e2[0], tr.e2[1], tr.e2[2]);
      bool accepted = false;
      do {
        const f3 rayEdge2 = cross(ray.d, edge2);
        const float det = dot(edge1, rayEdge2);
        if (double(fabsf(det)) < 1e-12) break;          // math_base.hpp:11 epsilon is a double
        const float invDet = 1.0f / det;
        const f3 b = ray.o - p0;
        const float u = dot(b, rayEdge2) * invDet;
        if (u < 0.0f || u > 1.0f) break;
        const f3 bEdge1 = cross(b, edge1);
        const float v = dot(ray.d, bEdge1) * invDet;
        if (v < 0.0f || u + v > 1.0f) break;
        const float t = dot(edge2, bEdge1) * invDet;
        if (t <= tMin || hit.t <= t) break;
        // (a transparent surface is an ordinary hit for a closest-hit ray: only NEE rays treat it specially)
        // and once a shadow ray is occluded its attenuation is never used)
        if ((MODE & TRAV_FAST) &&
            (tr.matFlags & ((NEE && !(occludedBefore || didHit)) ? (MAT_HAS_ALPHA | MAT_TRANSPARENT) : MAT_HAS_ALPHA))) {
          actx.deferred = true;
          return false;
        }
        // (lean shadow walk, already occluded: the interval stays what it was at the first hit — see the leaf loop's end)
        if (kPrune && (occludedBefore || didHit)) break;
        if (!(MODE & TRAV_FAST) && (tr.matFlags & (MAT_HAS_ALPHA | MAT_TRANSPARENT))) {
          // slow path: alpha cut-outs and NEE-transparent surfaces
          f2 uv; f3 n;
          interpUVN(sc, tr.triIdx, u, v, uv, n);
          const MaterialDev& mt = sc.materials[tr.material];
          if (tr.matFlags & MAT_HAS_ALPHA) {
            float alpha = matAlpha(sc, mt, uv);
            if (alpha < 1.0f && get1D(*actx.sampler, actx.cfg) > alpha) break;
          }
          if (NEE && (tr.matFlags & MAT_TRANSPARENT)) {
            attenuation *= absDot(n, ray.d) * matBase(sc, mt, uv);
            break;
          }
        }
        hit.t = t; hit.u = u; hit.v = v; hit.tri = tr.triIdx; hit.node = nodeIdx;
        hit.backSide = (det < 0 ? 1u : 0u) | (tr.material << 1);
        accepted = true;
      } while (false);
      didHit |= accepted;
      // The reference leaves a leaf's loop after the first triangle once the shadow ray has a hit IN THIS MESH. The lean shadow
      // walk does so only at the hit itself (up to there it IS the reference's walk); from then on it looks for alpha candidates in
      // a superset of what the reference can still test: every triangle of every leaf it visits, in the interval as it stood at
      // the first hit. It accepts nothing further — a later hit is one the reference may never test (its own interval is
      // shorter by then, or it is past the first triangle of a leaf), and accepting it would shorten the interval below the
      // reference's or start the first-triangle rule in a mesh where the reference still tests every triangle: alpha candidates
      // the reference draws for would then go unseen (found by the scene fuzz, tests/test_fuzz_scenes.py).
      if (NEE && (kPrune ? accepted : didHit)) break;
    }
    if (stackIdx == 0) break;
    uint32_t link;
    stackPop(stk, --stackIdx, link, d);
    leftFirst = link & ((1u << kSpanShift) - 1u); span = link >> kSpanShift;
  }
#undef YART_VISIT
  return didHit;
}

// World ray -> object space of scene node `idx`: the reference re-derives the ray at
// every level from its parent's object-space ray (ray-integrator.cpp:26-29).
YART_HD void objectRay(const SceneDev& sc, uint32_t idx, f3 o, f3 d, f3& oo, f3& od) {
  if (sc.nodes[idx].pad[0] & 1u) {
    // every transform on the chain is the identity: each 4x4 product reduces to x + 0.0f
    // (accumulation from +0 turns -0 into +0, everything else is unchanged), and that is
    // idempotent, so one application equals the reference's depth+1 applications
    oo = o + 0.0f; od = d + 0.0f;
    return;
  }
  uint32_t chain[kMaxNodeDepth];
  uint32_t n = 0;
  for (int32_t i = int32_t(idx); i >= 0 && n < kMaxNodeDepth; i = sc.nodes[i].parent) chain[n++] = uint32_t(i);
  for (uint32_t k = n; k-- > 0;) {
    const NodeDev& nd = sc.nodes[chain[k]];
    o = mulPoint(nd.xf.inv, o);
    d = mulVector(nd.xf.inv, d);
  }
  oo = o; od = d;
}

// Object-space ray of consecutive scene nodes without re-walking the ancestor chain: the
// reference hands each child the parent's object-space ray (ray-integrator.cpp:26-29), so
// siblings share it. One cached (parent node, ray) pair covers a pre-order walk of groups
// of siblings; identity chains need no cache (world ray + 0.0f).
struct NodeRayCache {
  int32_t node = -2;
  f3 o, d;
};
YART_HD void nodeObjectRay(const SceneDev& sc, uint32_t idx, const NodeDev& nd, f3 wo, f3 wd,
                           NodeRayCache& cache, f3& oo, f3& od) {
  if (nd.pad[0] & 1u) { oo = wo + 0.0f; od = wd + 0.0f; return; }
  f3 po, pd;
  if (nd.pad[0] & 2u) { po = wo + 0.0f; pd = wd + 0.0f; }
  else if (nd.parent == cache.node) { po = cache.o; pd = cache.d; }
  else {
    objectRay(sc, uint32_t(nd.parent), wo, wd, po, pd);
    cache.node = nd.parent; cache.o = po; cache.d = pd;
  }
  (void)idx;
  oo = mulPoint(nd.xf.inv, po);
  od = mulVector(nd.xf.inv, pd);
}

// testNode (ray-integrator.cpp:20-54) as a pre-order walk. hit.t carries tMax in.
template <bool NEE, int MODE = TRAV_GENERAL>
YART_HD bool traverseScene(const SceneDev& sc, f3 o, f3 d, float tMin, HitRec& hit, f3& attenuation,
                           const TravStack& stk, AlphaCtx& actx) {
  bool didHit = false;
  uint32_t i = 0;
  YART_COUNT(nTrav, 1);
  NodeRayCache cache;
  RayO ray;
  bool rayIsWorld = false;       // `ray` holds makeRay(o + 0, d + 0): shared by every identity-chain node
  while (i < sc.nNodes) {
    const NodeDev& nd = sc.nodes[i];
    if ((MODE & TRAV_IDENTITY) || (nd.pad[0] & 1u)) {
      if (!rayIsWorld) { ray = makeRay(o + 0.0f, d + 0.0f); rayIsWorld = true; }
    } else if (!(MODE & TRAV_IDENTITY)) {
      // transformed node: conservative cull in world space first (host_scene.hpp: padded world box,
      // here a padded [0, hit.t] interval) — no 4x4 products, no slab set-up for the instances a ray
      // passes by; the reference's own object-space test follows for the survivors
      if (!rayIsWorld) { ray = makeRay(o + 0.0f, d + 0.0f); rayIsWorld = true; }
      const f4 wlo = sc.nodeWorld[2u * i], whi = sc.nodeWorld[2u * i + 1u];
      const float wmin[3] = {wlo.x, wlo.y, wlo.z}, wmax[3] = {whi.x, whi.y, whi.z};
      float dw;
      YART_COUNT(nBox, 1);
      if (!testBox(ray, 0.0f, hit.t + (fabsf(hit.t) * 1e-4f + 1e-3f), wmin, wmax, dw)) { i = nd.skip; continue; }
      f3 oo, od;
      nodeObjectRay(sc, i, nd, o, d, cache, oo, od);
      ray = makeRay(oo, od);
      rayIsWorld = false;
    }
    float dd;
    YART_COUNT(nBox, 1);
    if (!testBox(ray, tMin, hit.t, nd.bmin, nd.bmax, dd) || hit.t < dd) { i = nd.skip; continue; }
    if (nd.mesh >= 0) {
      const MeshDev& mesh = sc.meshes[nd.mesh];
      if (!(NEE && (MODE & TRAV_FAST) && didHit && !mesh.hasAlpha)) {      // pruning, see traverseMesh
        didHit |= traverseMesh<NEE, MODE>(sc, mesh, i, ray, tMin, hit, attenuation, stk, actx, didHit);
        if ((MODE & TRAV_FAST) && actx.deferred) return false;
      }
    }
    i++;
  }
  return didHit;
}

// testTriangle's hit fields + testMesh (ray-integrator.cpp:56-82) + the object->world
// chain of testNode (:50-52) for the final hit.
// the triangle's index within its mesh (what the reference's Hit reports), from the scene-wide one of the hit record
YART_HD uint32_t localTri(const SceneDev& sc, const HitRec& r) { return r.tri - sc.meshes[sc.nodes[r.node].mesh].triOffset; }
YART_HD Hit finalizeHit(const SceneDev& sc, const HitRec& r, f3 o, f3 d) {
  f3 oo, od;
  objectRay(sc, r.node, o, d, oo, od);
  Hit h;
  h.t = r.t;
  f3 n;
  interpUVN(sc, r.tri, r.u, r.v, h.uv, n);
  h.p = oo + (r.t * od);                                   // ray(t), ray.hpp:27-29
  h.backSide = (r.backSide & 1u) != 0;
  const ShadeTri& st = sc.shadeTris[r.tri];        // (HitRec::tri is scene-wide: no node -> mesh -> offset chain in front of this gather)
  // the material index travels with the hit (every traversal variant packs it above the back-side bit): the material
  // record is fetched without waiting for the triangle's shade record
  h.material = r.backSide >> 1;
  const MaterialDev& mt = sc.materials[h.material];
  // testMesh: tangents with barycentrics (w,u,v), normal map, tangent rebuilt from n x Y
  const float w = 1.0f - r.u - r.v;
  f4 tg;
  tg.x = (w * st.t[0][0] + r.u * st.t[1][0]) + r.v * st.t[2][0];
  tg.y = (w * st.t[0][1] + r.u * st.t[1][1]) + r.v * st.t[2][1];
  tg.z = (w * st.t[0][2] + r.u * st.t[1][2]) + r.v * st.t[2][2];
  tg.w = (w * st.t[0][3] + r.u * st.t[1][3]) + r.v * st.t[2][3];
  n = bsdfNormal(sc, mt, n, tg, h.uv);
  f3 tang;
  if (absDot(n, mk3(0, 1, 0)) > 0.999f) tang = mk3(1, 0, 0);
  else tang = normalized(cross(n, mk3(0, 1, 0)));
  h.lightIdx = st.light;
  // up the node chain: p as Point, n as Normal (renormalised per level), tg as Vector
  f3 p = h.p;
  for (int32_t i = int32_t(r.node); i >= 0; i = sc.nodes[i].parent) {
    const NodeDev& a = sc.nodes[i];
    if (a.pad[0] & 1u) {
      // identity from here to the root: p, tg map to x + 0.0f; the normal is still
      // renormalised once per level (transform.hpp:71), which is not idempotent
      p = p + 0.0f; tang = tang + 0.0f;
      for (uint32_t k = 0; k <= a.depth; k++) n = normalized(n + 0.0f);
      break;
    }
    p = mulPoint(a.xf.fwd, p);
    n = mulNormalT(a.xf.inv, n);       // m_normalTransform = transpose(float3x3(inverse))
    tang = mulVector(a.xf.fwd, tang);
  }
  h.p = p; h.n = n; h.tg = tang;
  return h;
}

}  // namespace yart_hip
