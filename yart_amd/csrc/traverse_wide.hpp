// traverse_wide.hpp — the pieces of the lean kernels' walk of their own 8-wide trees (bvh8_build.hpp): the node test,
// the slot-order tables, the triangle acceptance rules. Used by trace_lean_wide.inc (the SIMT loop) and by the scalar
// walk at the end of this file (tests/hostsim: the same rules on the CPU, against the reference-order walk).
//
// What makes a result of this walk the reference's (cpu/ray-integrator.cpp:84-229), although tree and order differ:
//  * the triangle test is the reference's Möller-Trumbore, operation for operation (:163-229);
//  * closest hit: among the triangles of a mesh that pass it the smallest t wins in any order, EXCEPT two candidates at
//    exactly the same t (the reference keeps the first in ITS order: hit.t <= t rejects) -> hand-over;
//  * the reference only finds a triangle whose leaf (and its ancestors) pass its own box test (:231-261) before hit.t
//    drops below their entry distance. Boxes nest and the reference's slab arithmetic (mul, add, max/min folds) is
//    monotone in the box, so "the triangle's reference LEAF passes with [tMin, t] and entry < t" implies that every
//    ancestor passed: checked for every accepted candidate with the reference's own arithmetic, else hand-over;
//  * this walk is conservative (bvh8_build.hpp: grid boxes contain the reference's boxes plus the arithmetic slack, cull
//    interval widened by kWideCullRel / kWideCullAbs): a triangle the reference tests and this walk's current hit does
//    not already beat is tested here;
//  * alpha-tested triangles draw a sampler dimension in the reference when they pass the test inside the current
//    interval, in its order; NEE-transparent ones multiply an attenuation in its order. Both live in a second tree (A):
//    a ray that crosses one of them inside (tMin, hit.t at mesh entry) is handed over before the opaque tree (O) is
//    walked (NEE: a transparent crossing only matters if the ray turns out unoccluded);
//  * rays whose slab set-up is not finite (a zero direction component: the reference's 0 * inf NaNs, :231-261), or
//    outside the origin bound the trees were padded for (MeshDev::wideRo), or with a NaN candidate t -> hand-over.
// A handed-over ray is traced from the root by the general kernel on the reference's tree: results are per ray.
#pragma once
#include "traverse.hpp"

namespace yart_hip {

constexpr float kWideCullRel = 1.001f, kWideCullAbs = 0.001f;   // a child is culled when its entry lies beyond hit.t * rel + abs
constexpr float kWideIdirMax = 0x1p60f;

struct WideSetup {            // per ray and mesh
  f3 o, idir;
  uint32_t shX, shY, shZ;     // 16 where the direction component is negative, else 0: a child's plane word of an axis (low plane | high
                              // plane << 16) rotated right by it has the ray's NEAR plane in its low half and the FAR plane in its high half
  uint32_t octinv;            // 7 ^ (sign bits of the direction): slot ^ octinv = traversal priority (7 first)
};
// false: the ray must be handed over (see above)
YART_HD bool wideSetup(const RayO& ray, float ro, WideSetup& w) {
  w.o = ray.o; w.idir = ray.idir;
  w.shX = ray.sx ? 16u : 0u; w.shY = ray.sy ? 16u : 0u; w.shZ = ray.sz ? 16u : 0u;
  w.octinv = 7u ^ (ray.sx | (ray.sy << 1) | (ray.sz << 2));
  // (comparisons are false for NaN: a NaN anywhere fails)
  return fabsf(ray.idir.x) < kWideIdirMax && fabsf(ray.idir.y) < kWideIdirMax && fabsf(ray.idir.z) < kWideIdirMax &&
         fabsf(ray.o.x) <= ro && fabsf(ray.o.y) <= ro && fabsf(ray.o.z) <= ro;
}

// hit byte (slot order) -> triangle bits 3 * slot .. 3 * slot + 2; inner hits (slot order) -> priority order (bit = slot ^ octinv)
YART_HD uint32_t wideSpread(uint32_t b) {
  uint32_t r = 0;
  for (uint32_t s = 0; s < 8; s++) if ((b >> s) & 1u) r |= 7u << (3u * s);
  return r;
}
YART_HD uint32_t widePerm(uint32_t octinv, uint32_t b) {
  uint32_t r = 0;
  for (uint32_t s = 0; s < 8; s++) if ((b >> s) & 1u) r |= 1u << (s ^ octinv);
  return r;
}

// fma(float(half k of w), a, b) with one rounding
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float widePlaneLo(uint32_t w, float a, float b) {
  float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ float widePlaneHi(uint32_t w, float a, float b) {
  float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ uint32_t wideRotr(uint32_t w, uint32_t sh) { return __builtin_amdgcn_alignbit(w, w, sh); }
// max / min of four (the operands come out of inline assembly: fmaxf would first canonicalise each of them with a v_max_f32 x, x;
// no NaN reaches these: wideSetup)
__device__ __forceinline__ float wideMax4(float a, float b, float c, float d) {
  float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(d)); return r;
}
__device__ __forceinline__ float wideMin4(float a, float b, float c, float d) {
  float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(d)); return r;
}
#else
inline float wideHalfValue(uint32_t h) {                     // zero and normal halves (the planes are integers 0..2047)
  const uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
  if (e == 0u) return 0.0f;
  return __builtin_bit_cast(float, ((e + 112u) << 23) | (m << 13));
}
inline float widePlaneLo(uint32_t w, float a, float b) { return __builtin_fmaf(wideHalfValue(w & 0xffffu), a, b); }
inline float widePlaneHi(uint32_t w, float a, float b) { return __builtin_fmaf(wideHalfValue(w >> 16), a, b); }
inline uint32_t wideRotr(uint32_t w, uint32_t sh) { return sh ? ((w >> sh) | (w << (32u - sh))) : w; }
inline float wideMax4(float a, float b, float c, float d) { return fmaxf(fmaxf(fmaxf(a, b), c), d); }
inline float wideMin4(float a, float b, float c, float d) { return fminf(fminf(fminf(a, b), c), d); }
#endif

#if defined(__HIPCC__)
// min over the eight lanes of a group (lanes 8g .. 8g + 7, all of them active), returned in all of them: two quad permutes and a
// mirror of the half row, each fused into its v_min_f32 (DPP): no LDS, no permute unit
__device__ __forceinline__ float coopMin8(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1, 0, 3, 2]
  v = fminf(v, a);
  a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));         // quad_perm [2, 3, 0, 1]
  v = fminf(v, a);
  a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));        // row_half_mirror
  v = fminf(v, a);
#endif
  return v;
}
#endif

// Per node and ray: the grid step and origin folded into the slab line t = q * s + b of each axis.
struct WideNodeRay { float sx, sy, sz, bx, by, bz; };
YART_HD WideNodeRay wideNodeRay(const u4& h0, const WideSetup& r) {
  WideNodeRay n;
  // one grid step for the three axes (the largest extent decides it: bvh8_build.hpp)
  const float step = __builtin_bit_cast(float, (h0.w & 0xffu) << 23);
  n.sx = r.idir.x * step; n.sy = r.idir.y * step; n.sz = r.idir.z * step;
  n.bx = (__builtin_bit_cast(float, h0.x) - r.o.x) * r.idir.x;
  n.by = (__builtin_bit_cast(float, h0.y) - r.o.y) * r.idir.y;
  n.bz = (__builtin_bit_cast(float, h0.z) - r.o.z) * r.idir.z;
  return n;
}
// One child box (its three plane words) against the ray's interval [tMin, tMax]: true = missed. The ONE place the slab arithmetic of
// the 8-wide trees is written: the cooperative kernel runs it on one lane per child, the scalar walk below in a loop.
YART_HD bool wideChildMissed(uint32_t wx, uint32_t wy, uint32_t wz, const WideSetup& r, const WideNodeRay& n, float tMin, float tMax) {
  const uint32_t rx = wideRotr(wx, r.shX), ry = wideRotr(wy, r.shY), rz = wideRotr(wz, r.shZ);
  const float tnx = widePlaneLo(rx, n.sx, n.bx), tfx = widePlaneHi(rx, n.sx, n.bx);
  const float tny = widePlaneLo(ry, n.sy, n.by), tfy = widePlaneHi(ry, n.sy, n.by);
  const float tnz = widePlaneLo(rz, n.sz, n.bz), tfz = widePlaneHi(rz, n.sz, n.bz);
  const float lo = wideMax4(tnx, tny, tnz, tMin);
  const float hi = wideMin4(tfx, tfy, tfz, tMax);
  return (__builtin_bit_cast(uint32_t, hi - lo) >> 31) != 0u;    // sign bit of hi - lo
}

struct WideNodeHits {
  uint32_t hitByte;          // slots whose box the ray's interval meets
  uint32_t imask, childBase, triBase, triValid;
};
// (base: uniform pointer, off: 32-bit byte offset per lane — global_load with a scalar base and a 32-bit vector offset)
YART_HD u4 wideLd16(const uint8_t* base, uint32_t off) { return *reinterpret_cast<const u4*>(base + off); }
YART_HD uint32_t wideLd4(const uint8_t* base, uint32_t off) { return *reinterpret_cast<const uint32_t*>(base + off); }
constexpr uint32_t kWideChildOffset = 32u, kWideChildBytes = 12u;   // Wide8Node: header, then 12 bytes per slot

// One node against one ray, slot by slot (the scalar form). tMax: the cull bound (already widened by the caller).
YART_HD WideNodeHits wideTestNode(const uint8_t* nodes, uint32_t nidx, const WideSetup& r, float tMin, float tMax) {
  const uint32_t nb = nidx << 7;                                 // (fewer than 2^24 nodes: host_scene.hpp)
  const u4 h0 = wideLd16(nodes, nb), h1 = wideLd16(nodes, nb + 16u);
  const WideNodeRay n = wideNodeRay(h0, r);
  WideNodeHits o;
  o.hitByte = 0;
  for (uint32_t s = 0; s < 8u; s++) {
    const uint32_t cb = nb + kWideChildOffset + kWideChildBytes * s;
    if (!wideChildMissed(wideLd4(nodes, cb), wideLd4(nodes, cb + 4u), wideLd4(nodes, cb + 8u), r, n, tMin, tMax)) o.hitByte |= 1u << s;
  }
  o.imask = h0.w >> 24; o.childBase = h1.x; o.triBase = h1.y; o.triValid = h1.z;
  return o;
}

// the reference's box test of the reference's leaf that holds an accepted candidate (see the head of this file)
YART_HD bool wideLeafCheck(const SceneDev& sc, const RayO& ray, float tMin, float t, uint32_t matFlags) {
  const BvhNode lb = sc.bvhNodes[matFlags >> kWideRefLeafShift];
  float d0;
  return testBox(ray, tMin, t, lb.bmin, lb.bmax, d0) && d0 < t;
}

enum : uint32_t { WIDE_REJECT = 0, WIDE_ACCEPT = 1, WIDE_TRANSPARENT = 2,
                  WIDE_HANDOVER = 4, WIDE_HAND_ALPHA = 4, WIDE_HAND_TIE = 5, WIDE_HAND_CHECK = 6, WIDE_HAND_NAN = 7 };   // >= WIDE_HANDOVER: hand the ray over

// The reference's Moeller-Trumbore test (cpu/ray-integrator.cpp:163-196), operation for operation, up to the point where it has t:
// 0 = rejected by the determinant or the barycentrics, 1 = t / u / v / det are the reference's values, 2 = t is NaN (the reference
// accepts a NaN t: both of its interval comparisons are false).
struct WideCand { float t, u, v, det; };
YART_HD uint32_t wideTriTest(const LeafTri& tr, const f3& o, const f3& d, WideCand& c) {
  const f3 p0 = mk3(tr.p0[0], tr.p0[1], tr.p0[2]);
  const f3 edge1 = mk3(tr.e1[0], tr.e1[1], tr.e1[2]);
  const f3 edge2 = mk3(tr.e2[0], tr.e2[1], tr.e2[2]);
  const f3 rayEdge2 = cross(d, edge2);
  const float det = dot(edge1, rayEdge2);
  if (double(fabsf(det)) < 1e-12) return 0u;
  const float invDet = 1.0f / det;
  const f3 b = o - p0;
  const float u = dot(b, rayEdge2) * invDet;
  if (u < 0.0f || u > 1.0f) return 0u;
  const f3 bEdge1 = cross(b, edge1);
  const float v = dot(d, bEdge1) * invDet;
  if (v < 0.0f || u + v > 1.0f) return 0u;
  const float t = dot(edge2, bEdge1) * invDet;
  c.t = t; c.u = u; c.v = v; c.det = det;
  return t == t ? 1u : 2u;
}

// One triangle of a tree O (no alpha-tested triangles) / of a tree A (alpha-tested and NEE-transparent ones only), one after the
// other (the scalar walk; the cooperative kernel applies the same rules to the candidates of a round together).
// O: closest hit -> hit updated on WIDE_ACCEPT; NEE -> WIDE_ACCEPT = occluded (hit untouched), transparent triangles are no occluders.
// A: WIDE_HAND_ALPHA for a crossing that matters, WIDE_TRANSPARENT for a transparent crossing of a shadow ray, else WIDE_REJECT.
template <bool NEE, bool TREE_A>
YART_HD uint32_t wideTriangle(const SceneDev& sc, const LeafTri& tr, const RayO& ray, float tMin, HitRec& hit, bool meshDidHit, uint32_t nodeI) {
  if (!TREE_A && NEE && (tr.matFlags & MAT_TRANSPARENT)) return WIDE_REJECT;
  if (TREE_A && !NEE && !(tr.matFlags & MAT_HAS_ALPHA)) return WIDE_REJECT;     // (transparent: an ordinary surface for a closest-hit ray, found in O)
  WideCand c;
  const uint32_t r = wideTriTest(tr, ray.o, ray.d, c);
  if (r == 0u) return WIDE_REJECT;
  if (r == 2u) return WIDE_HAND_NAN;
  const float t = c.t;
  if (TREE_A) {
    if (t <= tMin || hit.t <= t) return WIDE_REJECT;
    if (tr.matFlags & MAT_HAS_ALPHA) return WIDE_HAND_ALPHA;
    return WIDE_TRANSPARENT;
  }
  if (t <= tMin || hit.t < t) return WIDE_REJECT;
  if (hit.t == t) return (!NEE && meshDidHit) ? WIDE_HAND_TIE : WIDE_REJECT;   // a tie within the mesh: the reference's order decides
  if (!wideLeafCheck(sc, ray, tMin, t, tr.matFlags)) return WIDE_HAND_CHECK;
  if (!NEE) {
    hit.t = t; hit.u = c.u; hit.v = c.v; hit.tri = tr.triIdx; hit.node = nodeI;
    hit.backSide = (c.det < 0 ? 1u : 0u) | (tr.material << 1);
  }
  return WIDE_ACCEPT;
}

// ---------------------------------------------------------------------------------------------------------------
// Scalar form of the walk (one ray, one mesh): what trace_lean_wide.inc does per lane. tests/hostsim runs it on the CPU
// against traverseMesh, the reference-order walk.
struct WideWalkStats { uint64_t nodes = 0, tris = 0, handAlpha = 0, handTie = 0, handCheck = 0, handGuard = 0; };
// returns false: hand the ray over. occludedBefore (NEE): an earlier mesh already occluded the ray (only alpha crossings matter).
template <bool NEE>
YART_HD bool wideTraverseMesh(const SceneDev& sc, const MeshDev& mesh, uint32_t nodeI, const RayO& ray, float tMin, HitRec& hit,
                              bool& meshDidHit, bool occludedBefore, bool& crossedTransparent, WideWalkStats* st) {
  WideSetup ws;
  if (!wideSetup(ray, mesh.wideRo, ws)) { if (st) st->handGuard++; return false; }
  meshDidHit = false;
  uint64_t stack[64];
  for (int phase = mesh.wideRootA != kNoWide ? 0 : 1; phase < 2; phase++) {
    const uint32_t root = phase == 0 ? mesh.wideRootA : mesh.wideRootO;
    if (root == kNoWide) continue;
    if (phase == 1 && NEE && occludedBefore) break;
    uint32_t sp = 0;
    // the root as a group of one: slot = 7 ^ octinv, imask all ones -> node index = base + slot
    const uint32_t slot0 = 7u ^ ws.octinv;
    uint32_t gBase = root - slot0, gBits = 0x800000ffu, tBase = 0, tBits = 0, tValid = 0;
    for (;;) {
      if (tBits == 0u) {
        if ((gBits >> 24) == 0u) {
          if (sp == 0u) break;
          const uint64_t e = stack[--sp];
          gBase = uint32_t(e); gBits = uint32_t(e >> 32);
        }
        const uint32_t pbit = 31u - uint32_t(__builtin_clz(gBits));
        gBits &= ~(1u << pbit);
        const uint32_t slot = (pbit - 24u) ^ ws.octinv;
        const uint32_t nidx = gBase + uint32_t(__builtin_popcount(gBits & 0xffu & ((1u << slot) - 1u)));
        if (gBits >> 24) stack[sp++] = uint64_t(gBase) | (uint64_t(gBits) << 32);
        // closest hit: cull beyond the current hit (widened); NEE and tree A: the interval is fixed
        const float tCull = (NEE || phase == 0) ? hit.t : hit.t * kWideCullRel + kWideCullAbs;
        const WideNodeHits h = wideTestNode(sc.wideNodes, nidx, ws, tMin, tCull);
        if (st) st->nodes++;
        gBase = h.childBase; gBits = (widePerm(ws.octinv, h.hitByte & h.imask) << 24) | h.imask;
        tBase = h.triBase; tValid = h.triValid; tBits = wideSpread(h.hitByte) & h.triValid;
      }
      while (tBits) {
        const uint32_t bit = uint32_t(__builtin_ctz(tBits));
        tBits &= tBits - 1u;
        const LeafTri tr = sc.wideTris[tBase + uint32_t(__builtin_popcount(tValid & ((1u << bit) - 1u)))];
        if (st) st->tris++;
        uint32_t r;
        if (phase == 0) r = wideTriangle<NEE, true>(sc, tr, ray, tMin, hit, meshDidHit, nodeI);
        else r = wideTriangle<NEE, false>(sc, tr, ray, tMin, hit, meshDidHit, nodeI);
        if (r >= WIDE_HANDOVER) {
          if (st) { if (r == WIDE_HAND_ALPHA) st->handAlpha++; else if (r == WIDE_HAND_TIE) st->handTie++; else st->handCheck++; }
          return false;
        }
        if (r == WIDE_TRANSPARENT) { if (!occludedBefore) crossedTransparent = true; }
        if (r == WIDE_ACCEPT) {
          meshDidHit = true;
          if (NEE) return true;                              // occluded: nothing in this mesh can change the result (tree A came first)
        }
      }
    }
  }
  return true;
}

}  // namespace yart_hip
