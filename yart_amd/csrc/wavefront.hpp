// wavefront.hpp — the MIS + NEE path loop of integrator.hpp split at its two trace
// calls into queue-driven stages (reference cpu/mis-integrator.cpp:13-148):
//
//   generate  camera ray, path state                       (ray-integrator.cpp:11-18)
//   extend    closest-hit traversal of every live path     (testNode, :20-54)
//   shade     miss / emission / BSDF sample / NEE set-up / throughput update; Russian roulette of the paths that cast no
//             shadow ray (mis-integrator.cpp:27-102, 111-124)
//   shadow    any-hit traversal of the shadow rays (:135-148); the NEE contribution (:125-133) is added to the path's
//             radiance where the traversal result is committed (wfShadowCommit) — no pass of its own
//   roulette  for the paths that cast a shadow ray: the Russian roulette + next-bounce decision (:96-102) AFTER the shadow
//             traversal, which may consume sampler dimensions first. Needed from the second bounce on only (the roulette
//             starts at depth 2): 16-48 bytes per path instead of the 120 a full post pass moved
//
// Path state lives in HBM as float4-packed SoA arrays indexed by path slot, so that a
// wave's 64 lanes read 1 KiB contiguous per field group. Every arithmetic expression is
// the one integrator.hpp uses; only the control flow differs, so results are identical.
#pragma once
#include "integrator.hpp"

namespace yart_hip {

struct WfState {
  // ray0 = {o.xyz, d.x}  ray1 = {d.yz, lastPdf, accRoughness}
  // thr = {att.xyz, dim(u32)}   acc = {L.xyz, flags(u32)}
  // hit0 = {t, u, v, tri(u32)}  hit1 = {hit word (wfHitWord), morton.lo, morton.hi, sampler-table column (u32)}
  // sh0 = {to.xyz, -}  sh1 = {attPre.xyz, denom}  sh2 = {Lif.xyz, cosTerm}   (read where a shadow ray's result is committed)
  f4 *ray0, *ray1, *thr, *acc, *hit0, *hit1, *sh0, *sh1, *sh2;
};
// exact test counters of the instrumented build (libyart_hip_count.so); empty otherwise
struct WfTally {
#if defined(YART_COUNT_TRAVERSAL)
  uint32_t box = 0, tri = 0, trav = 0, shade = 0;
  uint32_t resumed = 0;        // handed-over rays the general kernels took up where the lean kernel stood
  uint32_t waste = 0;          // box tests the lean kernels spent on rays they then handed to the general kernels
#endif
#if defined(YART_TRACE_STATS)
  // debug build: wave-iteration / active-lane counts per phase of the wave tracer
  // (pairs: [2k] wave iterations, [2k+1] lanes active in them)
  uint32_t ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
};
#if defined(YART_TRACE_STATS)
// to be executed by every active lane of a divergent region: the lowest active lane counts
// the wave iteration, every active lane counts itself
#define WF_PHASE(t, k)                                                                  \
  do {                                                                                  \
    const unsigned long long _m = __ballot(true);                                       \
    (t).ph[2 * (k)] += (int(threadIdx.x & 63u) == __ffsll((long long) _m) - 1) ? 1u : 0u; \
    (t).ph[2 * (k) + 1] += 1u;                                                          \
  } while (0)
#else
#define WF_PHASE(t, k) ((void)0)
#endif
#if defined(YART_COUNT_TRAVERSAL)
#define WF_TALLY_TRAV(t, ac) ((t).box += (ac).nBox, (t).tri += (ac).nTri, (t).trav += (ac).nTrav)
#define WF_TALLY_SHADE(t) ((t).shade++)
#else
#define WF_TALLY_TRAV(t, ac) ((void)0)
#define WF_TALLY_SHADE(t) ((void)0)
#endif

// Debug build -DYART_SHADE_REGIONS=1 (tools/shade_regions.py): where a wave of k_wf_shade spends its cycles. SR_MARK(k) at
// the END of a code region charges the wave's cycles since its previous mark to region k (first active lane, accumulators
// in LDS, so no registers are taken from the kernel) and counts the lanes that ran it. Empty otherwise.
#if defined(YART_SHADE_REGIONS) && defined(__HIPCC__)
constexpr int kShadeRegions = 16;
static __device__ unsigned long long g_shadeRegion[3 * kShadeRegions];   // [k] cycles, [16 + k] visits, [32 + k] lanes
#endif
#if defined(YART_SHADE_REGIONS) && defined(__HIP_DEVICE_COMPILE__)
__shared__ unsigned long long srAcc[16][3 * kShadeRegions];    // (one row per wave of the workgroup: the shade kernel runs 12)
__shared__ unsigned long long srLast[16];
#define SR_MARK(k)                                                                       \
  do {                                                                                   \
    const unsigned long long _m = __ballot(true);                                        \
    if (int(threadIdx.x & 63u) == __ffsll((long long) _m) - 1) {                         \
      const unsigned long long _t = clock64();                                           \
      unsigned long long* _a = srAcc[threadIdx.x >> 6];                                  \
      _a[k] += _t - srLast[threadIdx.x >> 6]; _a[kShadeRegions + (k)] += 1ull;           \
      _a[2 * kShadeRegions + (k)] += (unsigned long long) __popcll(_m);                  \
      srLast[threadIdx.x >> 6] = clock64();                                              \
    }                                                                                    \
  } while (0)
#else
#define SR_MARK(k) ((void)0)
#endif

enum : uint32_t { WF_SPECULAR = 1u << 8, WF_REGULARIZED = 1u << 9, WF_MISS = 1u << 10, WF_DEPTH_MASK = 0xffu,
                  // bits 16..23: unoccluded NEE rays of this path so far (mis-integrator.cpp:126 counts them as rays): with the
                  // depth they give the path's ray count when it ends (Renderer::TileData.rays, renderer.hpp:40-50)
                  WF_NEE_SHIFT = 16, WF_NEE_ONE = 1u << 16, WF_NEE_MASK = 0xffu << 16,
                  // set by shade for a path with a shadow ray in flight: FINAL = the bounce budget is used up, whoever commits the
                  // shadow ray's result writes the path's radiance out; ATT_NAN = the throughput before the bounce is not finite,
                  // so that even an occluded light sample changes the radiance (L += attenuation * 0, mis-integrator.cpp:80)
                  WF_FINAL = 1u << 11, WF_ATT_NAN = 1u << 12 };

// hit word of the path state: scene node (bits 0..19) | shade class = material index, or kWfClassMiss
// (bits 20..30) | back side (bit 31). The class is what k_wf_shade buckets its waves by.
constexpr uint32_t kWfNodeBits = 20, kWfClassBits = 11, kWfClassMiss = (1u << kWfClassBits) - 1u;
YART_HD uint32_t wfHitWord(const HitRec& h, bool didHit) {
  if (!didHit) return kWfClassMiss << kWfNodeBits;
  return h.node | ((h.backSide >> 1) << kWfNodeBits) | ((h.backSide & 1u) << 31);
}
YART_HD float asF(uint32_t u) { return __builtin_bit_cast(float, u); }
YART_HD uint32_t asU(float f) { return __builtin_bit_cast(uint32_t, f); }
YART_HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

// Streaming access to the path state: every word is read / written once per kernel, so it is marked
// non-temporal to keep it from displacing BVH nodes, leaf records and textures in L2.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(YART_NO_NT_STATE)
// The state pointers reach the kernels through a struct argument and, with compaction, out of memory (WF_DYN): the compiler does not know
// what they point to and would use flat loads / stores, which count on the vector-memory AND the LDS counter (a wait for either waits for
// both). Everything these three helpers touch — path state, per-sample radiance, resume records — is global memory, and they say so.
typedef float wf_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) wf_v4 wf_v4_global;
typedef __attribute__((address_space(1))) float wf_f_global;
__device__ __forceinline__ f4 wfLd(const f4* p) {
  const wf_v4 v = __builtin_nontemporal_load((const wf_v4_global*) reinterpret_cast<const wf_v4*>(p));
  return mk4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void wfSt(f4* p, f4 v) {
  wf_v4 q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
  __builtin_nontemporal_store(q, (wf_v4_global*) reinterpret_cast<wf_v4*>(p));
}
__device__ __forceinline__ void wfSt1(float* p, float v) { __builtin_nontemporal_store(v, (wf_f_global*) p); }
__device__ __forceinline__ float wfLd1(const float* p) { return *(const wf_f_global*) p; }     // (a word that is read again soon: cached)
#else
YART_HD f4 wfLd(const f4* p) { return *p; }
YART_HD void wfSt(f4* p, f4 v) { *p = v; }
YART_HD void wfSt1(float* p, float v) { *p = v; }
YART_HD float wfLd1(const float* p) { return *p; }
#endif

struct WfPath {             // register image of one path
  f3 o, d, att, L;
  float lastPdf, accRoughness;
  uint32_t flags, slot;
  Sampler smp;
};
YART_HD WfPath wfLoad(const WfState& s, uint32_t i) {
  const f4 r0 = wfLd(s.ray0 + i), r1 = wfLd(s.ray1 + i), t = wfLd(s.thr + i), a = wfLd(s.acc + i), h1 = wfLd(s.hit1 + i);
  WfPath p;
  p.o = mk3(r0.x, r0.y, r0.z); p.d = mk3(r0.w, r1.x, r1.y); p.lastPdf = r1.z; p.accRoughness = r1.w;
  p.att = mk3(t.x, t.y, t.z); p.smp.dim = asU(t.w); p.L = mk3(a.x, a.y, a.z); p.flags = asU(a.w);
  p.smp.morton = uint64_t(asU(h1.y)) | (uint64_t(asU(h1.z)) << 32); p.smp.pix = asU(h1.w); p.slot = i;
  return p;
}
YART_HD void wfStoreRay(const WfState& s, uint32_t i, const WfPath& p) {
  wfSt(s.ray0 + i, mk4(p.o.x, p.o.y, p.o.z, p.d.x));
  wfSt(s.ray1 + i, mk4(p.d.y, p.d.z, p.lastPdf, p.accRoughness));
}
YART_HD void wfStoreThr(const WfState& s, uint32_t i, const WfPath& p) {
  wfSt(s.thr + i, mk4(p.att.x, p.att.y, p.att.z, asF(p.smp.dim)));
  wfSt(s.acc + i, mk4(p.L.x, p.L.y, p.L.z, asF(p.flags)));
}
YART_HD uint32_t wfPathRays(uint32_t flags) { return (flags & WF_DEPTH_MASK) + ((flags & WF_NEE_MASK) >> WF_NEE_SHIFT); }

// Russian roulette and loop condition (mis-integrator.cpp:96-102 + the while at :21).
// Returns true if the path continues to the next bounce.
YART_HD bool wfRoulette(const RenderConst& rc, WfPath& p) {
  const uint32_t depth = p.flags & WF_DEPTH_MASK;
  if (depth > 1 && maxComponent(p.att) < 1.0f) {
    float q = stdmax(0.0f, 1.0f - maxComponent(p.att));
    if (get1D(p.smp, rc.sampler) < q) return false;
    p.att /= 1.0f - q;
  }
  return depth < rc.maxDepth;
}

// generate: RayIntegrator::sample up to the first trace
YART_HD void wfGenerate(const RenderConst& rc, const uint32_t* sobol, const CameraDev& cam, uint32_t px,
                        uint32_t py, uint32_t sample, uint32_t pix, const WfState& s, uint32_t i) {
  WfPath p;
  startPixelSample(p.smp, rc.sampler, px, py, sample);
  p.smp.pix = pix;
  f2 uvFilm = get2D(p.smp, rc.sampler, sobol);
  f2 uvLens = get2D(p.smp, rc.sampler, sobol);
  cameraRay(cam, px, py, uvFilm, uvLens, p.o, p.d);
  p.att = mk3(1.0f); p.L = mk3(0.0f); p.lastPdf = 0.0f; p.accRoughness = 0.0f; p.flags = 0; p.slot = i;
  wfStoreRay(s, i, p);
  wfStoreThr(s, i, p);
  s.hit1[i] = mk4(0.0f, asF(uint32_t(p.smp.morton)), asF(uint32_t(p.smp.morton >> 32)), asF(pix));
}

// The Russian roulette + next-bounce decision of a path that cast a shadow ray (mis-integrator.cpp:96-102), after the shadow
// traversal (whose alpha tests may have drawn sampler dimensions: thr.w is current). `depth` = bounces done, uniform over
// the launch and >= 2 here. Returns true if the path continues; otherwise its radiance and ray count go to `out`.
// Where a finished path's radiance goes: record slotMap[slot] of `out` (no map: record `slot`). The map of the compacted tail
// states is only read. `pool`: the map is the path pool's (wavefront_kernels.inc: k_wf_refill) — the slots' path index —, and the
// slot is then marked free (kWfFreeSlot): the next round starts a new path in every free slot.
constexpr uint32_t kWfFreeSlot = 0xffffffffu;
YART_HD void wfRetire(f4* out, uint32_t* slotMap, uint32_t slot, f4 record, bool pool) {
  uint32_t idx = slot;
  if (slotMap) { idx = slotMap[slot]; if (pool) slotMap[slot] = kWfFreeSlot; }
  wfSt(out + idx, record);
}
YART_HD bool wfRouletteAfterShadow(const RenderConst& rc, const WfState& s, uint32_t slot, uint32_t depth, f4* out, uint32_t* slotMap, bool pool) {
  const f4 t = wfLd(s.thr + slot);
  WfPath p;
  p.att = mk3(t.x, t.y, t.z);
  if (!(maxComponent(p.att) < 1.0f)) return true;              // no draw (wfRoulette), and depth < maxDepth or the path were FINAL
  const f4 h1 = wfLd(s.hit1 + slot);
  p.smp.dim = asU(t.w); p.smp.morton = uint64_t(asU(h1.y)) | (uint64_t(asU(h1.z)) << 32); p.smp.pix = asU(h1.w);
  p.flags = depth;
  if (wfRoulette(rc, p)) {
    wfSt(s.thr + slot, mk4(p.att.x, p.att.y, p.att.z, asF(p.smp.dim)));
    return true;
  }
  const f4 a = wfLd(s.acc + slot);
  wfRetire(out, slotMap, slot, mk4(a.x, a.y, a.z, asF(wfPathRays(asU(a.w)))), pool);
  return false;
}


// What a finished shadow ray leaves behind (mis-integrator.cpp:125-133 + the caller's `L += attenuation * Ld`, :80):
// unoccluded -> L += attPre * (Lif * attOcc * cos / denom), one more ray; occluded -> L += attPre * 0, which only a non-finite
// throughput makes visible (WF_ATT_NAN); a path whose bounce budget is used up (WF_FINAL) is written out here.
// Returns 1 if the ray counts (unoccluded).
YART_HD uint32_t wfShadowCommit(const WfState& s, uint32_t slot, bool occluded, f3 attOcc, f4* out, uint32_t* slotMap, bool pool) {
  f4 a = wfLd(s.acc + slot);
  uint32_t flags = asU(a.w);
  if (occluded && !(flags & (WF_FINAL | WF_ATT_NAN))) return 0u;
  f3 L = mk3(a.x, a.y, a.z);
  if (!occluded) {
    const f4 s1 = wfLd(s.sh1 + slot), s2 = wfLd(s.sh2 + slot);
    L += mk3(s1.x, s1.y, s1.z) * (mk3(s2.x, s2.y, s2.z) * attOcc * s2.w / s1.w);
    flags += WF_NEE_ONE;
  } else if (flags & WF_ATT_NAN) {
    const f4 s1 = wfLd(s.sh1 + slot);
    L += mk3(s1.x, s1.y, s1.z) * mk3(0.0f);
  }
  if (flags & WF_FINAL) wfRetire(out, slotMap, slot, mk4(L.x, L.y, L.z, asF(wfPathRays(flags))), pool);
  else wfSt(s.acc + slot, mk4(L.x, L.y, L.z, asF(flags)));
  return occluded ? 0u : 1u;
}

// extend, general variant (one ray per lane): every triangle kind, any node transforms; only the sampler dimension can
// change (alpha tests)
YART_HD void wfExtend(const SceneDev& sc, const RenderConst& rc, const TravStack& stk, const WfState& s,
                      uint32_t i, WfTally& tally) {
  const f4 r0 = s.ray0[i], r1 = s.ray1[i];
  f4 h1 = s.hit1[i];
  Sampler smp;
  smp.dim = asU(s.thr[i].w);
  smp.morton = uint64_t(asU(h1.y)) | (uint64_t(asU(h1.z)) << 32); smp.pix = asU(h1.w);
  const uint32_t dim0 = smp.dim;
  HitRec hr;
  hr.t = kInf; hr.u = hr.v = 0; hr.tri = 0; hr.node = 0; hr.backSide = 0;
  f3 dummy = mk3(1.0f);
  AlphaCtx ac; ac.sampler = &smp; ac.cfg = rc.sampler;
  bool hit = traverseScene<false>(sc, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), 0.001f, hr, dummy, stk, ac);
  WF_TALLY_TRAV(tally, ac);
  s.hit0[i] = mk4(hit ? hr.t : -1.0f, hr.u, hr.v, asF(hr.tri));
  h1.x = asF(wfHitWord(hr, hit));
  s.hit1[i] = h1;
  if (smp.dim != dim0) s.thr[i].w = asF(smp.dim);
}

// extend, fast variant (traverse.hpp TRAV_FAST [| TRAV_IDENTITY]): no sampler state, no alpha
// code. Returns false when the ray met an alpha / transparent candidate: nothing has been
// written then and the caller queues the path for wfExtend.
template <int MODE>
YART_HD bool wfExtendFast(const SceneDev& sc, const TravStack& stk, const WfState& s, uint32_t i, WfTally& tally) {
  const f4 r0 = s.ray0[i], r1 = s.ray1[i];
  HitRec hr;
  hr.t = kInf; hr.u = hr.v = 0; hr.tri = 0; hr.node = 0; hr.backSide = 0;
  f3 dummy = mk3(1.0f);
  AlphaCtx ac; ac.sampler = nullptr; ac.cfg = SamplerConfig{};
  const bool hit = traverseScene<false, MODE>(sc, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), 0.001f, hr, dummy, stk, ac);
  WF_TALLY_TRAV(tally, ac);
  if (ac.deferred) return false;
  s.hit0[i] = mk4(hit ? hr.t : -1.0f, hr.u, hr.v, asF(hr.tri));
  s.hit1[i].x = asF(wfHitWord(hr, hit));
  return true;
}

// shadow ray of the path in slot i, one ray per lane (mis-integrator.cpp:137-146); the result is committed here
// (wfShadowCommit). Fast variant: same contract as wfExtendFast. `rays` counts the unoccluded ones.
template <int MODE>
YART_HD bool wfShadow(const SceneDev& sc, const RenderConst& rc, const TravStack& stk, const WfState& s,
                      uint32_t i, WfTally& tally, f4* out, uint32_t* slotMap, bool pool, uint32_t& rays) {
  const f4 r0 = s.ray0[i], s0 = s.sh0[i];
  const f3 from = mk3(r0.x, r0.y, r0.z), to = mk3(s0.x, s0.y, s0.z);
  const f3 dir = normalized(to - from);                     // :140
  HitRec hr;
  hr.t = length(to - from) - 0.001f;                        // :144
  hr.u = hr.v = 0; hr.tri = 0; hr.node = 0; hr.backSide = 0;
  f3 attOcc = mk3(1.0f);
  Sampler smp; smp.dim = 0; smp.morton = 0;
  uint32_t dim0 = 0;
  AlphaCtx ac; ac.sampler = &smp; ac.cfg = rc.sampler;
  if (!(MODE & TRAV_FAST)) {
    const f4 h1 = s.hit1[i];
    smp.dim = dim0 = asU(s.thr[i].w);
    smp.morton = uint64_t(asU(h1.y)) | (uint64_t(asU(h1.z)) << 32); smp.pix = asU(h1.w);
  }
  const bool occluded = traverseScene<true, MODE>(sc, from, dir, 0.001f, hr, attOcc, stk, ac);
  WF_TALLY_TRAV(tally, ac);
  if ((MODE & TRAV_FAST) && ac.deferred) return false;
  if (!(MODE & TRAV_FAST) && smp.dim != dim0) s.thr[i].w = asF(smp.dim);
  rays += wfShadowCommit(s, i, occluded, attOcc, out, slotMap, pool);
  return true;
}

enum WfShadeResult { WF_TERMINATED = 0, WF_CONTINUE = 1, WF_SHADOW = 2 };

// shade: everything between the two trace calls of one bounce. `rays` counts path segments.
YART_HD WfShadeResult wfShade(const SceneDev& sc, const RenderConst& rc, const uint32_t* sobol, const WfState& s,
                              uint32_t i, WfPath& p, uint32_t& rays, WfTally& tally) {
  p = wfLoad(s, i);
  const f4 h0 = wfLd(s.hit0 + i);
  const uint32_t nodeBack = asU(s.hit1[i].x);
  const uint32_t depth = p.flags & WF_DEPTH_MASK;
  const bool specularBounce = (p.flags & WF_SPECULAR) != 0, regularized = (p.flags & WF_REGULARIZED) != 0;
  rays++;
  SR_MARK(0);                                                   // path state loaded
  if (h0.x < 0.0f) {                                            // miss (mis-integrator.cpp:27-43)
    for (uint32_t k = 0; k < sc.nInfinite; k++) {
      const LightDev& l = sc.lights[sc.infiniteLights[k]];
      f3 Le = lightLe(sc, l, octahedralUV(p.d));
      if (depth == 0 || specularBounce) p.L += p.att * Le;
      else {
        float pdfLight = lightPdf(sc, l, p.d);
        float wBSDF = p.lastPdf / (p.lastPdf + pdfLight);
        p.L += p.att * wBSDF * Le;
      }
    }
    p.L += p.att * rc.background;
    SR_MARK(1);                                                 // miss: environment lookup + MIS
    return WF_TERMINATED;
  }
  HitRec hr;
  hr.t = h0.x; hr.u = h0.y; hr.v = h0.z; hr.tri = asU(h0.w); hr.node = nodeBack & ((1u << kWfNodeBits) - 1u);
  hr.backSide = (nodeBack >> 31) | (((nodeBack >> kWfNodeBits) & kWfClassMiss) << 1);
  Hit hit = finalizeHit(sc, hr, p.o, p.d);
  SR_MARK(2);                                                   // finalizeHit
  WF_TALLY_SHADE(tally);
  const MaterialDev& mt = sc.materials[hit.material];
  f2 u = get2D(p.smp, rc.sampler, sobol);
  float uc = get1D(p.smp, rc.sampler);
  // A draw is a pure function of (pixel, sample, dimension): the sampler's state is the dimension counter alone. The third
  // draw (uc2, parametric.cpp:226-251) only chooses between the clearcoat / metallic / dielectric lobes; it is evaluated
  // below, once the material is known, and only where it can decide anything.
  Sampler smpUc2 = p.smp;
  p.smp.dim++;
  SR_MARK(3);                                                   // sampler: one 2D + two 1D draws
  const f3 wo = -p.d;
  // one shading frame and one material fetch for sample / f / pdf (core/bsdf.cpp:5-58 builds the
  // same frame and fetches the same texels in each of the three calls)
  const Frame fr = shadingFrame(hit.n, hit.tg);
  const f3 woLocal = wtl(fr, wo);
  const MatEval me = matEvaluate(sc, mt, hit.uv);
  // c = m = t = 0 (no clearcoat, metal or transmission — the bulk of most scenes): the three selection probabilities are 0
  // and `uc2 < p` is false for every uc2 in [0, 1) (bsdfSampleImplE), so its ~170 instructions of index hashing are skipped
  float uc2 = 0.0f;
  if (!(me.c == 0.0f && me.m == 0.0f && me.t == 0.0f)) uc2 = get1D(smpUc2, rc.sampler);
  SR_MARK(4);                                                   // shading frame + material / texture fetch
  BsdfSample res = bsdfSampleImplE(sc, mt, me, woLocal, hit.uv, u, uc, uc2, regularized);
  res.wi = ltw(fr, res.wi);
  SR_MARK(5);                                                   // BSDF sample
  if (res.scatter & SC_EMITTED) {
    if (depth == 0 || specularBounce) p.L += p.att * res.Le;
    else if (hit.lightIdx != -1) {
      const LightDev& l = sc.lights[hit.lightIdx];
      float pdfLight = lightPdf(sc, l, wo) * length2(p.o - hit.p) * lightSamplerP(sc, uint32_t(hit.lightIdx)) /
                       absDot(wo, hit.n);                       // lastHit.p == origin of the current ray
      float wBSDF = p.lastPdf / (p.lastPdf + pdfLight);
      p.L += p.att * wBSDF * res.Le;
    }
  }
  if (!(res.scatter & (SC_REFLECTED | SC_TRANSMITTED))) return WF_TERMINATED;

  // Everything of the continuing path that does not depend on the light sample is computed — and the new ray
  // stored — before the NEE block, so that the BSDF sample does not have to wait in registers (or scratch) for
  // the end of it. The old throughput stays in p.att until then (NEE's MIS terms use it).
  const f3 fcos = res.f * absDot(res.wi, hit.n);
  f3 newAtt = p.att * (fcos / res.pdf);
  if (hit.backSide) newAtt *= matAttenuation(mt, hit.t);
  {
    WfPath q;
    q.o = hit.p; q.d = res.wi; q.lastPdf = res.pdf; q.accRoughness = p.accRoughness + res.roughness;
    wfStoreRay(s, i, q);
    uint32_t fl = ((depth + 1) & WF_DEPTH_MASK) | (p.flags & WF_NEE_MASK);
    if (res.scatter & SC_SPECULAR) fl |= WF_SPECULAR;
    if (q.accRoughness > 0.5f) fl |= WF_REGULARIZED;
    p.flags = fl;
  }
  SR_MARK(6);                                                   // emission MIS, throughput, new ray stored
  const bool nee = !(res.scatter & (SC_EMITTED | SC_SPECULAR));

  bool shadow = false;
  if (nee) {                                                    // L += attenuation * Ld(...)   (:79-80)
    if (sc.nLights != 0) {                                      // Ld set-up (:111-124)
      // the light-choice draw: with ONE infinite light and no area light PowerLightSampler::sample returns that light with
      // probability 1 whatever u is (light-sampler.cpp:52-78: pInfinite = 1, index min(0, .)) — the dimension is consumed, the
      // value is not evaluated
      float ucl = 0.0f;
      if (sc.nArea == 0u && sc.nInfinite == 1u) p.smp.dim++;
      else ucl = get1D(p.smp, rc.sampler);
      f2 ul = get2D(p.smp, rc.sampler, sobol);
      SR_MARK(7);                                               // sampler: NEE draws (1D + 2D)
      float pl;
      uint32_t li = lightSamplerSample(sc, ucl, pl);
      const LightDev& l = sc.lights[li];
      LightSample ls = lightSample(sc, l, hit.p, ul);
      const f3 wiLocal = wtl(fr, ls.wi);
      SR_MARK(8);                                               // light choice + light sample (environment importance sampling)
      f3 f = bsdfFImplE(sc, mt, me, woLocal, wiLocal);
      SR_MARK(9);                                               // BSDF f for the light direction
      if (length2(f) != 0.0f) {
        // evaluated eagerly (pure); the reference evaluates them after the occlusion test
        float pdfBSDF = bsdfPdfImplE(sc, mt, me, woLocal, wiLocal);
        float pdfLight = pl * ls.pdf / absDot(ls.n, ls.wi);
        if (l.type == LIGHT_AREA) pdfLight *= length2(hit.p - ls.p);
        const f3 Lif = ls.Li * f;
        wfSt(s.sh0 + i, mk4(ls.p.x, ls.p.y, ls.p.z, 0.0f));
        wfSt(s.sh1 + i, mk4(p.att.x, p.att.y, p.att.z, pdfBSDF + pdfLight));
        wfSt(s.sh2 + i, mk4(Lif.x, Lif.y, Lif.z, absDot(ls.wi, hit.n)));
        shadow = true;
        // for whoever commits the shadow ray's result (wfShadowCommit): the bounce budget is used up -> the path's radiance is
        // written out there; a throughput that is not finite makes even an occluded sample count (attenuation * 0 = NaN)
        if ((p.flags & WF_DEPTH_MASK) >= rc.maxDepth) p.flags |= WF_FINAL;
        {
          const f3 z = p.att * mk3(0.0f);
          if (!(z.x == 0.0f && z.y == 0.0f && z.z == 0.0f)) p.flags |= WF_ATT_NAN;
        }
        SR_MARK(10);                                            // BSDF pdf, shadow-ray set-up stored
      }
    }
    // Ld returned {} (no lights / f == 0): the reference still executes L += attenuation * 0,
    // which matters only when the throughput is already inf/NaN — kept for bit parity
    if (!shadow) p.L += p.att * mk3(0.0f);
  }
  p.att = newAtt;
  return shadow ? WF_SHADOW : WF_CONTINUE;
}

}  // namespace yart_hip
