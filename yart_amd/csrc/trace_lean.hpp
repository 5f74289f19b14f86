// trace_lean.hpp — lean traversal with in-wave ray replacement (device only).
//
// Per ray this is traverse.hpp's TRAV_FAST walk, operation for operation (same scene-node
// order, same ordered BVH traversal, same leaf order, alpha / transparent candidates hand the
// ray to the general kernel). What changes is how a 64-wide wave is kept busy: bounced rays have
// heavy-tailed traversal lengths (mean ≈ 37 inner steps, the slowest of 64 ≈ 170), so in the
// one-ray-per-lane kernel 70-80 % of the lanes wait for the wave's slowest ray. Here
//
//   (A) lanes whose ray has finished take new queue entries once at least kLeanRefill lanes are out
//       of the BVH (one ballot + one atomicAdd per refill), and get their scene-node candidate mask,
//   (B) the scene-graph walk of all lanes that stand between two meshes runs to the point where
//       each of them has entered a mesh or finished its ray,
//   (C) "while-while": inner / pop steps until every lane inside a BVH stands at a leaf it must
//       test, then the leaves; repeated until fewer than 64 - kRefill lanes are inside a BVH.
//
// Lane state is small (object-space ray, hit, BVH cursor, stack index): the world ray of a lane
// that re-enters the walk after a transformed node is re-read from the path state.
#pragma once
#if defined(__HIPCC__)
#include "traverse.hpp"

namespace yart_hip {

#ifndef YART_LEAN_REFILL
#define YART_LEAN_REFILL 52
#endif
// measured on the C3 scene (reduced workload, extend / shadow ms): refill 16 -> 40.7 / 41.7, 32 -> 33.1 / 32.3,
// 52 -> 31.4 / 30.6 (with inner-min 4); inner-min 1 -> 35.5 / 37.9, 8 -> 29.7 / 28.1, 12 -> 29.4 / 27.0;
// the one-ray-per-lane kernels: 35.1 / 35.5. Few, well filled refill + walk rounds matter more than
// keeping every lane busy; what is gained is the tail (the last lanes of a batch overlap the next one).
constexpr uint32_t kLeanRefill = YART_LEAN_REFILL;      // refill when at least this many lanes are outside a BVH
#ifndef YART_LEAN_INNER_MIN
#define YART_LEAN_INNER_MIN 12
#endif
constexpr uint32_t kLeanInnerMin = YART_LEAN_INNER_MIN;
#ifndef YART_RETRY_REFILL
#define YART_RETRY_REFILL 52
#endif
#ifndef YART_LEAN_CHUNK_MAX
#define YART_LEAN_CHUNK_MAX 256u
#endif   // leave the inner loop when fewer lanes than this still step

__device__ __forceinline__ uint32_t leanChunk(uint32_t count, uint32_t nWaves) {
  const uint32_t over = count > nWaves * 64u ? count - nWaves * 64u : 0u;
  const uint32_t c = ((over / (nWaves * 4u)) + 63u) & ~63u;
  return c < 64u ? 64u : (c > YART_LEAN_CHUNK_MAX ? YART_LEAN_CHUNK_MAX : c);
}

// One top-up of a wave's private range of the queue. The size shrinks as the queue runs out (guided self-scheduling): a launch ends
// when its LAST wave has finished its last range, and with ranges of 256 entries to the end that wave works ~1 ms after the others
// have run dry — per launch, i.e. ~16 times per batch (profiles/r4_batch_sweep.txt: ~10 ms per batch of any size). How much is left
// is estimated from the end of the wave's own previous range (`lastEnd`; the waves advance together: the true cursor is at most
// nWaves * chunk further) — NOT read from the cursor: that word already takes one atomic per top-up from every wave, and a second
// access per top-up doubled the traversal kernels' time (measured).
__device__ __forceinline__ uint32_t leanTopUp(uint32_t* cursor, uint32_t count, uint32_t nWaves, uint32_t chunk, uint32_t lastEnd, bool isLeader, int leader, uint32_t& size) {
  uint32_t sz = chunk;
  if (chunk > 64u) {                                            // (wave-uniform)
    const uint32_t remaining = count > lastEnd ? count - lastEnd : 0u;
    const uint32_t g = ((remaining / (nWaves * 4u)) + 63u) & ~63u;
    sz = g < 64u ? 64u : (g > chunk ? chunk : g);
  }
  uint32_t c = 0;
  if (isLeader) c = atomicAdd(cursor, sz);
  size = sz;
  return uint32_t(__shfl(int(c), leader));
}

struct LeanRay { f3 o, d; float tMax; Sampler smp; };        // smp: general variant only (alpha tests)

// Fetch(slot) -> LeanRay (world ray of the path in that slot; deterministic, may be called again)
// Commit(slot, hit, didHit, attenuation, samplerDim): called for all lanes that finished since the last refill, together;
// Retry(pred, slot) appends to the retry queue (wave-wide call). MODE without TRAV_FAST = the general walk (alpha tests
// inline, no hand-over).
template <bool NEE, int MODE, class Fetch, class Commit, class Retry>
__device__ __forceinline__ void traceLean(const SceneDev& sc, const SamplerConfig& scfg, const TravStack& stk, const uint32_t* queue,
                                          uint32_t count, uint32_t* cursor, Fetch fetch, Commit commit,
                                          Retry retry, WfTally& tally) {
  constexpr bool kFast = (MODE & TRAV_FAST) != 0;             // else: the general walk (alpha tests, NEE attenuation)
  // the general walk over the retry queue (resumed rays: they start at a leaf and their REMAINING walks differ far more than whole
  // walks do) may take new rays earlier than the lean kernels: YART_RETRY_REFILL (profiles/r5_retry_refill.txt)
  constexpr uint32_t kRefill = kFast ? kLeanRefill : uint32_t(YART_RETRY_REFILL);
  constexpr uint32_t kRefillHere = kRefill;
  Sampler smp; smp.morton = 0; smp.dim = 0; smp.pix = 0;
  f3 attenuation = mk3(1.0f);
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long laneLt = (1ull << lane) - 1ull;
  const uint32_t waveId = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
  const float tMin = 0.001f;
  bool has = false, inMesh = false, exhausted = false, firstFill = true, pendingRetry = false, done = false;
  // the wave's private range of the queue and the size it is topped up by: 64 .. 256 entries, at least four top-ups per wave
  // so that the last ranges do not leave one wave walking alone
  uint32_t chunkNext = 0, chunkEnd = 0;
  const uint32_t chunk = leanChunk(count, nWaves);
  uint32_t recNext = 0, recEnd = 0;                            // the wave's private range of resume records (64 per atomic)
  bool didHit = false, meshDidHit = false, rayIsWorld = false;
  uint32_t slot = 0, nodeI = 0, leftFirst = 0, span = 0, stackIdx = 0;
  float d = 0.0f;
  RayO ray = makeRay(mk3(0.0f), mk3(1.0f));
  HitRec hit; hit.t = 0; hit.u = hit.v = 0; hit.tri = 0; hit.node = 0; hit.backSide = 0;
  const BvhNode* nodes = sc.bvhNodes;
  const LeafTri* leaves = sc.leafTris;
  bool meshHasAlpha = false;
  // scene nodes this ray can reach at all: bit n survives if the padded world box of n and of all
  // its ancestors is hit within [0, tMax] (conservative, see traverseScene); used for scenes of fewer than 64 nodes (the all-ones mask marks a new ray)
  unsigned long long cand = 0;
#if defined(YART_COUNT_TRAVERSAL)
  AlphaCtx actx; actx.sampler = nullptr;     // only its counters are used (YART_COUNT)
  uint32_t boxAtFetch = 0;
#endif

#define LEAN_VISIT() (d < hit.t && (!(NEE && kFast) || !(didHit || meshDidHit) || (leftFirst & kLinkAlphaBit)))
  for (;;) {
    // ------------------------------------------------------------------ (A) retry hand-over + refill
    if (kFast) {
      // A ray that met an alpha / transparent candidate goes to the general kernel WITH ITS WALK: scene node, candidate mask, hit so
      // far, the leaf it stands at and the traversal stack (traverse.hpp: resume record). The general kernel takes the walk up at
      // that leaf — testing a leaf again from its first triangle changes nothing: what was accepted is now rejected by hit.t <= t,
      // what was rejected is rejected again — instead of repeating it from the root (76 % of the general kernels' box tests were
      // such repeats, profiles/r3_ab_top_cache.txt). Not for a shadow ray that is occluded already: from then on the lean walk
      // skips subtrees without alpha-tested triangles, which the general walk — the reference's — does not.
      uint32_t word = slot;
      if (stk.rec != nullptr && __ballot(pendingRetry) != 0ull) {
        const bool can = pendingRetry && stackIdx <= kResumeStack && !(NEE && (didHit || meshDidHit));
        const unsigned long long mc = __ballot(can);
        const uint32_t need = uint32_t(__popcll(mc));
        if (need != 0u) {
          if (recEnd - recNext < need) {                          // (wave-uniform; what is left of the old range is not used)
            const int leader = __ffsll((long long) mc) - 1;
            uint32_t c = 0;
            if (int(lane) == leader) c = atomicAdd(stk.recCursor, 64u);
            recNext = __shfl(c, leader); recEnd = recNext + 64u;
          }
          const uint32_t idx = recNext + uint32_t(__popcll(mc & laneLt));
          recNext += need;
          if (can && idx < stk.recCap) {
            f4* r = stk.rec + size_t(idx) * kResumeWords;
            wfSt(r + 0, mk4(asF(slot), asF(nodeI), asF((didHit ? 1u : 0u) | (meshDidHit ? 2u : 0u) | (stackIdx << 8)), asF(packLink(leftFirst, span))));
            wfSt(r + 1, mk4(d, hit.t, asF(uint32_t(cand)), asF(uint32_t(cand >> 32))));
            // (a shadow ray that has a hit is not resumed; a closest-hit ray without one has nothing to say here)
            if (!NEE && (didHit || meshDidHit)) wfSt(r + 2, mk4(hit.u, hit.v, asF(hit.tri), asF(hit.node | (hit.backSide << kWfNodeBits))));
            for (uint32_t k = 0; k < stackIdx; k += 2u) {
              const uint64_t e0 = stackPeek(stk, k), e1 = k + 1u < stackIdx ? stackPeek(stk, k + 1u) : 0ull;
              wfSt(r + 3 + (k >> 1), mk4(asF(uint32_t(e0)), asF(uint32_t(e0 >> 32)), asF(uint32_t(e1)), asF(uint32_t(e1 >> 32))));
            }
            word = idx | kResumeFlag;
          }
        }
      }
      retry(pendingRetry, word);
    } else {
      retry(pendingRetry, slot);
    }
#if defined(YART_COUNT_TRAVERSAL)
    if (pendingRetry) tally.waste += actx.nBox - boxAtFetch;
#endif
    pendingRetry = false;
    if (has) WF_PHASE(tally, 5);                               // outer rounds / lanes holding a ray
    const unsigned long long idle = __ballot(!has);
    const uint32_t nIdle = uint32_t(__popcll(idle));
    // results of the rays that finished since the last refill are committed together, just before their lanes take new
    // rays (>= kLeanRefill lanes: what a commit loads / stores is issued for most of the wave at once, not lane by lane)
    if ((nIdle >= kRefill || exhausted) && done) { commit(slot, hit, didHit, attenuation, smp.dim); done = false; }
    if (nIdle == 64u && exhausted) break;
    if (!exhausted && nIdle >= kRefill) {
      // queue positions for the idle lanes: the first 64 by wave index; later ones from the wave's private range of the queue,
      // which is topped up `chunk` entries at a time from the shared cursor (one atomic per chunk, not per refill: a single L2
      // word takes ~90 atomics per microsecond, and 10 M refills per launch ran into exactly that; consecutive refills of a
      // wave also stay in one neighbourhood of the queue — the samples of neighbouring pixels — which its L1 likes)
      const uint32_t rank = uint32_t(__popcll(idle & laneLt));
      uint32_t k;
      if (firstFill) {                                          // by wave index, no atomic
        firstFill = false;
        k = waveId * 64u + rank;
        if (nWaves * 64u >= count) exhausted = true;
      } else {
        const uint32_t rem = chunkEnd - chunkNext;
        uint32_t fresh = 0, got = chunk;
        if (rem < nIdle) {                                      // (wave-uniform)
          const int leader = __ffsll((long long) idle) - 1;
          fresh = nWaves * 64u + leanTopUp(cursor, count, nWaves, chunk, chunkEnd, int(lane) == leader, leader, got);
        }
        k = rank < rem ? chunkNext + rank : fresh + (rank - rem);
        if (rem < nIdle) { chunkNext = fresh + (nIdle - rem); chunkEnd = fresh + got; }
        else chunkNext += nIdle;
        if (chunkNext >= count) exhausted = true;               // (ranges are handed out in increasing order: nothing is left behind it)
      }
      if (!has) {
        if (k < count && queue[k] != kWfFreeSlot) {             // (path pool: the queue is the slots in order, free ones marked)
          WF_PHASE(tally, 6);                                   // refills / rays fetched
          const uint32_t word = queue[k];
          const bool hasRec = !kFast && (word & kResumeFlag) != 0u && (word & ~kResumeFlag) < stk.recCap;
          const f4* rec = stk.rec + size_t(word & ~kResumeFlag) * kResumeWords;
          f4 w0 = mk4(0.0f, 0.0f, 0.0f, 0.0f);
          if (hasRec) w0 = wfLd(rec);
          slot = hasRec ? asU(w0.x) : word;
          // (a record that does not name a mesh node of this scene is not followed: the ray is traced from the root)
          const bool resumed = hasRec && asU(w0.y) < sc.nNodes && sc.nodes[asU(w0.y) < sc.nNodes ? asU(w0.y) : 0u].mesh >= 0;
          const LeanRay r = fetch(slot);
          ray = makeRay(r.o + 0.0f, r.d + 0.0f); rayIsWorld = true;
          hit.t = r.tMax; hit.u = hit.v = 0; hit.tri = 0; hit.node = 0; hit.backSide = 0;
          has = true; inMesh = false; nodeI = 0; didHit = false;
          if (!kFast) { smp = r.smp; attenuation = mk3(1.0f); }
          YART_COUNT(nTrav, 1);
#if defined(YART_COUNT_TRAVERSAL)
          boxAtFetch = actx.nBox;
#endif
          cand = ~0ull;
          if (resumed) {
            // the walk as the lean kernel left it: inside the mesh of scene node nodeI, at a leaf, with its stack (at most
            // kResumeStack entries: they fit the LDS part of this kernel's stack)
            const f4 w1 = wfLd(rec + 1);
            nodeI = asU(w0.y);
            const uint32_t fl = asU(w0.z), link = asU(w0.w);
            didHit = (fl & 1u) != 0u; meshDidHit = (fl & 2u) != 0u; stackIdx = fl >> 8;
            leftFirst = link & ((1u << kSpanShift) - 1u); span = link >> kSpanShift;
            d = w1.x; hit.t = w1.y;
            cand = uint64_t(asU(w1.z)) | (uint64_t(asU(w1.w)) << 32);
            if (!NEE && (didHit || meshDidHit)) {
              const f4 w2 = wfLd(rec + 2);
              hit.u = w2.x; hit.v = w2.y; hit.tri = asU(w2.z);
              hit.node = asU(w2.w) & ((1u << kWfNodeBits) - 1u); hit.backSide = asU(w2.w) >> kWfNodeBits;
            }
            for (uint32_t j = 0; j < stackIdx; j += 2u) {
              const f4 e = wfLd(rec + 3 + (j >> 1));
              stackPoke(stk, j, uint64_t(asU(e.x)) | (uint64_t(asU(e.y)) << 32));
              if (j + 1u < stackIdx) stackPoke(stk, j + 1u, uint64_t(asU(e.z)) | (uint64_t(asU(e.w)) << 32));
            }
            const NodeDev& nd = sc.nodes[nodeI];
            if (!((MODE & TRAV_IDENTITY) || (nd.pad[0] & 1u))) {
              f3 oo, od;
              objectRay(sc, nodeI, r.o, r.d, oo, od);
              ray = makeRay(oo, od); rayIsWorld = false;
            }
            const MeshDev& mesh = sc.meshes[nd.mesh];
            nodes = sc.bvhNodes + mesh.nodeOffset;
            leaves = sc.leafTris + mesh.leafOffset;
            meshHasAlpha = mesh.hasAlpha != 0;
            inMesh = true;
            YART_COUNT(nResumed, 1);
          }
        }
      }
      // candidate masks of the new rays: one pass over the node boxes (wave-uniform addresses)
      const bool fresh = has && cand == ~0ull;
      if (fresh) cand = sc.nNodes >= 64u ? ~0ull : ((1ull << sc.nNodes) - 1ull);
      for (uint32_t n = 0; n < sc.nNodes; n++) {
        // (a node no new ray of the wave can still reach — its parent's box was missed by all of them — costs a scalar branch)
        if (__ballot(fresh && ((cand >> n) & 1ull)) == 0ull) continue;
        const f4 wlo = sc.nodeWorld[2u * n], whi = sc.nodeWorld[2u * n + 1u];
        if (fresh && ((cand >> n) & 1ull)) {
          WF_PHASE(tally, 4);                                   // candidate-mask box tests
          const float wmin[3] = {wlo.x, wlo.y, wlo.z}, wmax[3] = {whi.x, whi.y, whi.z};
          float dw;
          YART_COUNT(nBox, 1);
          if (!testBox(ray, 0.0f, hit.t + (fabsf(hit.t) * 1e-4f + 1e-3f), wmin, wmax, dw))
            cand &= ~((unsigned long long) __builtin_bit_cast(uint32_t, wlo.w) |
                      ((unsigned long long) __builtin_bit_cast(uint32_t, whi.w) << 32));
        }
      }
    }

    // ------------------------------------------------------------------ (B) scene-graph walk
    {
      while (has && !inMesh) {                                  // (lanes leave this loop one by one)
        WF_PHASE(tally, 3);                                     // walk steps
        const unsigned long long rest = nodeI < 64u ? (cand >> nodeI) : 0ull;
        if (rest == 0ull) {                                     // testNode of the root has returned
          done = true;                                          // (committed at the next refill)
          has = false;
        } else {
          nodeI += uint32_t(__builtin_ctzll(rest));             // next node the ray can reach
          const NodeDev& nd = sc.nodes[nodeI];
          bool skip = false;
          if (!((MODE & TRAV_IDENTITY) || (nd.pad[0] & 1u))) {
            // transformed node (its padded world box is known to be hit): the exact object-space ray
            const LeanRay r = fetch(slot);                      // the exact world ray (ray.o/d carry +0.0f)
            f3 oo, od;
            objectRay(sc, nodeI, r.o, r.d, oo, od);
            ray = makeRay(oo, od); rayIsWorld = false;
          } else if (!rayIsWorld) {
            const LeanRay r = fetch(slot); ray = makeRay(r.o + 0.0f, r.d + 0.0f); rayIsWorld = true;
          }
          float dd;
          if (!skip) {
            YART_COUNT(nBox, 1);
            if (!testBox(ray, tMin, hit.t, nd.bmin, nd.bmax, dd) || hit.t < dd) skip = true;
          }
          if (skip) nodeI = nd.skip;                            // (bits of the subtree may remain set: skipped by index)
          else {
            bool entered = false;
            if (nd.mesh >= 0) {
              const MeshDev& mesh = sc.meshes[nd.mesh];
              if (!(NEE && kFast && didHit && !mesh.hasAlpha)) { // pruning of occluded shadow rays (traverse.hpp)
                nodes = sc.bvhNodes + mesh.nodeOffset;
                leaves = sc.leafTris + mesh.leafOffset;
                meshHasAlpha = mesh.hasAlpha != 0;
                const BvhNode root = nodes[0];
                YART_COUNT(nBox, 1);
                if (testBox(ray, tMin, hit.t, root.bmin, root.bmax, d)) {     // testBVH entry
                  inMesh = true; entered = true;
                  leftFirst = root.leftFirst; span = root.span; stackIdx = 0; meshDidHit = false;
                }
              }
            }
            if (!entered) nodeI++;
          }
        }
      }
    }

#include "trace_lean_bvh2.inc"
  }
#undef LEAN_VISIT
  (void)meshHasAlpha;
#if defined(YART_COUNT_TRAVERSAL)
  tally.box += actx.nBox; tally.tri += actx.nTri; tally.trav += actx.nTrav; tally.resumed += actx.nResumed;
#else
  (void)tally;
#endif
}

}  // namespace yart_hip
#endif  // __HIPCC__
