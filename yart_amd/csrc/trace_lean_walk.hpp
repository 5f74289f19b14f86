// trace_lean_walk.hpp — trace_lean.hpp for scenes of many nodes (kLeanWalkNodes and more; device only).
//
// Same algorithm, without the per-ray candidate mask: the mask is built by one wave-uniform pass over ALL node boxes per
// refill round, which is what a scene of a dozen nodes wants and what makes a scene of a thousand cost a thousand box tests
// per ray whatever it hits (measured: 20 -> 1064 nodes, same ray count: traversal 8.6 -> 385 ms). Here every lane walks the
// node list on its own, as the reference's recursion does: the node's padded world box first (the test a mask bit stands
// for), a miss jumps over the subtree through the node's skip link, a hit goes on to the exact test in the node's space.
// A ray costs the nodes it visits, not the nodes the scene has.
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
#include "trace_lean_chunked.hpp"

namespace yart_hip {

// Fetch(slot) -> LeanRay (world ray of the path in that slot; deterministic, may be called again)
// Commit(slot, hit, didHit, attenuation, samplerDim); Retry(pred, slot) appends to the retry queue
// (wave-wide call). MODE without TRAV_FAST = the general walk (alpha tests inline, no hand-over).
template <bool NEE, int MODE, class Fetch, class Commit, class Retry>
__device__ __forceinline__ void traceLeanWalk(const SceneDev& sc, const SamplerConfig& scfg, const TravStack& stk, const uint32_t* queue,
                                          uint32_t count, uint32_t* cursor, Fetch fetch,
                                          Commit commit, Retry retry, WfTally& tally) {
  constexpr bool kFast = (MODE & TRAV_FAST) != 0;             // else: the general walk (alpha tests, NEE attenuation)
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
  bool didHit = false, meshDidHit = false, rayIsWorld = false;
  uint32_t slot = 0, nodeI = 0, leftFirst = 0, span = 0, stackIdx = 0;
  float d = 0.0f;
  RayO ray = makeRay(mk3(0.0f), mk3(1.0f));
  HitRec hit; hit.t = 0; hit.u = hit.v = 0; hit.tri = 0; hit.node = 0; hit.backSide = 0;
  const BvhNode* nodes = sc.bvhNodes;
  const LeafTri* leaves = sc.leafTris;
  bool meshHasAlpha = false;
#if defined(YART_COUNT_TRAVERSAL)
  AlphaCtx actx; actx.sampler = nullptr;     // only its counters are used (YART_COUNT)
#endif

#define LEAN_VISIT() (d < hit.t && (!(NEE && kFast) || !(didHit || meshDidHit) || (leftFirst & kLinkAlphaBit)))
  for (;;) {
    // ------------------------------------------------------------------ (A) retry hand-over + refill
    retry(pendingRetry, slot);
    pendingRetry = false;
    if (has) WF_PHASE(tally, 5);                               // outer rounds / lanes holding a ray
    const unsigned long long idle = __ballot(!has);
    const uint32_t nIdle = uint32_t(__popcll(idle));
    // results of the rays that finished since the last refill are committed together, just before their lanes take new
    // rays (>= kLeanRefill lanes: what a commit loads / stores is issued for most of the wave at once, not lane by lane)
    if ((nIdle >= kLeanRefill || exhausted) && done) { commit(slot, hit, didHit, attenuation, smp.dim); done = false; }
    if (nIdle == 64u && exhausted) break;
    if (!exhausted && nIdle >= kLeanRefill) {
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
          slot = queue[k];
          const LeanRay r = fetch(slot);
          ray = makeRay(r.o + 0.0f, r.d + 0.0f); rayIsWorld = true;
          hit.t = r.tMax; hit.u = hit.v = 0; hit.tri = 0; hit.node = 0; hit.backSide = 0;
          has = true; inMesh = false; nodeI = 0; didHit = false;
          if (!kFast) { smp = r.smp; attenuation = mk3(1.0f); }
          YART_COUNT(nTrav, 1);
        }
      }
    }

    // ------------------------------------------------------------------ (B) scene-graph walk
    {
      while (has && !inMesh) {                                  // (lanes leave this loop one by one)
        WF_PHASE(tally, 3);                                     // walk steps
        if (nodeI >= sc.nNodes) {                               // testNode of the root has returned
          done = true;                                          // (committed at the next refill)
          has = false;
          break;
        }
        // the node's padded world box (conservative: a ray that misses it within [0, hit.t] fails the exact test below, for
        // this node and for every node of its subtree)
        if (!rayIsWorld) { const LeanRay r = fetch(slot); ray = makeRay(r.o + 0.0f, r.d + 0.0f); rayIsWorld = true; }
        {
          const f4 wlo = sc.nodeWorld[2u * nodeI], whi = sc.nodeWorld[2u * nodeI + 1u];
          const float wmin[3] = {wlo.x, wlo.y, wlo.z}, wmax[3] = {whi.x, whi.y, whi.z};
          float dw;
          WF_PHASE(tally, 4);                                   // padded-box tests
          YART_COUNT(nBox, 1);
          if (!testBox(ray, 0.0f, hit.t + (fabsf(hit.t) * 1e-4f + 1e-3f), wmin, wmax, dw)) { nodeI = __builtin_bit_cast(uint32_t, wlo.w); continue; }   // (the skip link rides in the box record)
        }
        {
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
          if (skip) nodeI = nd.skip;
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

constexpr uint32_t kRefillHere = kLeanRefill;
#include "trace_lean_bvh2.inc"
  }
#undef LEAN_VISIT
  (void)meshHasAlpha;
#if defined(YART_COUNT_TRAVERSAL)
  tally.box += actx.nBox; tally.tri += actx.nTri; tally.trav += actx.nTrav;
#else
  (void)tally;
#endif
}

// scenes of at least this many nodes use this form (yart_hip.hip); measured on instanced scenes (tools/many_nodes.py, 960x540x16):
// instances all over the room, every group box spanning it — 76 nodes: chunked masks 18.0 ms of traversal, this walk 25.2; 267: 74 /
// 76; 1064: 385 / 246 —, groups of instances that sit together — 267 nodes: 13.1 / 9.8; 1064: 41.5 / 20.4; 4252: 255 / 78
constexpr uint32_t kLeanWalkNodes = 512;

}  // namespace yart_hip
#endif  // __HIPCC__
