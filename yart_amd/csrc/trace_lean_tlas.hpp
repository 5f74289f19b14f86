// trace_lean_tlas.hpp — trace_lean.hpp for scenes of many nodes, with a spatial top-level hierarchy (device only).
//
// trace_lean_chunked.hpp with one change: a ray's 64-node candidate windows are not found by testing the window's 64 node
// boxes — linear in the node count whatever the ray hits — but read from a per-lane bitset that ONE query of a spatial
// hierarchy over the mesh nodes' padded world boxes (SceneDev::tlas, built in host_scene.hpp) fills when the lane takes the
// ray: a hit leaf sets the bit of its mesh node and of that node's ancestors. The hierarchy only filters (any superset of
// the nodes the ray can reach will do): the walk still visits the candidates in the reference's pre-order and applies the
// exact tests in the nodes' own spaces, so every result is the one the other forms give.
//
// The bitset has nodeBitWords 64-bit words per lane in global memory (lane-interleaved) and a summary word in a register:
// bit g = one of the words g G .. g G + G - 1 is not zero, G = ceil(nodeBitWords / 64) (1 up to 4096 nodes). A lane clears
// the word groups it set before it takes the next ray and when it leaves the kernel: the buffer is all zero between launches.
#pragma once
#if defined(__HIPCC__)
#include "trace_lean_walk.hpp"

namespace yart_hip {

// Fetch(slot) -> LeanRay (world ray of the path in that slot; deterministic, may be called again)
// Commit(slot, hit, didHit, attenuation, samplerDim); Retry(pred, slot) appends to the retry queue
// (wave-wide call). MODE without TRAV_FAST = the general walk (alpha tests inline, no hand-over).
template <bool NEE, int MODE, class Fetch, class Commit, class Retry>
__device__ __forceinline__ void traceLeanTlas(const SceneDev& sc, const SamplerConfig& scfg, const TravStack& stk, const uint32_t* queue,
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
  // Scene nodes this ray can reach at all, 64 at a time: bit k of `cand` = node candBase + k, set if its
  // padded world box and those of all its ancestors are hit within [0, hit.t] (conservative, see
  // traverseScene). Built by one wave-uniform pass over the chunk's node boxes; a missed node's subtree
  // is jumped over through its skip link (skipUntil).
  unsigned long long cand = 0, summary = 0;
  uint32_t candBase = 0;
  bool needMask = false;
  const uint32_t bitStride = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long* bits = sc.nodeBits + gtid;              // word w of this lane: bits[w * bitStride]
  const uint32_t wordsPerBit = (sc.nodeBitWords + 63u) / 64u;  // G
  auto clearBits = [&]() {
    while (summary) {
      const uint32_t g = uint32_t(__builtin_ctzll(summary));
      for (uint32_t w = g * wordsPerBit; w < (g + 1u) * wordsPerBit && w < sc.nodeBitWords; w++) bits[size_t(w) * bitStride] = 0ull;
      summary &= summary - 1ull;
    }
  };
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
          // the nodes this ray can reach: one query of the hierarchy (the lane's traversal stack is free: the ray is new)
          clearBits();
          if (sc.nTlas != 0u) {
            const float tFar = hit.t + (fabsf(hit.t) * 1e-4f + 1e-3f);
            uint32_t cur = 0, sp = 0;
            for (;;) {
              const TlasNode tn = sc.tlas[cur];
              float dw;
              YART_COUNT(nBox, 1);
              bool next = false;
              if (testBox(ray, 0.0f, tFar, tn.lo, tn.hi, dw)) {
                if (tn.b == 0u) { stackPush(stk, sp++, tn.a + 1u, 0.0f); cur = tn.a; next = true; }
                else
                  for (int32_t n = int32_t(tn.a); n >= 0; n = sc.nodes[n].parent) {     // the mesh node and its ancestors
                    const uint32_t w = uint32_t(n) >> 6;
                    const unsigned long long bit = 1ull << (uint32_t(n) & 63u);
                    const uint32_t g = w / wordsPerBit;
                    const unsigned long long have = (summary >> g) & 1ull ? bits[size_t(w) * bitStride] : 0ull;   // (an unflagged group is all zero)
                    if (have & bit) break;                                              // (marked by an earlier leaf, and so are its ancestors)
                    bits[size_t(w) * bitStride] = have | bit;
                    summary |= 1ull << g;
                  }
              }
              if (!next) {
                if (sp == 0u) break;
                float unused;
                stackPop(stk, --sp, cur, unused);
              }
            }
          }
          candBase = 0;
          needMask = true;
        }
      }
    }

    for (;;) {
    // ------------------------------------------------------------------ (A') candidate windows: the next non-empty word of the bitset
    if (has && needMask) {
      uint32_t w = nodeI >> 6;
      unsigned long long word = 0ull;
      while (w < sc.nodeBitWords) {
        const uint32_t g = w / wordsPerBit;
        const unsigned long long rest = summary >> g;
        if (rest == 0ull) { w = sc.nodeBitWords; break; }
        if (!(rest & 1ull)) { w = (g + uint32_t(__builtin_ctzll(rest))) * wordsPerBit; continue; }   // on to the next flagged group
        word = bits[size_t(w) * bitStride];
        if (word != 0ull) break;
        w++;
      }
      if (w >= sc.nodeBitWords) { nodeI = sc.nNodes; cand = 0ull; candBase = 0u; }     // nothing left the ray can reach
      else {
        cand = word;
        candBase = w << 6;
        if (nodeI < candBase) nodeI = candBase;
      }
      needMask = false;
    }

    // ------------------------------------------------------------------ (B) scene-graph walk
    {
      while (has && !inMesh && !needMask) {                     // (lanes leave this loop one by one)
        WF_PHASE(tally, 3);                                     // walk steps
        if (nodeI >= sc.nNodes) {                               // testNode of the root has returned
          done = true;                                          // (committed at the next refill)
          has = false;
          break;
        }
        if (nodeI >= candBase + 64u) {               // beyond this window: the next non-empty one first
          needMask = true;
          break;
        }
        const unsigned long long rest = cand >> (nodeI - candBase);
        if (rest == 0ull) { nodeI = candBase + 64u; continue; }
        {
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

    if (__ballot(has && needMask) == 0ull) break;               // a lane moved on to the next node chunk: mask, walk again
    }

constexpr uint32_t kRefillHere = kLeanRefill;
#include "trace_lean_bvh2.inc"
  }
  clearBits();                                                  // (the buffer is all zero between launches)
#undef LEAN_VISIT
  (void)meshHasAlpha;
#if defined(YART_COUNT_TRAVERSAL)
  tally.box += actx.nBox; tally.tri += actx.nTri; tally.trav += actx.nTrav;
#else
  (void)tally;
#endif
}

// the kernels' entry: NODES 0 = one candidate mask (fewer than 64 nodes), 4 = the same with the scene walk's tables in LDS (at most 16 nodes and meshes: the kernels copy
// them), 1 = chunked masks, 2 = per-lane walk, 3 = windows from the top-level hierarchy
template <bool NEE, int MODE, int NODES, class Fetch, class Commit, class Retry>
__device__ __forceinline__ void traceLeanAny(const SceneDev& sc, const SamplerConfig& scfg, const TravStack& stk,
                                             const uint32_t* queue, uint32_t count, uint32_t* cursor,
                                             Fetch fetch, Commit commit, Retry retry, WfTally& tally) {
  if (NODES == 3) traceLeanTlas<NEE, MODE>(sc, scfg, stk, queue, count, cursor, fetch, commit, retry, tally);
  else if (NODES == 2) traceLeanWalk<NEE, MODE>(sc, scfg, stk, queue, count, cursor, fetch, commit, retry, tally);
  else if (NODES == 1) traceLeanChunked<NEE, MODE>(sc, scfg, stk, queue, count, cursor, fetch, commit, retry, tally);
  else traceLean<NEE, MODE>(sc, scfg, stk, queue, count, cursor, fetch, commit, retry, tally);
}

}  // namespace yart_hip
#endif  // __HIPCC__
