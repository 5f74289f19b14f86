// trace_wave.hpp — wave-level persistent ray tracer (device only): the same per-ray
// operation sequence as traverse.hpp::traverseScene (reference cpu/ray-integrator.cpp:20-261),
// restructured for 64-wide wavefronts:
//
//  * dynamic refill — a lane whose ray has finished takes the next queue entry as soon as
//    enough lanes are idle (one ballot + one atomicAdd per refill), instead of the wave
//    waiting for its slowest ray (measured lane utilisation of the one-ray-per-lane form:
//    0.32 closest-hit, 0.22 shadow);
//  * while-while — inner-node / pop steps run in a tight loop until every lane of the wave
//    stands at a leaf it must test (or has left the mesh); leaf triangle tests then run
//    with all those lanes converged. Per ray nothing is reordered.
//  * identity-chain fast path for the scene-graph walk: a node whose whole ancestor chain
//    is the identity maps the ray with "x + 0.0f" (what the reference's 4x4 products reduce
//    to for finite inputs, including the -0 -> +0 of the accumulate-from-zero).
//
// A lane's ray is a small state machine:
//   WALK  at scene node `nodeI` (pre-order walk with skip links)
//   MESH  inside the BVH of node `nodeI`: current (leftFirst, span, d), LDS/global stack
#pragma once
#if defined(__HIPCC__)
#include "traverse.hpp"

namespace yart_hip {

struct TraceJob {            // what a lane needs to trace one ray
  f3 o, d;                   // world-space ray
  float tMax;                // hit.t on entry (inf for closest hit, dist - 0.001 for shadow rays)
  uint32_t slot;             // path slot (result address)
  Sampler smp;               // for the stochastic alpha test
};
struct TraceResult {
  float t, u, v;
  uint32_t tri, node, backSide;
  bool hit;
  f3 attenuation;            // NEE only: product over transparent surfaces
  uint32_t dim;              // sampler dimension after the traversal
};

constexpr uint32_t kRefillMin = 20;     // refill when at least this many lanes are idle
constexpr uint32_t kInnerBurst = 32;    // max inner/pop steps per scheduling round
constexpr uint32_t kInnerMin = 12;      // leave the inner loop when fewer lanes than this still step

// Fetch(k, job): load queue entry k.  Commit(job, result): store the result of a finished ray.
template <bool NEE, class Fetch, class Commit>
__device__ __forceinline__ void traceWave(const SceneDev& sc, const SamplerConfig& scfg, const TravStack& stk,
                                          uint32_t* cursor, uint32_t count, Fetch fetch, Commit commit,
                                          WfTally& tally) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long laneLt = (1ull << lane) - 1ull;
  const float tMin = 0.001f;
  const uint32_t waveId = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
  bool has = false, exhausted = false, inMesh = false, firstFill = true;
  TraceJob job;
  job.o = mk3(0); job.d = mk3(0); job.tMax = 0; job.slot = 0; job.smp.dim = 0; job.smp.morton = 0;
  RayO ray = makeRay(mk3(0), mk3(1));
  HitRec hit; hit.t = 0; hit.u = hit.v = 0; hit.tri = 0; hit.node = 0; hit.backSide = 0;
  f3 attenuation = mk3(1.0f);
  bool didHit = false, meshDidHit = false, rayIsWorld = false;
  uint32_t nodeI = 0, leftFirst = 0, span = 0, stackIdx = 0;
  float d = 0.0f;
  const BvhNode* nodes = sc.bvhNodes;
  const LeafTri* leaves = sc.leafTris;
  uint32_t meshIdx = 0;
  AlphaCtx actx; actx.sampler = &job.smp; actx.cfg = scfg;
  NodeRayCache cache;
  cache.o = mk3(0); cache.d = mk3(0);

  // Wave-uniform scheduling: every iteration the wave runs ONE phase, chosen by how many
  // lanes are waiting for it; lanes in other states wait. Per ray the operation order is
  // untouched (each lane walks its own state machine), only the interleaving changes.
  for (;;) {
    const bool walking = has && !inMesh;
    const bool atLeaf = has && inMesh && span > 0 && d < hit.t;
    const bool stepping = has && inMesh && !atLeaf;
    const unsigned long long idle = __ballot(!has);
    const uint32_t nIdle = uint32_t(__popcll(idle));
    const uint32_t nWalk = uint32_t(__popcll(__ballot(walking)));
    const uint32_t nLeaf = uint32_t(__popcll(__ballot(atLeaf)));
    const uint32_t nStep = uint32_t(__popcll(__ballot(stepping)));
    if (nIdle == 64u && exhausted) break;

    // ------------------------------------------------------------------ refill
    if (!exhausted && (nIdle >= kRefillMin || nIdle == 64u)) {
      const int leader = __ffsll((long long) idle) - 1;
      uint32_t base;
      if (firstFill) {                                          // by wave index, no atomic
        firstFill = false;
        base = waveId * 64u;
        if (nWaves * 64u >= count) exhausted = true;
      } else {
        base = 0;
        if (int(lane) == leader) base = atomicAdd(cursor, nIdle);
        base = nWaves * 64u + __shfl(base, leader);
        if (base + nIdle >= count) exhausted = true;           // wave-uniform
      }
      if (!has) {
        const uint32_t k = base + uint32_t(__popcll(idle & laneLt));
        if (k < count) {
          WF_PHASE(tally, 6);                                   // refills / rays fetched
          fetch(k, job);
          has = true; inMesh = false; nodeI = 0; didHit = false; cache.node = -2; rayIsWorld = false;
          hit.t = job.tMax; hit.u = hit.v = 0; hit.tri = 0; hit.node = 0; hit.backSide = 0;
          attenuation = mk3(1.0f);
          YART_COUNT(nTrav, 1);
        }
      }
      continue;
    }
    if (has) WF_PHASE(tally, 0);                               // scheduling rounds / lanes holding a ray

    if (nStep > 0 && nStep >= nWalk && nStep >= nLeaf) {
      // ---------------------------------------------------------------- inner / pop steps
      for (uint32_t burst = 0; burst < kInnerBurst; burst++) {
        const bool go = has && inMesh && !(span > 0 && d < hit.t);
        const uint32_t nGo = uint32_t(__popcll(__ballot(go)));
        if (nGo == 0 || (burst > 0 && nGo < kInnerMin)) break;
        if (go) {
          WF_PHASE(tally, 2);
          bool pop = true;
          if (d < hit.t) {                                      // inner node: test both children
            const BvhNode* pair = nodes + (leftFirst & kLinkIndexMask);
            const BvhNode c1 = pair[0], c2 = pair[1];
            YART_COUNT(nBox, 2);
            float d1, d2;
            bool hit1, hit2;
            testBox2(ray, tMin, hit.t, c1, c2, hit1, hit2, d1, d2);
            if (hit1 || hit2) {
              const bool firstNear = hit1 && !(hit2 && d1 > d2);
              if (hit1 && hit2)
                stackPush(stk, stackIdx++, firstNear ? (c2.leftFirst | (c2.span << kSpanShift))
                                                     : (c1.leftFirst | (c1.span << kSpanShift)),
                          firstNear ? d2 : d1);
              d = firstNear ? d1 : d2;
              leftFirst = firstNear ? c1.leftFirst : c2.leftFirst;
              span = firstNear ? c1.span : c2.span;
              pop = false;
            }
          }
          if (pop) {
            if (stackIdx == 0) {                                // testBVH returns: back to the walk
              inMesh = false; didHit |= meshDidHit; nodeI++;
            } else {
              uint32_t link;
              stackPop(stk, --stackIdx, link, d);
              leftFirst = link & ((1u << kSpanShift) - 1u); span = link >> kSpanShift;
            }
          }
        }
      }
    } else if (nLeaf > 0 && nLeaf >= nWalk) {
      // ---------------------------------------------------------------- leaf: triangles in index order
      if (atLeaf) {
        WF_PHASE(tally, 5);
        const MeshDev& mesh = sc.meshes[meshIdx];
        for (uint32_t i = 0; i < span; i++) {
          WF_PHASE(tally, 3);
          const LeafTri tr = leaves[(leftFirst & kLinkIndexMask) + i];
          YART_COUNT(nTri, 1);
          const f3 p0 = mk3(tr.p0[0], tr.p0[1], tr.p0[2]);
          const f3 edge1 = mk3(tr.e1[0], tr.e1[1], tr.e1[2]);
          const f3 edge2 = mk3(tr.e2[0], tr.e2[1], tr.e2[2]);
          bool accepted = false;
          do {
            const f3 rayEdge2 = cross(ray.d, edge2);
            const float det = dot(edge1, rayEdge2);
            if (double(fabsf(det)) < 1e-12) break;
            const float invDet = 1.0f / det;
            const f3 b = ray.o - p0;
            const float u = dot(b, rayEdge2) * invDet;
            if (u < 0.0f || u > 1.0f) break;
            const f3 bEdge1 = cross(b, edge1);
            const float v = dot(ray.d, bEdge1) * invDet;
            if (v < 0.0f || u + v > 1.0f) break;
            const float t = dot(edge2, bEdge1) * invDet;
            if (t <= tMin || hit.t <= t) break;
            if (tr.matFlags & (MAT_HAS_ALPHA | MAT_TRANSPARENT)) {
              WF_PHASE(tally, 4);                               // alpha / transparent slow path
              f2 uv; f3 n;
              interpUVN(sc, mesh, tr.triIdx, u, v, uv, n);
              const MaterialDev& mt = sc.materials[tr.material];
              if (tr.matFlags & MAT_HAS_ALPHA) {
                float alpha = matAlpha(sc, mt, uv);
                if (alpha < 1.0f && get1D(job.smp, scfg) > alpha) break;
              }
              if (NEE && (tr.matFlags & MAT_TRANSPARENT)) {
                attenuation *= absDot(n, ray.d) * matBase(sc, mt, uv);
                break;
              }
            }
            hit.t = t; hit.u = u; hit.v = v; hit.tri = tr.triIdx; hit.node = nodeI;
            hit.backSide = det < 0 ? 1u : 0u;
            accepted = true;
          } while (false);
          meshDidHit |= accepted;
          if (NEE && meshDidHit) break;
        }
        if (stackIdx == 0) {
          inMesh = false; didHit |= meshDidHit; nodeI++;
        } else {
          uint32_t link;
          stackPop(stk, --stackIdx, link, d);
          leftFirst = link & ((1u << kSpanShift) - 1u); span = link >> kSpanShift;
        }
      }
    } else {
      // ---------------------------------------------------------------- scene-graph walk: one node
      if (walking) {
        WF_PHASE(tally, 1);
        if (nodeI < sc.nNodes) {
          const NodeDev& nd = sc.nodes[nodeI];
          if (nd.pad[0] & 1u) {                                 // identity chain: the world ray (+0), made once
            if (!rayIsWorld) { ray = makeRay(job.o + 0.0f, job.d + 0.0f); rayIsWorld = true; }
          } else {
            f3 oo, od;
            nodeObjectRay(sc, nodeI, nd, job.o, job.d, cache, oo, od);
            ray = makeRay(oo, od);
            rayIsWorld = false;
          }
          float dd;
          YART_COUNT(nBox, 1);
          if (!testBox(ray, tMin, hit.t, nd.bmin, nd.bmax, dd) || hit.t < dd) {
            nodeI = nd.skip;
          } else {
            bool entered = false;
            if (nd.mesh >= 0) {
              const MeshDev& mesh = sc.meshes[nd.mesh];
              meshIdx = uint32_t(nd.mesh);
              nodes = sc.bvhNodes + mesh.nodeOffset;
              leaves = sc.leafTris + mesh.leafOffset;
              const BvhNode root = nodes[0];
              YART_COUNT(nBox, 1);
              if (testBox(ray, tMin, hit.t, root.bmin, root.bmax, d)) {      // testBVH entry (:95)
                inMesh = true; leftFirst = root.leftFirst; span = root.span; stackIdx = 0; meshDidHit = false;
                entered = true;
              }
            }
            if (!entered) nodeI++;
          }
        }
        if (!inMesh && nodeI >= sc.nNodes) {                    // walk finished: ray done
          TraceResult r;
          r.t = hit.t; r.u = hit.u; r.v = hit.v; r.tri = hit.tri; r.node = hit.node; r.backSide = hit.backSide;
          r.hit = didHit; r.attenuation = attenuation; r.dim = job.smp.dim;
          commit(job, r);
          has = false;
        }
      }
    }
  }
  (void)tally;
#if defined(YART_COUNT_TRAVERSAL)
  tally.box += actx.nBox; tally.tri += actx.nTri; tally.trav += actx.nTrav;
#endif
}

}  // namespace yart_hip
#endif  // __HIPCC__
