// yart_hip.hip — gfx950 kernels and the C ABI of libyart_hip.so.
//
// Stands where reference cpu/tile-renderer.hpp:118-309 (TileRenderer::renderImpl /
// finishTile) and cpu/integrator.cpp:5-28 (Integrator::render) stand: tiles of the
// image are rendered for `waveSamples` samples per pixel, each pixel's samples go
// through GMoN, and waves are blended into the HDR framebuffer. The thread pool of
// the reference becomes a persistent grid; the per-sample work is in integrator.hpp.
//
// Kernels in this file
//   k_render_mega   one lane = one (pixel, sample) path from camera to termination
//                   (BASELINE config "megakernel integrator"); persistent waves pull
//                   64 paths at a time from an atomic cursor; LDS traversal stack.
//   k_gmon_blend    one wave = one pixel: bucket sums in sample order, GMoN value,
//                   blend into the HDR buffer (integrator.cpp:17-25, tile-renderer.hpp:220-232).
//   k_probe_*       diagnostics used by the parity tests.
// The wavefront (queue-based) pipeline lives in wavefront.hip.inc.
// This file is compiled four times (csrc/Makefile), YART_TU selecting what a translation unit emits — the kernels are templates and
// instantiate where they are referenced, so the four objects build in parallel and each holds a quarter of the device code:
//   0  the C ABI, host orchestration, and every kernel not named below (streaming passes, megakernel, probes, BVH build)
//   1  the lean closest-hit kernels k_wf_extend_lean<MODE, NODES>          2  the lean any-hit kernels k_wf_shadow_lean<MODE, NODES>
//   3  the general kernels: retry (resumed walks), one-ray-per-lane lean and general forms        4  the shade kernel k_wf_shade<SORT, FIT, ENV1>
// Units 1-4 export their kernels as type-erased host stubs (yart_hip::tu::*, below); unit 0 launches them through those pointers.
#ifndef YART_TU
#define YART_TU 0
#endif
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/yart_hip.h"
#include "estimator.hpp"
#include "host_scene.hpp"
#include "integrator.hpp"
#include "scene_file.hpp"
#include "gltf_reader.hpp"
#include "wavefront.hpp"
#include "trace_lean.hpp"
#include "trace_lean_chunked.hpp"
#include "trace_lean_walk.hpp"
#include "trace_lean_tlas.hpp"
#include "tonemap.hpp"
#include "trace_ranges.hpp"

using namespace yart_hip;

namespace {

constexpr int kBlock = 256;            // 4 waves per workgroup
#ifndef YART_STREAM_BLOCKS
#define YART_STREAM_BLOCKS 8           // workgroups per CU of the streaming kernels (generate, post, compact); shade + post stage at 1080p x 64 spp: 4 -> 105.5, 8 -> 106.1, 16 -> 106.4, 32 -> 106.6 ms
#endif
constexpr int kLdsStack = 24;          // traversal stack entries kept in LDS per lane (8 B each)
constexpr int kSpillDepth = int(kRefStackDepth) - kLdsStack;
constexpr int kSpillDepthMax = int(kRefStackDepth);   // spill area sized for the shallowest LDS stack
constexpr uint64_t kDefaultBatchPaths = 1ull << 28;   // YartRenderParams::max_batch_paths = 0: 268 M paths (batch-synchronous: 67 GB; path pool: 4.3 GB of per-sample records)
constexpr uint64_t kDefaultPoolPaths = 1ull << 25;    // YartRenderParams::pool_paths = 0: 33.5 M slots, 5.6 GB
constexpr int kPoolLag = 4;                            // the host looks at the counters of the round before the previous one (ring of 4)
constexpr int kNumCounters = 32;       // [0] rays, [1..4] instrumented tallies, [8..31] debug statistics

thread_local std::string g_lastError;

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIP_CHECK(expr)                                                                     \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  void ensure(size_t count) {
    if (count <= n) return;
    release();
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
    n = count;
  }
  void upload(const std::vector<T>& v) {
    ensure(std::max<size_t>(v.size(), 1));
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  }
};

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
#if YART_TU == 0
struct MegaArgs {
  SceneDev sc;
  CameraDev cam;
  RenderConst rc;
  const uint32_t* pixels;      // packed x | y << 16, tile-major order
  uint32_t nPixels, spp, sampleOffset, pad;
  f4* L;                    // 3 floats per (pixel, sample)
  uint32_t* cursor;
  unsigned long long* rays;
  uint64_t* spill;             // kSpillDepth entries per launched thread, lane-interleaved
};

__global__ void __launch_bounds__(kBlock) k_render_mega(MegaArgs a) {
  __shared__ uint64_t ldsStack[kLdsStack * kBlock];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nthreads = gridDim.x * blockDim.x;
  PathCtx cx;
  cx.sc = &a.sc;
  cx.sobol = reinterpret_cast<const uint32_t*>(a.sc.lut + LutDev::sobol);
  cx.stk.lds = (lds_u64*)(ldsStack + threadIdx.x); cx.stk.ldsStride = kBlock; cx.stk.ldsDepth = kLdsStack;
  cx.stk.spill = a.spill + gtid; cx.stk.spillStride = nthreads;
  cx.rc = a.rc;
  const uint32_t total = a.nPixels * a.spp;
  uint32_t rays = 0;
  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.cursor, 64u);
    base = __shfl(base, 0);
    if (base >= total) break;            // wave-uniform exit: every wave drains the cursor
    const uint32_t w = base + lane;
    if (w < total) {
      const uint32_t pi = w / a.spp, s = w - pi * a.spp;
      const uint32_t pk = a.pixels[pi];
      const uint32_t before = rays;
      f3 L = samplePixel(cx, a.cam, pk & 0xffffu, pk >> 16, s + a.sampleOffset, rays);
      a.L[w] = mk4(L.x, L.y, L.z, asF(rays - before));
    }
  }
  // one atomic per wave
  unsigned long long r = rays;
  for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o);
  if (lane == 0 && r) atomicAdd(a.rays, r);
#if defined(YART_COUNT_TRAVERSAL)
  unsigned long long c[4] = {cx.nTrav, cx.nBox, cx.nTri, cx.nShade};
  for (int k = 0; k < 4; k++) {
    unsigned long long v = c[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0 && v) atomicAdd(a.rays + 1 + k, v);
  }
#endif
}

struct GmonArgs {
  const f4* L;                 // per (pixel, sample): radiance.xyz, ray count
  uint32_t* pixRays;           // per pixel of the batch: the wave's ray count of that pixel (nullptr: not wanted)
  const uint32_t* pixels;
  uint32_t nPixels, spp, width;
  int kind;                    // EstimatorKind
  float exposureScale, wCurrent, wWave, pad1;
  float* hdr;                  // RGBA32F, width * height
};

// 16 lanes per pixel (lane b < m sums bucket b in increasing sample order, as the reference's
// accumulation does), 4 pixels per wave; the serial GMoN tail runs on one lane per pixel.
constexpr int kGmonLanes = 16;
static_assert(kGmonMax <= kGmonLanes, "one lane per bucket");
constexpr int kGmonPixPerBlock = kBlock / kGmonLanes;
__global__ void __launch_bounds__(kBlock) k_gmon_blend(GmonArgs a) {
  // bucket sums / counts live in LDS and are sorted there: private arrays with run-time indices would be scratch
  __shared__ f3 sAcc[kGmonPixPerBlock][kGmonMax];
  __shared__ uint32_t sCnt[kGmonPixPerBlock][kGmonMax];
  const uint32_t sub = threadIdx.x & (kGmonLanes - 1), lp = threadIdx.x / kGmonLanes;
  const uint32_t pi = blockIdx.x * kGmonPixPerBlock + lp;
  const bool valid = pi < a.nPixels;
  const int m = estimatorBuckets(a.kind, int32_t(a.spp));
  __shared__ uint32_t sRays[kGmonPixPerBlock][kGmonMax];
  if (valid && int(sub) < m) {
    f3 acc = mk3(0); uint32_t cnt = 0, rays = 0;
    const f4* p = a.L + size_t(pi) * a.spp;
    // bucket k mod m, increasing k; four samples' loads in flight, accumulated in order
    const uint32_t um = uint32_t(m);
    uint32_t s = sub;
    for (; s + 3u * um < a.spp; s += 4u * um) {
      f4 v[4];
      for (uint32_t j = 0; j < 4u; j++) v[j] = p[s + j * um];
      for (uint32_t j = 0; j < 4u; j++) {
        const f3 w = mk3(v[j].x, v[j].y, v[j].z) * a.exposureScale;
        if (estimatorAccepts(a.kind, w)) { acc += w; cnt++; }
        rays += __builtin_bit_cast(uint32_t, v[j].w);
      }
    }
    for (; s < a.spp; s += um) {
      const f4 q = p[s];
      f3 v = mk3(q.x, q.y, q.z) * a.exposureScale;
      if (estimatorAccepts(a.kind, v)) { acc += v; cnt++; }
      rays += __builtin_bit_cast(uint32_t, q.w);
    }
    sAcc[lp][sub] = acc;
    sCnt[lp][sub] = cnt;
    sRays[lp][sub] = rays;
  }
  __syncthreads();
  if (valid && sub == 0 && a.pixRays != nullptr) {
    uint32_t r = 0;
    for (int b = 0; b < m; b++) r += sRays[lp][b];
    a.pixRays[pi] = r;
  }
  if (valid && sub == 0) {
    f3 v = estimatorFinish(a.kind, sAcc[lp], sCnt[lp], m, a.spp);
    const uint32_t pk = a.pixels[pi];
    float* o = a.hdr + (size_t(pk >> 16) * a.width + (pk & 0xffffu)) * 4;
    // m_hdrBuffer = current * wCurrent + wave * wWave   (tile-renderer.hpp:230)
    o[0] = o[0] * a.wCurrent + v.x * a.wWave;
    o[1] = o[1] * a.wCurrent + v.y * a.wWave;
    o[2] = o[2] * a.wCurrent + v.z * a.wWave;
    o[3] = o[3] * a.wCurrent + 1.0f * a.wWave;
  }
}

// Renderer::TileData.rays (renderer.hpp:40-50; tile-renderer.hpp:183 passes the tile integrator's ray count): per pixel block of
// this rank the sum of its pixels' ray counts of the wave (k_gmon_blend's pixRays); one wave per block
__global__ void __launch_bounds__(64) k_tile_rays(const uint32_t* pixRays, const uint32_t* tileStart, const uint32_t* tileCount,
                                                  uint32_t firstTile, uint32_t nTiles, unsigned long long* out) {
  const uint32_t t = firstTile + blockIdx.x;
  if (blockIdx.x >= nTiles) return;
  unsigned long long r = 0;
  for (uint32_t i = threadIdx.x; i < tileCount[t]; i += 64u) r += pixRays[tileStart[t] + i];
  for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o);
  if (threadIdx.x == 0) out[t] = r;
}

// AgX tonemap of an RGBA32F frame (alpha kept as 1, tile-renderer.hpp:234-237) and the 8-bit
// encoding of output/ppm.cpp; one lane per pixel, 16 B in / 16 B (or 3 B) out: HBM-bound.
__global__ void __launch_bounds__(kBlock) k_tonemap_agx(const f4* in, f4* out, uint32_t n, int look) {
  const AgxLook lk = agxLook(look);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const f4 v = in[i];
    const f3 c = agxTonemap(mk3(v.x, v.y, v.z), lk);
    f4 o; o.x = c.x; o.y = c.y; o.z = c.z; o.w = 1.0f;
    out[i] = o;
  }
}
__global__ void __launch_bounds__(kBlock) k_encode_rgb8(const f4* in, uint8_t* out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const f4 v = in[i];
    out[3 * size_t(i)] = ppmByte(v.x); out[3 * size_t(i) + 1] = ppmByte(v.y); out[3 * size_t(i) + 2] = ppmByte(v.z);
  }
}

struct ProbeSampleArgs {
  SceneDev sc; CameraDev cam; RenderConst rc;
  const uint32_t* xys; uint32_t n; float* out; unsigned long long* rays; uint64_t* spill;
};
__global__ void __launch_bounds__(kBlock) k_probe_samples(ProbeSampleArgs a) {
  __shared__ uint64_t ldsStack[kLdsStack * kBlock];
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  PathCtx cx;
  cx.sc = &a.sc;
  cx.sobol = reinterpret_cast<const uint32_t*>(a.sc.lut + LutDev::sobol);
  cx.stk.lds = (lds_u64*)(ldsStack + threadIdx.x); cx.stk.ldsStride = kBlock; cx.stk.ldsDepth = kLdsStack;
  cx.stk.spill = a.spill + gtid; cx.stk.spillStride = gridDim.x * blockDim.x;
  cx.rc = a.rc;
  if (gtid >= a.n) return;
  uint32_t rays = 0;
  f3 L = samplePixel(cx, a.cam, a.xys[gtid * 3], a.xys[gtid * 3 + 1], a.xys[gtid * 3 + 2], rays);
  a.out[gtid * 3] = L.x; a.out[gtid * 3 + 1] = L.y; a.out[gtid * 3 + 2] = L.z;
  atomicAdd(a.rays, (unsigned long long) rays);
}

// the sampler alone (diagnostic): per case startPixelSample + a pattern of draws (1 = get1D, 2 = get2D); with `tab` set the
// draws go through the per-render sampler tables exactly as the wavefront kernels' do
struct ProbeSamplerArgs { SamplerConfig cfg; const uint32_t* sobol; const uint32_t* cases; uint32_t n, nDraws, nOut, pad; const uint8_t* pattern; float* out; };
__global__ void __launch_bounds__(kBlock) k_probe_sampler(ProbeSamplerArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  Sampler s;
  startPixelSample(s, a.cfg, a.cases[3 * i], a.cases[3 * i + 1], a.cases[3 * i + 2]);
  s.pix = i;                                                 // (sampler tables: one pixel column per case)
  float* o = a.out + size_t(i) * a.nOut;
  for (uint32_t k = 0; k < a.nDraws; k++) {
    if (a.pattern[k] == 2) { const f2 v = get2D(s, a.cfg, a.sobol); *o++ = v.x; *o++ = v.y; }
    else *o++ = get1D(s, a.cfg);
  }
}

struct ProbeHitArgs { SceneDev sc; const float* rays; uint32_t n; float* out; uint64_t* spill; };
__global__ void __launch_bounds__(kBlock) k_probe_hits(ProbeHitArgs a) {
  __shared__ uint64_t ldsStack[kLdsStack * kBlock];
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  TravStack stk;
  stk.lds = (lds_u64*)(ldsStack + threadIdx.x); stk.ldsStride = kBlock; stk.ldsDepth = kLdsStack;
  stk.spill = a.spill + gtid; stk.spillStride = gridDim.x * blockDim.x;
  if (gtid >= a.n) return;
  const float* r = a.rays + size_t(gtid) * 6;
  f3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);
  HitRec hr; hr.t = kInf; hr.u = hr.v = 0; hr.tri = 0; hr.node = 0; hr.backSide = 0;
  f3 att = mk3(1.0f);
  Sampler dummy; dummy.dim = 0; dummy.morton = 0;
  AlphaCtx ac; ac.sampler = &dummy; ac.cfg.log2spp = 0; ac.cfg.nBase4Digits = 6;
  bool hit = traverseScene<false>(a.sc, o, d, 0.001f, hr, att, stk, ac);
  float* q = a.out + size_t(gtid) * 16;
  for (int i = 0; i < 16; i++) q[i] = 0.0f;
  q[0] = hit ? 1.0f : 0.0f;
  if (hit) {
    Hit h = finalizeHit(a.sc, hr, o, d);
    q[1] = h.t; q[2] = hr.u; q[3] = hr.v;
    q[4] = h.p.x; q[5] = h.p.y; q[6] = h.p.z; q[7] = h.n.x; q[8] = h.n.y; q[9] = h.n.z;
    q[10] = h.tg.x; q[11] = h.tg.y; q[12] = h.tg.z;
    q[13] = float(localTri(a.sc, hr)); q[14] = float(h.lightIdx); q[15] = h.backSide ? 1.0f : 0.0f;
  }
}

// 2x2 footprint records of one texture (scene_types.hpp TexDev::quadOffset), expanded on the device at upload from the plain
// texel arrays: record (x, y) = the four taps texture.cpp:21-35 reads for a lookup whose base texel is (x, y)
struct TexQuadArgs { const uint8_t* u8; const float* f32; TexDev t; uint8_t* out; };
__global__ void __launch_bounds__(kBlock) k_tex_quads(TexQuadArgs a) {
  const uint32_t w = a.t.width, h = a.t.height, C = a.t.channels;
  const uint32_t rec = texQuadRecordBytes(C, a.t.isFloat);
  const uint32_t stride = a.t.quadStride ? a.t.quadStride : rec;     // (a record inside a material's shared 64-byte record)
  const size_t n = size_t(w) * h;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const uint32_t x = uint32_t(i % w), y = uint32_t(i / w);
    const uint32_t x1 = x + 1u < w ? x + 1u : x, y1 = y + 1u < h ? y + 1u : y;      // (records of the last column / row are never read)
    const size_t idx[4] = {size_t(y) * w + x, size_t(y1) * w + x, size_t(y) * w + x1, size_t(y1) * w + x1};
    uint8_t* o = a.out + i * stride;
    if (a.t.isFloat) {
      float* of = reinterpret_cast<float*>(o);
      for (uint32_t k = 0; k < rec / 4u; k++) of[k] = 0.0f;
      for (uint32_t tap = 0; tap < 4u; tap++)
        for (uint32_t c = 0; c < C; c++) of[tap * C + c] = a.f32[size_t(a.t.offset) + C * idx[tap] + c];
    } else {
      uint32_t word[4];
      for (uint32_t tap = 0; tap < 4u; tap++) {
        word[tap] = 0u;
        for (uint32_t c = 0; c < C; c++) word[tap] |= uint32_t(a.u8[size_t(a.t.offset) + C * idx[tap] + c]) << (8u * c);
      }
      uint32_t* ow = reinterpret_cast<uint32_t*>(o);
      if (C >= 3u) { ow[0] = word[0]; ow[1] = word[1]; ow[2] = word[2]; ow[3] = word[3]; }
      else if (C == 2u) { ow[0] = word[0] | (word[1] << 16); ow[1] = word[2] | (word[3] << 16); }
      else ow[0] = word[0] | (word[1] << 8) | (word[2] << 16) | (word[3] << 24);
    }
  }
}

#endif  // YART_TU == 0

#include "wavefront_kernels.inc"
#if YART_TU == 0
#include "bvh_build_device.inc"
#endif

#if YART_TU != 0
}  // namespace

// the kernels of this unit, for unit 0 (function pointers to the host stubs; the argument type is the same struct in every unit)
namespace yart_hip { namespace tu {
typedef void (*AnyKernel)();
#define YART_ANY(K) reinterpret_cast<AnyKernel>(static_cast<void (*)(WfArgs)>(K))
#define YART_PICK_LEAN(KERNEL, M)                                                                                             \
  (nodesForm == 4 ? (ident ? YART_ANY((KERNEL<(M) | TRAV_IDENTITY, 4>)) : YART_ANY((KERNEL<(M), 4>)))                         \
   : nodesForm == 3 ? (ident ? YART_ANY((KERNEL<(M) | TRAV_IDENTITY, 3>)) : YART_ANY((KERNEL<(M), 3>)))                       \
   : nodesForm == 2 ? (ident ? YART_ANY((KERNEL<(M) | TRAV_IDENTITY, 2>)) : YART_ANY((KERNEL<(M), 2>)))                       \
   : nodesForm == 1 ? (ident ? YART_ANY((KERNEL<(M) | TRAV_IDENTITY, 1>)) : YART_ANY((KERNEL<(M), 1>)))                       \
                    : (ident ? YART_ANY((KERNEL<(M) | TRAV_IDENTITY, 0>)) : YART_ANY((KERNEL<(M), 0>))))
#if YART_TU == 1
AnyKernel extendLean(int nodesForm, bool ident) { return YART_PICK_LEAN(k_wf_extend_lean, TRAV_FAST); }
#elif YART_TU == 2
AnyKernel shadowLean(int nodesForm, bool ident) { return YART_PICK_LEAN(k_wf_shadow_lean, TRAV_FAST); }
#elif YART_TU == 3
AnyKernel extendRetry(int nodesForm) {
  return nodesForm == 4 ? YART_ANY(k_wf_extend_retry_lean<4>) : nodesForm == 3 ? YART_ANY(k_wf_extend_retry_lean<3>)
       : nodesForm == 2 ? YART_ANY(k_wf_extend_retry_lean<2>) : nodesForm == 1 ? YART_ANY(k_wf_extend_retry_lean<1>) : YART_ANY(k_wf_extend_retry_lean<0>);
}
AnyKernel shadowRetry(int nodesForm) {
  return nodesForm == 4 ? YART_ANY(k_wf_shadow_retry_lean<4>) : nodesForm == 3 ? YART_ANY(k_wf_shadow_retry_lean<3>)
       : nodesForm == 2 ? YART_ANY(k_wf_shadow_retry_lean<2>) : nodesForm == 1 ? YART_ANY(k_wf_shadow_retry_lean<1>) : YART_ANY(k_wf_shadow_retry_lean<0>);
}
AnyKernel extendFast(bool ident) { return ident ? YART_ANY((k_wf_extend_fast<TRAV_FAST | TRAV_IDENTITY>)) : YART_ANY(k_wf_extend_fast<TRAV_FAST>); }
AnyKernel shadowFast(bool ident) { return ident ? YART_ANY((k_wf_shadow_fast<TRAV_FAST | TRAV_IDENTITY>)) : YART_ANY(k_wf_shadow_fast<TRAV_FAST>); }
AnyKernel extendGeneral(bool retry) { return retry ? YART_ANY(k_wf_extend<true>) : YART_ANY(k_wf_extend<false>); }
AnyKernel shadowGeneral(bool retry) { return retry ? YART_ANY(k_wf_shadow<true>) : YART_ANY(k_wf_shadow<false>); }
#elif YART_TU == 4
AnyKernel shade(bool sort, bool fit, bool env1) {
  return sort ? (env1 ? YART_ANY((k_wf_shade<true, true, true>)) : fit ? YART_ANY((k_wf_shade<true, true, false>)) : YART_ANY((k_wf_shade<true, false, false>)))
              : (env1 ? YART_ANY((k_wf_shade<false, true, true>)) : fit ? YART_ANY((k_wf_shade<false, true, false>)) : YART_ANY((k_wf_shade<false, false, false>)));
}
#if defined(YART_SHADE_REGIONS)
void shadeRegionsTake(unsigned long long* v48) {          // (measurement builds: the kernel's region counters live in this unit)
  (void)hipMemcpyFromSymbol(v48, HIP_SYMBOL(g_shadeRegion), 48 * sizeof(unsigned long long));
  const unsigned long long zero[48] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shadeRegion), zero, sizeof(zero));
}
#endif
#endif
#undef YART_PICK_LEAN
#undef YART_ANY
#if defined(YART_COUNT_TRAVERSAL)
// (instrumented build: every unit tallies the texel bytes of ITS kernels' lookups; unit 0 sums them)
#define YART_CAT2(a, b) a##b
#define YART_CAT(a, b) YART_CAT2(a, b)
void YART_CAT(texTapReset, YART_TU)() { const unsigned long long zero = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_texTapBytes), &zero, sizeof(zero)); }
unsigned long long YART_CAT(texTapRead, YART_TU)() { unsigned long long v = 0; (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_texTapBytes), sizeof(v)); return v; }
#endif
}}  // namespace yart_hip::tu

#else  // YART_TU == 0: everything below

}  // namespace
namespace yart_hip { namespace tu {
typedef void (*AnyKernel)();
AnyKernel extendLean(int nodesForm, bool ident);
AnyKernel shadowLean(int nodesForm, bool ident);
AnyKernel extendRetry(int nodesForm);
AnyKernel shadowRetry(int nodesForm);
AnyKernel extendFast(bool ident);
AnyKernel shadowFast(bool ident);
AnyKernel extendGeneral(bool retry);
AnyKernel shadowGeneral(bool retry);
AnyKernel shade(bool sort, bool fit, bool env1);
#if defined(YART_SHADE_REGIONS)
void shadeRegionsTake(unsigned long long* v48);
#endif
#if defined(YART_COUNT_TRAVERSAL)
void texTapReset1(); void texTapReset2(); void texTapReset3(); void texTapReset4();
unsigned long long texTapRead1(); unsigned long long texTapRead2(); unsigned long long texTapRead3(); unsigned long long texTapRead4();
#endif
}}
namespace {
typedef void (*WfKernelFn)(WfArgs);
inline WfKernelFn wfKernel(yart_hip::tu::AnyKernel k) { return reinterpret_cast<WfKernelFn>(k); }

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct Timer {
  hipEvent_t a{}, b{};
  Timer() { HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b)); }
  ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};

}  // namespace

struct YartScene {
  int device = 0;
  HostImage host;
  SceneDev dev{};
  int numCUs = 256;
  bool texQuadsOn = true;      // the textures' 2x2 footprint records are on the device (uploadScene: they fit the budget)
  // device copies of the scene image
  DevBuf<f4> resumeRec; DevBuf<ShadeTri> shadeTris; DevBuf<BvhNode> bvhNodes; DevBuf<LeafTri> leafTris; DevBuf<u4> triVerts; DevBuf<int32_t> triLight;
  DevBuf<f4> vPos, vNormal, vTangent; DevBuf<f2> vUV; DevBuf<MeshDev> meshes; DevBuf<NodeDev> nodes;
  DevBuf<MaterialDev> materials; DevBuf<TexDev> textures; DevBuf<uint8_t> texU8; DevBuf<float> texF32; DevBuf<uint8_t> texQuads;
  DevBuf<LightDev> lights; DevBuf<EnvDev> envs; DevBuf<float> envData; DevBuf<uint32_t> envGuide; DevBuf<f4> nodeWorld; DevBuf<TlasNode> tlas; DevBuf<unsigned long long> nodeBits;
  DevBuf<uint32_t> infiniteLights, areaLights; DevBuf<float> areaPowerCdf; DevBuf<float> lut;
  DevBuf<uint8_t> matClass;                // host_scene.hpp::lobeClass per material
  // render scratch (grown on demand, reused across calls)
  DevBuf<uint32_t> pixels; DevBuf<f4> L; DevBuf<uint32_t> cursor; DevBuf<unsigned long long> counters;
  DevBuf<uint32_t> pixRays, tileStart, tileCount; DevBuf<unsigned long long> tileRays;   // per-block ray counts (tile callbacks only)
  std::vector<unsigned long long> tileRaysHost;
  DevBuf<uint64_t> spill; DevBuf<float> hdr; DevBuf<uint32_t> probeIn; DevBuf<float> probeOut;
  DevBuf<f4> wf[9];                        // wavefront path state (wavefront.hpp::WfState)
  DevBuf<f4> wfTail[1][9];                 // compacted state of the late bounces (1/2 of the batch; the second one is the batch-sized state itself)
  DevBuf<uint32_t> wfTailMap[2];
  DevBuf<WfDyn> wfDyn;
  DevBuf<unsigned long long> pathsLog;     // paths entering bounce b, summed over the batches of a render (YartStats::paths_at_bounce)
  DevBuf<uint32_t> poolMap;                // path pool: the path (index of L) each slot carries, kWfFreeSlot = free
  uint32_t* poolHost = nullptr;            // pinned: the queue counters of the last rounds (the host's view of "is the batch done")
  ~YartScene() { if (poolHost) (void)hipHostFree(poolHost); }
  DevBuf<uint32_t> qA, qB, qS, qR, wfCounters; // wavefront queues
  DevBuf<uint64_t> smpEntries, smpHash; DevBuf<uint32_t> smpSobol1;   // SamplerTables of the current render
  std::vector<uint32_t> pixelsHost;
  struct TileRec { uint32_t x, y, w, h, start, count; };     // a pixel block of this rank: its rectangle and its range of pixelsHost
  std::vector<TileRec> tiles;
  unsigned long long lastCounters[32] = {0};
  uint32_t pixW = 0, pixH = 0, pixTile = 0, pixRank = 0, pixWorld = 0;
  std::mutex mu;
};

namespace {

void uploadScene(YartScene& s) {
  const HostImage& h = s.host;
  s.shadeTris.upload(h.shadeTris);
  s.bvhNodes.upload(h.bvhNodes); s.leafTris.upload(h.leafTris); s.triVerts.upload(h.triVerts);
  s.triLight.upload(h.triLight); s.vPos.upload(h.vPos); s.vNormal.upload(h.vNormal);
  s.vTangent.upload(h.vTangent); s.vUV.upload(h.vUV); s.meshes.upload(h.meshes); s.nodes.upload(h.nodes);
  s.materials.upload(h.materials); s.textures.upload(h.textures); s.texU8.upload(h.texU8);
  s.texF32.upload(h.texF32); s.lights.upload(h.lights); s.envs.upload(h.envs); s.envData.upload(h.envData); s.envGuide.upload(h.envGuide); s.nodeWorld.upload(h.nodeWorld); s.tlas.upload(h.tlas);
  s.infiniteLights.upload(h.infiniteLights); s.areaLights.upload(h.areaLights);
  s.areaPowerCdf.upload(h.areaPowerCdf); s.lut.upload(h.lut);
  {
    std::vector<uint8_t> cls;
    for (const MaterialDev& m : h.materials) cls.push_back(lobeClass(m));
    while (cls.size() % 4u) cls.push_back(0);          // (the shade kernel copies the classes into LDS a word at a time)
    s.matClass.upload(cls);
  }
  // The textures' 2x2 footprint records: expanded here from the plain texel arrays just uploaded — unless they do not fit a
  // budget: more than a third of the free device memory (or YART_TEX_QUADS_MAX_MB megabytes) and the kernels take their bilinear
  // taps from the plain texel arrays (bsdf.hpp::texQuad: sc.texQuads == nullptr; same taps, same arithmetic, same frame).
  bool quads = true;
  {
    size_t freeB = 0, totalB = 0;
    HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
    size_t budget = freeB / 3;
    if (const char* e = std::getenv("YART_TEX_QUADS_MAX_MB")) budget = size_t(std::max<long long>(0, std::atoll(e))) << 20;
    if (h.texQuadUnits * 16u > budget) quads = false;
  }
  s.texQuadsOn = quads;
  s.texQuads.ensure(quads ? std::max<size_t>(h.texQuadUnits, 1) * 16u : 16u);
  for (const TexDev& t : h.textures) {
    if (!quads || t.width == 0u || t.quadOffset == kNoTexQuads) continue;
    TexQuadArgs qa{s.texU8.p, s.texF32.p, t, s.texQuads.p + size_t(t.quadOffset) * 16u};
    const size_t n = size_t(t.width) * t.height;
    hipLaunchKernelGGL(k_tex_quads, dim3(uint32_t(std::min<size_t>((n + kBlock - 1) / kBlock, 65535u))), dim3(kBlock), 0, nullptr, qa);
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipDeviceSynchronize());
  SceneDev d = h.view();       // counts and totals; pointers replaced below
  d.shadeTris = s.shadeTris.p;
  d.bvhNodes = s.bvhNodes.p; d.leafTris = s.leafTris.p; d.triVerts = s.triVerts.p; d.triLight = s.triLight.p;
  d.vPos = s.vPos.p; d.vNormal = s.vNormal.p; d.vTangent = s.vTangent.p; d.vUV = s.vUV.p;
  d.meshes = s.meshes.p; d.nodes = s.nodes.p; d.materials = s.materials.p; d.textures = s.textures.p;
  d.texU8 = s.texU8.p; d.texF32 = s.texF32.p; d.texQuads = s.texQuadsOn ? s.texQuads.p : nullptr; d.lights = s.lights.p; d.envs = s.envs.p;
  d.envData = s.envData.p; d.envGuide = s.envGuide.p; d.nodeWorld = s.nodeWorld.p; d.tlas = s.tlas.p; d.nTlas = h.tlas.size() > 1 || (h.tlas.size() == 1 && h.tlas[0].b) ? uint32_t(h.tlas.size()) : 0u; d.infiniteLights = s.infiniteLights.p; d.areaLights = s.areaLights.p;
  d.areaPowerCdf = s.areaPowerCdf.p; d.lut = s.lut.p;
  s.dev = d;
}

int resolveDevice(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return -1;
  if (device < 0) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return -1;
    return cur;
  }
  return device < count ? device : -1;
}

// the device build of one mesh's BVH (same bytes as the host build; a mesh it refuses — NaN coordinates — is left to the host)
static bool deviceMeshBvh(void* ctx, const float* positions, uint32_t nVerts, const uint32_t* faces, uint32_t stride, uint32_t nFaces,
                          std::vector<BvhNode>& nodes, std::vector<uint32_t>& indices) {
  // any failure of the device build (out of device memory for its scratch arrays, ...) leaves the mesh to the host builder
  try {
    return devbvh::build(*static_cast<int*>(ctx), positions, nVerts, faces, stride, nFaces, nodes, indices, nullptr);
  } catch (const std::exception&) {
    (void)hipGetLastError();
    nodes.clear(); indices.clear();
    return false;
  }
}
YartScene* createScene(const YartSceneDesc& desc, int device, uint32_t sceneFlags = 0) {
  int dev = resolveDevice(device);
  if (dev < 0) throw HipError("no usable HIP device (libyart_hip has no CPU fallback)");
  HIP_CHECK(hipSetDevice(dev));
  auto s = std::make_unique<YartScene>();
  s->device = dev;
  // the meshes' BVHs are built on the device unless the caller (YART_SCENE_HOST_BVH) or the environment (YART_HOST_BVH) asks
  // for the host builder; the two give the same bytes, and every GPU test that compares a frame with the reference's checks it
  if ((sceneFlags & YART_SCENE_HOST_BVH) || std::getenv("YART_HOST_BVH")) s->host = buildHostImage(desc);
  else s->host = buildHostImage(desc, deviceMeshBvh, &dev);
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  s->numCUs = prop.multiProcessorCount;
  uploadScene(*s);
  return s.release();
}

uint32_t morton2(uint32_t x, uint32_t y) { return uint32_t(encodeMorton2(x, y)); }

// Pixels of the tiles this rank owns, tile-major (row-major inside a tile). Tiles are
// the reference's unit of parallel work (tile-renderer.hpp:126-144); across ranks they
// are dealt round-robin in Morton order.
std::vector<uint32_t> makePixelList(uint32_t W, uint32_t H, uint32_t tile, uint32_t rank, uint32_t world,
                                    std::vector<YartScene::TileRec>* tilesOut) {
  const uint32_t tx = (W + tile - 1) / tile, ty = (H + tile - 1) / tile;
  std::vector<std::pair<uint32_t, uint32_t>> order;   // (morton, linear tile)
  for (uint32_t y = 0; y < ty; y++)
    for (uint32_t x = 0; x < tx; x++) order.push_back({morton2(x, y), y * tx + x});
  std::sort(order.begin(), order.end());
  std::vector<uint32_t> pixels;
  if (tilesOut) tilesOut->clear();
  for (size_t k = 0; k < order.size(); k++) {
    if (k % world != rank) continue;
    const uint32_t x0 = (order[k].second % tx) * tile, y0 = (order[k].second / tx) * tile;
    const uint32_t x1 = std::min(W, x0 + tile), y1 = std::min(H, y0 + tile);
    if (tilesOut) tilesOut->push_back({x0, y0, x1 - x0, y1 - y0, uint32_t(pixels.size()), (x1 - x0) * (y1 - y0)});
    for (uint32_t y = y0; y < y1; y++)
      for (uint32_t x = x0; x < x1; x++) pixels.push_back(x | (y << 16));
  }
  return pixels;
}
void buildPixelList(YartScene& s, uint32_t W, uint32_t H, uint32_t tile, uint32_t rank, uint32_t world) {
  if (s.pixW == W && s.pixH == H && s.pixTile == tile && s.pixRank == rank && s.pixWorld == world) return;
  s.pixelsHost = makePixelList(W, H, tile, rank, world, &s.tiles);
  s.pixels.upload(s.pixelsHost);
  {
    std::vector<uint32_t> start, count;
    for (const YartScene::TileRec& t : s.tiles) { start.push_back(t.start); count.push_back(t.count); }
    s.tileStart.upload(start); s.tileCount.upload(count);
  }
  s.pixW = W; s.pixH = H; s.pixTile = tile; s.pixRank = rank; s.pixWorld = world;
}

void validate(const YartCameraDesc* cam, const YartRenderParams* p) {
  require(cam && p, "camera / params pointer is null");
  require(cam->width > 0 && cam->height > 0 && cam->width < 65536 && cam->height < 65536,
          "image size must be in [1, 65535]");
  require(p->samples > 0 && p->first_wave_samples > 0 && p->max_wave_samples > 0, "sample counts must be > 0");
  require(p->tile_size > 0 && p->tile_size <= 4096, "tile_size out of range");
  require(p->shard_tile <= 4096, "shard_tile out of range");
  require(p->world_size > 0 && p->rank < p->world_size, "rank / world_size");
  // (the path flags keep the depth in 8 bits and the count of unoccluded NEE rays in the next 8: wavefront.hpp WF_DEPTH_MASK / WF_NEE_MASK)
  require(p->max_depth > 0 && p->max_depth <= 255, "max_depth must be in [1, 255]");
  require(p->start_sample < p->samples && (p->stop_sample == 0 || (p->stop_sample > p->start_sample && p->stop_sample <= p->samples)),
          "start_sample / stop_sample out of range");
  require(p->estimator <= YART_ESTIMATOR_GMONB, "estimator must be one of YART_ESTIMATOR_*");
}

RenderConst makeRenderConst(const YartRenderParams& p) {
  RenderConst rc;
  // the sampler is constructed with the TOTAL sample count and the tile size
  // (tile-renderer.hpp:153-156)
  rc.sampler = makeSamplerConfig(p.samples, p.tile_size);
  rc.maxDepth = p.max_depth;
  rc.background = mk3(p.background[0], p.background[1], p.background[2]);
  return rc;
}

int persistentGrid(const YartScene& s, const void* kernel, int cap, int block = kBlock) {
  int perCU = 0;
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, block, 0));
  if (perCU < 1) perCU = 1;
  if (perCU > cap) perCU = cap;
  return s.numCUs * perCU;
}

// HIP-event stopwatch for one class of launches on the render stream: record(begin/end)
// around each launch, resolve() after the stream has been synchronised.
struct StageTimer {
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  double ms = 0.0;
  uint32_t launches = 0;
  ~StageTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
  hipEvent_t next() {
    if (used == ev.size()) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); ev.push_back(e); }
    return ev[used++];
  }
  void begin(hipStream_t st) { HIP_CHECK(hipEventRecord(next(), st)); }
  void end(hipStream_t st) { HIP_CHECK(hipEventRecord(next(), st)); launches++; }
  void resolve() {
    for (size_t i = 0; i + 1 < used; i += 2) {
      float t = 0; HIP_CHECK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      ms += t;
    }
    used = 0;
  }
};

// Called after every batch of a wave, once its pixels are final for that wave in the output frame (the stream has been
// synchronised): the range [c0, c0 + n) of this rank's pixel list. Returning true stops the render after this batch.
struct BatchInfo { uint32_t c0, n, wave, waveSamples, samplesTaken, totalSamples; };
typedef std::function<bool(const BatchInfo&)> BatchHook;

bool renderToDevice(YartScene& s, const YartCameraDesc& camDesc, const YartRenderParams& p, float* dOut,
                    hipStream_t stream, YartStats* stats, const BatchHook* hook = nullptr) {
  bool aborted = false;
  auto wall0 = std::chrono::high_resolution_clock::now();
  TraceRange rgRender("yart:render");
  HIP_CHECK(hipSetDevice(s.device));
  const uint32_t W = camDesc.width, H = camDesc.height;
  const CameraDev cam = makeCamera(camDesc);
  const RenderConst rc = makeRenderConst(p);
  const bool mega = (p.flags & YART_FLAG_MEGAKERNEL) != 0;
  uint32_t effFlags = p.flags;              // after the defaults (reported in YartStats::pipeline_flags)
  if (!mega && !(effFlags & YART_FLAG_NO_SHADE_SORT)) effFlags |= YART_FLAG_SHADE_SORT;
  buildPixelList(s, W, H, p.shard_tile ? p.shard_tile : p.tile_size, p.rank, p.world_size);
  const uint32_t nPix = uint32_t(s.pixelsHost.size());

  const uint64_t startSample = p.start_sample, stopSample = p.stop_sample ? p.stop_sample : p.samples;
  if (startSample == 0) HIP_CHECK(hipMemsetAsync(dOut, 0, size_t(W) * H * 4 * sizeof(float), stream));
  s.cursor.ensure(1); s.counters.ensure(kNumCounters); s.pathsLog.ensure(16);
  HIP_CHECK(hipMemsetAsync(s.pathsLog.p, 0, 16 * sizeof(unsigned long long), stream));
  HIP_CHECK(hipMemsetAsync(s.counters.p, 0, kNumCounters * sizeof(unsigned long long), stream));
#if defined(YART_COUNT_TRAVERSAL)
  { const unsigned long long zero = 0; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_texTapBytes), &zero, sizeof(zero))); tu::texTapReset1(); tu::texTapReset2(); tu::texTapReset3(); tu::texTapReset4(); }
#endif

  // lean traversal kernels when the scene allows them (every node transform chain the identity ->
  // identity-only variant); YART_FLAG_GENERAL_TRACE forces the general kernels for everything
  const bool general = (p.flags & YART_FLAG_GENERAL_TRACE) != 0;
  const bool ident = s.host.allIdentity;
  const bool refill = (p.flags & YART_FLAG_NO_REFILL) == 0;
  // trace_lean.hpp keeps one 64-bit node candidate mask per ray and uses the all-ones mask as its "new ray"
  // marker, which a ray that can reach all of exactly 64 nodes would keep: 64 nodes and more go to the chunked form
  const bool chunked = s.host.nodes.size() >= 64;
  // 64 nodes and more: candidate windows from the top-level hierarchy (3; measured the fastest form at every size from 65 to 4252
  // nodes, profiles/r2_many_nodes.txt). Without it (no mesh nodes, more than 16384 nodes = 2 KB of bitset per lane, or debug bit
  // 262144): chunked masks below kLeanWalkNodes nodes, the per-lane walk from there on (debug bit 65536: the walk at any size)
  const bool tlasOk = s.dev.nTlas != 0u && s.host.nodes.size() <= 16384u && !(effFlags & (262144u | 65536u));
  const bool leanLds = !chunked && s.host.nodes.size() <= kLeanSceneNodes && s.host.meshes.size() <= kLeanSceneNodes;
  const int nodesForm = !chunked ? (leanLds ? 4 : 0) : tlasOk ? 3 : ((effFlags & 65536u) || s.host.nodes.size() >= kLeanWalkNodes) ? 2 : 1;
  auto pickExtend = [&]() -> void (*)(WfArgs) {
    if (!refill) return wfKernel(tu::extendFast(ident));
    return wfKernel(tu::extendLean(nodesForm, ident));
  };
  auto pickShadow = [&]() -> void (*)(WfArgs) {
    if (!refill) return wfKernel(tu::shadowFast(ident));
    return wfKernel(tu::shadowLean(nodesForm, ident));
  };
  auto kExtendFast = pickExtend();
  auto kShadowFast = pickShadow();
  auto kRetryE = wfKernel(tu::extendRetry(nodesForm));
  auto kRetryS = wfKernel(tu::shadowRetry(nodesForm));
  auto kExtendGen = wfKernel(tu::extendGeneral(false)), kExtendGenRetry = wfKernel(tu::extendGeneral(true));
  auto kShadowGen = wfKernel(tu::shadowGeneral(false)), kShadowGenRetry = wfKernel(tu::shadowGeneral(true));
  const int gridMega = persistentGrid(s, reinterpret_cast<const void*>(k_render_mega), 3);
  const int gridExtendFast = persistentGrid(s, reinterpret_cast<const void*>(kExtendFast), 8);
  const int gridShadowFast = persistentGrid(s, reinterpret_cast<const void*>(kShadowFast), 8);
  const int gridExtend = persistentGrid(s, reinterpret_cast<const void*>(kExtendGen), 8);
  const int gridShadow = persistentGrid(s, reinterpret_cast<const void*>(kShadowGen), 8);
  // the shade kernel's LDS copies of the scene's small tables: FIT when the sampler tables are in use (below) and every table fits its slot
  const bool samplerTables = !mega && !(p.flags & YART_FLAG_DIRECT_SAMPLER) && nPix > 0 && uint64_t(p.samples) <= (1ull << rc.sampler.log2spp);
  const bool shadeFit = samplerTables && s.dev.nMaterials <= kShadeMatSlots && s.dev.nTextures <= kShadeTexSlots && s.dev.nLights <= kShadeLightSlots &&
                        s.dev.nEnvs <= kShadeEnvSlots && s.dev.nNodes <= kShadeNodeSlots && s.dev.nInfinite <= kShadeLightSlots;
  const bool envOnly = shadeFit && s.dev.nArea == 0u && s.dev.nInfinite == 1u && s.dev.nLights == 1u;    // (variant of the FIT kernels only)
  auto kShade = wfKernel(tu::shade((effFlags & YART_FLAG_SHADE_SORT) != 0, shadeFit, envOnly));
  const int gridShade = persistentGrid(s, reinterpret_cast<const void*>(kShade), 8, kShadeBlock);
  const int gridRetryE = persistentGrid(s, reinterpret_cast<const void*>(kRetryE), 8);
  const int gridRetryS = persistentGrid(s, reinterpret_cast<const void*>(kRetryS), 8);
  int gridMax = std::max(gridMega, std::max(gridExtend, gridShadow));
  gridMax = std::max(gridMax, std::max(gridExtendFast, gridShadowFast));
  gridMax = std::max(gridMax, std::max(gridRetryE, gridRetryS));
  s.spill.ensure(size_t(gridMax) * kBlock * kSpillDepthMax);
  // the lanes' node bitsets of the top-level-hierarchy form (trace_lean_tlas.hpp): all zero between launches
  const uint32_t nodeBitWords = nodesForm == 3 ? uint32_t((s.host.nodes.size() + 63u) / 64u) : 0u;
  if (nodeBitWords) {
    const size_t need = size_t(gridMax) * kBlock * nodeBitWords;
    if (s.nodeBits.n < need) { s.nodeBits.ensure(need); HIP_CHECK(hipMemsetAsync(s.nodeBits.p, 0, need * 8, stream)); }
  }

  // Batch = the pixels (x all samples of a wave) rendered together: max_batch_paths, by default kDefaultBatchPaths — a fixed
  // number, no longer a share of the free device memory (round 3 sized ONE batch to 60 % of it: 150 GB for the C3 frame); only if
  // that does not fit the device is it cut down to what does. The default pipeline is batch-synchronous: one slot per path of the
  // batch (251 bytes with the compacted tail state), every bounce one launch per stage over the paths still alive. What a smaller
  // batch costs (profiles/r4_ab_path_pool.txt): ~10 ms per batch of any size — the tail of every launch, when the GPU waits for
  // the slowest rays of the last waves — so the C3 frame in 2 / 4 / 8 / 16 batches is 1.4 / 4.5 / 9.7 / 19 % slower than in one.
  // YART_FLAG_PATH_POOL: the batch runs through a POOL of pool_paths slots with path regeneration instead (168 bytes per slot +
  // 16 per path of the batch): bounded memory at any frame size, every launch pool-sized until the batch runs out — and ~25 %
  // slower on the C3 frame, because a wave's lanes then hold paths of every generation (same file).
  const uint32_t maxWave = std::min(p.max_wave_samples, p.samples);
  const uint32_t waveCap = std::max(std::min(p.first_wave_samples, p.samples), maxWave);
  const bool pool = !mega && (p.flags & YART_FLAG_PATH_POOL) != 0;
  uint32_t resumeCap = 0;
  const bool compact = !mega && !pool && !(p.flags & YART_FLAG_NO_COMPACTION);
  uint64_t maxPaths = p.max_batch_paths ? p.max_batch_paths : kDefaultBatchPaths;
  uint64_t poolFit = ~0ull;                     // path pool: the slots the device can hold next to the batch's radiance records
  if (!mega) {
    // (safety only: a device that cannot hold the batch renders smaller ones)
    size_t freeB = 0, totalB = 0;
    HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
    if (const char* e = std::getenv("YART_FAKE_FREE_MB")) freeB = size_t(std::max<long long>(1, std::atoll(e))) << 20;   // (tests of the clamp)
    uint64_t held = uint64_t(s.L.n) * 16 + (uint64_t(s.qA.n) + s.qB.n + s.qS.n + s.qR.n) * 4 + uint64_t(s.resumeRec.n) * 16;
    for (auto& b : s.wf) held += uint64_t(b.n) * 16;
    for (auto& t : s.wfTail) for (auto& b : t) held += uint64_t(b.n) * 16;
    held += (uint64_t(s.wfTailMap[0].n) + s.wfTailMap[1].n) * 4;
    held += uint64_t(s.smpEntries.n) * 8;      // the sampler tables of the previous render stay allocated
    if (std::getenv("YART_FAKE_FREE_MB")) held = 0;
    // what grows with the batch: 9 x 16 B of path state + 4 queue words + 16 B of radiance = 176 B per path; with compaction two
    // a tail state of 1/2 of the batch (9 x 16 B + a slot map word) and a slot map of 1/4 = 75 B more; the resume records of an eighth of
    // the paths (kResumeWords x 16 B each = 24 B per path). What does not: the sampler tables (8 B x dims per PIXEL of the rank),
    // the resume records' per-wave ranges and the traversal spill area — taken off the budget first.
    const uint64_t perPath = (compact ? 251 : 176) + ((p.flags & YART_FLAG_NO_RESUME) ? 0 : (kResumeWords * 16 + 7) / 8);
    const uint64_t dimsEst = std::min<uint32_t>(256u, (4u + 8u * p.max_depth + 16u + 7u) & ~7u);
    const uint64_t fixedB = uint64_t(nPix) * dimsEst * 8 + uint64_t(gridMax) * kBlock * (kResumeWords * 16 + uint64_t(kSpillDepthMax) * 8);
    const uint64_t budget = (uint64_t(freeB) + held) * 8 / 10;
    const uint64_t avail = budget > fixedB ? budget - fixedB : 0;
    if (!pool) {
      const uint64_t fits = std::max<uint64_t>(avail / perPath, 1u << 16);
      maxPaths = std::min<uint64_t>(std::min<uint64_t>(maxPaths, fits), kWfMaxPaths);
    } else {
      // pool: 16 B of radiance per path of the batch + 168 B (+ resume records) per slot of the pool: the batch gets at most half
      // of the budget, the pool what is left
      const uint64_t fits = std::max<uint64_t>(avail / 2 / 16, 1u << 16);
      maxPaths = std::min<uint64_t>(maxPaths, fits);
      const uint64_t left = avail > std::min<uint64_t>(maxPaths, uint64_t(nPix ? nPix : 1) * waveCap) * 16 ? avail - std::min<uint64_t>(maxPaths, uint64_t(nPix ? nPix : 1) * waveCap) * 16 : 0;
      poolFit = std::max<uint64_t>(left / (176 + (kResumeWords * 16 + 7) / 8), 64);
    }
  }
  maxPaths = std::min<uint64_t>(maxPaths, (1ull << 31) - 64);
  uint32_t chunk = uint32_t(std::min<uint64_t>(nPix ? nPix : 1, std::max<uint64_t>(maxPaths / waveCap, 1)));
  if (nPix > chunk) {                            // batches of equal size (the last one is not a sliver)
    const uint32_t nb = (nPix + chunk - 1) / chunk;
    chunk = (nPix + nb - 1) / nb;
  }
  s.L.ensure(size_t(chunk) * waveCap);
  uint32_t poolSlots = 0;
  if (!mega) {
    const size_t npBatch = size_t(chunk) * waveCap;
    size_t np = npBatch;
    if (pool) {
      const size_t want = std::min<uint64_t>(p.pool_paths ? p.pool_paths : kDefaultPoolPaths, poolFit);
      np = std::max<size_t>(64, (std::min(want, npBatch) + 63) & ~size_t(63));
      poolSlots = uint32_t(np);
      s.poolMap.ensure(np);
      if (!s.poolHost) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.poolHost), kPoolLag * WC_COUNT * sizeof(uint32_t)));
    }
    for (auto& b : s.wf) b.ensure(np);
    if (compact) {
      // Two dense "tail" states take the survivors in turn: the first is an array set of half the batch; the second is the BATCH-SIZED
      // state itself — by the time a second compaction happens the paths live in the first tail and the large arrays hold nothing
      // that is still read, so the survivors go back to their front (round 5: 36 B per path less than a third array set).
      for (auto& b : s.wfTail[0]) b.ensure(np / 2 + 64);
      s.wfTailMap[0].ensure(np / 2 + 64);
      s.wfTailMap[1].ensure(np / 4 + 64);
      s.wfDyn.ensure(1);
    }
    s.qA.ensure(np); s.qB.ensure(np); s.qS.ensure(np); s.qR.ensure(np); s.wfCounters.ensure(WC_COUNT);
    // resume records of the rays the lean kernels hand to the general ones (traverse.hpp: 192 B each): room for an eighth of
    // the slots (C3 hands over 6-7 % of its rays; a ray that finds no record is restarted, as all of them were before).
    // YART_RESUME_CAP: records (tests of the fallback), 0 = restarts only.
    if (!(p.flags & YART_FLAG_NO_RESUME)) {
      // (+ one range of 64 per wave of the largest grid: a wave takes its records 64 at a time and may leave a range unfinished)
      size_t cap = std::max<size_t>(np / 8, std::min<size_t>(np, 1u << 16)) + size_t(gridMax) * kBlock;
      if (const char* e = std::getenv("YART_RESUME_CAP")) cap = size_t(std::max<long long>(0, std::atoll(e)));
      cap = std::min<size_t>(cap, 0x7fffff00u);
      if (cap) s.resumeRec.ensure(cap * kResumeWords);
      resumeCap = uint32_t(cap);
    }
  }

  Timer tAll;
  StageTimer tMega, tExtend, tShade, tConnect, tGmon, tLean, tShadeK, tShadowLean;
  uint32_t waves = 0;
  uint64_t renderedSamples = 0;
  HIP_CHECK(hipEventRecord(tAll.a, stream));

  // sampler tables (sampler.hpp::SamplerTables) for the wavefront pipeline; they require every
  // sample index to fit the sampler's log2spp bits (log2Int rounds to nearest, e.g. 90 spp -> 6)
  RenderConst rcw = rc;
  if (samplerTables) {
    const uint32_t dims = std::min<uint32_t>(256u, (4u + 8u * p.max_depth + 16u + 7u) & ~7u);     // (+ 3 <= kShadeHashSlots)
    s.smpEntries.ensure(size_t(dims) * nPix); s.smpHash.ensure(dims + 3); s.smpSobol1.ensure(8 * 256);
    SamplerTabArgs ta{};
    ta.cfg = rc.sampler; ta.pixels = s.pixels.p; ta.nPixels = nPix; ta.dims = dims;
    ta.entries = s.smpEntries.p; ta.hash = s.smpHash.p; ta.sobol1 = s.smpSobol1.p;
    ta.matrix52 = reinterpret_cast<const uint32_t*>(s.dev.lut + LutDev::sobol);
    TraceRange rg("yart:sampler_tables", stream);
    tShade.begin(stream);
    hipLaunchKernelGGL(k_sampler_tables, dim3(s.numCUs * 8), dim3(kBlock), 0, stream, ta);
    HIP_CHECK(hipGetLastError());
    tShade.end(stream);
    rcw.sampler.tab.entries = s.smpEntries.p; rcw.sampler.tab.hash = s.smpHash.p; rcw.sampler.tab.sobol1 = s.smpSobol1.p;
    rcw.sampler.tab.dims = dims; rcw.sampler.tab.stride = nPix;
  }

  // wave schedule of tile-renderer.hpp:121-124, 284-289
  uint64_t remaining = p.samples;
  uint64_t waveSamples = std::min<uint64_t>(p.first_wave_samples, p.samples);
  uint64_t currentWave = 0;
  while (waveSamples > 0) {
    const uint64_t takenBefore = p.samples - remaining, takenAfter = takenBefore + waveSamples;
    const float wCurrent = float(takenBefore) / float(takenAfter);
    const float wWave = float(waveSamples) / float(takenAfter);
    // resumable accumulation: waves outside [start_sample, stop_sample) are not rendered by this call
    const bool inRange = takenBefore >= startSample && takenBefore < stopSample;
    if (!inRange && takenBefore < startSample && takenAfter > startSample) throw std::invalid_argument("start_sample is not a wave boundary");
    if (inRange && takenAfter > stopSample) throw std::invalid_argument("stop_sample is not a wave boundary");
    if (inRange) { waves++; renderedSamples += waveSamples; }
    size_t tileDone = 0;                     // blocks of this wave whose ray counts have been summed (tile callbacks only)
    for (uint32_t c0 = 0; inRange && !aborted && c0 < nPix; c0 += chunk) {
      const uint32_t n = std::min(chunk, nPix - c0);
      if (mega) {
        HIP_CHECK(hipMemsetAsync(s.cursor.p, 0, sizeof(uint32_t), stream));
        MegaArgs a{};
        a.sc = s.dev; a.cam = cam; a.rc = rc; a.pixels = s.pixels.p + c0; a.nPixels = n;
        a.spp = uint32_t(waveSamples); a.sampleOffset = uint32_t(takenBefore); a.L = s.L.p;
        a.cursor = s.cursor.p; a.rays = s.counters.p; a.spill = s.spill.p;
        TraceRange rgMega("yart:megakernel", stream);
        tMega.begin(stream);
        hipLaunchKernelGGL(k_render_mega, dim3(gridMega), dim3(kBlock), 0, stream, a);
        HIP_CHECK(hipGetLastError());
        tMega.end(stream);
      } else if (pool) {
        WfArgs a{};
        a.sc = s.dev; a.cam = cam; a.rc = rcw; a.pixBase = c0;
        a.st.ray0 = s.wf[0].p; a.st.ray1 = s.wf[1].p; a.st.thr = s.wf[2].p; a.st.acc = s.wf[3].p;
        a.st.hit0 = s.wf[4].p; a.st.hit1 = s.wf[5].p; a.st.sh0 = s.wf[6].p; a.st.sh1 = s.wf[7].p; a.st.sh2 = s.wf[8].p;
        a.qA = s.qA.p; a.qB = s.qB.p; a.qS = s.qS.p; a.qR = s.qR.p; a.counters = s.wfCounters.p;
        a.pixels = s.pixels.p + c0; a.nPaths = n * uint32_t(waveSamples); a.spp = uint32_t(waveSamples);
        a.sampleOffset = uint32_t(takenBefore); a.L = s.L.p; a.stats = s.counters.p; a.spill = s.spill.p;
        a.matClass = s.matClass.p;
        a.resumeRec = resumeCap ? s.resumeRec.p : nullptr; a.resumeCap = resumeCap;
        a.sc.nodeBits = s.nodeBits.p; a.sc.nodeBitWords = nodeBitWords;
        a.slotMap = s.poolMap.p;
        a.poolSlots = uint32_t(std::min<uint64_t>(poolSlots, (uint64_t(a.nPaths) + 63u) & ~uint64_t(63)));
        const uint32_t init[WC_COUNT] = {a.poolSlots, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(s.wfCounters.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_wf_pool_init, dim3(s.numCUs * 4), dim3(kBlock), 0, stream, s.poolMap.p, a.poolSlots);
        HIP_CHECK(hipGetLastError());
        // Rounds: every round starts new paths in the free slots and takes every live path one bounce further. The host does not
        // know when the batch is done; it reads the counters of the round before the previous one (a copy into pinned memory and
        // an event per round, never waited for) and stops launching when that round left no live path and nothing to start:
        // two or three empty rounds at the end instead of a host synchronisation per round.
        struct RoundEvents {                      // (destroyed on every way out of the round loop, a thrown HipError included)
          hipEvent_t ev[kPoolLag] = {};
          ~RoundEvents() { for (auto e : ev) if (e) (void)hipEventDestroy(e); }
        } roundEv;
        hipEvent_t* evRound = roundEv.ev;
        for (int k = 0; k < kPoolLag; k++) HIP_CHECK(hipEventCreateWithFlags(&evRound[k], hipEventDisableTiming));
        const uint64_t maxRounds = (uint64_t(a.nPaths) / a.poolSlots + 2u) * (rc.maxDepth + 1u) + 16u;   // (a path lives at most maxDepth rounds)
        bool done = false;
        for (uint64_t round = 0; !done; round++) {
          if (round > maxRounds) throw HipError("path pool: the batch did not finish within its bound of rounds");
          tShade.begin(stream);
          hipLaunchKernelGGL(k_wf_refill, dim3(s.numCUs * YART_STREAM_BLOCKS), dim3(kBlock), 0, stream, a);
          HIP_CHECK(hipGetLastError());
          tShade.end(stream);
          {   // the live slots of this round, for the host (WC_NEXT: k_wf_refill)
            uint32_t* snap = s.poolHost + size_t(round % kPoolLag) * WC_COUNT;
            HIP_CHECK(hipMemcpyAsync(snap, s.wfCounters.p, WC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipEventRecord(evRound[round % kPoolLag], stream));
          }
          tExtend.begin(stream);
          if (general) {
            hipLaunchKernelGGL(kExtendGen, dim3(gridExtend), dim3(kBlock), 0, stream, a);
          } else {
            tLean.begin(stream);
            hipLaunchKernelGGL(kExtendFast, dim3(gridExtendFast), dim3(kBlock), 0, stream, a);
            tLean.end(stream);
            if (refill) hipLaunchKernelGGL(kRetryE, dim3(gridRetryE), dim3(kBlock), 0, stream, a);
            else hipLaunchKernelGGL(kExtendGenRetry, dim3(gridExtend), dim3(kBlock), 0, stream, a);
            hipLaunchKernelGGL(k_wf_reset_retry, dim3(1), dim3(64), 0, stream, s.wfCounters.p);
          }
          HIP_CHECK(hipGetLastError());
          tExtend.end(stream);
          tShade.begin(stream);
          tShadeK.begin(stream);
          hipLaunchKernelGGL(kShade, dim3(gridShade), dim3(kShadeBlock), 0, stream, a);
          HIP_CHECK(hipGetLastError());
          tShadeK.end(stream);
          tShade.end(stream);
          tConnect.begin(stream);
          if (general) {
            hipLaunchKernelGGL(kShadowGen, dim3(gridShadow), dim3(kBlock), 0, stream, a);
          } else {
            tShadowLean.begin(stream);
            hipLaunchKernelGGL(kShadowFast, dim3(gridShadowFast), dim3(kBlock), 0, stream, a);
            tShadowLean.end(stream);
            if (refill) hipLaunchKernelGGL(kRetryS, dim3(gridRetryS), dim3(kBlock), 0, stream, a);
            else hipLaunchKernelGGL(kShadowGenRetry, dim3(gridShadow), dim3(kBlock), 0, stream, a);
          }
          HIP_CHECK(hipGetLastError());
          tConnect.end(stream);
          tShade.begin(stream);
          hipLaunchKernelGGL(k_wf_roulette, dim3(s.numCUs * YART_STREAM_BLOCKS), dim3(kBlock), 0, stream, a);
          HIP_CHECK(hipGetLastError());
          tShade.end(stream);
          hipLaunchKernelGGL(k_wf_pool_advance, dim3(1), dim3(64), 0, stream, s.wfCounters.p, a.poolSlots);
          HIP_CHECK(hipGetLastError());
          if (round >= 2) {
            const uint64_t old = round - 2;
            // (rounds far ahead of the device would only queue launches: wait for the round before the previous one)
            HIP_CHECK(hipEventSynchronize(evRound[old % kPoolLag]));
            const uint32_t* c = s.poolHost + size_t(old % kPoolLag) * WC_COUNT;
            if (c[WC_NEXT] == 0u) done = true;                  // no live slot after its refill: nothing left to start either
          }
        }
      } else {
        WfArgs a{};
        a.sc = s.dev; a.cam = cam; a.rc = rcw; a.pixBase = c0;
        a.st.ray0 = s.wf[0].p; a.st.ray1 = s.wf[1].p; a.st.thr = s.wf[2].p; a.st.acc = s.wf[3].p;
        a.st.hit0 = s.wf[4].p; a.st.hit1 = s.wf[5].p; a.st.sh0 = s.wf[6].p; a.st.sh1 = s.wf[7].p; a.st.sh2 = s.wf[8].p;
        a.qA = s.qA.p; a.qB = s.qB.p; a.qS = s.qS.p; a.qR = s.qR.p; a.counters = s.wfCounters.p;
        a.pixels = s.pixels.p + c0; a.nPaths = n * uint32_t(waveSamples); a.spp = uint32_t(waveSamples);
        a.sampleOffset = uint32_t(takenBefore); a.L = s.L.p; a.stats = s.counters.p; a.spill = s.spill.p;
        a.matClass = s.matClass.p;
        a.resumeRec = resumeCap ? s.resumeRec.p : nullptr; a.resumeCap = resumeCap;
        a.sc.nodeBits = s.nodeBits.p; a.sc.nodeBitWords = nodeBitWords;
        if (compact) {
          for (int t = 0; t < 2; t++) {
            f4** f = &a.tail[t].ray0;                  // the nine pointers of WfState, in declaration order
            for (int k = 0; k < 9; k++) f[k] = t == 0 ? s.wfTail[0][k].p : s.wf[k].p;      // (second tail: the batch-sized state, see above)
            a.tailMap[t] = s.wfTailMap[t].p;
            a.tailCap[t] = a.nPaths / (t == 0 ? 2u : 4u);
          }
          WfDyn d0{};
          d0.st = a.st; d0.slotMap = nullptr; d0.extent = a.nPaths; d0.inTail = 0;
          HIP_CHECK(hipMemcpyAsync(s.wfDyn.p, &d0, sizeof(d0), hipMemcpyHostToDevice, stream));
          a.dyn = s.wfDyn.p;
        }
        const uint32_t init[WC_COUNT] = {a.nPaths, 0, 0, 0, 0, 0, 0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(s.wfCounters.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
        TraceRange rgGen("yart:generate", stream);
        tShade.begin(stream);
        hipLaunchKernelGGL(k_wf_generate, dim3(s.numCUs * YART_STREAM_BLOCKS), dim3(kBlock), 0, stream, a);
        HIP_CHECK(hipGetLastError());
        tShade.end(stream);
        rgGen.end();
        for (uint32_t bounce = 0; bounce < rc.maxDepth; bounce++) {
          a.bounce = bounce;
          char rgName[32]; std::snprintf(rgName, sizeof(rgName), "yart:bounce %u", bounce);
          TraceRange rgBounce(rgName);
          TraceRange rgExtend("yart:extend", stream);
          tExtend.begin(stream);
          if (general) {
            hipLaunchKernelGGL(kExtendGen, dim3(gridExtend), dim3(kBlock), 0, stream, a);
          } else {
            tLean.begin(stream);
            hipLaunchKernelGGL(kExtendFast, dim3(gridExtendFast), dim3(kBlock), 0, stream, a);
            tLean.end(stream);
            if (refill) hipLaunchKernelGGL(kRetryE, dim3(gridRetryE), dim3(kBlock), 0, stream, a);
            else hipLaunchKernelGGL(kExtendGenRetry, dim3(gridExtend), dim3(kBlock), 0, stream, a);
            hipLaunchKernelGGL(k_wf_reset_retry, dim3(1), dim3(64), 0, stream, s.wfCounters.p);
          }
          HIP_CHECK(hipGetLastError());
          tExtend.end(stream);
          rgExtend.end();
          TraceRange rgShade("yart:shade", stream);
          tShade.begin(stream);
          tShadeK.begin(stream);
          hipLaunchKernelGGL(kShade, dim3(gridShade), dim3(kShadeBlock), 0, stream, a);
          HIP_CHECK(hipGetLastError());
          tShadeK.end(stream);
          tShade.end(stream);
          rgShade.end();
          TraceRange rgShadow("yart:shadow", stream);
          tConnect.begin(stream);
          if (general) {
            hipLaunchKernelGGL(kShadowGen, dim3(gridShadow), dim3(kBlock), 0, stream, a);
          } else {
            tShadowLean.begin(stream);
            hipLaunchKernelGGL(kShadowFast, dim3(gridShadowFast), dim3(kBlock), 0, stream, a);
            tShadowLean.end(stream);
            if (refill) hipLaunchKernelGGL(kRetryS, dim3(gridRetryS), dim3(kBlock), 0, stream, a);
            else hipLaunchKernelGGL(kShadowGenRetry, dim3(gridShadow), dim3(kBlock), 0, stream, a);
          }
          HIP_CHECK(hipGetLastError());
          tConnect.end(stream);
          rgShadow.end();
          // Russian roulette of the paths that cast a shadow ray: from the second bounce on (at depth 1 none applies: k_wf_shade
          // queued them itself), not after the last one (their radiance has been written out by the shadow kernels)
          if (bounce >= 1 && bounce + 1 < rc.maxDepth) {
            TraceRange rgR("yart:roulette", stream);
            tShade.begin(stream);
            hipLaunchKernelGGL(k_wf_roulette, dim3(s.numCUs * YART_STREAM_BLOCKS), dim3(kBlock), 0, stream, a);
            HIP_CHECK(hipGetLastError());
            tShade.end(stream);
          }
          hipLaunchKernelGGL(k_wf_advance, dim3(1), dim3(64), 0, stream, s.wfCounters.p, bounce + 1 < 16u ? s.pathsLog.p + bounce + 1 : nullptr);
          HIP_CHECK(hipGetLastError());
          std::swap(a.qA, a.qB);
          if (compact && bounce >= 1 && bounce + 1 < rc.maxDepth) {     // Russian roulette starts thinning at depth 2
            TraceRange rgC("yart:compact", stream);
            tShade.begin(stream);
            hipLaunchKernelGGL(k_wf_compact, dim3(s.numCUs * YART_STREAM_BLOCKS), dim3(kBlock), 0, stream, a);
            hipLaunchKernelGGL(k_wf_compact_commit, dim3(1), dim3(64), 0, stream, a);
            HIP_CHECK(hipGetLastError());
            tShade.end(stream);
          }
        }
      }
      GmonArgs g{};
      if (hook && *hook) { s.pixRays.ensure(std::max<uint32_t>(nPix, 1u)); g.pixRays = s.pixRays.p + c0; }
      g.L = s.L.p; g.pixels = s.pixels.p + c0; g.nPixels = n; g.spp = uint32_t(waveSamples); g.width = W;
      g.exposureScale = cam.exposureScale; g.wCurrent = wCurrent; g.wWave = wWave; g.hdr = dOut;
      g.kind = int(p.estimator);
      TraceRange rgBlend("yart:gmon_blend", stream);
      tGmon.begin(stream);
      hipLaunchKernelGGL(k_gmon_blend, dim3((n + kGmonPixPerBlock - 1) / kGmonPixPerBlock), dim3(kBlock), 0, stream, g);
      HIP_CHECK(hipGetLastError());
      tGmon.end(stream);
      HIP_CHECK(hipStreamSynchronize(stream));
      rgBlend.end();
      tMega.resolve(); tExtend.resolve(); tShade.resolve(); tConnect.resolve(); tGmon.resolve(); tLean.resolve(); tShadeK.resolve(); tShadowLean.resolve();
      if (hook && *hook) {
        // ray counts of the blocks this batch completed (their pixels' counts of this wave are all in pixRays now)
        size_t done = tileDone;
        while (done < s.tiles.size() && s.tiles[done].start + s.tiles[done].count <= c0 + n) done++;
        if (done > tileDone) {
          s.tileRays.ensure(s.tiles.size()); s.tileRaysHost.resize(s.tiles.size());
          hipLaunchKernelGGL(k_tile_rays, dim3(uint32_t(done - tileDone)), dim3(64), 0, stream, s.pixRays.p, s.tileStart.p, s.tileCount.p,
                             uint32_t(tileDone), uint32_t(done - tileDone), s.tileRays.p);
          HIP_CHECK(hipGetLastError());
          HIP_CHECK(hipMemcpyAsync(s.tileRaysHost.data() + tileDone, s.tileRays.p + tileDone, (done - tileDone) * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, stream));
          HIP_CHECK(hipStreamSynchronize(stream));
          tileDone = done;
        }
        const BatchInfo bi{c0, n, uint32_t(currentWave), uint32_t(waveSamples), uint32_t(takenAfter), p.samples};
        if ((*hook)(bi)) aborted = true;
      }
    }
    if (aborted) break;
    remaining -= waveSamples;
    uint64_t next = (currentWave > 0 || waveSamples > 1) ? std::min<uint64_t>(waveSamples * 2, p.max_wave_samples) : 1;
    waveSamples = std::min(next, remaining);
    currentWave++;
  }
  HIP_CHECK(hipEventRecord(tAll.b, stream));
  HIP_CHECK(hipEventSynchronize(tAll.b));
  float msAll = 0; HIP_CHECK(hipEventElapsedTime(&msAll, tAll.a, tAll.b));
  unsigned long long cnt[kNumCounters] = {0};
  HIP_CHECK(hipMemcpy(cnt, s.counters.p, sizeof(cnt), hipMemcpyDeviceToHost));
  for (int i = 0; i < kNumCounters; i++) s.lastCounters[i] = cnt[i];
  if (stats) {
    *stats = YartStats{};
    stats->samples = uint64_t(nPix) * renderedSamples;
    stats->rays = cnt[0];
    stats->traversals = cnt[1]; stats->box_tests = cnt[2]; stats->tri_tests = cnt[3]; stats->shaded_hits = cnt[4];
    stats->ms_device = msAll;
    stats->ms_traverse = mega ? tMega.ms : tExtend.ms + tConnect.ms;
    stats->launches_traverse = mega ? tMega.launches : tExtend.launches + tConnect.launches;
    stats->ms_extend = tExtend.ms; stats->ms_shade = tShade.ms; stats->ms_connect = tConnect.ms; stats->ms_gmon = tGmon.ms;
    stats->ms_extend_lean = tLean.ms; stats->launches_extend_lean = tLean.launches;
    stats->lean_traversals = cnt[5]; stats->lean_box_tests = cnt[6]; stats->lean_tri_tests = cnt[7];
    stats->ms_shade_kernel = tShadeK.ms; stats->launches_shade_kernel = tShadeK.launches;
    stats->ms_shadow_lean = tShadowLean.ms; stats->launches_shadow_lean = tShadowLean.launches;
    stats->shadow_lean_traversals = cnt[24]; stats->shadow_lean_box_tests = cnt[25]; stats->shadow_lean_tri_tests = cnt[26];
    stats->shade_entries = cnt[28]; stats->retry_extend_traversals = cnt[29]; stats->retry_shadow_traversals = cnt[30];
    stats->pipeline_flags = effFlags;
    {
      unsigned long long pl[16] = {0};
      HIP_CHECK(hipMemcpy(pl, s.pathsLog.p, sizeof(pl), hipMemcpyDeviceToHost));
      for (int b = 0; b < 16; b++) stats->paths_at_bounce[b] = pl[b];
      stats->paths_at_bounce[0] = mega || pool ? 0 : uint64_t(nPix) * renderedSamples;     // (every path enters bounce 0)
    }
#if defined(YART_COUNT_TRAVERSAL)
    {
      unsigned long long tb = 0;
      HIP_CHECK(hipMemcpyFromSymbol(&tb, HIP_SYMBOL(g_texTapBytes), sizeof(tb)));
      tb += tu::texTapRead1() + tu::texTapRead2() + tu::texTapRead3() + tu::texTapRead4();
      stats->texture_tap_bytes = tb;
    }
#endif
    stats->launches_extend = tExtend.launches; stats->launches_connect = tConnect.launches;
    stats->waves = waves;
    stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - wall0).count();
  }
  return aborted;
}

struct RcclFailure : std::runtime_error { using std::runtime_error::runtime_error; };

template <class F>
int guarded(F&& f) {
  try {
    f();
    return YART_OK;
  } catch (const RcclFailure& e) {
    g_lastError = e.what(); return YART_E_RCCL;
  } catch (const std::invalid_argument& e) {
    g_lastError = e.what(); return YART_E_INVALID;
  } catch (const HipError& e) {
    g_lastError = e.what();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return YART_E_NO_DEVICE;
    return YART_E_HIP;
  } catch (const std::exception& e) {
    g_lastError = e.what(); return YART_E_IO;
  }
}

}  // namespace

namespace {
std::unique_ptr<LoadedScene> importGltf(const char* path, const YartImportOptions* opts) {
  auto loaded = gltf::loadGltf(path);
  if (opts) {
    const float radius = opts->env_radius > 0.0f ? opts->env_radius : 100.0f;      // frontend main.cpp:82
    if (opts->env_hdr_path && opts->env_hdr_path[0]) gltf::addImageEnvironment(*loaded, opts->env_hdr_path, radius);
    if (opts->uniform_env) gltf::addUniformEnvironment(*loaded, opts->uniform_emission, radius);
  }
  return loaded;
}
}  // namespace

#include "multi_device.inc"

extern "C" {

int yart_hip_abi_version(void) { return YART_HIP_ABI_VERSION; }

int yart_hip_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

const char* yart_hip_last_error(void) { return g_lastError.c_str(); }

int yart_hip_scene_create(const YartSceneDesc* desc, int device, YartScene** out) {
  return guarded([&] {
    require(desc && out, "desc / out pointer is null");
    *out = createScene(*desc, device);
  });
}

int yart_hip_scene_create_flags(const YartSceneDesc* desc, int device, uint32_t scene_flags, YartScene** out) {
  return guarded([&] {
    require(desc && out, "desc / out pointer is null");
    *out = createScene(*desc, device, scene_flags);
  });
}

int yart_hip_scene_load(const char* path, int device, YartScene** out) {
  return guarded([&] {
    require(path && out, "path / out pointer is null");
    auto loaded = loadSceneFile(path);
    *out = createScene(loaded->desc, device);
  });
}

int yart_hip_scene_load_gltf(const char* path, const YartImportOptions* opts, int device, YartScene** out) {
  return guarded([&] {
    require(path && out, "path / out pointer is null");
    auto loaded = importGltf(path, opts);
    *out = createScene(loaded->desc, device);
  });
}

int yart_hip_gltf_to_yscn(const char* gltf_path, const YartImportOptions* opts, const char* yscn_path) {
  return guarded([&] {
    require(gltf_path && yscn_path, "path pointer is null");
    auto loaded = importGltf(gltf_path, opts);
    saveSceneFile(loaded->desc, yscn_path);
  });
}

void yart_hip_scene_destroy(YartScene* scene) {
  if (!scene) return;
  (void)hipSetDevice(scene->device);
  delete scene;
}

int yart_hip_render_device(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                           float* d_out_rgba, void* stream, YartStats* stats) {
  return guarded([&] {
    require(scene && d_out_rgba, "scene / output pointer is null");
    validate(cam, params);
    std::lock_guard<std::mutex> lock(scene->mu);
    renderToDevice(*scene, *cam, *params, d_out_rgba, static_cast<hipStream_t>(stream), stats);
  });
}

int yart_hip_render(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                    float* out_rgba, YartStats* stats) {
  return guarded([&] {
    require(scene && out_rgba, "scene / output pointer is null");
    validate(cam, params);
    std::lock_guard<std::mutex> lock(scene->mu);
    auto t0 = std::chrono::high_resolution_clock::now();
    HIP_CHECK(hipSetDevice(scene->device));
    const size_t n = size_t(cam->width) * cam->height * 4;
    scene->hdr.ensure(n);
    if (params->start_sample > 0) HIP_CHECK(hipMemcpy(scene->hdr.p, out_rgba, n * sizeof(float), hipMemcpyHostToDevice));
    renderToDevice(*scene, *cam, *params, scene->hdr.p, nullptr, stats);
    HIP_CHECK(hipMemcpy(out_rgba, scene->hdr.p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (stats)
      stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
  });
}

// One wave of the schedule at a time, and inside a wave one batch at a time (tile-renderer.hpp:200-309: finishTile
// fires onRenderTileComplete per tile and onRenderWaveComplete after a wave's last tile).
static int renderProgressive(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params, float* out_rgba,
                             YartStats* stats, YartWaveCallback on_wave, YartTileCallback on_tile, void* user) {
  bool aborted = false;
  const int rc = guarded([&] {
    require(scene && out_rgba, "scene / output pointer is null");
    validate(cam, params);
    std::lock_guard<std::mutex> lock(scene->mu);
    auto t0 = std::chrono::high_resolution_clock::now();
    HIP_CHECK(hipSetDevice(scene->device));
    const uint32_t W = cam->width, H = cam->height;
    const size_t n = size_t(W) * H * 4;
    scene->hdr.ensure(n);
    const uint32_t stop = params->stop_sample ? params->stop_sample : params->samples;
    if (params->start_sample > 0) HIP_CHECK(hipMemcpy(scene->hdr.p, out_rgba, n * sizeof(float), hipMemcpyHostToDevice));
    else if (on_tile) std::memset(out_rgba, 0, n * sizeof(float));       // tiles arrive one by one: the rest of the frame is defined
    YartStats total{};
    // the wave schedule of tile-renderer.hpp:264-289 (renderToDevice walks the same one): w0 = min(first, samples),
    // then min(2 w, max) — a first wave of one sample is followed by another single one
    uint64_t remaining = params->samples, wave = 0;
    uint64_t waveSamples = std::min<uint64_t>(params->first_wave_samples, params->samples);
    while (waveSamples > 0 && !aborted) {
      const uint32_t taken = uint32_t(params->samples - remaining);
      if (taken >= params->start_sample && taken < stop) {
        YartRenderParams q = *params;
        q.start_sample = taken; q.stop_sample = uint32_t(taken + waveSamples);
        YartStats st{};
        auto tw = std::chrono::high_resolution_clock::now();
        size_t nextTile = 0;                 // tiles are in pixel-list order: everything before nextTile has been reported
        BatchHook hook = [&](const BatchInfo& b) {
          // the tiles whose last pixel lies in this batch are final for this wave: copy them out and report them
          const auto& tiles = scene->tiles;
          bool stopNow = false;
          while (nextTile < tiles.size() && tiles[nextTile].start + tiles[nextTile].count <= b.c0 + b.n) {
            const YartScene::TileRec& t = tiles[nextTile++];
            const size_t off = (size_t(t.y) * W + t.x) * 4;
            HIP_CHECK(hipMemcpy2D(out_rgba + off, size_t(W) * 16, scene->hdr.p + off, size_t(W) * 16, size_t(t.w) * 16, t.h,
                                  hipMemcpyDeviceToHost));
            YartTileInfo ti{};
            ti.x = t.x; ti.y = t.y; ti.width = t.w; ti.height = t.h;
            ti.index = uint32_t(nextTile); ti.total = uint32_t(tiles.size());
            ti.wave = b.wave; ti.wave_samples = b.waveSamples; ti.samples_taken = b.samplesTaken; ti.total_samples = b.totalSamples;
            ti.rays = nextTile - 1 < scene->tileRaysHost.size() ? scene->tileRaysHost[nextTile - 1] : 0;
            ti.ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - tw).count();
            if (on_tile(user, &ti) != 0) stopNow = true;
          }
          return stopNow;
        };
        const bool stopped = renderToDevice(*scene, *cam, q, scene->hdr.p, nullptr, &st, on_tile ? &hook : nullptr);
        // after a wave the whole frame (with a tile callback: every tile has been copied already unless the render stopped)
        if (!on_tile || stopped) HIP_CHECK(hipMemcpy(out_rgba, scene->hdr.p, n * sizeof(float), hipMemcpyDeviceToHost));
        total.rays += st.rays; total.waves += st.waves; total.ms_device += st.ms_device; total.samples += st.samples;
        total.ms_extend += st.ms_extend; total.ms_shade += st.ms_shade; total.ms_connect += st.ms_connect; total.ms_gmon += st.ms_gmon;
        total.ms_traverse += st.ms_traverse; total.launches_traverse += st.launches_traverse;
        total.pipeline_flags = st.pipeline_flags;
        st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        if (stopped) aborted = true;
        else if (on_wave && on_wave(user, &st, uint32_t(wave), uint32_t(waveSamples), uint32_t(taken + waveSamples), params->samples) != 0)
          aborted = true;
      }
      remaining -= waveSamples;
      const uint64_t next = (wave > 0 || waveSamples > 1) ? std::min<uint64_t>(waveSamples * 2, params->max_wave_samples) : 1;
      waveSamples = std::min(next, remaining);
      wave++;
    }
    total.ms_total = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    if (stats) *stats = total;
  });
  return rc == YART_OK && aborted ? YART_ABORTED : rc;
}

int yart_hip_render_waves(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                          float* out_rgba, YartStats* stats, YartWaveCallback on_wave, void* user) {
  return renderProgressive(scene, cam, params, out_rgba, stats, on_wave, nullptr, user);
}

int yart_hip_render_tiles(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                          float* out_rgba, YartStats* stats, YartWaveCallback on_wave, YartTileCallback on_tile, void* user) {
  return renderProgressive(scene, cam, params, out_rgba, stats, on_wave, on_tile, user);
}

int yart_hip_multi_create(const YartSceneDesc* desc, const int* devices, uint32_t n_devices, YartMulti** out) {
  return guarded([&] {
    require(desc && out, "desc / out pointer is null");
    *out = createMulti(*desc, devices, n_devices);
  });
}

int yart_hip_multi_load(const char* path, const YartImportOptions* opts, const int* devices, uint32_t n_devices, YartMulti** out) {
  return guarded([&] {
    require(path && out, "path / out pointer is null");
    const std::string p(path);
    const bool gltf = p.size() > 4 && (p.rfind(".glb") == p.size() - 4 || p.rfind(".gltf") == p.size() - 5);
    auto loaded = gltf ? importGltf(path, opts) : loadSceneFile(path);
    *out = createMulti(loaded->desc, devices, n_devices);
  });
}

void yart_hip_multi_destroy(YartMulti* multi) { delete multi; }

int yart_hip_multi_rccl_selftest(int device, uint32_t n_floats) {
  return guarded([&] {
    require(n_floats > 0 && n_floats <= (1u << 26), "n_floats out of range");
    const int dev = resolveDevice(device);
    if (dev < 0) throw HipError("no usable HIP device (libyart_hip has no CPU fallback)");
    HIP_CHECK(hipSetDevice(dev));
    // the calls of mergeSlabs on a one-rank communicator: the rank sends a slab to itself (matching send / recv in one group)
    const RcclApi& R = rccl();
    ncclComm_t comm = nullptr;
    RCCL_CHECK(R.CommInitAll(&comm, 1, &dev));
    hipStream_t st = nullptr;
    DevBuf<float> a, b;
    std::vector<float> h(n_floats), back(n_floats, 0.0f);
    for (uint32_t i = 0; i < n_floats; i++) h[i] = float(i % 8191u) * 0.25f - 3.0f;
    try {
      HIP_CHECK(hipStreamCreate(&st));
      a.upload(h); b.ensure(n_floats);
      HIP_CHECK(hipMemsetAsync(b.p, 0, size_t(n_floats) * 4, st));
      RCCL_CHECK(R.GroupStart());
      ncclResult_t e = R.Send(a.p, n_floats, ncclFloat, 0, comm, st);
      if (e == ncclSuccess) e = R.Recv(b.p, n_floats, ncclFloat, 0, comm, st);
      const ncclResult_t ge = R.GroupEnd();                 // closed whatever the calls inside returned
      RCCL_CHECK(e);
      RCCL_CHECK(ge);
      HIP_CHECK(hipStreamSynchronize(st));
      HIP_CHECK(hipMemcpy(back.data(), b.p, size_t(n_floats) * 4, hipMemcpyDeviceToHost));
    } catch (...) {
      if (st) (void)hipStreamDestroy(st);
      (void)R.CommDestroy(comm);
      throw;
    }
    (void)hipStreamDestroy(st);
    RCCL_CHECK(R.CommDestroy(comm));
    if (std::memcmp(h.data(), back.data(), size_t(n_floats) * 4) != 0) throw RcclFailure("rccl selftest: the received slab differs from the sent one");
  });
}

int yart_hip_multi_device_count(const YartMulti* multi) { return multi ? int(multi->scenes.size()) : 0; }

int yart_hip_multi_failed_devices(const YartMulti* multi, int* replicas_out, uint32_t capacity) {
  if (!multi) return 0;
  int nf = 0;
  for (size_t i = 0; i < multi->failed.size(); i++)
    if (multi->failed[i]) { if (replicas_out && uint32_t(nf) < capacity) replicas_out[nf] = int(i); nf++; }
  if (nf) g_lastError = multi->failureNote;
  return nf;
}

int yart_hip_multi_render(YartMulti* multi, const YartCameraDesc* cam, const YartRenderParams* params, float* out_rgba,
                          YartStats* stats) {
  return guarded([&] {
    require(multi && out_rgba, "multi / output pointer is null");
    validate(cam, params);
    renderMulti(*multi, *cam, *params, out_rgba, stats);
  });
}

int yart_hip_multi_render_tiles(YartMulti* multi, const YartCameraDesc* cam, const YartRenderParams* params, float* out_rgba,
                                YartStats* stats, YartWaveCallback on_wave, YartTileCallback on_tile, void* user) {
  bool aborted = false;
  const int rc = guarded([&] {
    require(multi && out_rgba, "multi / output pointer is null");
    validate(cam, params);
    aborted = renderMultiProgressive(*multi, *cam, *params, out_rgba, stats, on_wave, on_tile, user) == YART_ABORTED;
  });
  return rc == YART_OK && aborted ? YART_ABORTED : rc;
}

int yart_hip_probe_samples(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                           uint32_t n, const uint32_t* xys, float* out_rgb, uint64_t* out_rays) {
  return guarded([&] {
    require(scene && xys && out_rgb, "null pointer");
    validate(cam, params);
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(scene->mu);
    YartScene& s = *scene;
    HIP_CHECK(hipSetDevice(s.device));
    const int grid = int((n + kBlock - 1) / kBlock);
    s.spill.ensure(size_t(grid) * kBlock * kSpillDepth);
    s.probeIn.ensure(size_t(n) * 3); s.probeOut.ensure(size_t(n) * 3); s.counters.ensure(8);
    HIP_CHECK(hipMemcpy(s.probeIn.p, xys, size_t(n) * 3 * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(s.counters.p, 0, 8 * sizeof(unsigned long long)));
    ProbeSampleArgs a{};
    a.sc = s.dev; a.cam = makeCamera(*cam); a.rc = makeRenderConst(*params);
    a.xys = s.probeIn.p; a.n = n; a.out = s.probeOut.p; a.rays = s.counters.p; a.spill = s.spill.p;
    hipLaunchKernelGGL(k_probe_samples, dim3(grid), dim3(kBlock), 0, nullptr, a);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out_rgb, s.probeOut.p, size_t(n) * 3 * 4, hipMemcpyDeviceToHost));
    if (out_rays) {
      unsigned long long r = 0;
      HIP_CHECK(hipMemcpy(&r, s.counters.p, sizeof(r), hipMemcpyDeviceToHost));
      *out_rays = r;
    }
  });
}

int yart_hip_probe_hits(YartScene* scene, uint32_t n, const float* rays, float* out) {
  return guarded([&] {
    require(scene && rays && out, "null pointer");
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(scene->mu);
    YartScene& s = *scene;
    HIP_CHECK(hipSetDevice(s.device));
    const int grid = int((n + kBlock - 1) / kBlock);
    s.spill.ensure(size_t(grid) * kBlock * kSpillDepth);
    DevBuf<float> in, res;
    in.ensure(size_t(n) * 6); res.ensure(size_t(n) * 16);
    HIP_CHECK(hipMemcpy(in.p, rays, size_t(n) * 6 * 4, hipMemcpyHostToDevice));
    ProbeHitArgs a{};
    a.sc = s.dev; a.rays = in.p; a.n = n; a.out = res.p; a.spill = s.spill.p;
    hipLaunchKernelGGL(k_probe_hits, dim3(grid), dim3(kBlock), 0, nullptr, a);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, res.p, size_t(n) * 16 * 4, hipMemcpyDeviceToHost));
  });
}

int yart_hip_probe_sampler(YartScene* scene, uint32_t spp, uint32_t tile, uint32_t n, const uint32_t* cases, uint32_t n_draws,
                           const uint8_t* pattern, int use_tables, float* out) {
  return guarded([&] {
    require(scene && cases && pattern && out && n > 0 && n_draws > 0 && n_draws <= 64 && spp > 0 && tile > 0, "probe_sampler: bad argument");
    YartScene& s = *scene;
    std::lock_guard<std::mutex> lk(s.mu);
    HIP_CHECK(hipSetDevice(s.device));
    uint32_t nOut = 0;
    for (uint32_t k = 0; k < n_draws; k++) { require(pattern[k] == 1 || pattern[k] == 2, "probe_sampler: pattern entries are 1 or 2"); nOut += pattern[k]; }
    for (uint32_t i = 0; i < n; i++) require(cases[3 * i] < 65536u && cases[3 * i + 1] < 65536u && cases[3 * i + 2] < spp, "probe_sampler: pixel / sample out of range");
    DevBuf<uint32_t> dCases; DevBuf<uint8_t> dPat; DevBuf<float> dOut;
    dCases.upload(std::vector<uint32_t>(cases, cases + size_t(n) * 3)); dPat.upload(std::vector<uint8_t>(pattern, pattern + n_draws));
    dOut.ensure(size_t(n) * nOut);
    ProbeSamplerArgs a{};
    a.cfg = makeSamplerConfig(spp, tile);
    a.sobol = reinterpret_cast<const uint32_t*>(s.dev.lut + LutDev::sobol);
    a.cases = dCases.p; a.n = n; a.nDraws = n_draws; a.nOut = nOut; a.pattern = dPat.p; a.out = dOut.p;
    DevBuf<uint32_t> dPix; DevBuf<uint64_t> entries, hash; DevBuf<uint32_t> sobol1;
    if (use_tables) {
      // the tables of a render whose pixel list is the cases' pixels (k_sampler_tables, as renderToDevice builds them)
      require(uint64_t(spp) <= (1ull << a.cfg.log2spp), "probe_sampler: the sampler tables need spp <= 2^log2spp");
      std::vector<uint32_t> pix(n);
      for (uint32_t i = 0; i < n; i++) pix[i] = cases[3 * i] | (cases[3 * i + 1] << 16);
      dPix.upload(pix);
      const uint32_t dims = 256u;
      entries.ensure(size_t(dims) * n); hash.ensure(dims + 3); sobol1.ensure(8 * 256);
      SamplerTabArgs ta{};
      ta.cfg = a.cfg; ta.pixels = dPix.p; ta.nPixels = n; ta.dims = dims; ta.entries = entries.p; ta.hash = hash.p; ta.sobol1 = sobol1.p;
      ta.matrix52 = a.sobol;
      hipLaunchKernelGGL(k_sampler_tables, dim3(64), dim3(kBlock), 0, nullptr, ta);
      HIP_CHECK(hipGetLastError());
      a.cfg.tab.entries = entries.p; a.cfg.tab.hash = hash.p; a.cfg.tab.sobol1 = sobol1.p; a.cfg.tab.dims = dims; a.cfg.tab.stride = n;
    }
    hipLaunchKernelGGL(k_probe_sampler, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, a);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, dOut.p, size_t(n) * nOut * 4, hipMemcpyDeviceToHost));
  });
}

int yart_hip_tonemap_agx(const float* d_hdr_rgba, uint32_t width, uint32_t height, int look, float* d_ldr_rgba,
                         void* stream) {
  return guarded([&] {
    require(d_hdr_rgba && d_ldr_rgba && width > 0 && height > 0 && look >= 0 && look <= 2, "tonemap: bad argument");
    const uint32_t n = width * height;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_tonemap_agx, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                       reinterpret_cast<const f4*>(d_hdr_rgba), reinterpret_cast<f4*>(d_ldr_rgba), n, look);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));
  });
}

int yart_hip_encode_rgb8(const float* d_rgba, uint32_t width, uint32_t height, uint8_t* d_rgb8, void* stream) {
  return guarded([&] {
    require(d_rgba && d_rgb8 && width > 0 && height > 0, "encode: bad argument");
    const uint32_t n = width * height;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_encode_rgb8, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                       reinterpret_cast<const f4*>(d_rgba), d_rgb8, n);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));
  });
}

int yart_hip_tonemap_host(const float* hdr_rgba, uint32_t width, uint32_t height, int look, float* ldr_rgba,
                          uint8_t* rgb8) {
  return guarded([&] {
    require(hdr_rgba && width > 0 && height > 0 && look >= -1 && look <= 2, "tonemap: bad argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw HipError("no HIP device");
    const size_t n = size_t(width) * height;
    DevBuf<float> in, out; DevBuf<uint8_t> bytes;
    in.ensure(n * 4); out.ensure(n * 4); bytes.ensure(n * 3);
    HIP_CHECK(hipMemcpy(in.p, hdr_rgba, n * 16, hipMemcpyHostToDevice));
    const float* src = in.p;
    if (look >= 0) {                                           // look -1: no tonemapper (tile-renderer.hpp:238-240)
      hipLaunchKernelGGL(k_tonemap_agx, dim3((uint32_t(n) + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr,
                         reinterpret_cast<const f4*>(in.p), reinterpret_cast<f4*>(out.p), uint32_t(n), look);
      HIP_CHECK(hipGetLastError());
      src = out.p;
    }
    hipLaunchKernelGGL(k_encode_rgb8, dim3((uint32_t(n) + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr,
                       reinterpret_cast<const f4*>(src), bytes.p, uint32_t(n));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    if (ldr_rgba) HIP_CHECK(hipMemcpy(ldr_rgba, src, n * 16, hipMemcpyDeviceToHost));
    if (rgb8) HIP_CHECK(hipMemcpy(rgb8, bytes.p, n * 3, hipMemcpyDeviceToHost));
  });
}

int yart_hip_debug_counters(YartScene* scene, uint64_t* out32) {
  return guarded([&] {
    require(scene && out32, "null pointer");
    for (int i = 0; i < kNumCounters; i++) out32[i] = scene->lastCounters[i];
  });
}

/* debug build -DYART_SHADE_REGIONS=1 only (tools/shade_regions.py): cycles / visits / lanes per code region of k_wf_shade,
 * summed over the renders since the last call (16 regions x 3); -1 otherwise */
int yart_hip_debug_shade_regions(uint64_t* out48) {
#if defined(YART_SHADE_REGIONS)
  return guarded([&] {
    require(out48 != nullptr, "null pointer");
    unsigned long long v[3 * 16];
    tu::shadeRegionsTake(v);
    for (int i = 0; i < 48; i++) out48[i] = v[i];
  });
#else
  (void)out48;
  return YART_E_INVALID;
#endif
}

int yart_hip_bvh_info(YartScene* scene, uint32_t mesh, uint32_t* n_nodes, uint32_t* n_tris) {
  return guarded([&] {
    require(scene && n_nodes && n_tris, "null pointer");
    require(mesh < scene->host.meshes.size(), "mesh index out of range");
    *n_nodes = scene->host.meshes[mesh].nNodes;
    *n_tris = scene->host.meshes[mesh].nTris;
  });
}

int yart_hip_bvh_copy(YartScene* scene, uint32_t mesh, uint32_t* nodes_out, uint32_t* indices_out) {
  return guarded([&] {
    require(scene && nodes_out && indices_out, "null pointer");
    require(mesh < scene->host.meshes.size(), "mesh index out of range");
    const MeshDev& m = scene->host.meshes[mesh];
    // read the node array back from the DEVICE copy: this is what the kernels traverse
    HIP_CHECK(hipSetDevice(scene->device));
    HIP_CHECK(hipMemcpy(nodes_out, scene->bvhNodes.p + m.nodeOffset, size_t(m.nNodes) * sizeof(BvhNode),
                        hipMemcpyDeviceToHost));
    for (uint32_t n = 0; n < m.nNodes; n++) nodes_out[size_t(n) * 8 + 6] &= kLinkIndexMask;   // the reference's index only
    std::memcpy(indices_out, scene->host.bvhIndices[mesh].data(), size_t(m.nTris) * 4);
  });
}

int yart_hip_bvh_build_device(int device, const float* positions, uint32_t n_verts, const uint32_t* faces, uint32_t face_stride,
                              uint32_t n_faces, uint32_t* nodes_out, uint32_t* indices_out, uint32_t* n_nodes, double* ms_device) {
  return guarded([&] {
    require(positions && faces && nodes_out && indices_out && n_nodes, "null pointer");
    require(n_faces >= 1 && n_faces <= kLinkIndexMask && face_stride >= 3, "bvh build: bad triangle count or stride");
    for (size_t k = 0; k < size_t(n_faces) * face_stride; k += face_stride)
      for (int c = 0; c < 3; c++) require(faces[k + c] < n_verts, "bvh build: vertex index out of range");
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> indices;
    require(devbvh::build(device, positions, n_verts, faces, face_stride, n_faces, nodes, indices, ms_device),
            "bvh build: refused on the device (NaN coordinate or tree deeper than 192 levels): build on the host");
    *n_nodes = uint32_t(nodes.size());
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(BvhNode));
    std::memcpy(indices_out, indices.data(), indices.size() * 4);
  });
}

int yart_hip_bvh_build_host(const float* positions, uint32_t n_verts, const uint32_t* faces, uint32_t face_stride, uint32_t n_faces,
                            uint32_t threads, uint32_t* nodes_out, uint32_t* indices_out, uint32_t* n_nodes, double* ms_host) {
  return guarded([&] {
    require(positions && faces && nodes_out && indices_out && n_nodes, "null pointer");
    require(n_faces >= 1 && n_faces <= kLinkIndexMask && face_stride >= 3, "bvh build: bad triangle count or stride");
    for (size_t k = 0; k < size_t(n_faces) * face_stride; k += face_stride)
      for (int c = 0; c < 3; c++) require(faces[k + c] < n_verts, "bvh build: vertex index out of range");
    const auto t0 = std::chrono::steady_clock::now();
    SahBvhBuilder b;
    b.setThreads(threads);
    b.build(positions, faces, face_stride, n_faces);
    if (ms_host) *ms_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *n_nodes = uint32_t(b.nodes.size());
    std::memcpy(nodes_out, b.nodes.data(), b.nodes.size() * sizeof(BvhNode));
    std::memcpy(indices_out, b.indices.data(), b.indices.size() * 4);
  });
}

}  // extern "C"
#endif  // YART_TU == 0
