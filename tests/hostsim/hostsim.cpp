// hostsim.cpp — TEST INFRASTRUCTURE, never part of libyart_hip.so.
//
// Compiles the product's device headers (yart_amd/csrc/*.hpp) as plain host C++
// (YART_HD expands to `inline`) and evaluates the known-answer-test program of
// oracle/kat_common.hpp with them, so that every device function can be compared
// bit for bit with the compiled reference (oracle/_ref/yart_ref kat) in the
// GPU-less build container. It also renders small images on CPU threads with the
// same per-sample code, which lets `pytest -m "not gpu"` bound the difference
// between the kernel source and the reference before a GPU is involved.
// It is not a fallback: the C ABI has no path to this code (tests/test_no_fallback.py
// checks that the library fails without a device).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include "../../oracle/kat_common.hpp"
#include "../../oracle/params.hpp"
#include "../../yart_amd/csrc/estimator.hpp"
#include "../../yart_amd/csrc/host_scene.hpp"
#include "../../yart_amd/csrc/tonemap.hpp"
#include "../../yart_amd/csrc/integrator.hpp"
#include "../../yart_amd/csrc/scene_file.hpp"

using namespace yart_hip;

static YartCameraDesc cameraDesc(const params::Params& p) {
  YartCameraDesc c{};
  c.width = p.width; c.height = p.height; c.focal_length = p.focal; c.f_number = p.fnumber;
  c.sensor[0] = p.sensor[0]; c.sensor[1] = p.sensor[1];
  for (int i = 0; i < 3; i++) { c.position[i] = p.eye[i]; c.target[i] = p.target[i]; c.up[i] = p.up[i]; }
  c.exposure = p.exposure; c.aperture_sides = p.apertureSides;
  return c;
}

struct Ctx {
  HostImage im;
  SceneDev sc;
  CameraDev cam;
  RenderConst rc;
};

static PathCtx pathCtx(const Ctx& c, uint64_t* stack) {
  PathCtx px;
  px.sc = &c.sc;
  px.sobol = reinterpret_cast<const uint32_t*>(c.sc.lut + LutDev::sobol);
  px.stk.lds = stack; px.stk.ldsStride = 1; px.stk.ldsDepth = kRefStackDepth;
  px.stk.spill = nullptr; px.stk.spillStride = 0;
  px.rc = c.rc;
  return px;
}

static int doKat(Ctx& c, const params::Params& p, const std::string& outPath) {
  kat::Writer w(outPath);
  {
    std::vector<uint64_t> h, mb, mo;
    std::vector<int64_t> l2;
    for (uint32_t d = 0; d < 48; d++) h.push_back(hashDim(d));
    for (uint64_t v : kat::mixInputs()) mb.push_back(mixBits(v));
    for (auto xy : kat::mortonInputs()) mo.push_back(encodeMorton2(xy.first, xy.second));
    for (float v : kat::log2Inputs()) l2.push_back(log2IntHost(v));
    w.u64("hash32", h); w.u64("mixbits", mb); w.u64("morton", mo); w.i64("log2int", l2);
  }
  const uint32_t* sobol = reinterpret_cast<const uint32_t*>(c.sc.lut + LutDev::sobol);
  {
    std::vector<float> out;
    for (const auto& k : kat::samplerCases()) {
      SamplerConfig cfg = makeSamplerConfig(k.spp, k.tile);
      Sampler s;
      startPixelSample(s, cfg, k.px, k.py, k.sample);
      for (int q : kat::samplerPattern()) {
        if (q == 2) { f2 v = get2D(s, cfg, sobol); out.push_back(v.x); out.push_back(v.y); }
        else out.push_back(get1D(s, cfg));
      }
    }
    w.f32("sampler", out);
  }
  {
    std::vector<float> e, ea, be, bea, ge, gea;
    const float* lut = c.sc.lut;
    for (const auto& q : kat::lutInputs()) {
      e.push_back(ggxE(lut, q.c, q.r));
      ea.push_back(ggxEavg(lut, q.r));
      be.push_back(ggxBaseE(lut, q.f0, q.r, q.c));
      bea.push_back(ggxBaseEavg(lut, q.f0, q.r));
      ge.push_back(ggxGlassE(lut, q.ior, q.r, std::fabs(q.c)));
      gea.push_back(0.0f);   // ggxGlassEavg is not on the path (unused by parametric.cpp)
    }
    w.f32("ggxE", e); w.f32("ggxEavg", ea); w.f32("ggxBaseE", be);
    w.f32("ggxBaseEavg", bea); w.f32("ggxGlassE", ge); w.f32("ggxGlassEavg", gea);
  }
  {
    std::vector<float> out;
    kat::Lcg rng(7);
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      for (int k = 0; k < 4; k++) {
        f2 uf = mk2(0, 0), ul = mk2(0, 0);
        uf.x = rng.next(); uf.y = rng.next(); ul.x = rng.next(); ul.y = rng.next();
        f3 o, d;
        cameraRay(c.cam, p.probePixels[i], p.probePixels[i + 1], uf, ul, o, d);
        out.push_back(o.x); out.push_back(o.y); out.push_back(o.z);
        out.push_back(d.x); out.push_back(d.y); out.push_back(d.z);
      }
    }
    w.f32("camera_rays", out);
  }
  {
    std::vector<uint64_t> out;
    for (size_t m = 0; m < c.im.meshes.size(); m++) {
      const MeshDev& md = c.im.meshes[m];
      out.push_back(md.nNodes);
      std::vector<BvhNode> plain(c.im.bvhNodes.begin() + md.nodeOffset, c.im.bvhNodes.begin() + md.nodeOffset + md.nNodes);
      for (auto& n : plain) n.leftFirst &= kLinkIndexMask;       // the device image carries an extra flag bit
      out.push_back(kat::fnv1a(plain.data(), size_t(md.nNodes) * 32));
      out.push_back(kat::fnv1a(c.im.bvhIndices[m].data(), c.im.bvhIndices[m].size() * 4));
    }
    w.u64("bvh", out);
  }
  uint64_t stack[kRefStackDepth];
  PathCtx px = pathCtx(c, stack);
  {
    std::vector<float> fo, ro6;
    std::vector<int64_t> io;
    // the reference's probe integrator holds ONE sampler, constructed and never started on a pixel (ref_driver.cpp: pixel 0,
    // sample 0, dimension 0): stochastic alpha tests of the probe rays draw from it one after the other
    Sampler dummy; dummy.dim = 0; dummy.morton = 0; dummy.pix = 0;
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      f3 o, d;
      cameraRay(c.cam, p.probePixels[i], p.probePixels[i + 1], mk2(0.5f, 0.5f), mk2(0.5f, 0.5f), o, d);
      ro6.push_back(o.x); ro6.push_back(o.y); ro6.push_back(o.z); ro6.push_back(d.x); ro6.push_back(d.y); ro6.push_back(d.z);
      HitRec hr; hr.t = kInf; hr.u = hr.v = 0; hr.tri = 0; hr.node = 0; hr.backSide = 0;
      f3 att = mk3(1.0f);
      AlphaCtx ac; ac.sampler = &dummy; ac.cfg = c.rc.sampler;
      bool hit = traverseScene<false>(c.sc, o, d, 0.001f, hr, att, px.stk, ac);
      io.push_back(hit);
      if (!hit) { io.push_back(-1); io.push_back(-1); io.push_back(0); for (int k = 0; k < 12; k++) fo.push_back(0); continue; }
      Hit h = finalizeHit(c.sc, hr, o, d);
      io.push_back(localTri(c.sc, hr)); io.push_back(h.lightIdx); io.push_back(h.backSide);
      fo.push_back(h.t); fo.push_back(h.uv.x); fo.push_back(h.uv.y);
      fo.push_back(h.p.x); fo.push_back(h.p.y); fo.push_back(h.p.z);
      fo.push_back(h.n.x); fo.push_back(h.n.y); fo.push_back(h.n.z);
      fo.push_back(h.tg.x); fo.push_back(h.tg.y); fo.push_back(h.tg.z);
    }
    w.i64("hits_i", io); w.f32("hits_f", fo); w.f32("hit_rays", ro6);
  }
  {
    std::vector<float> fo;
    std::vector<int64_t> io;
    const f3 n = mk3(0, 0, 1), t = mk3(1, 0, 0);
    size_t nm = c.im.nMaterials;
    for (size_t m = 0; m < nm; m++) {
      const MaterialDev& mt = c.im.materials[m];
      kat::Lcg rng(1000 + uint32_t(m));
      for (int k = 0; k < kat::bsdfCasesPerMaterial; k++) {
        float r[12];
        for (int q = 0; q < 6; q++) r[q] = rng.sym();
        for (int q = 6; q < 12; q++) r[q] = rng.next();
        f3 wo = normalized(mk3(r[0], r[1], r[2]));
        f3 wi = normalized(mk3(r[3], r[4], r[5]));
        f2 uv = mk2(r[6] * 3.0f - 1.0f, r[7] * 3.0f - 1.0f);
        f2 u = mk2(r[8], r[9]);
        float uc = r[10], uc2 = r[11];
        bool reg = k & 1;
        f3 f = bsdfF(c.sc, mt, wo, wi, n, t, uv);
        float pdf = bsdfPdf(c.sc, mt, wo, wi, n, t, uv);
        BsdfSample s = bsdfSample(c.sc, mt, wo, n, t, uv, u, uc, uc2, reg);
        fo.push_back(f.x); fo.push_back(f.y); fo.push_back(f.z);
        fo.push_back(pdf);
        io.push_back(s.scatter);
        fo.push_back(s.f.x); fo.push_back(s.f.y); fo.push_back(s.f.z);
        fo.push_back(s.Le.x); fo.push_back(s.Le.y); fo.push_back(s.Le.z);
        fo.push_back(s.wi.x); fo.push_back(s.wi.y); fo.push_back(s.wi.z);
        fo.push_back(s.pdf); fo.push_back(s.roughness);
        fo.push_back(matAlpha(c.sc, mt, uv));
        f3 base = matBase(c.sc, mt, uv);
        fo.push_back(base.x); fo.push_back(base.y); fo.push_back(base.z);
        f4 t4; t4.x = 1; t4.y = 0; t4.z = 0; t4.w = 1;
        f3 sn = bsdfNormal(c.sc, mt, n, t4, uv);
        fo.push_back(sn.x); fo.push_back(sn.y); fo.push_back(sn.z);
        f3 att = matAttenuation(mt, uc * 4.0f);
        fo.push_back(att.x); fo.push_back(att.y); fo.push_back(att.z);
      }
      io.push_back((mt.flags & MAT_TRANSPARENT) ? 1 : 0);
    }
    w.i64("bsdf_i", io); w.f32("bsdf_f", fo);
  }
  {
    std::vector<float> fo;
    std::vector<int64_t> io;
    size_t nl = c.sc.nLights;
    kat::Lcg rng(4242);
    for (size_t li : kat::lightSubset(nl)) {
      const LightDev& l = c.sc.lights[li];
      fo.push_back(l.power);
      for (int k = 0; k < 4; k++) {
        float r0 = rng.sym(), r1 = rng.next(), r2 = rng.sym(), r3 = rng.next(), r4 = rng.next();
        f3 pp = mk3(r0 * 4.0f, r1 * 8.0f, r2 * 4.0f);
        f2 u = mk2(r3, r4);
        LightSample s = lightSample(c.sc, l, pp, u);
        fo.push_back(s.Li.x); fo.push_back(s.Li.y); fo.push_back(s.Li.z);
        fo.push_back(s.wi.x); fo.push_back(s.wi.y); fo.push_back(s.wi.z);
        fo.push_back(s.p.x); fo.push_back(s.p.y); fo.push_back(s.p.z);
        fo.push_back(s.n.x); fo.push_back(s.n.y); fo.push_back(s.n.z);
        fo.push_back(s.pdf);
        float w0 = rng.sym(), w1 = rng.sym(), w2 = rng.sym();
        f3 wi = normalized(mk3(w0, w1, w2));
        fo.push_back(lightPdf(c.sc, l, wi));
        f3 le = lightLe(c.sc, l, octahedralUV(wi));
        fo.push_back(le.x); fo.push_back(le.y); fo.push_back(le.z);
      }
      fo.push_back(lightSamplerP(c.sc, uint32_t(li)));
    }
    if (nl > 0) {
      for (int k = 0; k < 32; k++) {
        float u = rng.next();
        float pl;
        uint32_t which = lightSamplerSample(c.sc, u, pl);
        io.push_back(which);
        fo.push_back(pl);
      }
    }
    w.i64("lights_i", io); w.f32("lights_f", fo);
  }
  {
    std::vector<float> rad, pix;
    uint32_t rays = 0;
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      int m = gmonBuckets(int32_t(p.spp));
      f3 acc[kGmonMax]; uint32_t cnt[kGmonMax];
      for (int b = 0; b < kGmonMax; b++) { acc[b] = mk3(0); cnt[b] = 0; }
      for (uint32_t s = 0; s < p.spp; s++) {
        f3 L = samplePixel(px, c.cam, p.probePixels[i], p.probePixels[i + 1], s, rays);
        rad.push_back(L.x); rad.push_back(L.y); rad.push_back(L.z);
        f3 v = L * c.cam.exposureScale;
        int b = int(s % uint32_t(m));
        if (gmonAccepts(v)) { acc[b] += v; cnt[b]++; }
      }
      f3 v = gmonFinish(acc, cnt, m);
      pix.push_back(v.x); pix.push_back(v.y); pix.push_back(v.z);
    }
    w.f32("radiance", rad); w.f32("gmon", pix);
    std::vector<uint64_t> rc{rays};
    w.u64("probe_rays", rc);
  }
  w.close();
  return 0;
}

static int doRender(Ctx& c, const params::Params& p, const std::string& outPath) {
  const uint32_t W = p.width, H = p.height;
  std::vector<float> img(size_t(W) * H * 4, 0.0f);
  std::atomic<uint32_t> nextRow{0};
  std::atomic<uint64_t> totalRays{0};
  unsigned nt = p.threads ? p.threads : std::thread::hardware_concurrency();
  auto t0 = std::chrono::high_resolution_clock::now();
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&] {
      uint64_t stack[kRefStackDepth];
      PathCtx px = pathCtx(c, stack);
      uint32_t rays = 0;
      int m = gmonBuckets(int32_t(p.spp));
      for (;;) {
        uint32_t y = nextRow++;
        if (y >= H) break;
        for (uint32_t x = 0; x < W; x++) {
          f3 acc[kGmonMax]; uint32_t cnt[kGmonMax];
          for (int b = 0; b < kGmonMax; b++) { acc[b] = mk3(0); cnt[b] = 0; }
          for (uint32_t s = 0; s < p.spp; s++) {
            f3 v = samplePixel(px, c.cam, x, y, s, rays) * c.cam.exposureScale;
            int b = int(s % uint32_t(m));
            if (gmonAccepts(v)) { acc[b] += v; cnt[b]++; }
          }
          f3 v = gmonFinish(acc, cnt, m);
          float* o = &img[(size_t(y) * W + x) * 4];
          // single wave: hdr = hdr*0 + wave*1 (tile-renderer.hpp:220-232)
          o[0] = 0.0f * 0.0f + v.x * 1.0f; o[1] = 0.0f * 0.0f + v.y * 1.0f; o[2] = 0.0f * 0.0f + v.z * 1.0f;
          o[3] = 1.0f;
        }
      }
      totalRays += rays;
    });
  for (auto& t : th) t.join();
  double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  FILE* f = std::fopen(outPath.c_str(), "wb");
  if (!f) return 2;
  std::fwrite(img.data(), 4, img.size(), f);
  std::fclose(f);
  std::printf("{\"rays\": %llu, \"seconds\": %.6f, \"msamples_per_s\": %.6f, \"threads\": %u}\n",
              (unsigned long long) totalRays.load(), sec, double(W) * H * p.spp / sec * 1e-6, nt);
  return 0;
}

// selftest: the table-driven forms of the device headers against their direct forms, on the host.
//  * ZSobol index: SamplerTables entry + remaining digits == the reference's digit loop, for every
//    log2spp / tile size combination, random pixels, dimensions and samples;
//  * Sobol' dimension-1 byte tables == the bit loop;
//  * CDF search with a guide table == the plain lower-bound search, on random and degenerate CDFs.
static int doSelfTest() {
  struct { uint64_t s = 0x5eed5eedull; uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return uint32_t(s >> 32); } } rng;
  uint64_t checked = 0;
  std::vector<uint32_t> matrix52(52);
  for (uint32_t k = 0; k < 52; k++) matrix52[k] = sobolDim1Column(k);
  std::vector<uint32_t> byteTab(8 * 256);
  for (uint32_t k = 0; k < 8 * 256; k++) {
    uint32_t b = k >> 8, v = k & 255u, x = 0;
    for (uint32_t j = 0; j < 8; j++) if (((v >> j) & 1u) && 8 * b + j < 52) x ^= matrix52[8 * b + j];
    byteTab[k] = x;
  }
  for (uint32_t spp : {1u, 2u, 3u, 4u, 8u, 12u, 16u, 32u, 48u, 64u, 128u, 256u, 512u, 1024u, 2048u}) {
    for (uint32_t tile : {8u, 64u, 100u, 512u, 4096u}) {
      SamplerConfig cfg = makeSamplerConfig(spp, tile);
      if (uint64_t(spp) > (1ull << cfg.log2spp)) continue;      // the renderer disables the tables here
      const uint32_t dims = 40, nPix = 24;
      std::vector<uint32_t> px(nPix), py(nPix);
      std::vector<uint64_t> entries(size_t(dims) * nPix), hash(dims + 3);
      for (uint32_t i = 0; i < nPix; i++) { px[i] = rng.next() % 4000u; py[i] = rng.next() % 3000u; }
      for (uint32_t d = 0; d < dims; d++)
        for (uint32_t i = 0; i < nPix; i++) entries[size_t(i) * dims + d] = samplerTableEntry(cfg, encodeMorton2(px[i], py[i]), d);
      for (uint32_t d = 0; d < dims + 3; d++) hash[d] = hashDim(d);
      SamplerConfig tcfg = cfg;
      tcfg.tab.entries = entries.data(); tcfg.tab.hash = hash.data(); tcfg.tab.sobol1 = byteTab.data();
      tcfg.tab.dims = dims; tcfg.tab.stride = nPix;
      for (uint32_t i = 0; i < nPix; i++) {
        for (uint32_t rep = 0; rep < 40; rep++) {
          Sampler a, b;
          startPixelSample(a, cfg, px[i], py[i], rng.next() % spp);
          a.dim = rng.next() % (dims + 6);                      // also beyond the table: fallback
          b = a; b.pix = i;
          if (getSampleIndexDirect(a, cfg) != getSampleIndex(b, tcfg)) {
            std::fprintf(stderr, "selftest: sample index mismatch spp=%u tile=%u dim=%u\n", spp, tile, a.dim);
            return 3;
          }
          Sampler a2 = a, b2 = b;
          f2 u0 = get2D(a2, cfg, matrix52.data()), u1 = get2D(b2, tcfg, matrix52.data());
          float v0 = get1D(a2, cfg), v1 = get1D(b2, tcfg);
          if (std::memcmp(&u0, &u1, sizeof(u0)) != 0 || std::memcmp(&v0, &v1, 4) != 0 || a2.dim != b2.dim) {
            std::fprintf(stderr, "selftest: sample value mismatch spp=%u tile=%u dim=%u\n", spp, tile, a.dim);
            return 3;
          }
          checked++;
        }
      }
    }
  }
  // v % 24 of the permutation-row selection: the multiply-free form against the operator, random and edge 40-bit values
  for (uint64_t k = 0; k < 400000; k++) {
    uint64_t v = (uint64_t(rng.next()) << 8 | (rng.next() & 0xffu)) & 0xffffffffffull;
    if (k < 64) v = k;
    else if (k < 128) v = 0xffffffffffull - (k - 64);
    else if (k < 192) v = (1ull << (k - 128 < 40 ? k - 128 : 39)) - (k & 1);
    if (mod24of40(v) != uint32_t(v % 24ull)) { std::fprintf(stderr, "selftest: mod24of40(%llu)\n", (unsigned long long) v); return 3; }
    checked++;
  }
  // CDF search
  for (uint32_t n : {1u, 2u, 3u, 7u, 64u, 100u, 1000u}) {
    for (int kind = 0; kind < 4; kind++) {
      std::vector<float> func(n), cdf(n + 1);
      for (uint32_t k = 0; k < n; k++) {
        float f = float(rng.next() % 1000u) / 1000.0f;
        if (kind == 1) f = (k % 5 == 0) ? f : 0.0f;             // long zero runs (flat CDF segments)
        if (kind == 2) f = (k == n / 2) ? 1.0f : 0.0f;          // a single spike
        if (kind == 3) f = 0.0f;                                // all zero -> uniform CDF
        func[k] = f;
      }
      cdf[0] = 0.0f;
      for (uint32_t k = 1; k <= n; k++) cdf[k] = cdf[k - 1] + func[k - 1] / float(n);
      const float integral = cdf[n];
      for (uint32_t k = 1; k <= n; k++) cdf[k] = integral == 0.0f ? float(k) / float(n) : cdf[k] / integral;
      uint32_t K = 1; while (K < n) K <<= 1;
      std::vector<uint32_t> guide;
      uint32_t idx = 1;
      for (uint32_t j = 0; j <= K; j++) {
        const float uj = float(j) / float(K);
        while (idx < n && cdf[idx] < uj) idx++;
        guide.push_back(idx);
      }
      // the interleaved records the environment sampling walks (lights.hpp::pc1dSampleRecords), strides 2 and 4
      std::vector<float> rec2(size_t(n + 1) * 2, 0.0f), rec4(size_t(n + 1) * 4, 0.0f);
      for (uint32_t k = 0; k <= n; k++) {
        rec2[size_t(k) * 2] = rec4[size_t(k) * 4] = cdf[k];
        if (k < n) rec2[size_t(k) * 2 + 1] = rec4[size_t(k) * 4 + 1] = func[k];
      }
      for (uint32_t rep = 0; rep < 4000; rep++) {
        float u = float(rng.next() & 0xffffffu) * 0x1p-24f;
        if (rep % 7 == 0) u = cdf[rng.next() % (n + 1)];        // exactly on a CDF value
        if (u >= 1.0f) u = kOneMinusEpsilon;
        float pdf0, pdf1; uint32_t o0, o1;
        float x0 = pc1dSample(func.data(), cdf.data(), n, integral, 0.0f, 1.0f, u, pdf0, o0);
        float x1 = pc1dSample(func.data(), cdf.data(), n, integral, 0.0f, 1.0f, u, pdf1, o1, guide.data(), K);
        if (o0 != o1 || std::memcmp(&x0, &x1, 4) != 0 || std::memcmp(&pdf0, &pdf1, 4) != 0) {
          std::fprintf(stderr, "selftest: guided CDF search mismatch n=%u kind=%d u=%a (%u vs %u)\n", n, kind, u, o0, o1);
          return 3;
        }
        float pdf2, pdf3; uint32_t o2, o3;
        const float x2 = pc1dSampleRecords<2>(rec2.data(), n, integral, cdf[1], u, pdf2, o2, guide.data(), K);
        const float x3 = pc1dSampleRecords<4>(rec4.data(), n, integral, cdf[1], u, pdf3, o3, nullptr, 0);
        if (o0 != o2 || o0 != o3 || std::memcmp(&x0, &x2, 4) != 0 || std::memcmp(&x0, &x3, 4) != 0 || std::memcmp(&pdf0, &pdf2, 4) != 0 ||
            std::memcmp(&pdf0, &pdf3, 4) != 0) {
          std::fprintf(stderr, "selftest: record-form CDF search mismatch n=%u kind=%d u=%a\n", n, kind, u);
          return 3;
        }
        checked++;
      }
    }
  }
  // device log2f / powf (libm_pow.hpp) against this machine's libm: a dense random sweep (the exhaustive
  // one — every float, ten exponents, 0 mismatches — takes minutes and was run once)
  {
    const float ys[] = {1.0f, 0.8f, 1.35f, 2.2f, 1.0f / 2.2f};
    for (uint32_t rep = 0; rep < 1500000; rep++) {
      uint32_t bits = rng.next();
      if (rep % 3 == 0) bits = (bits & 0x007fffffu) | ((118u + rng.next() % 14u) << 23);   // around [2^-9, 2^5)
      const float x = __builtin_bit_cast(float, bits);
      const float a = libm_pow::log2f_(x), b = std::log2(x);
      bool ok = (a != a && b != b) || std::memcmp(&a, &b, 4) == 0;
      for (float y : ys) {
        const float c = libm_pow::powf_(x, y), d = std::pow(x, y);
        ok = ok && ((c != c && d != d) || std::memcmp(&c, &d, 4) == 0);
      }
      if (!ok) { std::fprintf(stderr, "selftest: log2f / powf emulation differs from libm at x=%a\n", x); return 3; }
      checked++;
    }
  }
  // byte / 255.0f of the texture fetch (bsdf.hpp byteToUnit) against the IEEE divide, all 256 numerators
  for (uint32_t b = 0; b < 256; b++) {
    volatile float x = float(b), c = 255.0f;
    const float q = x / c, r = byteToUnit(b);
    if (std::memcmp(&q, &r, 4) != 0) { std::fprintf(stderr, "selftest: byteToUnit(%u) differs from %u / 255.0f\n", b, b); return 3; }
    checked++;
  }
  std::printf("{\"selftest\": \"ok\", \"checked\": %llu}\n", (unsigned long long) checked);
  return 0;
}

// tonemap: csrc/tonemap.hpp (the device functions, compiled for the host) over an RGBA32F file
static int doTonemap(const char* in, unsigned w, unsigned h, const std::string& look, const char* outF32, const char* outPpm) {
  std::vector<float> px(size_t(w) * h * 4);
  FILE* f = std::fopen(in, "rb");
  if (!f || std::fread(px.data(), 4, px.size(), f) != px.size()) { std::fprintf(stderr, "tonemap: short input\n"); return 2; }
  std::fclose(f);
  const int lk = look == "golden" ? 1 : look == "punchy" ? 2 : look == "none" ? 0 : -1;
  std::vector<uint8_t> bytes(size_t(w) * h * 3);
  for (size_t i = 0; i < size_t(w) * h; i++) {
    if (lk >= 0) {
      const f3 o = agxTonemap(mk3(px[4 * i], px[4 * i + 1], px[4 * i + 2]), agxLook(lk));
      px[4 * i] = o.x; px[4 * i + 1] = o.y; px[4 * i + 2] = o.z; px[4 * i + 3] = 1.0f;
    }
    for (int c = 0; c < 3; c++) bytes[3 * i + c] = ppmByte(px[4 * i + c]);
  }
  f = std::fopen(outF32, "wb"); std::fwrite(px.data(), 4, px.size(), f); std::fclose(f);
  f = std::fopen(outPpm, "wb"); std::fprintf(f, "P6\n%u %u\n255\n", w, h); std::fwrite(bytes.data(), 1, bytes.size(), f); std::fclose(f);
  return 0;
}

// estimator: csrc/estimator.hpp (the functions k_gmon_blend runs, compiled for the host) over groups of `spp`
// RGB samples; sample k goes to bucket k mod m in increasing k, exactly as the kernel's lanes do.
static int doEstimator(int kind, unsigned spp, const char* in, const char* out) {
  FILE* f = std::fopen(in, "rb");
  if (!f) { std::fprintf(stderr, "estimator: cannot open input\n"); return 2; }
  std::vector<float> smp;
  float buf[3];
  while (std::fread(buf, 4, 3, f) == 3) smp.insert(smp.end(), buf, buf + 3);
  std::fclose(f);
  const size_t groups = smp.size() / 3 / spp;
  std::vector<float> res(groups * 3);
  const int m = estimatorBuckets(kind, int32_t(spp));
  for (size_t g = 0; g < groups; g++) {
    f3 acc[kGmonMax]; uint32_t cnt[kGmonMax];
    for (int b = 0; b < m; b++) {
      acc[b] = mk3(0); cnt[b] = 0;
      for (unsigned k = unsigned(b); k < spp; k += unsigned(m)) {
        const float* p = &smp[(g * spp + k) * 3];
        const f3 v = mk3(p[0], p[1], p[2]);
        if (estimatorAccepts(kind, v)) { acc[b] += v; cnt[b]++; }
      }
    }
    const f3 v = estimatorFinish(kind, acc, cnt, m, spp);
    res[3 * g] = v.x; res[3 * g + 1] = v.y; res[3 * g + 2] = v.z;
  }
  f = std::fopen(out, "wb"); std::fwrite(res.data(), 4, res.size(), f); std::fclose(f);
  return 0;
}


// bvhcheck: the task-parallel BVH build against the plain recursion, every mesh of a scene, byte for byte
static int doBvhCheck(const char* scenePath, unsigned threads) {
  auto loaded = loadSceneFile(scenePath);
  double msSerial = 0, msParallel = 0;
  size_t nodes = 0, tris = 0;
  uint32_t maxLeaf = 0;
  for (uint32_t m = 0; m < loaded->desc.n_meshes; m++) {
    const YartMeshDesc& md = loaded->desc.meshes[m];
    SahBvhBuilder a, b;
    a.setThreads(1); b.setThreads(threads);
    auto t0 = std::chrono::steady_clock::now();
    a.build(md.positions, md.faces, 4, md.n_faces);
    auto t1 = std::chrono::steady_clock::now();
    b.build(md.positions, md.faces, 4, md.n_faces);
    auto t2 = std::chrono::steady_clock::now();
    msSerial += std::chrono::duration<double, std::milli>(t1 - t0).count();
    msParallel += std::chrono::duration<double, std::milli>(t2 - t1).count();
    if (a.nodes.size() != b.nodes.size() || std::memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(BvhNode)) != 0 ||
        a.indices != b.indices) {
      std::fprintf(stderr, "bvhcheck: mesh %u differs between 1 and %u threads\n", m, threads);
      return 3;
    }
    nodes += a.nodes.size(); tris += md.n_faces;
    for (const BvhNode& n : a.nodes) maxLeaf = std::max(maxLeaf, n.span);
  }
  const float share = buildHostImage(loaded->desc).dominantLobeShare;
  std::printf("{\"bvhcheck\": \"ok\", \"dominant_lobe_share\": %.4f, \"meshes\": %u, \"triangles\": %zu, \"nodes\": %zu, \"max_leaf_span\": %u, \"threads\": %u, \"ms_serial\": %.1f, \"ms_parallel\": %.1f}\n",
              share, loaded->desc.n_meshes, tris, nodes, maxLeaf, threads, msSerial, msParallel);
  return 0;
}

// loadstress: N concurrent host callers of the scene loader and the scene build (what N threads holding their own YartScene
// handles do on the host before anything reaches a device): .yscn parse, flattening, the task-parallel SAH build with its own
// worker pool inside every caller. Every caller must arrive at the same image (node array hashed). Run under
// -fsanitize=thread by tests/test_sanitizers.py: shared mutable state between callers would show as a race.
static int doLoadStress(const char* scenePath, unsigned callers) {
  std::vector<uint64_t> sig(callers, 0);
  std::vector<std::string> err(callers);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < callers; t++)
    th.emplace_back([&, t] {
      try {
        auto loaded = loadSceneFile(scenePath);
        HostImage im = buildHostImage(loaded->desc);
        uint64_t h = 0xcbf29ce484222325ull;
        const uint8_t* b = reinterpret_cast<const uint8_t*>(im.bvhNodes.data());
        for (size_t k = 0; k < im.bvhNodes.size() * sizeof(BvhNode); k++) h = (h ^ b[k]) * 0x100000001b3ull;
        const uint8_t* l = reinterpret_cast<const uint8_t*>(im.leafTris.data());
        for (size_t k = 0; k < im.leafTris.size() * sizeof(LeafTri); k++) h = (h ^ l[k]) * 0x100000001b3ull;
        sig[t] = h;
      } catch (const std::exception& e) { err[t] = e.what(); }
    });
  for (auto& x : th) x.join();
  for (unsigned t = 0; t < callers; t++) {
    if (!err[t].empty()) { std::fprintf(stderr, "loadstress: caller %u: %s\n", t, err[t].c_str()); return 2; }
    if (sig[t] != sig[0]) { std::fprintf(stderr, "loadstress: caller %u built a different image\n", t); return 3; }
  }
  std::printf("{\"loadstress\": \"ok\", \"callers\": %u, \"signature\": \"%016llx\"}\n", callers, (unsigned long long) sig[0]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 2 && std::string(argv[1]) == "selftest") return doSelfTest();
  if (argc == 4 && std::string(argv[1]) == "loadstress") return doLoadStress(argv[2], unsigned(std::atoi(argv[3])));
  if (argc == 4 && std::string(argv[1]) == "bvhcheck") return doBvhCheck(argv[2], unsigned(std::atoi(argv[3])));
  if (argc == 6 && std::string(argv[1]) == "estimator")
    return doEstimator(std::atoi(argv[2]), unsigned(std::atoi(argv[3])), argv[4], argv[5]);
  if (argc == 8 && std::string(argv[1]) == "tonemap")
    return doTonemap(argv[2], unsigned(std::atoi(argv[3])), unsigned(std::atoi(argv[4])), argv[5], argv[6], argv[7]);
  if (argc != 5) {
    std::fprintf(stderr, "usage: hostsim kat|render <scene.yscn> <params.txt> <out>\n");
    return 1;
  }
  try {
    auto loaded = loadSceneFile(argv[2]);
    auto p = params::load(argv[3]);
    Ctx c;
    c.im = buildHostImage(loaded->desc);
    c.sc = c.im.view();
    c.cam = makeCamera(cameraDesc(p));
    c.rc.sampler = makeSamplerConfig(p.spp, p.tile);
    c.rc.maxDepth = p.depth;
    c.rc.background = mk3(p.background[0], p.background[1], p.background[2]);
    std::string mode = argv[1];
    if (mode == "kat") return doKat(c, p, argv[4]);
    if (mode == "render") return doRender(c, p, argv[4]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "hostsim: %s\n", e.what());
    return 2;
  }
  return 1;
}
