// oracle_main.cpp — TEST INFRASTRUCTURE: command-line front end of the CPU restatement.
//   yart_oracle kat    <scene.yscn> <params.txt> <out.json>   same KAT program as ref_driver.cpp
//   yart_oracle render <scene.yscn> <params.txt> <out.f32>    tile-threaded render (the reference's
//                      scheme: 64x64 tiles popped from a shared queue, tile-renderer.hpp:118-309)
// LUT tables are read from tests/golden/ref_tables.bin (dumped from the compiled reference;
// override with YART_ORACLE_LUTS).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <unistd.h>

#include "kat_common.hpp"
#include "params.hpp"
#include "yart_oracle.hpp"

using namespace orc;

static std::vector<float> loadLuts() {
  std::string path;
  if (const char* e = std::getenv("YART_ORACLE_LUTS")) path = e;
  else {
    char buf[4096];
    ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
    std::string exe = n > 0 ? std::string(buf, size_t(n)) : std::string(".");
    path = exe.substr(0, exe.rfind('/')) + "/../../tests/golden/ref_tables.bin";
  }
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open LUT tables " + path);
  std::vector<float> t(14112);
  if (std::fread(t.data(), 4, t.size(), f) != t.size()) throw std::runtime_error("short LUT file");
  std::fclose(f);
  return t;
}

static Camera makeCamera(const params::Params& p) {
  Camera c(p.width, p.height, p.focal, p.fnumber, p.sensor[0], p.sensor[1], V3(p.eye[0], p.eye[1], p.eye[2]),
           V3(p.target[0], p.target[1], p.target[2]), V3(p.up[0], p.up[1], p.up[2]));
  c.exposure = p.exposure; c.apertureSides = p.apertureSides;
  return c;
}

static void push3(std::vector<float>& v, V3 a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }

static int doKat(const Scene& sc, const yscn::SceneFile& sf, const params::Params& p, const std::string& out) {
  kat::Writer w(out);
  Camera cam = makeCamera(p);
  {
    std::vector<uint64_t> h, mb, mo; std::vector<int64_t> l2;
    for (uint32_t d = 0; d < 48; d++) h.push_back(hashU32(d));
    for (uint64_t v : kat::mixInputs()) mb.push_back(mixBits(v));
    for (auto xy : kat::mortonInputs()) mo.push_back(encodeMorton2(xy.first, xy.second));
    for (float v : kat::log2Inputs()) l2.push_back(log2Int(v));
    w.u64("hash32", h); w.u64("mixbits", mb); w.u64("morton", mo); w.i64("log2int", l2);
  }
  {
    std::vector<float> o;
    for (const auto& c : kat::samplerCases()) {
      SobolSampler s(c.spp, c.tile);
      s.startPixelSample(c.px, c.py, c.sample);
      for (int k : kat::samplerPattern()) {
        if (k == 2) { V2 v = s.get2D(); o.push_back(v.x); o.push_back(v.y); } else o.push_back(s.get1D());
      }
    }
    w.f32("sampler", o);
  }
  {
    std::vector<float> e, ea, be, bea, ge, gea;
    for (const auto& q : kat::lutInputs()) {
      e.push_back(sc.luts.ggxE(q.c, q.r)); ea.push_back(sc.luts.ggxEavg(q.r));
      be.push_back(sc.luts.ggxBaseE(q.f0, q.r, q.c)); bea.push_back(sc.luts.ggxBaseEavg(q.f0, q.r));
      ge.push_back(sc.luts.ggxGlassE(q.ior, q.r, std::abs(q.c))); gea.push_back(0.0f);
    }
    w.f32("ggxE", e); w.f32("ggxEavg", ea); w.f32("ggxBaseE", be); w.f32("ggxBaseEavg", bea);
    w.f32("ggxGlassE", ge); w.f32("ggxGlassEavg", gea);
  }
  {
    std::vector<float> o;
    kat::Lcg rng(7);
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2)
      for (int k = 0; k < 4; k++) {
        float a0 = rng.next(), a1 = rng.next(), a2 = rng.next(), a3 = rng.next();
        V3 ro, rd;
        cam.ray(p.probePixels[i], p.probePixels[i + 1], V2{a0, a1}, V2{a2, a3}, ro, rd);
        push3(o, ro); push3(o, rd);
      }
    w.f32("camera_rays", o);
  }
  {
    std::vector<uint64_t> o;
    for (const auto& m : sc.meshes) {
      std::vector<uint32_t> nodes(m->nodesUsed * 8), idx(m->idx.size());
      for (size_t i = 0; i < m->nodesUsed; i++) {
        float v[6] = {m->nodes[i].b.mn.x, m->nodes[i].b.mn.y, m->nodes[i].b.mn.z,
                      m->nodes[i].b.mx.x, m->nodes[i].b.mx.y, m->nodes[i].b.mx.z};
        std::memcpy(&nodes[i * 8], v, 24);
        nodes[i * 8 + 6] = m->nodes[i].leftFirst; nodes[i * 8 + 7] = m->nodes[i].span;
      }
      for (size_t i = 0; i < idx.size(); i++) idx[i] = uint32_t(m->idx[i]);
      o.push_back(m->nodesUsed);
      o.push_back(kat::fnv1a(nodes.data(), nodes.size() * 4));
      o.push_back(kat::fnv1a(idx.data(), idx.size() * 4));
    }
    w.u64("bvh", o);
  }
  SobolSampler sampler(p.spp, p.tile);
  Integrator integ;
  integ.scene = &sc; integ.cam = &cam; integ.sampler = &sampler; integ.maxDepth = p.depth;
  integ.background = V3(p.background[0], p.background[1], p.background[2]);
  {
    std::vector<float> fo, ro6; std::vector<int64_t> io;
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      V3 ro, rd;
      cam.ray(p.probePixels[i], p.probePixels[i + 1], V2{0.5f, 0.5f}, V2{0.5f, 0.5f}, ro, rd);
      push3(ro6, ro); push3(ro6, rd);
      Hit h;
      bool hit = integ.testNode(Ray(ro, rd), 0.001f, h, *sc.root);
      io.push_back(hit);
      if (!hit) { io.push_back(-1); io.push_back(-1); io.push_back(0); for (int c = 0; c < 12; c++) fo.push_back(0); continue; }
      io.push_back(h.idx); io.push_back(h.lightIdx); io.push_back(h.backSide);
      fo.push_back(h.t); fo.push_back(h.uv.x); fo.push_back(h.uv.y);
      push3(fo, h.p); push3(fo, h.n); push3(fo, h.tg);
    }
    w.i64("hits_i", io); w.f32("hits_f", fo); w.f32("hit_rays", ro6);
  }
  {
    std::vector<float> fo; std::vector<int64_t> io;
    const V3 n(0, 0, 1), t(1, 0, 0);
    for (size_t m = 0; m < sc.materials.size(); m++) {
      const Material& b = sc.materials[m];
      kat::Lcg rng(1000 + uint32_t(m));
      for (int k = 0; k < kat::bsdfCasesPerMaterial; k++) {
        float r[12];
        for (int q = 0; q < 6; q++) r[q] = rng.sym();
        for (int q = 6; q < 12; q++) r[q] = rng.next();
        V3 wo = normalized(V3(r[0], r[1], r[2])), wi = normalized(V3(r[3], r[4], r[5]));
        V2 uv{r[6] * 3.0f - 1.0f, r[7] * 3.0f - 1.0f}, u{r[8], r[9]};
        float uc = r[10], uc2 = r[11];
        push3(fo, b.f(wo, wi, n, t, uv));
        fo.push_back(b.pdf(wo, wi, n, t, uv));
        BSDFSample s = b.sample(wo, n, t, uv, u, uc, uc2, k & 1);
        io.push_back(s.scatter);
        push3(fo, s.f); push3(fo, s.Le); push3(fo, s.wi);
        fo.push_back(s.pdf); fo.push_back(s.roughness);
        fo.push_back(b.alpha(uv));
        push3(fo, b.baseAt(uv));
        push3(fo, b.normal(n, V4{1, 0, 0, 1}, uv));
        push3(fo, b.attenuation(uc * 4.0f));
      }
      io.push_back(b.transparent());
    }
    w.i64("bsdf_i", io); w.f32("bsdf_f", fo);
  }
  {
    std::vector<float> fo; std::vector<int64_t> io;
    size_t nl = sc.lights.size();
    kat::Lcg rng(4242);
    for (size_t li : kat::lightSubset(nl)) {
      const Light& l = *sc.lights[li];
      fo.push_back(l.power());
      for (int k = 0; k < 4; k++) {
        float r0 = rng.sym(), r1 = rng.next(), r2 = rng.sym(), r3 = rng.next(), r4 = rng.next();
        LightSample s = l.sample(V3(r0 * 4.0f, r1 * 8.0f, r2 * 4.0f), V2{r3, r4});
        push3(fo, s.Li); push3(fo, s.wi); push3(fo, s.p); push3(fo, s.n); fo.push_back(s.pdf);
        float w0 = rng.sym(), w1 = rng.sym(), w2 = rng.sym();
        V3 wi = normalized(V3(w0, w1, w2));
        fo.push_back(l.pdf(wi));
        push3(fo, l.Le(octahedralUV(wi)));
      }
      fo.push_back(sc.lightP(li));
    }
    if (nl > 0)
      for (int k = 0; k < 32; k++) {
        float u = rng.next(), pl;
        const Light* l = sc.sampleLight(u, pl);
        int64_t which = -1;
        for (size_t i = 0; i < nl; i++) if (sc.lights[i].get() == l) which = int64_t(i);
        io.push_back(which); fo.push_back(pl);
      }
    w.i64("lights_i", io); w.f32("lights_f", fo);
  }
  {
    std::vector<float> rad, pix;
    float ev = std::exp2(cam.exposure);
    integ.rays = 0;
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      GMoN est(int32_t(p.spp), 15);
      for (uint32_t s = 0; s < p.spp; s++) {
        V3 L = integ.samplePixel(p.probePixels[i], p.probePixels[i + 1], s);
        push3(rad, L);
        est.add(L * ev);
      }
      push3(pix, est.value());
    }
    w.f32("radiance", rad); w.f32("gmon", pix);
    std::vector<uint64_t> rc{integ.rays};
    w.u64("probe_rays", rc);
  }
  (void)sf;
  w.close();
  return 0;
}

static int doRender(const Scene& sc, const params::Params& p, const std::string& outPath) {
  const uint32_t W = p.width, H = p.height, T = p.tile;
  Camera cam = makeCamera(p);
  std::vector<float> hdr(size_t(W) * H * 4, 0.0f);
  struct Tile { uint32_t x, y, w, h; };
  std::vector<Tile> tiles;
  for (uint32_t y = 0; y < (H + T - 1) / T; y++)
    for (uint32_t x = 0; x < (W + T - 1) / T; x++)
      tiles.push_back({x * T, y * T, std::min(T, W - x * T), std::min(T, H - y * T)});
  // shard_* keys: only the pixel blocks the library deals to rank R of N (blocks of shard_tile pixels numbered in
  // Morton order of their block coordinates, block k to rank k % N) — include/yart_hip.h YartRenderParams
  const uint32_t ST = p.shardTile ? p.shardTile : T, bx = (W + ST - 1) / ST, by = (H + ST - 1) / ST;
  std::vector<uint8_t> mine(size_t(bx) * by, 1);
  if (p.shardWorld > 1) {
    std::vector<std::pair<uint64_t, uint32_t>> order;
    for (uint32_t y = 0; y < by; y++)
      for (uint32_t x = 0; x < bx; x++) order.push_back({encodeMorton2(x, y), y * bx + x});
    std::sort(order.begin(), order.end());
    for (size_t k = 0; k < order.size(); k++) mine[order[k].second] = (k % p.shardWorld) == p.shardRank;
  }
  uint64_t nPixelsMine = 0;
  for (uint32_t y = 0; y < H; y++)
    for (uint32_t x = 0; x < W; x++) nPixelsMine += mine[size_t(y / ST) * bx + x / ST];
  unsigned nt = p.threads ? p.threads : std::thread::hardware_concurrency();
  std::atomic<uint64_t> totalRays{0}, totalBox{0}, totalTri{0}, totalTrav{0}, totalShade{0}, neeBox{0}, neeTri{0}, neeTrav{0};
  auto t0 = std::chrono::high_resolution_clock::now();
  // wave schedule, tile-renderer.hpp:121-124, 284-289
  uint64_t remaining = p.spp, wave = std::min(p.firstWave, p.spp), current = 0;
  const float ev = std::exp2(cam.exposure);
  while (wave > 0) {
    const uint64_t before = p.spp - remaining, after = before + wave;
    const float wCur = float(before) / float(after), wWave = float(wave) / float(after);
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
      th.emplace_back([&] {
        SobolSampler sampler(p.spp, T);
        Integrator integ;
        integ.scene = &sc; integ.cam = &cam; integ.sampler = &sampler; integ.maxDepth = p.depth;
        integ.background = V3(p.background[0], p.background[1], p.background[2]);
        for (;;) {
          size_t ti = next++;
          if (ti >= tiles.size()) break;
          const Tile& tl = tiles[ti];
          for (uint32_t j = 0; j < tl.h; j++)
            for (uint32_t i = 0; i < tl.w; i++) {
              if (!mine[size_t((j + tl.y) / ST) * bx + (i + tl.x) / ST]) continue;
              GMoN est(int32_t(wave), 15);                                   // integrator.cpp:17
              for (uint32_t s = 0; s < wave; s++)
                est.add(integ.samplePixel(i + tl.x, j + tl.y, s + uint32_t(before)) * ev);
              V3 v = est.value();
              float* o = &hdr[(size_t(j + tl.y) * W + (i + tl.x)) * 4];
              const float wv[4] = {v.x, v.y, v.z, 1.0f};
              for (int c = 0; c < 4; c++) o[c] = o[c] * wCur + wv[c] * wWave;   // tile-renderer.hpp:230
            }
        }
        totalRays += integ.rays; totalBox += integ.nBox; totalTri += integ.nTri; totalTrav += integ.nTrav; totalShade += integ.nShade;
        neeBox += integ.nBoxNee; neeTri += integ.nTriNee; neeTrav += integ.nTravNee;
      });
    for (auto& t : th) t.join();
    remaining -= wave;
    uint64_t nextWave = (current > 0 || wave > 1) ? std::min<uint64_t>(wave * 2, p.maxWave) : 1;
    wave = std::min(nextWave, remaining);
    current++;
  }
  double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  FILE* f = std::fopen(outPath.c_str(), "wb");
  if (!f) return 2;
  std::fwrite(hdr.data(), 4, hdr.size(), f);
  std::fclose(f);
  std::printf("{\"rays\": %llu, \"seconds\": %.6f, \"msamples_per_s\": %.6f, \"threads\": %u, \"pixels\": %llu, "
              "\"traversals\": %llu, \"box_tests\": %llu, \"tri_tests\": %llu, \"shaded_hits\": %llu, "
              "\"shadow_traversals\": %llu, \"shadow_box_tests\": %llu, \"shadow_tri_tests\": %llu}\n",
              (unsigned long long) totalRays.load(), sec, double(nPixelsMine) * p.spp / sec * 1e-6, nt,
              (unsigned long long) nPixelsMine, (unsigned long long) totalTrav.load(), (unsigned long long) totalBox.load(),
              (unsigned long long) totalTri.load(), (unsigned long long) totalShade.load(),
              (unsigned long long) neeTrav.load(), (unsigned long long) neeBox.load(), (unsigned long long) neeTri.load());
  return 0;
}

// ---- tonemap mode: AgX + PPM bytes (reference core/tonemapping.hpp:14-92, output/ppm.cpp:7-21) -------------
// Scalar restatement with the reference's arithmetic order; double literals narrow to float where the
// reference's `scalar * vec` does (vec.hpp:282-288), log2 / pow are libm's (std::log2, std::pow on float).
namespace agx {
struct V { float x, y, z; };
static V mul(const float* m, V v) {             // float3x3 * float3, accumulate from 0 (mat.hpp:562-573)
  return {((0.0f + m[0] * v.x) + m[1] * v.y) + m[2] * v.z, ((0.0f + m[3] * v.x) + m[4] * v.y) + m[5] * v.z,
          ((0.0f + m[6] * v.x) + m[7] * v.y) + m[8] * v.z};
}
static float mn(float m, float n) { return m < n ? m : n; }      // math_base.hpp:83-92
static float mx(float m, float n) { return m > n ? m : n; }
static float contrast(float x) {
  float x2 = x * x, x4 = x2 * x2;
  return (((((((x4 * 15.5f) * x2) - ((x4 * 40.14f) * x)) + (x4 * 31.96f)) - ((x2 * 6.868f) * x)) + (x2 * 0.4298f)) +
          (x * 0.1191f)) - 0.00232f;
}
static V tonemap(V hdr, int look) {
  static const float A[9] = {float(0.842479062253094), float(0.0784335999999992), float(0.0792237451477643),
                             float(0.0423282422610123), float(0.878468636469772), float(0.0791661274605434),
                             float(0.0423756549057051), float(0.0784336), float(0.879142973793104)};
  static const float I[9] = {float(1.19687900512017), float(-0.0980208811401368), float(-0.0990297440797205),
                             float(-0.0528968517574562), float(1.15190312990417), float(-0.0989611768448433),
                             float(-0.0529716355144438), float(-0.0980434501171241), float(1.15107367264116)};
  const float minEv = -12.47393f, maxEv = 4.026069f;
  float slope[3] = {1, 1, 1}, power[3] = {1, 1, 1}, sat = 1.0f;
  if (look == 1) { slope[1] = 0.9f; slope[2] = 0.5f; power[0] = power[1] = power[2] = 0.8f; sat = 0.8f; }
  if (look == 2) { power[0] = power[1] = power[2] = 1.35f; sat = 1.4f; }
  V v = mul(A, hdr);
  float c[3] = {v.x, v.y, v.z};
  for (float& q : c) q = contrast((mn(maxEv, mx(minEv, std::log2(q))) - minEv) / (maxEv - minEv));
  const float luma = ((0.0f + c[0] * 0.2126f) + c[1] * 0.7152f) + c[2] * 0.0722f;     // dot, vec.hpp
  for (int k = 0; k < 3; k++) c[k] = luma + sat * (std::pow(c[k] * slope[k] + 0.0f, power[k]) - luma);
  v = mul(I, V{c[0], c[1], c[2]});
  float o[3] = {v.x, v.y, v.z};
  for (float& q : o) q = std::pow(mn(1.0f, mx(0.0f, q)), 2.2f);
  return {o[0], o[1], o[2]};
}
static uint8_t byteOf(float v) {
  const float gamma = 1.0f / 2.2f;
  float m = std::pow(v, gamma);
  m = m < 0.0f ? 0.0f : (1.0f < m ? 1.0f : m);
  if (!(m == m)) return 0;                        // x86 float -> uint8 of NaN
  return uint8_t(m * 255.999f);
}
}  // namespace agx

static int doTonemap(const char* in, unsigned w, unsigned h, const std::string& look, const char* outF32, const char* outPpm) {
  std::vector<float> px(size_t(w) * h * 4);
  FILE* f = std::fopen(in, "rb");
  if (!f || std::fread(px.data(), 4, px.size(), f) != px.size()) throw std::runtime_error("tonemap: short input");
  std::fclose(f);
  const int lk = look == "golden" ? 1 : look == "punchy" ? 2 : look == "none" ? 0 : -1;
  if (lk < 0 && look != "-") throw std::runtime_error("tonemap: unknown look");
  std::vector<uint8_t> bytes(size_t(w) * h * 3);
  for (size_t i = 0; i < size_t(w) * h; i++) {
    if (lk >= 0) {
      agx::V o = agx::tonemap({px[4 * i], px[4 * i + 1], px[4 * i + 2]}, lk);
      px[4 * i] = o.x; px[4 * i + 1] = o.y; px[4 * i + 2] = o.z; px[4 * i + 3] = 1.0f;
    }
    for (int c = 0; c < 3; c++) bytes[3 * i + c] = agx::byteOf(px[4 * i + c]);
  }
  f = std::fopen(outF32, "wb"); std::fwrite(px.data(), 4, px.size(), f); std::fclose(f);
  f = std::fopen(outPpm, "wb"); std::fprintf(f, "P6\n%u %u\n255\n", w, h); std::fwrite(bytes.data(), 1, bytes.size(), f); std::fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 8 && std::string(argv[1]) == "tonemap") {
    try { return doTonemap(argv[2], unsigned(std::atoi(argv[3])), unsigned(std::atoi(argv[4])), argv[5], argv[6], argv[7]); }
    catch (const std::exception& e) { std::fprintf(stderr, "yart_oracle: %s\n", e.what()); return 2; }
  }
  if (argc != 5) {
    std::fprintf(stderr, "usage: yart_oracle kat|render <scene.yscn> <params.txt> <out>\n");
    return 1;
  }
  try {
    auto luts = loadLuts();
    auto sf = yscn::load(argv[2]);
    auto p = params::load(argv[3]);
    Scene sc = buildScene(sf, luts.data());
    for (auto& m : sc.materials) m.lut = &sc.luts;     // the Scene object may have moved
    std::string mode = argv[1];
    if (mode == "kat") return doKat(sc, sf, p, argv[4]);
    if (mode == "render") return doRender(sc, p, argv[4]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "yart_oracle: %s\n", e.what());
    return 2;
  }
  return 1;
}
