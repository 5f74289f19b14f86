// ref_driver.cpp — drives the *compiled reference* (teofum/yart, sources compiled
// where they lie under /root/reference by oracle/Makefile) through its public C++
// API. TEST INFRASTRUCTURE: produces ground-truth framebuffers and known-answer
// vectors (tests/golden/) and is the "reference" CPU baseline of bench.py. Nothing
// in the product links or calls this.
//
// It stands where reference src/main.cpp:19-107 stands: build a Scene, a Camera,
// a TileRenderer<SobolSampler<FastOwenScrambler>, MISIntegrator>, renderSync().
// The scene comes from a .yscn container (oracle/yscn.hpp) instead of the glTF
// loader (src/gltf/gltf.cpp needs fastgltf, which is not vendored).
//
// Modes:
//   yart_ref render <scene.yscn> <params.txt> <out.f32>     raw RGBA32F, W*H*4 floats
//   yart_ref kat    <scene.yscn> <params.txt> <out.json>    known-answer vectors
//   yart_ref luts   <out.bin>                               raw dump of the 8 LUT tables
//   yart_ref bvh    <scene.yscn> <mesh> <out.bin>           node array + index array
//   yart_ref tonemap <in.f32> <w> <h> <look|-> <out.f32> <out.ppm>   AgX (none|golden|punchy, "-": no
//                   tonemapper) applied as tile-renderer.hpp:234-240 does, then output::writePPM
//   yart_ref texture <image file> <C> <type 0|1|2> <c0,c1,..> <out.bin>   loadTexture<C>(bytes, len, type,
//                   channels) of core/texture.hpp:62-92 (stb_image decode + gamma-2 re-encode): u32 w, h, C + bytes
//   yart_ref estimator <kind 0 GMoN|1 Mean|2 MoN|3 GMoNb> <spp> <in.f32> <out.f32>   the reference's estimator classes
//                   (core/estimator.hpp, constructed as cpu/integrator.cpp:17-18 does: (spp, 15) / (spp)) fed groups of
//                   spp RGB samples in order; one getValue() per group
//   yart_ref writejpg <in image> <quality> <out.jpg>        stb_image_write's JPEG encoder (vendored by the reference
//                   next to stb_image) on the decoded RGB of the input: quality <= 90 gives 4:2:0, above 4:4:4
//   yart_ref hdr    <file.hdr> <out.bin>                    loadTextureHDR (core/texture.cpp:5-20): u32 w, h + floats
//   yart_ref xform  <in.txt> <out.bin>   per input line "tx ty tz qx qy qz qw sx sy sz parent": the node Transform
//                   of gltf.cpp:284-291 (float4x4::translation * rotationFromQuat * float4x4::scaling, inverse by
//                   Transform(float4x4)) and `node.transform * globalTransform` (:293) along the parent chain, with
//                   the reference's own float4x4 / Transform classes. rotationFromQuat is a file-static function of
//                   gltf.cpp (a TU that needs fastgltf), so its 10 lines are restated here on the reference's types.
//                   Output per line: local fwd, local inv, global fwd, global inv (16 floats each).
#include <core/core.hpp>
#include <cpu/tile-renderer.hpp>
#include <cpu/mis-integrator.hpp>
#include <bsdf/parametric.hpp>
#include <bsdf/luts.hpp>
#include <core/tonemapping.hpp>
#include <core/estimator.hpp>
#include <output/ppm.hpp>
#include <stb_image_write.h>

#include <cstdio>
#include <fstream>

#include "yscn.hpp"
#include "params.hpp"
#include "kat_common.hpp"

using namespace yart;
using namespace yart::math;

static uint32_t g_maxDepth = 30;

// The only way to choose a bounce depth: TileRenderer constructs the integrator
// itself (tile-renderer.hpp:157) and m_maxDepth is a public field of
// RayIntegrator (ray-integrator.hpp:14).
struct MISDepth : cpu::MISIntegrator {
  MISDepth(Buffer& b, const Camera& c, Sampler& s) noexcept : cpu::MISIntegrator(b, c, s) {
    m_maxDepth = g_maxDepth;
  }
  // Probes for the KAT mode (protected members of RayIntegrator / Integrator).
  float3 probeSample(uint2 p, uint32_t s) {
    m_sampler.startPixelSample(p, s);
    return sample(p.x(), p.y());
  }
  bool probeHit(const Ray& r, cpu::Hit& h) { return testNode(r, 0.001f, h, scene->root()); }
  // setup() is private in MISIntegrator; Integrator::render() (public) calls it and
  // then loops over an empty default samplingBounds.
  void probeSetup() { render(); }
};

static float4x4 mat16(const float* m) {
  std::array<float, 16> a;
  for (int i = 0; i < 16; i++) a[i] = m[i];
  return float4x4(a);
}

struct BuiltScene {
  std::unique_ptr<Scene> scene;
  std::vector<std::unique_ptr<HDRTexture>> hdr;   // env maps are owned by the caller (main.cpp:81)
  std::vector<const BSDF*> materials;
};

static Node buildNode(const yscn::SceneFile& sf, Scene& scene, size_t idx) {
  const auto& rec = sf.nodes[idx];
  Node node = rec.mesh >= 0 ? Node(&scene.mesh(size_t(rec.mesh))) : Node();
  node.transform = Transform(mat16(rec.fwd), mat16(rec.inv));
  for (size_t c = idx + 1; c < sf.nodes.size(); c++)
    if (sf.nodes[c].parent == int32_t(idx)) node.appendChild(buildNode(sf, scene, c));
  return node;
}

static BuiltScene build(const yscn::SceneFile& sf) {
  BuiltScene out;
  Node root;
  out.scene = std::make_unique<Scene>(std::move(root));
  Scene& scene = *out.scene;

  // Textures (already in the in-memory form texture.hpp:62-99 produces)
  std::vector<const void*> tex(sf.textures.size(), nullptr);
  std::vector<const HDRTexture*> texHdr(sf.textures.size(), nullptr);
  for (size_t i = 0; i < sf.textures.size(); i++) {
    const auto& t = sf.textures[i];
    TextureType type = TextureType(t.type);
    if (t.dtype == 1) {
      auto h = std::make_unique<HDRTexture>(t.width, t.height, type);
      h->data = t.f32;
      texHdr[i] = h.get();
      out.hdr.push_back(std::move(h));
    } else if (t.channels == 1) {
      auto p = std::make_unique<MonoTexture>(t.width, t.height, type);
      p->data = t.u8; tex[i] = scene.addTexture(std::move(p));
    } else if (t.channels == 2) {
      auto p = std::make_unique<SDRTexture<2>>(t.width, t.height, type);
      p->data = t.u8; tex[i] = scene.addTexture(std::move(p));
    } else if (t.channels == 3) {
      auto p = std::make_unique<RGBTexture>(t.width, t.height, type);
      p->data = t.u8; tex[i] = scene.addTexture(std::move(p));
    } else {
      auto p = std::make_unique<RGBATexture>(t.width, t.height, type);
      p->data = t.u8; tex[i] = scene.addTexture(std::move(p));
    }
  }
  auto T = [&](int32_t i) -> const void* { return i < 0 ? nullptr : tex[size_t(i)]; };

  for (const auto& m : sf.materials) {
    auto* bsdf = new ParametricBSDF(
      float3(m.base[0], m.base[1], m.base[2]),
      (const RGBATexture*) T(m.texBase),
      (const SDRTexture<2>*) T(m.texMR),
      (const MonoTexture*) T(m.texTransmission),
      (const RGBTexture*) T(m.texNormal),
      (const MonoTexture*) T(m.texClearcoat),
      (const RGBTexture*) T(m.texEmission),
      m.metallic, m.roughness, m.transmission, m.ior, m.anisotropic, m.anisoRotation,
      m.clearcoat, m.clearcoatRoughness,
      float3(m.emission[0], m.emission[1], m.emission[2]),
      m.normalScale, m.thinTransmission != 0,
      float3(m.volumeColor[0], m.volumeColor[1], m.volumeColor[2]), m.volumeDensity);
    out.materials.push_back(bsdf);
    scene.addMaterial(std::unique_ptr<BSDF>(bsdf));
  }

  size_t meshCount = 0;
  for (const auto& m : sf.meshes) {
    std::vector<float3> v(m.nVertices);
    std::vector<VertexData> vd(m.nVertices);
    std::vector<Face> f(m.nFaces);
    for (uint32_t i = 0; i < m.nVertices; i++) {
      v[i] = float3(m.positions[3 * i], m.positions[3 * i + 1], m.positions[3 * i + 2]);
      vd[i].normal = float3(m.normals[3 * i], m.normals[3 * i + 1], m.normals[3 * i + 2]);
      vd[i].tangent = float4(m.tangents[4 * i], m.tangents[4 * i + 1], m.tangents[4 * i + 2],
                             m.tangents[4 * i + 3]);
      vd[i].texCoords = float2(m.uvs[2 * i], m.uvs[2 * i + 1]);
    }
    for (uint32_t i = 0; i < m.nFaces; i++)
      f[i] = Face{m.faces[4 * i], m.faces[4 * i + 1], m.faces[4 * i + 2], m.faces[4 * i + 3]};
    scene.addMesh(std::make_unique<Mesh>(Mesh(v, vd, f)));
    Mesh& mesh = scene.mesh(meshCount++);
    for (uint32_t i = 0; i < m.nFaces; i++) mesh.lightIdx(i) = m.faceLight[i];
  }

  // Node tree: node 0 is the root (gltf.cpp:345-355 creates it empty, identity)
  if (!sf.nodes.empty()) {
    scene.root().transform = Transform(mat16(sf.nodes[0].fwd), mat16(sf.nodes[0].inv));
    for (size_t c = 1; c < sf.nodes.size(); c++)
      if (sf.nodes[c].parent == 0) scene.root().appendChild(buildNode(sf, scene, c));
  }

  for (const auto& l : sf.lights) {
    Transform tr(mat16(l.fwd), mat16(l.inv));
    float3 em(l.emission[0], l.emission[1], l.emission[2]);
    if (l.type == 0) {
      Mesh* mesh = &scene.mesh(size_t(l.mesh));
      AreaLight light(&mesh->triangles()[l.tri], mesh, em, tr);
      light.twoSided = l.twoSided != 0;
      scene.addLight(std::move(light));
    } else if (l.type == 1) {
      scene.addLight(UniformInfiniteLight(l.radius, em));
    } else {
      ImageInfiniteLight light(l.radius, texHdr[size_t(l.texture)]);
      light.transform = tr;
      scene.addLight(std::move(light));
    }
  }
  return out;
}

static Camera makeCamera(const params::Params& p) {
  Camera cam({p.width, p.height}, p.focal, p.fnumber, {p.sensor[0], p.sensor[1]});
  cam.exposure = p.exposure;
  cam.apertureSides = p.apertureSides;
  cam.moveAndLookAt({p.eye[0], p.eye[1], p.eye[2]}, {p.target[0], p.target[1], p.target[2]},
                    {p.up[0], p.up[1], p.up[2]});
  return cam;
}

using RefRenderer = cpu::TileRenderer<SobolSampler<FastOwenScrambler>, MISDepth>;

static int doRender(const std::string& scenePath, const std::string& paramPath,
                    const std::string& outPath) {
  auto sf = yscn::load(scenePath);
  auto p = params::load(paramPath);
  if (p.shardWorld > 1) { std::fprintf(stderr, "yart_ref: TileRenderer renders every tile (shard_* keys are the oracle restatement's)\n"); return 2; }
  auto built = build(sf);
  Camera cam = makeCamera(p);
  g_maxDepth = p.depth;

  RefRenderer r(Buffer(p.width, p.height), cam);
  r.scene = built.scene.get();
  r.samples = p.spp;
  r.firstWaveSamples = p.firstWave;
  r.maxWaveSamples = p.maxWave;
  r.tileSize = p.tile;
  if (p.threads) r.threadCount = p.threads;
  else {
    // No thread count asked for: at most one worker per four tiles (at least one). TileRenderer's workers read m_currentWave
    // without the lock (tile-renderer.hpp:160-190); a worker that only starts when two short waves are already over waits for a
    // wave number that has passed, and renderSync() never returns — seen with one-tile frames on a many-core host (2 of 4000).
    const uint32_t ts = p.tile ? p.tile : 64u;
    const uint64_t tiles = uint64_t((p.width + ts - 1) / ts) * ((p.height + ts - 1) / ts);
    r.threadCount = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(r.threadCount, tiles / 4)));
  }
  r.backgroundColor = float3(p.background[0], p.background[1], p.background[2]);

  // The ray count reported is the sum of the per-tile counts the renderer hands its tile callback (called under the buffer lock, one
  // tile at a time). RenderData::totalRays is the same sum added up WITHOUT the lock (tile-renderer.hpp:217-218: two plain `+=` of
  // a uint64_t from every worker): with several workers it comes out short now and then — exactly one 64x64 tile's rays missing in
  // 10 of 60 full-HD renders on a 256-thread host. Printed next to it as total_rays_unsynchronised.
  uint64_t tileRaySum = 0;
  r.onRenderTileComplete = [&](Renderer::RenderData, Renderer::TileData t) { tileRaySum += t.rays; };
  auto t0 = std::chrono::high_resolution_clock::now();
  auto d = r.renderSync();
  auto t1 = std::chrono::high_resolution_clock::now();
  double sec = std::chrono::duration<double>(t1 - t0).count();

  std::vector<float> img(size_t(p.width) * p.height * 4);
  for (uint32_t y = 0; y < p.height; y++)
    for (uint32_t x = 0; x < p.width; x++)
      for (int c = 0; c < 4; c++) img[(size_t(y) * p.width + x) * 4 + c] = d.buffer(x, y)[c];
  FILE* f = std::fopen(outPath.c_str(), "wb");
  if (!f) { std::fprintf(stderr, "cannot write %s\n", outPath.c_str()); return 2; }
  std::fwrite(img.data(), sizeof(float), img.size(), f);
  std::fclose(f);

  double msamples = double(p.width) * p.height * p.spp / sec * 1e-6;
  std::printf("{\"rays\": %llu, \"total_rays_unsynchronised\": %llu, \"seconds\": %.6f, \"msamples_per_s\": %.6f, \"threads\": %u}\n",
              (unsigned long long) tileRaySum, (unsigned long long) d.totalRays, sec, msamples, r.threadCount);
  return 0;
}

static int doLuts(const std::string& outPath) {
  FILE* f = std::fopen(outPath.c_str(), "wb");
  if (!f) return 2;
  std::fwrite(lut::table_ggx_E, sizeof(float), 32 * 32, f);
  std::fwrite(lut::table_ggx_Eavg, sizeof(float), 32, f);
  std::fwrite(lut::table_ggx_base_E, sizeof(float), 16 * 16 * 16, f);
  std::fwrite(lut::table_ggx_base_Eavg, sizeof(float), 16 * 16, f);
  std::fwrite(lut::table_ggx_glass_E, sizeof(float), 16 * 16 * 16, f);
  std::fwrite(lut::table_ggx_glass_Eavg, sizeof(float), 16 * 16, f);
  std::fwrite(lut::table_ggx_glass_inv_E, sizeof(float), 16 * 16 * 16, f);
  std::fwrite(lut::table_ggx_glass_inv_Eavg, sizeof(float), 16 * 16, f);
  // Sobol generator matrix, dimension 1 (the only non-trivial one the sampler
  // reads: sampler.hpp:142-153 uses dim 0 = bit reversal, dim 1 = matrices[52..103])
  std::fwrite(sobol::matrices + sobol::sobolMatrixSize, sizeof(uint32_t), sobol::sobolMatrixSize, f);
  std::fclose(f);
  return 0;
}

static size_t countNodes(const BVH& bvh) {
  // m_nodesUsed is protected; every allocated node is reachable from the root.
  size_t maxIdx = 0;
  std::vector<size_t> st{0};
  while (!st.empty()) {
    size_t i = st.back(); st.pop_back();
    maxIdx = std::max(maxIdx, i);
    if (bvh[i].span == 0) { st.push_back(bvh[i].left); st.push_back(bvh[i].left + 1); }
  }
  return maxIdx + 1;
}

static void packNodes(const Mesh& mesh, std::vector<uint32_t>& nodes, std::vector<uint32_t>& idx) {
  const BVH& bvh = mesh.bvh();
  size_t n = countNodes(bvh);
  nodes.resize(n * 8);
  for (size_t i = 0; i < n; i++) {
    const BVHNode& b = bvh[i];
    float v[6] = {b.bounds.min[0], b.bounds.min[1], b.bounds.min[2],
                  b.bounds.max[0], b.bounds.max[1], b.bounds.max[2]};
    std::memcpy(&nodes[i * 8], v, 24);
    nodes[i * 8 + 6] = b.left;
    nodes[i * 8 + 7] = b.span;
  }
  idx.resize(mesh.triangles().size());
  for (size_t i = 0; i < idx.size(); i++) idx[i] = bvh.idx(i);
}

static int doBvh(const std::string& scenePath, size_t meshIdx, const std::string& outPath) {
  auto sf = yscn::load(scenePath);
  auto built = build(sf);
  std::vector<uint32_t> nodes, idx;
  packNodes(built.scene->mesh(meshIdx), nodes, idx);
  FILE* f = std::fopen(outPath.c_str(), "wb");
  if (!f) return 2;
  uint32_t hdr[2] = {uint32_t(nodes.size() / 8), uint32_t(idx.size())};
  std::fwrite(hdr, 4, 2, f);
  std::fwrite(nodes.data(), 4, nodes.size(), f);
  std::fwrite(idx.data(), 4, idx.size(), f);
  std::fclose(f);
  return 0;
}

static int doKat(const std::string& scenePath, const std::string& paramPath,
                 const std::string& outPath) {
  auto sf = yscn::load(scenePath);
  auto p = params::load(paramPath);
  auto built = build(sf);
  Scene& scene = *built.scene;
  Camera cam = makeCamera(p);
  g_maxDepth = p.depth;
  kat::Writer w(outPath);

  // --- integer routines (rng.hpp:25-100, math.hpp:102-134, math_base.hpp:156-170)
  {
    std::vector<uint64_t> h, mb, mo;
    std::vector<int64_t> l2;
    for (uint32_t d = 0; d < 48; d++) h.push_back(hash(d));
    for (uint64_t v : kat::mixInputs()) mb.push_back(mixBits(v));
    for (auto xy : kat::mortonInputs()) mo.push_back(encodeMorton2(xy.first, xy.second));
    for (float v : kat::log2Inputs()) l2.push_back(log2Int(v));
    w.u64("hash32", h); w.u64("mixbits", mb); w.u64("morton", mo); w.i64("log2int", l2);
  }

  // --- sampler streams (sampler.hpp:72-174, scrambler.hpp:53-69)
  {
    std::vector<float> out;
    for (const auto& c : kat::samplerCases()) {
      SobolSampler<FastOwenScrambler> s(c.spp, {c.tile, c.tile});
      s.startPixelSample({c.px, c.py}, c.sample);
      for (int k : kat::samplerPattern()) {
        if (k == 2) { float2 v = s.get2D(); out.push_back(v.x()); out.push_back(v.y()); }
        else out.push_back(s.get1D());
      }
    }
    w.f32("sampler", out);
  }

  // --- LUT lookups incl. negative cosines (luts.hpp:33-191)
  {
    std::vector<float> e, ea, be, bea, ge, gea;
    for (const auto& q : kat::lutInputs()) {
      e.push_back(lut::ggxE(q.c, q.r));
      ea.push_back(lut::ggxEavg(q.r));
      be.push_back(lut::ggxBaseE(q.f0, q.r, q.c));
      bea.push_back(lut::ggxBaseEavg(q.f0, q.r));
      ge.push_back(lut::ggxGlassE(q.ior, q.r, std::abs(q.c)));
      gea.push_back(lut::ggxGlassEavg(q.ior, q.r));
    }
    w.f32("ggxE", e); w.f32("ggxEavg", ea); w.f32("ggxBaseE", be);
    w.f32("ggxBaseEavg", bea); w.f32("ggxGlassE", ge); w.f32("ggxGlassEavg", gea);
  }

  // --- camera rays (camera.hpp:138-164)
  {
    std::vector<float> out;
    kat::Lcg rng(7);
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      for (int k = 0; k < 4; k++) {
        float a0 = rng.next(), a1 = rng.next(), a2 = rng.next(), a3 = rng.next();   // sequenced draws
        float2 uf(a0, a1), ul(a2, a3);
        Ray r = cam.getRay({p.probePixels[i], p.probePixels[i + 1]}, uf, ul);
        for (int c = 0; c < 3; c++) out.push_back(r.origin[c]);
        for (int c = 0; c < 3; c++) out.push_back(r.dir[c]);
      }
    }
    w.f32("camera_rays", out);
  }

  // --- BVH structure (bvh.hpp:41-184, 273-347): counts + FNV-1a of the packed arrays
  {
    std::vector<uint64_t> out;
    for (size_t m = 0; m < sf.meshes.size(); m++) {
      std::vector<uint32_t> nodes, idx;
      packNodes(scene.mesh(m), nodes, idx);
      out.push_back(nodes.size() / 8);
      out.push_back(kat::fnv1a(nodes.data(), nodes.size() * 4));
      out.push_back(kat::fnv1a(idx.data(), idx.size() * 4));
    }
    w.u64("bvh", out);
  }

  // Integrator probe (needs a buffer/sampler like tile-renderer.hpp:152-157 builds)
  Buffer tb(p.tile, p.tile);
  SobolSampler<FastOwenScrambler> sampler(p.spp, {p.tile, p.tile});
  MISDepth integ(tb, cam, sampler);
  integ.scene = &scene;
  integ.backgroundColor = float3(p.background[0], p.background[1], p.background[2]);
  integ.probeSetup();

  // --- closest hits for centre-of-pixel primary rays (ray-integrator.cpp:20-261)
  {
    std::vector<float> fo, ro6;        // hit_rays: the rays themselves (origin, direction), for the device-side check
    std::vector<int64_t> io;
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      Ray r = cam.getRay({p.probePixels[i], p.probePixels[i + 1]}, {0.5f, 0.5f}, {0.5f, 0.5f});
      for (int c = 0; c < 3; c++) ro6.push_back(r.origin[c]);
      for (int c = 0; c < 3; c++) ro6.push_back(r.dir[c]);
      cpu::Hit h;
      bool hit = integ.probeHit(r, h);
      io.push_back(hit);
      if (!hit) { io.push_back(-1); io.push_back(-1); io.push_back(0); for (int c = 0; c < 12; c++) fo.push_back(0); continue; }
      io.push_back(h.idx); io.push_back(h.lightIdx); io.push_back(h.backSide);
      fo.push_back(h.t); fo.push_back(h.uv[0]); fo.push_back(h.uv[1]);
      for (int c = 0; c < 3; c++) fo.push_back(h.p[c]);
      for (int c = 0; c < 3; c++) fo.push_back(h.n[c]);
      for (int c = 0; c < 3; c++) fo.push_back(h.tg[c]);
    }
    w.i64("hits_i", io); w.f32("hits_f", fo); w.f32("hit_rays", ro6);
  }

  // --- BSDF f / pdf / sample per material (bsdf.cpp:5-58, parametric.cpp:84-838)
  {
    std::vector<float> fo;
    std::vector<int64_t> io;
    const float3 n(0, 0, 1), t(1, 0, 0);
    for (size_t m = 0; m < built.materials.size(); m++) {
      const BSDF& b = *built.materials[m];
      kat::Lcg rng(1000 + uint32_t(m));
      for (int k = 0; k < kat::bsdfCasesPerMaterial; k++) {
        float r[12];
        for (int q = 0; q < 6; q++) r[q] = rng.sym();       // sequenced draws (argument order is unspecified)
        for (int q = 6; q < 12; q++) r[q] = rng.next();
        float3 wo = normalized(float3(r[0], r[1], r[2]));
        float3 wi = normalized(float3(r[3], r[4], r[5]));
        float2 uv(r[6] * 3.0f - 1.0f, r[7] * 3.0f - 1.0f);
        float2 u(r[8], r[9]);
        float uc = r[10], uc2 = r[11];
        bool reg = k & 1;
        float3 f = b.f(wo, wi, n, t, uv);
        float pdf = b.pdf(wo, wi, n, t, uv);
        BSDFSample s = b.sample(wo, n, t, uv, u, uc, uc2, reg);
        for (int c = 0; c < 3; c++) fo.push_back(f[c]);
        fo.push_back(pdf);
        io.push_back(s.scatter);
        for (int c = 0; c < 3; c++) fo.push_back(s.f[c]);
        for (int c = 0; c < 3; c++) fo.push_back(s.Le[c]);
        for (int c = 0; c < 3; c++) fo.push_back(s.wi[c]);
        fo.push_back(s.pdf); fo.push_back(s.roughness);
        fo.push_back(b.alpha(uv));
        float3 base = b.base(uv);
        for (int c = 0; c < 3; c++) fo.push_back(base[c]);
        float3 sn = b.normal(n, float4(1, 0, 0, 1), uv);
        for (int c = 0; c < 3; c++) fo.push_back(sn[c]);
        float3 att = b.attenuation(uc * 4.0f);
        for (int c = 0; c < 3; c++) fo.push_back(att[c]);
      }
      io.push_back(b.transparent());
    }
    w.i64("bsdf_i", io); w.f32("bsdf_f", fo);
  }

  // --- lights and light sampler (light.cpp:16-243, light-sampler.cpp:32-93)
  {
    std::vector<float> fo;
    std::vector<int64_t> io;
    PowerLightSampler ls;
    ls.init(&scene);
    size_t nl = scene.nLights();
    kat::Lcg rng(4242);
    for (size_t li : kat::lightSubset(nl)) {
      const Light& l = scene.light(li);
      fo.push_back(l.power());
      for (int k = 0; k < 4; k++) {
        float r0 = rng.sym(), r1 = rng.next(), r2 = rng.sym(), r3 = rng.next(), r4 = rng.next();
        float3 pp(r0 * 4.0f, r1 * 8.0f, r2 * 4.0f);
        float2 u(r3, r4);
        LightSample s = l.sample(pp, float3(0, 1, 0), u, 0.0f);
        for (int c = 0; c < 3; c++) fo.push_back(s.Li[c]);
        for (int c = 0; c < 3; c++) fo.push_back(s.wi[c]);
        for (int c = 0; c < 3; c++) fo.push_back(s.p[c]);
        for (int c = 0; c < 3; c++) fo.push_back(s.n[c]);
        fo.push_back(s.pdf);
        float w0 = rng.sym(), w1 = rng.sym(), w2 = rng.sym();
        float3 wi = normalized(float3(w0, w1, w2));
        fo.push_back(l.pdf(wi));
        float3 le = l.Le(octahedralUV(wi));
        for (int c = 0; c < 3; c++) fo.push_back(le[c]);
      }
      fo.push_back(ls.p(float3(), float3(), li));
    }
    if (nl > 0) {
      for (int k = 0; k < 32; k++) {
        float u = rng.next();
        SampledLight s = ls.sample(float3(), float3(), u);
        int64_t which = -1;
        for (size_t i = 0; i < nl; i++) if (&scene.light(i) == &s.light) which = int64_t(i);
        io.push_back(which);
        fo.push_back(s.p);
      }
    }
    w.i64("lights_i", io); w.f32("lights_f", fo);
  }

  // --- per-sample radiance for the probe pixels (mis-integrator.cpp:13-148) and the
  //     GMoN pixel value (integrator.cpp:15-25, estimator.hpp:148-198)
  {
    std::vector<float> rad, pix;
    float ev = std::exp2(cam.exposure);
    for (size_t i = 0; i + 1 < p.probePixels.size(); i += 2) {
      GMoNEstimator est(int32_t(p.spp), 15);
      for (uint32_t s = 0; s < p.spp; s++) {
        float3 L = integ.probeSample({p.probePixels[i], p.probePixels[i + 1]}, s);
        for (int c = 0; c < 3; c++) rad.push_back(L[c]);
        est.addSample(L * ev);
      }
      float3 v = est.getValue();
      for (int c = 0; c < 3; c++) pix.push_back(v[c]);
    }
    w.f32("radiance", rad); w.f32("gmon", pix);
    std::vector<uint64_t> rc{integ.rayCount()};
    w.u64("probe_rays", rc);
  }

  w.close();
  return 0;
}

static int doTonemap(const char* in, unsigned w, unsigned h, const std::string& look, const char* outF32, const char* outPpm) {
  std::ifstream f(in, std::ios::binary);
  std::vector<float> hdr(size_t(w) * h * 4);
  f.read(reinterpret_cast<char*>(hdr.data()), std::streamsize(hdr.size() * 4));
  if (!f) throw std::runtime_error("tonemap: short input");
  tonemap::AgX agx;
  if (look == "golden") agx.look = tonemap::AgX::golden;
  else if (look == "punchy") agx.look = tonemap::AgX::punchy;
  else if (look != "none" && look != "-") throw std::runtime_error("tonemap: unknown look");
  const tonemap::Tonemap* tm = look == "-" ? nullptr : &agx;
  Buffer buf(w, h);
  for (unsigned y = 0; y < h; y++)
    for (unsigned x = 0; x < w; x++) {
      const float* px = &hdr[(size_t(y) * w + x) * 4];
      const float4 v(px[0], px[1], px[2], px[3]);
      buf(x, y) = tm ? float4((*tm)(float3(v)), 1.0f) : v;      // tile-renderer.hpp:234-240
    }
  std::ofstream o(outF32, std::ios::binary);
  for (unsigned y = 0; y < h; y++)
    for (unsigned x = 0; x < w; x++) {
      const float4& v = buf(x, y);
      const float q[4] = {v[0], v[1], v[2], v[3]};
      o.write(reinterpret_cast<const char*>(q), 16);
    }
  std::ofstream ppm(outPpm, std::ios::binary);
  output::writePPM(ppm, buf);
  return 0;
}

template <size_t C>
static int textureDump(const std::vector<uint8_t>& file, TextureType type, const std::vector<uint32_t>& ch, const char* outPath) {
  std::array<uint32_t, C> channels{};
  for (size_t i = 0; i < C; i++) channels[i] = ch[i];
  SDRTexture<C> t = loadTexture<C>(file.data(), int32_t(file.size()), type, channels);
  std::ofstream o(outPath, std::ios::binary);
  const uint32_t hdr[3] = {t.width(), t.height(), uint32_t(C)};
  o.write(reinterpret_cast<const char*>(hdr), 12);
  o.write(reinterpret_cast<const char*>(t.data.data()), std::streamsize(size_t(hdr[0]) * hdr[1] * C));
  return 0;
}

static int doTexture(const char* inPath, int C, int type, const char* chans, const char* outPath) {
  std::ifstream f(inPath, std::ios::binary);
  if (!f) throw std::runtime_error(std::string("cannot open ") + inPath);
  std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<uint32_t> ch;
  for (const char* p = chans; *p;) { ch.push_back(uint32_t(std::strtoul(p, const_cast<char**>(&p), 10))); if (*p == ',') p++; }
  if (int(ch.size()) != C) throw std::runtime_error("channel list length != C");
  const TextureType tt = type == 0 ? TextureType::LinearRGB : type == 1 ? TextureType::sRGB : TextureType::NonColor;
  switch (C) {
    case 1: return textureDump<1>(file, tt, ch, outPath);
    case 2: return textureDump<2>(file, tt, ch, outPath);
    case 3: return textureDump<3>(file, tt, ch, outPath);
    case 4: return textureDump<4>(file, tt, ch, outPath);
  }
  throw std::runtime_error("C must be 1..4");
}

static int doEstimator(int kind, unsigned spp, const char* inPath, const char* outPath) {
  std::ifstream f(inPath, std::ios::binary);
  if (!f) throw std::runtime_error(std::string("cannot open ") + inPath);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const float* smp = reinterpret_cast<const float*>(raw.data());
  const size_t groups = raw.size() / 12 / spp;
  std::ofstream o(outPath, std::ios::binary);
  for (size_t g = 0; g < groups; g++) {
    std::unique_ptr<Estimator> e;
    switch (kind) {
      case 0: e = std::make_unique<GMoNEstimator>(int32_t(spp), 15); break;
      case 1: e = std::make_unique<MeanEstimator>(spp); break;
      case 2: e = std::make_unique<MoNEstimator>(int32_t(spp), 15); break;
      case 3: e = std::make_unique<GMoNbEstimator>(int32_t(spp), 15); break;
      default: throw std::runtime_error("estimator kind must be 0..3");
    }
    for (unsigned k = 0; k < spp; k++) {
      const float* p = smp + (g * spp + k) * 3;
      e->addSample(float3(p[0], p[1], p[2]));
    }
    const float3 v = e->getValue();
    const float q[3] = {v[0], v[1], v[2]};
    o.write(reinterpret_cast<const char*>(q), 12);
  }
  return 0;
}

static int doWriteJpg(const char* inPath, int quality, const char* outPath) {
  int w, h, n;
  unsigned char* px = stbi_load(inPath, &w, &h, &n, 3);
  if (!px) throw std::runtime_error(std::string("cannot decode ") + inPath);
  const int ok = stbi_write_jpg(outPath, w, h, 3, px, quality);
  stbi_image_free(px);
  if (!ok) throw std::runtime_error("stbi_write_jpg failed");
  return 0;
}

static int doHdr(const char* inPath, const char* outPath) {
  HDRTexture t = loadTextureHDR(inPath);
  std::ofstream o(outPath, std::ios::binary);
  const uint32_t hdr[2] = {t.width(), t.height()};
  o.write(reinterpret_cast<const char*>(hdr), 8);
  o.write(reinterpret_cast<const char*>(t.data.data()), std::streamsize(size_t(hdr[0]) * hdr[1] * 3 * sizeof(float)));
  return 0;
}

static float4x4 quatRotation(const float q[4]) {     // gltf.cpp:6-19, on the reference's float4x4
  float qr = q[3], qi = q[0], qj = q[1], qk = q[2];
  float4x4 half{
    0.5f - (qj * qj + qk * qk), (qi * qj - qr * qk), (qi * qk + qr * qj), 0.0f,
    (qi * qj + qr * qk), 0.5f - (qi * qi + qk * qk), (qj * qk - qr * qi), 0.0f,
    (qi * qk - qr * qj), (qj * qk + qr * qi), 0.5f - (qi * qi + qj * qj), 0.0f,
    0.0f, 0.0f, 0.0f, 0.5f
  };
  return half * 2.0f;
}

static int doXform(const char* inPath, const char* outPath) {
  std::ifstream in(inPath);
  if (!in) throw std::runtime_error(std::string("cannot open ") + inPath);
  std::ofstream o(outPath, std::ios::binary);
  std::vector<Transform> globals;
  float t[3], q[4], s[3];
  int parent;
  // Transform keeps its matrices private: column j is read back as M * e_j (mat.hpp:561-572; exact up to the
  // sign of zero entries, which "0 + m * 1" turns positive)
  auto dump = [&](const Transform& x, bool inverse) {
    float v[16];
    for (size_t j = 0; j < 4; j++) {
      float4 e(0.0f);
      e[j] = 1.0f;
      const float4 c = inverse ? x.inverse(e) : x(e);
      for (size_t i = 0; i < 4; i++) v[i * 4 + j] = c[i];
    }
    o.write(reinterpret_cast<const char*>(v), sizeof(v));
  };
  while (in >> t[0] >> t[1] >> t[2] >> q[0] >> q[1] >> q[2] >> q[3] >> s[0] >> s[1] >> s[2] >> parent) {
    const float4x4 m = float4x4::translation(float3(t[0], t[1], t[2])) * quatRotation(q) * float4x4::scaling(float3(s[0], s[1], s[2]));
    const Transform local(m);
    const Transform global = local * (parent >= 0 ? globals.at(size_t(parent)) : Transform());
    globals.push_back(global);
    dump(local, false); dump(local, true); dump(global, false); dump(global, true);
  }
  return 0;
}

int main(int argc, char** argv) {
  std::string mode = argc > 1 ? argv[1] : "";
  try {
    if (mode == "render" && argc == 5) return doRender(argv[2], argv[3], argv[4]);
    if (mode == "kat" && argc == 5) return doKat(argv[2], argv[3], argv[4]);
    if (mode == "luts" && argc == 3) return doLuts(argv[2]);
    if (mode == "bvh" && argc == 5) return doBvh(argv[2], size_t(std::atoi(argv[3])), argv[4]);
    if (mode == "texture" && argc == 7) return doTexture(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), argv[5], argv[6]);
    if (mode == "hdr" && argc == 4) return doHdr(argv[2], argv[3]);
    if (mode == "estimator" && argc == 6) return doEstimator(std::atoi(argv[2]), unsigned(std::atoi(argv[3])), argv[4], argv[5]);
    if (mode == "writejpg" && argc == 5) return doWriteJpg(argv[2], std::atoi(argv[3]), argv[4]);
    if (mode == "xform" && argc == 4) return doXform(argv[2], argv[3]);
    if (mode == "tonemap" && argc == 8)
      return doTonemap(argv[2], unsigned(std::atoi(argv[3])), unsigned(std::atoi(argv[4])), argv[5], argv[6], argv[7]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "yart_ref: %s\n", e.what());
    return 2;
  }
  std::fprintf(stderr, "usage: yart_ref render|kat <scene.yscn> <params.txt> <out> | luts <out> | bvh <scene> <mesh> <out>\n");
  return 1;
}
