// host_scene.hpp — host-side preparation of the flattened scene image:
// BVH build, leaf records, scene-graph flattening, material constants, light
// tables and environment-map distributions. Everything here runs once per scene
// (the reference does the same work in its constructors: Mesh::Mesh mesh.hpp:54-61,
// Node::appendChild scene.hpp:48-52, ParametricBSDF ctor parametric.cpp:11-68,
// AreaLight ctor light.cpp:16-34, ImageInfiniteLight ctor light.cpp:137-197,
// PowerLightSampler::init light-sampler.cpp:32-50, Camera::calcDerivedProperties
// camera.hpp:25-59) with the same float arithmetic, so the kernels start from
// bit-identical constants.
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <functional>
#include <vector>

#include "../../include/yart_hip.h"
#include "bvh_build.hpp"
#include "sampler.hpp"
#include "scene_types.hpp"

namespace yart_hip {

// Lobe class of a material: which of the branches of ParametricBSDF::sampleImpl / fImpl / pdfImpl
// (parametric.cpp:226-251) its hits can take — bit 0 transmissive, bit 1 clearcoat, bit 2 metallic.
inline uint8_t lobeClass(const MaterialDev& mt) {
  return uint8_t(((mt.cTrans > 0.0f || mt.texTransmission >= 0) ? 1u : 0u) |
                 ((mt.clearcoat > 0.0f || mt.texClearcoat >= 0) ? 2u : 0u) | (mt.cMetallic > 0.0f ? 4u : 0u));
}

struct HostImage {
  std::vector<ShadeTri> shadeTris;
  std::vector<BvhNode> bvhNodes;
  std::vector<LeafTri> leafTris;
  std::vector<u4> triVerts;
  std::vector<int32_t> triLight;
  std::vector<f4> vPos, vNormal, vTangent;
  std::vector<f2> vUV;
  std::vector<MeshDev> meshes;
  std::vector<NodeDev> nodes;
  std::vector<MaterialDev> materials;
  std::vector<TexDev> textures;
  std::vector<uint8_t> texU8;
  std::vector<float> texF32;
  size_t texQuadUnits = 0;                         // size of the device-side footprint records, in 16-byte units (TexDev::quadOffset)
  std::vector<LightDev> lights;
  std::vector<EnvDev> envs;
  std::vector<float> envData;
  std::vector<uint32_t> envGuide;                  // CDF search guide tables (EnvDev::guideOffset)
  std::vector<f4> nodeWorld;                       // conservative world-space node boxes (traverse.hpp)
  std::vector<TlasNode> tlas;                      // spatial hierarchy over the mesh nodes' boxes (scenes of 64 nodes and more)
  std::vector<uint32_t> infiniteLights, areaLights;
  std::vector<float> areaPowerCdf;
  std::vector<float> lut;                          // LutDev layout (incl. Sobol matrix bits)
  std::vector<std::vector<uint32_t>> bvhIndices;   // per mesh, for yart_hip_bvh_copy
  float totalPower = 0.0f;
  uint32_t maxNodeDepth = 0;
  uint32_t nLights = 0, nInfinite = 0, nArea = 0, nMaterials = 0;   // real counts (the vectors are padded)
  bool allIdentity = true;   // every node's transform chain is exactly the identity (TRAV_IDENTITY kernels)
  // Surface-area share of the most common lobe class (plain / metallic / clearcoat / transmissive combinations,
  // parametric.cpp:226-251 picks among them per hit) over the instanced triangles: a cheap stand-in for how mixed the
  // shade queue of this scene is. Below kShadeSortShare the shade kernel buckets its entries by material by default.
  float dominantLobeShare = 1.0f;

  SceneDev view() const {
    SceneDev s{};
    s.shadeTris = shadeTris.data();
    s.bvhNodes = bvhNodes.data(); s.leafTris = leafTris.data(); s.triVerts = triVerts.data();
    s.triLight = triLight.data(); s.vPos = vPos.data(); s.vNormal = vNormal.data();
    s.vTangent = vTangent.data(); s.vUV = vUV.data(); s.meshes = meshes.data(); s.nodes = nodes.data();
    s.materials = materials.data(); s.textures = textures.data(); s.texU8 = texU8.data();
    s.texF32 = texF32.data(); s.lights = lights.data(); s.envs = envs.data(); s.envData = envData.data();
    s.envGuide = envGuide.data(); s.nodeWorld = nodeWorld.data();
    s.tlas = tlas.data(); s.nTlas = uint32_t(tlas.size()); s.nodeBits = nullptr; s.nodeBitWords = 0;
    s.infiniteLights = infiniteLights.data(); s.areaLights = areaLights.data();
    s.areaPowerCdf = areaPowerCdf.data(); s.lut = lut.data();
    s.nMeshes = uint32_t(meshes.size()); s.nMaterials = uint32_t(materials.size()); s.nTextures = uint32_t(textures.size()); s.nEnvs = uint32_t(envs.size());
    s.nNodes = uint32_t(nodes.size()); s.nLights = nLights;
    s.nInfinite = nInfinite; s.nArea = nArea;
    s.totalPower = totalPower;
    return s;
  }
};

// LUT tables (bsdf/luts.cpp data, dumped by oracle/_ref into yart_amd/data/ggx_luts.bin
// and embedded at build time; see yart_amd/data/README.md)
const float* embeddedLutTables();      // 14112 floats, LutDev layout

inline void require(bool ok, const char* what) {
  if (!ok) throw std::invalid_argument(what);
}

// float4x4::rotation(angle, axis_z) -> float3x3 (mat.hpp:52-74, parametric.cpp:52-53)
inline void rotationZ3x3(float angle, float* m9) {
  const float a = angle;
  const float c = std::cos(a);
  const float s = std::sin(a);
  const float ax[3] = {0.0f, 0.0f, 1.0f};
  const float len = std::sqrt((ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2]);
  const float n[3] = {ax[0] / len, ax[1] / len, ax[2] / len};
  const float k = float(1.0 - double(c));          // "(1.0 - c) * nAxis": double subtract, then T(lhs)
  const float t[3] = {n[0] * k, n[1] * k, n[2] * k};
  m9[0] = c + t[0] * n[0];          m9[1] = t[1] * n[0] - s * n[2];   m9[2] = t[2] * n[0] + s * n[1];
  m9[3] = t[0] * n[1] + s * n[2];   m9[4] = c + t[1] * n[1];          m9[5] = t[2] * n[1] - s * n[0];
  m9[6] = t[0] * n[2] - s * n[1];   m9[7] = t[1] * n[2] + s * n[0];   m9[8] = c + t[2] * n[2];
}

inline float triangleAreaHost(f3 p0, f3 p1, f3 p2) {       // primitives.hpp:24-32
  return length(cross(p1 - p0, p2 - p0)) * 0.5f;
}

// optional builder for the BVH of one mesh (the device build of bvh_build_device.inc): fills nodes / indices and returns
// true, or returns false to leave the mesh to the host builder — the two give the same bytes
typedef bool (*MeshBvhFn)(void* ctx, const float* positions, uint32_t nVerts, const uint32_t* faces, uint32_t stride, uint32_t nFaces,
                          std::vector<BvhNode>& nodes, std::vector<uint32_t>& indices);
// Footprint copies of the interpolated LUTs (LutDev::fp*): what one lookup reads, side by side. `lut` holds the reference's
// tables (LutDev::E .. glassInvEavg); it is grown to LutDev::totalWithFootprints.
inline void appendLutFootprints(std::vector<float>& lut) {
  lut.resize(LutDev::totalWithFootprints);
  {
    float* L = lut.data();
    auto at = [&](uint32_t table, uint32_t i) { return L[table + i]; };
    for (uint32_t r = 0; r < 31; r++)
      for (uint32_t c = 0; c < 32; c++) {
        float* o = L + LutDev::fpE + (r * 32 + c) * 4;
        const uint32_t c1 = c + 1 < 32 ? c + 1 : c;              // (c = 31 is never a base index: sizeTClamp caps at 30)
        o[0] = at(LutDev::E, r * 32 + c); o[1] = at(LutDev::E, r * 32 + c1);
        o[2] = at(LutDev::E, (r + 1) * 32 + c); o[3] = at(LutDev::E, (r + 1) * 32 + c1);
      }
    for (uint32_t r = 0; r < 32; r++) {
      float* o = L + LutDev::fpEavg + r * 4;
      o[0] = at(LutDev::Eavg, r); o[1] = at(LutDev::Eavg, r + 1 < 32 ? r + 1 : r); o[2] = o[3] = 0.0f;
    }
    auto cube = [&](uint32_t src, uint32_t dst) {                 // record (i, j, k): src[(i+a)*256 + (j+b)*16 + (k+c)] at a*4 + b*2 + c
      for (uint32_t i = 0; i < 15; i++)
        for (uint32_t j = 0; j < 16; j++)
          for (uint32_t k = 0; k < 16; k++) {
            float* o = L + dst + ((i * 16 + j) * 16 + k) * 8;
            for (uint32_t a = 0; a < 2; a++)
              for (uint32_t b = 0; b < 2; b++)
                for (uint32_t c = 0; c < 2; c++) {
                  const uint32_t jj = j + b < 16 ? j + b : j, kk = k + c < 16 ? k + c : k;    // (15 is never a base index)
                  o[a * 4 + b * 2 + c] = at(src, ((i + a) * 16 + jj) * 16 + kk);
                }
          }
    };
    cube(LutDev::baseE, LutDev::fpBaseE); cube(LutDev::glassE, LutDev::fpGlassE); cube(LutDev::glassInvE, LutDev::fpGlassInvE);
    for (uint32_t i = 0; i < 15; i++)
      for (uint32_t j = 0; j < 16; j++) {
        float* o = L + LutDev::fpBaseEavg + (i * 16 + j) * 4;
        const uint32_t j1 = j + 1 < 16 ? j + 1 : j;
        o[0] = at(LutDev::baseEavg, i * 16 + j); o[1] = at(LutDev::baseEavg, i * 16 + j1);
        o[2] = at(LutDev::baseEavg, (i + 1) * 16 + j); o[3] = at(LutDev::baseEavg, (i + 1) * 16 + j1);
      }
  }
}

inline HostImage buildHostImage(const YartSceneDesc& d, MeshBvhFn bvhFn = nullptr, void* bvhCtx = nullptr) {
  require(d.n_nodes >= 1 && d.nodes, "scene needs a root node");
  require(d.n_materials == 0 || d.materials, "materials pointer is null");
  require(d.n_meshes == 0 || d.meshes, "meshes pointer is null");
  HostImage im;

  // ---- textures -------------------------------------------------------------
  for (uint32_t i = 0; i < d.n_textures; i++) {
    const YartTextureDesc& t = d.textures[i];
    require(t.data && t.width >= 2 && t.height >= 2 && t.channels >= 1 && t.channels <= 4,
            "texture: needs data, >= 2x2 texels, 1-4 channels");
    TexDev td{};
    td.width = t.width; td.height = t.height; td.channels = t.channels; td.type = t.type;
    td.isFloat = t.is_float ? 1u : 0u;
    size_t n = size_t(t.width) * t.height * t.channels;
    if (t.is_float) {
      td.offset = uint32_t(im.texF32.size());
      const float* p = static_cast<const float*>(t.data);
      im.texF32.insert(im.texF32.end(), p, p + n);
    } else {
      td.offset = uint32_t(im.texU8.size());
      const uint8_t* p = static_cast<const uint8_t*>(t.data);
      im.texU8.insert(im.texU8.end(), p, p + n);
      while (im.texU8.size() % 4) im.texU8.push_back(0);
    }
    td.quadOffset = kNoTexQuads;                               // its own footprint records: allocated below if something samples it directly
    im.textures.push_back(td);
  }
  auto texOk = [&](int32_t t, uint32_t ch, bool isFloat) {
    return t < 0 || (uint32_t(t) < d.n_textures && d.textures[t].channels == ch &&
                     (d.textures[t].is_float != 0) == isFloat);
  };

  // ---- materials (parametric.cpp:11-68) ---------------------------------------
  require(d.n_materials < 2047, "more than 2046 materials are not supported");   // wavefront.hpp::wfHitWord
  for (uint32_t i = 0; i < d.n_materials; i++) {
    const YartMaterialDesc& m = d.materials[i];
    require(texOk(m.tex_base, 4, false) && texOk(m.tex_mr, 2, false) && texOk(m.tex_transmission, 1, false) &&
            texOk(m.tex_normal, 3, false) && texOk(m.tex_clearcoat, 1, false) && texOk(m.tex_emission, 3, false),
            "material: texture index / channel count mismatch");
    MaterialDev md{};
    md.base = mk3(m.base[0], m.base[1], m.base[2]);
    md.emission = mk3(m.emission[0], m.emission[1], m.emission[2]);
    md.cTrans = m.transmission; md.cMetallic = m.metallic; md.ior = m.ior; md.roughness = m.roughness;
    md.anisotropic = m.anisotropic; md.clearcoat = m.clearcoat; md.clearcoatRoughness = m.clearcoat_roughness;
    md.volumeColor = mk3(m.volume_color[0], m.volume_color[1], m.volume_color[2]);
    md.volumeDensity = m.volume_density;
    md.texBase = m.tex_base; md.texMR = m.tex_mr; md.texTransmission = m.tex_transmission;
    md.texNormal = m.tex_normal; md.texClearcoat = m.tex_clearcoat; md.texEmission = m.tex_emission;
    rotationZ3x3(-m.aniso_rotation, md.localRot);
    rotationZ3x3(m.aniso_rotation, md.invRot);
    uint32_t flags = 0;
    if (m.thin_transmission) flags |= MAT_THIN;
    if (m.tex_base >= 0) {                                  // parametric.cpp:61-63
      const YartTextureDesc& t = d.textures[m.tex_base];
      const uint8_t* p = static_cast<const uint8_t*>(t.data);
      size_t n = size_t(t.width) * t.height * 4;
      for (size_t k = 3; k < n; k += 4) if (p[k] < 255) { flags |= MAT_HAS_ALPHA; break; }
    }
    if (length2(md.emission) > 0.0f) flags |= MAT_HAS_EMISSION;          // :67
    if (m.thin_transmission && m.transmission > 0.0f) flags |= MAT_TRANSPARENT;   // :80-82
    md.flags = flags;
    im.materials.push_back(md);
  }
  // One 64-byte footprint record per texel position for the base-colour / normal / metal-rough maps of a material (TexDev::quadStride):
  // materials with at least two of the three, all 8-bit and of one size, get cloned texture entries that point into an array shared by
  // every material with the same triple. (Only the device-side footprint records are shared; the texel arrays and every value stay.)
  {
    std::vector<std::pair<std::array<int32_t, 3>, uint32_t>> bundles;      // (base, normal, metal-rough) -> quadOffset of the shared array
    const uint32_t nTex = uint32_t(im.textures.size());
    for (MaterialDev& md : im.materials) {
      int32_t* slot[3] = {&md.texBase, &md.texNormal, &md.texMR};
      const uint32_t at[3] = {0u, 1u, 2u};                                   // member offsets in 16-byte units: base 0, normal 16 B, metal-rough 32 B
      uint32_t w = 0, h = 0, present = 0; bool ok = true;
      for (int k = 0; k < 3; k++) {
        const int32_t t = *slot[k];
        if (t < 0) continue;
        if (uint32_t(t) >= nTex || im.textures[t].isFloat) { ok = false; break; }
        if (present && (im.textures[t].width != w || im.textures[t].height != h)) { ok = false; break; }
        w = im.textures[t].width; h = im.textures[t].height; present++;
      }
      if (!ok || present < 2) continue;
      const std::array<int32_t, 3> key = {*slot[0], *slot[1], *slot[2]};
      uint32_t base = 0; bool found = false;
      for (auto& b : bundles) if (b.first == key) { base = b.second; found = true; break; }
      if (!found) {
        const size_t units = size_t(w) * h * 4u;                             // 64 B per texel position
        require(im.texQuadUnits + units < (size_t(1) << 32), "textures: more than 64 GB of footprint records");
        base = uint32_t(im.texQuadUnits); im.texQuadUnits += units;
        bundles.push_back({key, base});
      }
      for (int k = 0; k < 3; k++) {
        if (*slot[k] < 0) continue;
        TexDev c = im.textures[*slot[k]];
        c.quadOffset = base + at[k]; c.quadStride = 64u;
        // (one clone per (bundle, member) would do; a clone per material keeps this simple: TexDev is 32 bytes)
        *slot[k] = int32_t(im.textures.size());
        im.textures.push_back(c);
      }
    }
  }

  // Own footprint records only for the textures something still samples directly: a material slot that was not bundled above, an
  // emission / transmission / clearcoat map, an image light's map. (A texture reached only through bundle clones needs none: its
  // records would be built and never read — 4x its texels.)
  {
    const uint32_t nTex = uint32_t(d.n_textures);
    std::vector<uint8_t> direct(nTex, 0);
    auto mark = [&](int32_t t) { if (t >= 0 && uint32_t(t) < nTex) direct[t] = 1; };
    for (const MaterialDev& md : im.materials) { mark(md.texBase); mark(md.texNormal); mark(md.texMR); mark(md.texTransmission); mark(md.texClearcoat); mark(md.texEmission); }
    for (uint32_t i = 0; i < d.n_lights; i++) mark(d.lights[i].texture);
    for (uint32_t t = 0; t < nTex; t++) {
      if (!direct[t]) continue;
      TexDev& td = im.textures[t];
      const size_t units = (size_t(td.width) * td.height * texQuadRecordBytes(td.channels, td.isFloat) + 15u) / 16u;
      require(im.texQuadUnits + units < (size_t(1) << 32) - 1u, "textures: more than 64 GB of footprint records");
      td.quadOffset = uint32_t(im.texQuadUnits);
      im.texQuadUnits += units;
    }
  }

  // ---- meshes: BVH + leaf records (mesh.hpp:27-61, bvh.hpp) --------------------
  for (uint32_t mi = 0; mi < d.n_meshes; mi++) {
    const YartMeshDesc& m = d.meshes[mi];
    require(m.positions && m.normals && m.tangents && m.uvs && m.faces && m.n_faces > 0 && m.n_vertices > 0,
            "mesh: null array or empty mesh");
    // traversal stack entries pack (index | alpha bit << 26 | span << 27): node and leaf indices < 2^26
    require(m.n_faces < (1u << 25), "mesh: more than 2^25 triangles per mesh are not supported");
    for (uint32_t f = 0; f < m.n_faces; f++) {
      require(m.faces[4 * f] < m.n_vertices && m.faces[4 * f + 1] < m.n_vertices &&
              m.faces[4 * f + 2] < m.n_vertices, "mesh: vertex index out of range");
      require(m.faces[4 * f + 3] < d.n_materials, "mesh: material index out of range");
    }
    MeshDev md{};
    // children live at (left, left+1) with left odd (root = 0, pairs allocated after it),
    // so an odd mesh base puts every sibling pair on a 64-byte boundary
    if (im.bvhNodes.size() % 2 == 0) im.bvhNodes.push_back(BvhNode{});
    md.nodeOffset = uint32_t(im.bvhNodes.size());
    md.leafOffset = uint32_t(im.leafTris.size());
    md.triOffset = uint32_t(im.triVerts.size());
    md.vertOffset = uint32_t(im.vPos.size());
    md.nTris = m.n_faces; md.nVerts = m.n_vertices;

    struct { std::vector<BvhNode> nodes; std::vector<uint32_t> indices; } b;
    if (!(bvhFn && bvhFn(bvhCtx, m.positions, m.n_vertices, m.faces, 4, m.n_faces, b.nodes, b.indices))) {
      SahBvhBuilder hb;
      // YART_BVH_THREADS: worker threads of the build (default: all hardware threads, at most 32; 1 = serial)
      const char* e = std::getenv("YART_BVH_THREADS");
      hb.setThreads(e ? unsigned(std::atoi(e)) : 0u);
      hb.build(m.positions, m.faces, 4, m.n_faces);
      b.nodes = std::move(hb.nodes); b.indices = std::move(hb.indices);
    }
    md.nNodes = uint32_t(b.nodes.size());
    im.bvhNodes.insert(im.bvhNodes.end(), b.nodes.begin(), b.nodes.end());
    for (uint32_t k = 0; k < m.n_faces; k++) {
      uint32_t t = b.indices[k];
      const float* p0 = m.positions + size_t(m.faces[4 * t]) * 3;
      const float* p1 = m.positions + size_t(m.faces[4 * t + 1]) * 3;
      const float* p2 = m.positions + size_t(m.faces[4 * t + 2]) * 3;
      LeafTri lt{};
      for (int c = 0; c < 3; c++) { lt.p0[c] = p0[c]; lt.e1[c] = p1[c] - p0[c]; lt.e2[c] = p2[c] - p0[c]; }
      lt.triIdx = md.triOffset + t;       // index of the ShadeTri record (scene-wide): the hit needs no mesh look-up to find it
      lt.material = m.faces[4 * t + 3];
      lt.matFlags = im.materials[lt.material].flags & (MAT_HAS_ALPHA | MAT_TRANSPARENT);
      im.leafTris.push_back(lt);
    }
    {   // alpha bit per BVH node, bottom-up (children are allocated after their parent)
      std::vector<uint8_t> hasAlpha(b.nodes.size(), 0);
      for (size_t n = b.nodes.size(); n-- > 0;) {
        const BvhNode& bn = b.nodes[n];
        if (bn.span > 0) {
          for (uint32_t k = 0; k < bn.span; k++)
            if (im.leafTris[md.leafOffset + bn.leftFirst + k].matFlags & MAT_HAS_ALPHA) hasAlpha[n] = 1;
          // the leaf's span, for stack entries whose 5-bit span field saturated (traverse.hpp::leafSpan)
          require(bn.span < (1u << (32 - kLeafSpanShift)), "mesh: a BVH leaf of 2^24 or more triangles is not supported");
          im.leafTris[md.leafOffset + bn.leftFirst].matFlags |= bn.span << kLeafSpanShift;
        } else {
          hasAlpha[n] = hasAlpha[bn.leftFirst] | hasAlpha[bn.leftFirst + 1];
        }
        if (hasAlpha[n]) im.bvhNodes[md.nodeOffset + n].leftFirst |= kLinkAlphaBit;
        // ... and the node's span, saturated, in the link word's top bits: what the traversal keeps of a node (its stack entries,
        // the lean inner step's `cur`) is this word as loaded — no pack per step (traverse.hpp::packLink is idempotent on it)
        im.bvhNodes[md.nodeOffset + n].leftFirst |= (bn.span < kSpanBig ? bn.span : kSpanBig) << kSpanShift;
      }
      md.hasAlpha = hasAlpha.empty() ? 0u : hasAlpha[0];
    }
    for (uint32_t f = 0; f < m.n_faces; f++) {
      u4 tv; tv.x = m.faces[4 * f]; tv.y = m.faces[4 * f + 1]; tv.z = m.faces[4 * f + 2]; tv.w = m.faces[4 * f + 3];
      im.triVerts.push_back(tv);
      im.triLight.push_back(m.face_light ? m.face_light[f] : -1);
      ShadeTri st{};
      const uint32_t vi[3] = {tv.x, tv.y, tv.z};
      for (int k = 0; k < 3; k++) {
        for (int c = 0; c < 3; c++) st.n[k][c] = m.normals[3 * size_t(vi[k]) + c];
        for (int c = 0; c < 4; c++) st.t[k][c] = m.tangents[4 * size_t(vi[k]) + c];
        for (int c = 0; c < 2; c++) st.uv[k][c] = m.uvs[2 * size_t(vi[k]) + c];
      }
      st.material = tv.w; st.light = m.face_light ? m.face_light[f] : -1;
      im.shadeTris.push_back(st);
    }
    for (uint32_t v = 0; v < m.n_vertices; v++) {
      f4 p; p.x = m.positions[3 * v]; p.y = m.positions[3 * v + 1]; p.z = m.positions[3 * v + 2]; p.w = 0;
      f4 n; n.x = m.normals[3 * v]; n.y = m.normals[3 * v + 1]; n.z = m.normals[3 * v + 2]; n.w = 0;
      f4 t; t.x = m.tangents[4 * v]; t.y = m.tangents[4 * v + 1]; t.z = m.tangents[4 * v + 2]; t.w = m.tangents[4 * v + 3];
      im.vPos.push_back(p); im.vNormal.push_back(n); im.vTangent.push_back(t);
      im.vUV.push_back(mk2(m.uvs[2 * v], m.uvs[2 * v + 1]));
    }
    im.meshes.push_back(md);
    im.bvhIndices.push_back(std::move(b.indices));
  }
  // The lean inner step lets lanes that only pop read nodes[1], nodes[2] of their mesh unconditionally (the root's children;
  // the values are discarded). For a mesh whose tree is a single leaf those slots lie behind the mesh — for the LAST mesh behind
  // the array: two zero records keep that read inside the allocation.
  im.bvhNodes.push_back(BvhNode{}); im.bvhNodes.push_back(BvhNode{});

  // ---- scene graph (scene.hpp:11-64): pre-order with skip links, children-inclusive bounds
  const uint32_t nn = d.n_nodes;
  im.nodes.resize(nn);
  std::vector<Bounds3> nb(nn);
  for (uint32_t i = 0; i < nn; i++) {
    const YartNodeDesc& n = d.nodes[i];
    require(i == 0 ? n.parent < 0 : (n.parent >= 0 && uint32_t(n.parent) < i), "nodes must be in pre-order");
    require(n.mesh < int32_t(d.n_meshes), "node: mesh index out of range");
    NodeDev& nd = im.nodes[i];
    std::memcpy(nd.xf.fwd, n.fwd, 64); std::memcpy(nd.xf.inv, n.inv, 64);
    nd.mesh = n.mesh; nd.parent = n.parent;
    nd.depth = i == 0 ? 0 : im.nodes[n.parent].depth + 1;
    {   // pad[0] bit 0: this node's transform and all its ancestors' are exactly the identity
      static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      bool ident = std::memcmp(n.fwd, I, 64) == 0 && std::memcmp(n.inv, I, 64) == 0;
      const bool parentIdent = i == 0 || (im.nodes[n.parent].pad[0] & 1u);
      if (i > 0) ident = ident && parentIdent;
      // bit 1: the PARENT's chain is the identity (or there is no parent): the ray in the
      // parent's object space is the world ray (+0.0f)
      nd.pad[0] = (ident ? 1u : 0u) | (parentIdent ? 2u : 0u);
      if (!ident) im.allIdentity = false;
    }
    require(nd.depth < kMaxNodeDepth, "scene graph deeper than 8 levels");
    require(nn < (1u << 20), "more than 2^20 scene nodes are not supported");   // wavefront.hpp::wfHitWord
    if (nd.depth > im.maxNodeDepth) im.maxNodeDepth = nd.depth;
    if (n.mesh >= 0) {                                          // Node(Mesh*), scene.hpp:17-22
      const YartMeshDesc& m = d.meshes[n.mesh];
      for (uint32_t v = 0; v < m.n_vertices; v++) nb[i].expand(m.positions + size_t(v) * 3);
    }
  }
  for (uint32_t i = nn; i-- > 1;) {                             // appendChild bottom-up, scene.hpp:48-52
    const NodeDev& c = im.nodes[i];
    const Bounds3& b = nb[i];
    float corners[8][3];
    int k = 0;                                                  // transform.hpp:75-90 corner order
    for (int xi = 0; xi < 2; xi++) for (int yi = 0; yi < 2; yi++) for (int zi = 0; zi < 2; zi++) {
      f3 p = mulPoint(c.xf.fwd, mk3(xi ? b.mx[0] : b.mn[0], yi ? b.mx[1] : b.mn[1], zi ? b.mx[2] : b.mn[2]));
      corners[k][0] = p.x; corners[k][1] = p.y; corners[k][2] = p.z; k++;
    }
    const float* ptr[8];
    for (int q = 0; q < 8; q++) ptr[q] = corners[q];
    nb[c.parent].join(boundsFromPoints(ptr, 8));
  }
  // NB: children are joined in reverse order here; min/max folding is order independent.
  for (uint32_t i = 0; i < nn; i++) {
    NodeDev& nd = im.nodes[i];
    for (int c = 0; c < 3; c++) { nd.bmin[c] = nb[i].mn[c]; nd.bmax[c] = nb[i].mx[c]; }
    uint32_t j = i + 1;
    while (j < nn && im.nodes[j].depth > nd.depth) j++;
    nd.skip = j;
  }

  // Conservative world-space box per node: the node's local box pushed through its forward chain in
  // double precision, then padded far beyond the rounding of the reference's object-space test
  // (2e-3 + 1e-4 * |coordinate|; the boxes themselves already carry the reference's +-0.001).
  // The device tests it first (world ray, no transform) and runs the reference's exact test only on
  // the survivors, so it can only skip work the exact test would also have skipped.
  im.nodeWorld.resize(size_t(nn) * 2);
  for (uint32_t i = 0; i < nn; i++) {
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    const NodeDev& nd = im.nodes[i];
    bool finite = true;
    for (int corner = 0; corner < 8; corner++) {
      double p[3] = {(corner & 4) ? nd.bmax[0] : nd.bmin[0], (corner & 2) ? nd.bmax[1] : nd.bmin[1],
                     (corner & 1) ? nd.bmax[2] : nd.bmin[2]};
      for (int32_t a = int32_t(i); a >= 0; a = im.nodes[a].parent) {
        const float* m = im.nodes[a].xf.fwd;                    // row-major 4x4, points: w = 1
        double q[3];
        for (int r = 0; r < 3; r++) q[r] = double(m[4 * r]) * p[0] + double(m[4 * r + 1]) * p[1] + double(m[4 * r + 2]) * p[2] + double(m[4 * r + 3]);
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
      }
      for (int c = 0; c < 3; c++) {
        if (!std::isfinite(p[c])) finite = false;
        mn[c] = std::min(mn[c], p[c]); mx[c] = std::max(mx[c], p[c]);
      }
    }
    f4 lo, hi;
    float* l = &lo.x; float* h = &hi.x;
    for (int c = 0; c < 3; c++) {
      const double pad = 2e-3 + 1e-4 * std::max(std::fabs(mn[c]), std::fabs(mx[c]));
      l[c] = finite ? float(mn[c] - pad) : -kInf;               // float() rounds to nearest: the pad dwarfs it
      h[c] = finite ? float(mx[c] + pad) : kInf;
    }
    // .w of the pair: bits of the node's pre-order subtree [i, skip) as a 64-bit mask (scenes of fewer than
    // 64 nodes; trace_lean.hpp keeps a per-ray candidate mask)
    uint64_t sub = 0;
    if (nn < 64) for (uint32_t k = i; k < nd.skip; k++) sub |= 1ull << k;
    // scenes of 64 nodes and more (trace_lean_chunked.hpp walks the node list per lane): lo.w = the node's skip link, so a
    // missed box needs no second load
    const uint32_t subLo = nn < 64 ? uint32_t(sub) : nd.skip, subHi = uint32_t(sub >> 32);
    std::memcpy(&lo.w, &subLo, 4); std::memcpy(&hi.w, &subHi, 4);
    im.nodeWorld[2 * i] = lo; im.nodeWorld[2 * i + 1] = hi;
  }

  // ---- spatial hierarchy over the mesh nodes' padded world boxes (scenes of 64 nodes and more): median splits of the box
  // centres along the widest axis, one node per leaf. Any tree would do: a ray's query only has to return every mesh node whose
  // box it hits (trace_lean_tlas.hpp), the exact tests then run in the reference's pre-order.
  if (nn >= 64) {
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < nn; i++) if (im.nodes[i].mesh >= 0) ids.push_back(i);
    if (!ids.empty()) {
      // (a node whose padded world box is not finite — lo = -inf, hi = +inf — would give a NaN centre, and a comparator over
      // NaNs is not a strict weak ordering: undefined behaviour in std::nth_element. Such centres count as 0; the hierarchy
      // only filters, any split is valid.)
      auto centre = [&](uint32_t n, int c) {
        const float v = 0.5f * ((&im.nodeWorld[2 * n].x)[c] + (&im.nodeWorld[2 * n + 1].x)[c]);
        return std::isfinite(v) ? v : 0.0f;
      };
      im.tlas.resize(2 * ids.size() - 1);
      uint32_t used = 1;
      std::function<void(uint32_t, uint32_t, uint32_t)> build = [&](uint32_t at, uint32_t lo, uint32_t hi) {
        TlasNode t{};
        for (int c = 0; c < 3; c++) { t.lo[c] = kInf; t.hi[c] = -kInf; }
        float cmin[3] = {kInf, kInf, kInf}, cmax[3] = {-kInf, -kInf, -kInf};
        for (uint32_t k = lo; k < hi; k++)
          for (int c = 0; c < 3; c++) {
            t.lo[c] = std::min(t.lo[c], (&im.nodeWorld[2 * ids[k]].x)[c]); t.hi[c] = std::max(t.hi[c], (&im.nodeWorld[2 * ids[k] + 1].x)[c]);
            cmin[c] = std::min(cmin[c], centre(ids[k], c)); cmax[c] = std::max(cmax[c], centre(ids[k], c));
          }
        if (hi - lo == 1) { t.a = ids[lo]; t.b = 1; im.tlas[at] = t; return; }
        int axis = 0;
        for (int c = 1; c < 3; c++) if (cmax[c] - cmin[c] > cmax[axis] - cmin[axis]) axis = c;
        const uint32_t mid = lo + (hi - lo) / 2;
        std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi,
                         [&](uint32_t x, uint32_t y) { const float cx = centre(x, axis), cy = centre(y, axis); return cx < cy || (cx == cy && x < y); });
        t.a = used; t.b = 0; used += 2;
        im.tlas[at] = t;
        build(t.a, lo, mid); build(t.a + 1, mid, hi);
      };
      build(0, 0, uint32_t(ids.size()));
    }
  }
  if (im.tlas.empty()) im.tlas.resize(1);

  // ---- lights (light.cpp, light-sampler.cpp:32-50) ------------------------------
  for (uint32_t i = 0; i < d.n_lights; i++) {
    const YartLightDesc& l = d.lights[i];
    LightDev ld{};
    ld.type = l.type; ld.mesh = l.mesh; ld.tri = l.tri; ld.twoSided = l.two_sided;
    ld.emission = mk3(l.emission[0], l.emission[1], l.emission[2]);
    ld.radius = l.radius; ld.texture = l.texture;
    std::memcpy(ld.xf.fwd, l.fwd, 64); std::memcpy(ld.xf.inv, l.inv, 64);
    if (l.type == LIGHT_AREA) {
      require(l.mesh >= 0 && uint32_t(l.mesh) < d.n_meshes && l.tri < d.meshes[l.mesh].n_faces,
              "area light: mesh / triangle out of range");
      const YartMeshDesc& m = d.meshes[l.mesh];
      auto P = [&](uint32_t v) { return mk3(m.positions[3 * v], m.positions[3 * v + 1], m.positions[3 * v + 2]); };
      f3 t0 = mulPoint(ld.xf.fwd, P(m.faces[4 * l.tri]));
      f3 t1 = mulPoint(ld.xf.fwd, P(m.faces[4 * l.tri + 1]));
      f3 t2 = mulPoint(ld.xf.fwd, P(m.faces[4 * l.tri + 2]));
      ld.area = triangleAreaHost(t0, t1, t2);                                   // light.cpp:26-33
      ld.power = length(ld.emission) * ld.area * kPi * (l.two_sided ? 2.0f : 1.0f);   // :36-38
      im.areaLights.push_back(i);
      im.areaPowerCdf.push_back(im.totalPower + ld.power);                      // light-sampler.cpp:46-47
      im.totalPower += ld.power;
    } else if (l.type == LIGHT_UNIFORM_INF) {
      ld.power = 4.0f * kPi * kPi * l.radius * l.radius * length(ld.emission);   // light.cpp:98-103
      im.infiniteLights.push_back(i);
    } else if (l.type == LIGHT_IMAGE_INF) {
      require(l.texture >= 0 && uint32_t(l.texture) < d.n_textures && d.textures[l.texture].is_float &&
              d.textures[l.texture].channels == 3, "image light: needs a float RGB texture");
      const YartTextureDesc& t = d.textures[l.texture];
      const float* px = static_cast<const float*>(t.data);
      const uint32_t w = t.width, h = t.height;                 // full (0,0)-(1,1) bounds: x0=y0=0
      EnvDev e{};
      e.w = w; e.h = h;
      e.funcOffset = uint32_t(im.envData.size());
      std::vector<float> func(size_t(w) * h);
      f3 Lavg = mk3(0);
      for (uint32_t y = 0; y < h; y++) {                        // light.cpp:156-170
        float v = (float(y) + 0.5f) / float(h);
        float z = 1.0f - v * 2.0f;
        float sinTheta = std::sqrt(1.0f - z * z);
        for (uint32_t x = 0; x < w; x++) {
          const float* s = px + (size_t(x) + size_t(y) * w) * 3;
          float value = (((0.0f + s[0]) + s[1]) + s[2]) / 3.0f;
          func[size_t(y) * w + x] = std::fabs(value * sinTheta);    // PiecewiseConstant1D takes |f|
          Lavg += mk3(s[0], s[1], s[2]);
        }
      }
      Lavg /= float(w * h);
      im.envData.insert(im.envData.end(), func.begin(), func.end());
      e.cdfOffset = uint32_t(im.envData.size());
      std::vector<float> rowInt(h);
      for (uint32_t y = 0; y < h; y++) {                        // sampling.hpp:122-143
        std::vector<float> cdf(w + 1);
        cdf[0] = 0.0f;
        for (uint32_t k = 1; k < w + 1; k++) cdf[k] = cdf[k - 1] + func[size_t(y) * w + k - 1] * (1.0f - 0.0f) / float(w);
        float integral = cdf[w];
        if (integral == 0.0f) for (uint32_t k = 1; k < w + 1; k++) cdf[k] = float(k) / float(w);
        else for (uint32_t k = 1; k < w + 1; k++) cdf[k] /= integral;
        rowInt[y] = integral;
        im.envData.insert(im.envData.end(), cdf.begin(), cdf.end());
      }
      e.rowIntOffset = uint32_t(im.envData.size());
      for (uint32_t y = 0; y < h; y++) im.envData.push_back(std::fabs(rowInt[y]));
      e.margCdfOffset = uint32_t(im.envData.size());
      {
        std::vector<float> cdf(h + 1);
        cdf[0] = 0.0f;
        for (uint32_t k = 1; k < h + 1; k++) cdf[k] = cdf[k - 1] + std::fabs(rowInt[k - 1]) * (1.0f - 0.0f) / float(h);
        float integral = cdf[h];
        if (integral == 0.0f) for (uint32_t k = 1; k < h + 1; k++) cdf[k] = float(k) / float(h);
        else for (uint32_t k = 1; k < h + 1; k++) cdf[k] /= integral;
        e.margIntegral = integral;
        im.envData.insert(im.envData.end(), cdf.begin(), cdf.end());
      }
      float phi0 = 0.0f * 2.0f * kPi, phi1 = 1.0f * 2.0f * kPi;              // light.cpp:192-196
      float theta0 = 0.0f * kPi, theta1 = 1.0f * kPi;
      e.surfaceArea = (phi1 - phi0) * (std::cos(theta0) - std::cos(theta1));
      ld.power = e.surfaceArea * kPi * l.radius * l.radius * (((0.0f + Lavg.x) + Lavg.y) + Lavg.z) / 3.0f;   // :206-209
      {   // guide tables: G[j] = first index in [1, n) whose cdf is >= j / K (n if none), K = 2^k >= n
        auto pow2ge = [](uint32_t n) { uint32_t k = 1; while (k < n) k <<= 1; return k; };
        auto appendGuide = [&](const float* cdf, uint32_t n, uint32_t K) {
          uint32_t idx = 1;
          for (uint32_t j = 0; j <= K; j++) {
            const float uj = float(j) / float(K);               // exact: K is a power of two <= 2^24
            while (idx < n && cdf[idx] < uj) idx++;
            im.envGuide.push_back(idx);
          }
        };
        // K = 4 n: a cell of the guide then rarely holds a breakpoint, and the search that follows is mostly zero or one step —
        // a wave pays the LONGEST search of its 64 lanes (profiles/r5_shade_env_search.txt: shade kernel -1.7 % against K = n for 50 MB
        // of tables at 2048^2; K = 16 n: -2.4 % for 250 MB). YART_ENV_GUIDE_MUL overrides (measurements).
        uint32_t mul = 4;
        if (const char* g = std::getenv("YART_ENV_GUIDE_MUL")) mul = uint32_t(std::max(1, std::min(16, std::atoi(g))));
        e.guideKw = pow2ge(w) * mul; e.guideKh = pow2ge(h) * mul;
        e.guideOffset = uint32_t(im.envGuide.size());
        appendGuide(im.envData.data() + e.margCdfOffset, h, e.guideKh);
        for (uint32_t y = 0; y < h; y++) appendGuide(im.envData.data() + e.cdfOffset + size_t(y) * (w + 1), w, e.guideKw);
      }
      {   // interleaved copies for the sampling walk (EnvDev::pairOffset)
        while (im.envData.size() % 4u) im.envData.push_back(0.0f);          // 16-byte aligned records
        e.pairOffset = uint32_t(im.envData.size());
        std::vector<float> pairs(size_t(h + 1) * 4u + size_t(h) * (w + 1) * 2u, 0.0f);
        const float* D = im.envData.data();
        for (uint32_t k = 0; k <= h; k++) {
          pairs[size_t(k) * 4u] = D[e.margCdfOffset + k];
          if (k < h) { pairs[size_t(k) * 4u + 1u] = D[e.rowIntOffset + k]; pairs[size_t(k) * 4u + 2u] = D[e.cdfOffset + size_t(k) * (w + 1) + 1u]; }
        }
        float* rows = pairs.data() + size_t(h + 1) * 4u;
        for (uint32_t y = 0; y < h; y++)
          for (uint32_t k = 0; k <= w; k++) {
            rows[(size_t(y) * (w + 1) + k) * 2u] = D[e.cdfOffset + size_t(y) * (w + 1) + k];
            if (k < w) rows[(size_t(y) * (w + 1) + k) * 2u + 1u] = D[e.funcOffset + size_t(y) * w + k];
          }
        im.envData.insert(im.envData.end(), pairs.begin(), pairs.end());
      }
      ld.envOffset = uint32_t(im.envs.size());
      im.envs.push_back(e);
      im.infiniteLights.push_back(i);
    } else {
      require(false, "light: unknown type");
    }
    im.lights.push_back(ld);
  }
  // hit.lightIdx (per-mesh numbering) is used as a global light index by the reference
  // (mis-integrator.cpp:65): validate it stays in range.
  for (int32_t li : im.triLight) require(li < int32_t(im.lights.size()), "face_light index out of range");

  // ---- LUTs + Sobol dimension-1 matrix ------------------------------------------
  im.lut.assign(embeddedLutTables(), embeddedLutTables() + LutDev::sobol);
  appendLutFootprints(im.lut);
  for (uint32_t k = 0; k < 52; k++) {
    uint32_t bits = sobolDim1Column(k);
    std::memcpy(&im.lut[LutDev::sobol + k], &bits, 4);
  }
  im.nLights = uint32_t(im.lights.size()); im.nInfinite = uint32_t(im.infiniteLights.size());
  im.nArea = uint32_t(im.areaLights.size()); im.nMaterials = uint32_t(im.materials.size());
  // never hand the kernels a null pointer for an empty table
  im.texU8.insert(im.texU8.end(), 8, uint8_t(0));   // texelWord reads two aligned words for 3-channel texels
  if (im.texF32.empty()) im.texF32.resize(1);
  if (im.textures.empty()) im.textures.push_back(TexDev{});
  if (im.lights.empty()) im.lights.push_back(LightDev{});
  if (im.envs.empty()) im.envs.push_back(EnvDev{});
  if (im.envData.empty()) im.envData.resize(1);
  if (im.envGuide.empty()) im.envGuide.resize(1);
  if (im.infiniteLights.empty()) im.infiniteLights.push_back(0);
  if (im.areaLights.empty()) im.areaLights.push_back(0);
  if (im.areaPowerCdf.empty()) im.areaPowerCdf.push_back(0.0f);
  if (im.materials.empty()) im.materials.push_back(MaterialDev{});
  {   // lobe-class mix by instanced surface area (object-space areas: a heuristic, not a measurement)
    std::vector<uint32_t> uses(im.meshes.size(), 0);
    for (uint32_t i = 0; i < nn; i++) if (d.nodes[i].mesh >= 0) uses[d.nodes[i].mesh]++;
    double area[8] = {0, 0, 0, 0, 0, 0, 0, 0}, total = 0;
    for (uint32_t mi = 0; mi < d.n_meshes; mi++) {
      const YartMeshDesc& m = d.meshes[mi];
      for (uint32_t f = 0; f < m.n_faces; f++) {
        const float* p0 = m.positions + size_t(m.faces[4 * f]) * 3;
        const float* p1 = m.positions + size_t(m.faces[4 * f + 1]) * 3;
        const float* p2 = m.positions + size_t(m.faces[4 * f + 2]) * 3;
        const double e1[3] = {double(p1[0]) - p0[0], double(p1[1]) - p0[1], double(p1[2]) - p0[2]};
        const double e2[3] = {double(p2[0]) - p0[0], double(p2[1]) - p0[1], double(p2[2]) - p0[2]};
        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        const double a = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz) * uses[mi];
        const MaterialDev& mt = im.materials[m.faces[4 * f + 3]];
        area[lobeClass(mt)] += a; total += a;
      }
    }
    double top = 0;
    for (double a : area) top = std::max(top, a);
    im.dominantLobeShare = total > 0 ? float(top / total) : 1.0f;
  }
  return im;
}

// Camera::calcDerivedProperties (camera.hpp:25-59) after moveAndLookAt (:123-130)
inline CameraDev makeCamera(const YartCameraDesc& c) {
  require(c.width > 0 && c.height > 0 && c.focal_length > 0, "camera: bad image size / focal length");
  CameraDev cd{};
  const float aspect = float(c.width) / float(c.height);
  const f3 position = mk3(c.position[0], c.position[1], c.position[2]);
  const f3 forward = mk3(c.target[0], c.target[1], c.target[2]) - position;
  f3 up = mk3(c.up[0], c.up[1], c.up[2]);
  if (length2(up) == 0.0f) up = mk3(0, 1, 0);
  float sensorAspect = c.sensor[0] / c.sensor[1];
  float croppedSensorHeight = c.sensor[0] / ymax(sensorAspect, aspect);
  float focusDistance = length(forward);
  float vh = focusDistance * croppedSensorHeight / c.focal_length;
  float vw = vh * aspect;
  up = normalized(up);
  f3 w = normalized(-forward);
  f3 u = cross(up, w);
  f3 v = cross(w, u);
  Frame fr = frameFromNormalTangent(w, u, 1.0f);
  f3 viewportU = u * vw;
  f3 viewportV = (-v) * vh;
  f3 viewportTopLeft = (position - w * focusDistance) - (viewportU + viewportV) * 0.5f;
  cd.pixelDeltaU = viewportU / float(c.width);
  cd.pixelDeltaV = viewportV / float(c.height);
  cd.topLeftPixel = viewportTopLeft + (cd.pixelDeltaU + cd.pixelDeltaV) * 0.5f;
  cd.apertureRadius = c.f_number ? (c.focal_length / 2000.0f) / c.f_number : 0.0f;
  cd.apertureSides = c.aperture_sides;
  cd.position = position;
  cd.frameX = fr.x; cd.frameY = fr.y; cd.frameZ = fr.z;
  cd.exposureScale = std::exp2(c.exposure);                   // integrator.cpp:23
  return cd;
}



}  // namespace yart_hip
