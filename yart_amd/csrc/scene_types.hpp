// scene_types.hpp — the flattened, pointer-free scene image the kernels read.
//
// The reference holds the scene as a pointer graph (core/scene.hpp:11-169,
// core/mesh.hpp:15-128, core/bvh.hpp:21-33, core/light.hpp, bsdf/parametric.hpp).
// For the GPU everything becomes index-addressed arrays in HBM:
//
//   bvhNodes   32-byte nodes exactly as the reference builds them (bounds, left|first,
//              span); the two children of an inner node are adjacent, so one 64-byte
//              aligned fetch (nodes[left], nodes[left+1]) serves an inner visit.
//   leafTris   48-byte records in BVH leaf order (the reference chases
//              m_indices -> Triangle -> 3 vertices, 52 B over 3 dependent loads):
//              {p0, triIdx | e1 = p1-p0, material flags | e2 = p2-p0, material}.
//   triVerts / triLight / vertex arrays   per original triangle / vertex, read once
//              per accepted hit by the shading code.
//   nodes      scene-graph nodes in pre-order with a skip link, so that the
//              reference's recursive testNode becomes a linear walk.
#pragma once
#include "ymath.hpp"

namespace yart_hip {

constexpr uint32_t kMaxNodeDepth = 8;   // scene-graph nesting the node walk supports

enum : uint32_t { TEX_LINEAR = 0, TEX_SRGB = 1, TEX_NONCOLOR = 2 };
enum : uint32_t { LIGHT_AREA = 0, LIGHT_UNIFORM_INF = 1, LIGHT_IMAGE_INF = 2 };
enum : uint32_t {
  MAT_THIN = 1u, MAT_HAS_ALPHA = 2u, MAT_HAS_EMISSION = 4u, MAT_TRANSPARENT = 8u
};

struct TexDev {            // core/texture.hpp:21-49
  uint32_t offset;         // element offset into texU8 (bytes) or texF32 (floats)
  uint32_t width, height, channels;
  uint32_t type;           // TEX_*
  uint32_t isFloat;
  // 2x2 footprint records (device only, SceneDev::texQuads; built by k_tex_quads at upload): record (x, y) holds the four
  // bilinear taps (x, y), (x, y+1), (x+1, y), (x+1, y+1) side by side, so that one lookup is ONE aligned access of one 64-byte
  // sector instead of taps in two texel rows. u8: 4 / 8 / 16 bytes per record for 1 / 2 / 3-4 channels (a tap = one word,
  // channel c in byte c); float: 4 taps x channels floats, padded to a power of two (RGB: 64 bytes). In units of 16 bytes.
  uint32_t quadOffset;
  // Bytes from one footprint record to the next; 0 = the record's own size. The base-colour, metal-rough and normal maps of a
  // material that have the same size are laid out as ONE 64-byte record per texel position (base at byte 0, normal at 16,
  // metal-rough at 32: host_scene.hpp clones the TexDev entries of such a material and points them into the shared array), so
  // that the three lookups of a shaded hit — same uv, same position — touch one line and one page instead of three.
  uint32_t quadStride;
};
constexpr uint32_t kNoTexQuads = 0xffffffffu;   // TexDev::quadOffset of a texture without footprint records of its own (only sampled through bundle clones)
YART_HD uint32_t texQuadRecordBytes(uint32_t channels, uint32_t isFloat) {
  if (isFloat) return channels == 1 ? 16u : channels == 2 ? 32u : 64u;
  return channels == 1 ? 4u : channels == 2 ? 8u : 16u;
}

struct MaterialDev {       // bsdf/parametric.hpp:52-77
  f3 base; float cTrans;
  f3 emission; float cMetallic;
  float ior, roughness, anisotropic, clearcoat;
  float clearcoatRoughness; uint32_t flags; float volumeDensity; int32_t texBase;
  f3 volumeColor; int32_t texMR;
  int32_t texTransmission, texNormal, texClearcoat, texEmission;
  float localRot[9];       // float3x3(rotation(-anisoRotation, z)), parametric.cpp:52
  float invRot[9];         // float3x3(rotation(+anisoRotation, z)), parametric.cpp:53
  float pad[2];
};

// BvhNode::leftFirst as stored for the device (the traversal's LINK WORD): bits 27..31 min(span, 31), bits 0..25 the reference's index (left child / first
// leaf triangle), bit 26 set when the node's subtree contains a triangle with an alpha-tested
// material (used by the lean shadow kernel to end occluded rays early, traverse.hpp).
constexpr uint32_t kLinkIndexMask = (1u << 26) - 1u;
constexpr uint32_t kLinkAlphaBit = 1u << 26;
// traversal-stack link word: index | alpha bit | leaf span << 27 (5 bits, 31 = "31 or more": traverse.hpp::leafSpan
// then reads the true span from bits 8..31 of the matFlags word of the leaf's first LeafTri)
constexpr uint32_t kSpanShift = 27, kSpanBig = 31, kLeafSpanShift = 8;

struct BvhNode {           // core/bvh.hpp:21-33 (32 bytes)
  float bmin[3];
  float bmax[3];
  uint32_t leftFirst;
  uint32_t span;
};

struct LeafTri {           // 48 bytes, leaf order
  float p0[3]; uint32_t triIdx;      // MeshDev::triOffset + triangle: index into SceneDev::shadeTris
  float e1[3]; uint32_t matFlags;   // MAT_HAS_ALPHA | MAT_TRANSPARENT of the triangle's material; first record of a leaf: | span << 8
  float e2[3]; uint32_t material;
};

// Everything the shade stage needs about one triangle in one 128-byte line (vertex normals,
// tangents, uvs, material, light) — instead of an index fetch followed by nine scattered vertex
// attribute fetches (ten 64-byte sectors per shaded hit). Indexed like triVerts.
struct ShadeTri {
  float n[3][3];
  float t[3][4];
  float uv[3][2];
  uint32_t material; int32_t light;
  uint32_t pad[3];
};
static_assert(sizeof(ShadeTri) == 128, "ShadeTri is one cache line");

struct MeshDev {
  uint32_t nodeOffset;     // into bvhNodes
  uint32_t leafOffset;     // into leafTris
  uint32_t triOffset;      // into triVerts / triLight
  uint32_t vertOffset;     // into vertex arrays
  uint32_t nTris, nVerts, nNodes;
  uint32_t hasAlpha;       // some triangle of the mesh has an alpha-tested material
  uint32_t pad0[8];        // (words 8..10 held the roots of rounds 4-5's 8-wide trees: removed, profiles/r5_ab_coop_tree.txt)
};
static_assert(sizeof(MeshDev) == 64, "MeshDev is 64 bytes (the lean kernels copy the records into LDS)");

struct NodeDev {           // core/scene.hpp:11-64
  Xform xf;                // transform (fwd, inv)
  float bmin[3]; int32_t mesh;
  float bmax[3]; uint32_t skip;    // next node index when this subtree is culled
  int32_t parent; uint32_t depth; uint32_t pad[2];
};

// inner: a = left child (right = a + 1), b = 0; leaf: a = scene node, b = 1. Root = node 0.
struct TlasNode { float lo[3]; uint32_t a; float hi[3]; uint32_t b; };

struct LightDev {          // core/light.hpp
  uint32_t type; int32_t mesh; uint32_t tri; uint32_t twoSided;
  f3 emission; float area;         // AreaLight::m_area (of the transformed triangle)
  float power; float radius; int32_t texture; uint32_t envOffset;   // env table index
  Xform xf;
};

struct EnvDev {            // ImageInfiniteLight + PiecewiseConstant2D (light.cpp:137-197, sampling.hpp:118-196)
  uint32_t w, h;           // distribution resolution (= texture resolution for full bounds)
  uint32_t funcOffset;     // w*h floats   (|f|)
  uint32_t cdfOffset;      // (w+1)*h floats (conditional CDFs)
  uint32_t rowIntOffset;   // h floats (conditional integrals = marginal function)
  uint32_t margCdfOffset;  // h+1 floats
  float margIntegral;
  float surfaceArea;       // light.cpp:192-196
  // guide tables for the CDF searches (lights.hpp::pc1dSample): SceneDev::envGuide + guideOffset holds
  // guideKh + 1 entries for the marginal CDF, then h rows of guideKw + 1 entries; 0 = none
  uint32_t guideOffset, guideKw, guideKh;
  // the same tables interleaved for the SAMPLING walk (lights.hpp::envSample), so that what one step of it reads lies in one
  // sector: at envData + pairOffset first h + 1 marginal records {margCdf[k], rowInt[k], cdf_k[1], 0} (the row integral and
  // the row's cdf[1] — the reference's normalisation typo needs it — arrive with the marginal search's own last read), then
  // h rows of w + 1 records {cdf[k], func[k]} (func[w] = 0). Values are copies: same floats, same arithmetic.
  uint32_t pairOffset;
};

struct CameraDev {         // core/camera.hpp:13-59
  f3 position; float apertureRadius;
  f3 topLeftPixel; uint32_t apertureSides;
  f3 pixelDeltaU; float exposureScale;   // exp2(exposure), integrator.cpp:23
  f3 pixelDeltaV; float pad0;
  f3 frameX; float pad1;
  f3 frameY; float pad2;
  f3 frameZ; float pad3;
};

struct LutDev {            // bsdf/luts.hpp:14-24 — float offsets into lutData
  // E[32][32], Eavg[32], baseE[16][16][16], baseEavg[16][16], glassE[16^3], glassEavg[16^2],
  // glassInvE[16^3], glassInvEavg[16^2], then 52 uint32 of the Sobol dim-1 matrix
  static constexpr uint32_t E = 0, Eavg = 1024, baseE = 1056, baseEavg = 5152, glassE = 5408,
                            glassEavg = 9504, glassInvE = 9760, glassInvEavg = 13856,
                            sobol = 14112, total = 14164;
  // Footprint copies of the tables above (host_scene.hpp::appendLutFootprints): record (i, j[, k]) holds the 2 / 4 / 8 values
  // one interpolated lookup reads, in the order the interpolation formula takes them, so that a lookup is one or two aligned
  // 16-byte loads instead of 2-8 scattered ones. Same floats, same arithmetic. Float offsets, 16-byte aligned.
  static constexpr uint32_t fpE = 14164,                       // [31][32] x 4: E[ri][ci], E[ri][ci+1], E[ri+1][ci], E[ri+1][ci+1]
                            fpEavg = fpE + 31 * 32 * 4,        // [32] x 2 (padded to 4): Eavg[ri], Eavg[ri+1]
                            fpBaseE = fpEavg + 32 * 4,         // [15][16][16] x 8: baseE[f0i+a][ri+b][ci+c] at a*4 + b*2 + c
                            fpBaseEavg = fpBaseE + 15 * 16 * 16 * 8,   // [15][16] x 4
                            fpGlassE = fpBaseEavg + 15 * 16 * 4,       // [15][16][16] x 8: glassE[f0i+a][ci+b][ri+c], record (f0i, ci, ri)
                            fpGlassInvE = fpGlassE + 15 * 16 * 16 * 8,
                            totalWithFootprints = fpGlassInvE + 15 * 16 * 16 * 8;
};

// Everything a kernel needs, passed by value as a kernel argument (pointers into HBM).
struct SceneDev {
  const ShadeTri* shadeTris;
  const BvhNode* bvhNodes;
  const LeafTri* leafTris;
  const u4* triVerts;          // i0, i1, i2 (mesh-local vertex ids), material
  const int32_t* triLight;
  const f4* vPos;              // xyz, pad
  const f4* vNormal;           // xyz, pad
  const f4* vTangent;
  const f2* vUV;
  const MeshDev* meshes;
  const NodeDev* nodes;
  const MaterialDev* materials;
  const TexDev* textures;
  const uint8_t* texU8;
  const float* texF32;
  const uint8_t* texQuads;     // 2x2 footprint records of every texture (TexDev::quadOffset), nullptr on the host (tests/hostsim)
  const LightDev* lights;
  const EnvDev* envs;
  const float* envData;
  const uint32_t* envGuide;
  const f4* nodeWorld;       // 2 per scene node: padded WORLD-space AABB of the node's subtree (min, max)
  // scenes of many nodes (trace_lean_tlas.hpp): a spatial hierarchy over the mesh nodes' nodeWorld boxes — a filter only, the
  // exact tests still run in the reference's pre-order — and the lanes' node bitsets (scratch, set per launch): nodeBitWords
  // 64-bit words per lane, word w of thread t at nodeBits[w * threads + t]
  const struct TlasNode* tlas;
  unsigned long long* nodeBits;
  uint32_t nTlas, nodeBitWords;
  const uint32_t* infiniteLights;   // indices into lights
  const uint32_t* areaLights;       // indices into lights
  const float* areaPowerCdf;        // m_lightPowers, light-sampler.cpp:43-47
  const float* lut;                 // LutDev layout
  uint32_t nNodes, nLights, nInfinite, nArea;
  float totalPower;
  uint32_t nMeshes;
  uint32_t nMaterials, nTextures, nEnvs;   // sizes of the (padded) record arrays: what a kernel copies when it keeps them in LDS
};

struct Hit {                   // cpu/hit.hpp:8-17 after testNode returned
  float t;
  f2 uv;
  f3 p, n, tg;
  uint32_t material;
  int32_t lightIdx;
  bool backSide;
};

}  // namespace yart_hip
