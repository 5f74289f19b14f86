// bsdf.hpp — texture fetches, energy-compensation LUT lookups, GGX and the
// parametric PBR material (metallic / dielectric / glossy-diffuse / clearcoat).
//
// Restates, operation for operation, reference core/texture.hpp:105-161 +
// texture.cpp:21-35, bsdf/luts.hpp:33-191, core/bsdf.hpp:16-291, core/bsdf.cpp:5-58
// and bsdf/parametric.cpp:84-838. Quirks kept on purpose are listed in
// SURVEY.md Appendix A (items 6, 14, 17, 18, 20, 21).
#pragma once
#include "scene_types.hpp"

namespace yart_hip {

// ---------------------------------------------------------------------------
// Textures — repeat wrap, 4-tap bilinear, u8/255, gamma-2 decode for sRGB typed
// ---------------------------------------------------------------------------
struct TexTaps { uint32_t i00, i01, i10, i11; float u, v; };

// Instrumented build (libyart_hip_count.so) only: bytes of texel data the lookups of one render need —
// 4 taps x channels x (1 B | 4 B) per lookup, SURVEY §8(d)'s "4·C·taps" term of B_shade — summed per wave.
#if defined(YART_COUNT_TRAVERSAL) && defined(__HIPCC__)
static __device__ unsigned long long g_texTapBytes;     // (one per translation unit: yart_hip.hip sums its units')
#endif
#if defined(YART_COUNT_TRAVERSAL) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void texTally(const TexDev& t) {
  const uint32_t bytes = 4u * t.channels * (t.isFloat ? 4u : 1u);
  // lanes of one call may read different textures: one round per distinct byte count among the active lanes
  unsigned long long rem = __ballot(true), sum = 0;
  const int first = __ffsll((long long) rem) - 1;
  while (rem) {
    const uint32_t b = uint32_t(__shfl(int(bytes), __ffsll((long long) rem) - 1));
    const unsigned long long same = __ballot(bytes == b) & rem;
    sum += uint64_t(b) * uint64_t(__popcll(same));
    rem &= ~same;
  }
  if (int(threadIdx.x & 63u) == first) atomicAdd(&g_texTapBytes, sum);
}
#else
YART_HD void texTally(const TexDev&) {}
#endif

YART_HD TexTaps texTaps(const TexDev& t, f2 uv) {              // texture.cpp:21-35
  uv.x -= floorf(uv.x);
  uv.y -= floorf(uv.y);
  uv.x *= float(t.width - 1);
  uv.y *= float(t.height - 1);
  uint32_t wm = t.width - 2, hm = t.height - 2;
  uint32_t x = (float(wm) < uv.x) ? wm : uint32_t(uv.x);      // math::min<uint32_t,float>
  uint32_t y = (float(hm) < uv.y) ? hm : uint32_t(uv.y);
  uv.x -= float(x);
  uv.y -= float(y);
  TexTaps r;
  r.i00 = y * t.width + x;            // samples[0]
  r.i01 = (y + 1) * t.width + x;      // samples[1]
  r.i10 = y * t.width + (x + 1);      // samples[2]
  r.i11 = (y + 1) * t.width + (x + 1);
  r.u = uv.x; r.v = uv.y;
  return r;
}
YART_HD float bilerp1(float a0, float a1, float b0, float b1, float u, float v) {   // math_base.hpp:46-59
  return ((a0 * (1.0f - u) * (1.0f - v) + a1 * (1.0f - u) * v) + b0 * u * (1.0f - v)) + b1 * u * v;
}
// One texel of a u8 texture as a word (channel c in byte c): one aligned load for 1/2/4-channel
// texels (texture offsets are multiples of 4), two for 3-channel ones — instead of one byte
// load per channel and tap (a divergent load costs the same address-unit time whatever its width).
YART_HD uint32_t texelWord(const SceneDev& sc, const TexDev& t, uint32_t idx) {
  const uint32_t b = t.offset + t.channels * idx;
  if (t.channels == 4) return *reinterpret_cast<const uint32_t*>(sc.texU8 + b);
  if (t.channels == 2) return *reinterpret_cast<const uint16_t*>(sc.texU8 + b);
  if (t.channels == 1) return sc.texU8[b];
  const uint32_t* w = reinterpret_cast<const uint32_t*>(sc.texU8 + (b & ~3u));
  const uint64_t v = uint64_t(w[0]) | (uint64_t(w[1]) << 32);
  return uint32_t(v >> (8u * (b & 3u)));
}
// byte / 255.0f (texture.hpp:105-110) without the 11-instruction IEEE divide: for the 256 possible
// numerators q = b * RN(1/255) followed by one exact-residual correction IS the correctly rounded
// quotient (all 256 checked by hostsim selftest and tests/test_hostsim.py), 3 VALU instructions
// per channel and tap instead of 11.
YART_HD float byteToUnit(uint32_t b) {
  const float x = float(b), r = 1.0f / 255.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}
YART_HD float texelChannel(const TexDev& t, uint32_t word, uint32_t c) {
  float v = byteToUnit((word >> (8u * c)) & 0xffu);
  if (t.type == TEX_SRGB && c < 3) v = v * v;                  // texture.hpp:111-113
  return v;
}
struct TexQuad { uint32_t w00, w01, w10, w11; };               // the four taps of a u8 texture
YART_HD TexQuad texQuad(const SceneDev& sc, const TexDev& t, const TexTaps& k) {
  TexQuad q;
  if (sc.texQuads != nullptr) {
    // one footprint record = the same four texels (texture.cpp:21-35 picks them; k.i00 = y * width + x names the record)
    const uint8_t* base = sc.texQuads + size_t(t.quadOffset) * 16u;
    const uint32_t stride = t.quadStride ? t.quadStride : texQuadRecordBytes(t.channels, 0u);
    if (t.channels >= 3) {
      const u4 v = *reinterpret_cast<const u4*>(base + size_t(k.i00) * stride);
      q.w00 = v.x; q.w01 = v.y; q.w10 = v.z; q.w11 = v.w;
    } else if (t.channels == 2) {
      const uint32_t* p = reinterpret_cast<const uint32_t*>(base + size_t(k.i00) * stride);
      const uint32_t a = p[0], b = p[1];
      q.w00 = a & 0xffffu; q.w01 = a >> 16; q.w10 = b & 0xffffu; q.w11 = b >> 16;
    } else {
      const uint32_t a = *reinterpret_cast<const uint32_t*>(base + size_t(k.i00) * stride);
      q.w00 = a & 0xffu; q.w01 = (a >> 8) & 0xffu; q.w10 = (a >> 16) & 0xffu; q.w11 = a >> 24;
    }
    return q;
  }
  q.w00 = texelWord(sc, t, k.i00); q.w01 = texelWord(sc, t, k.i01);
  q.w10 = texelWord(sc, t, k.i10); q.w11 = texelWord(sc, t, k.i11);
  return q;
}
YART_HD float texelF(const SceneDev& sc, const TexDev& t, uint32_t idx, uint32_t c) {
  return sc.texF32[t.offset + t.channels * idx + c];
}
YART_HD float texSampleChannel(const SceneDev& sc, const TexDev& t, const TexTaps& k, const TexQuad& q, uint32_t c) {
  if (t.isFloat)
    return bilerp1(texelF(sc, t, k.i00, c), texelF(sc, t, k.i01, c), texelF(sc, t, k.i10, c), texelF(sc, t, k.i11, c),
                   k.u, k.v);
  return bilerp1(texelChannel(t, q.w00, c), texelChannel(t, q.w01, c), texelChannel(t, q.w10, c),
                 texelChannel(t, q.w11, c), k.u, k.v);
}
YART_HD TexQuad texQuadOrZero(const SceneDev& sc, const TexDev& t, const TexTaps& k) {
  if (t.isFloat) { TexQuad q; q.w00 = q.w01 = q.w10 = q.w11 = 0; return q; }
  return texQuad(sc, t, k);
}
YART_HD f3 texSample3(const SceneDev& sc, int32_t tex, f2 uv) {
  const TexDev t = sc.textures[tex];
  texTally(t);
  const TexTaps k = texTaps(t, uv);
  if (t.isFloat && t.channels == 3 && sc.texQuads != nullptr) {
    // float RGB (environment maps): the four taps' twelve floats are one 64-byte record (tap-major, then channel)
    const f4* r = reinterpret_cast<const f4*>(sc.texQuads + size_t(t.quadOffset) * 16u + size_t(k.i00) * 64u);
    const f4 a = r[0], b = r[1], c = r[2];           // a = {t00.rgb, t01.r}  b = {t01.gb, t10.rg}  c = {t10.b, t11.rgb}
    return mk3(bilerp1(a.x, a.w, b.z, c.y, k.u, k.v), bilerp1(a.y, b.x, b.w, c.z, k.u, k.v), bilerp1(a.z, b.y, c.x, c.w, k.u, k.v));
  }
  const TexQuad q = texQuadOrZero(sc, t, k);
  return mk3(texSampleChannel(sc, t, k, q, 0), texSampleChannel(sc, t, k, q, 1), texSampleChannel(sc, t, k, q, 2));
}
YART_HD f4 texSample4(const SceneDev& sc, int32_t tex, f2 uv) {
  const TexDev t = sc.textures[tex];
  texTally(t);
  const TexTaps k = texTaps(t, uv);
  const TexQuad q = texQuadOrZero(sc, t, k);
  f4 r;
  r.x = texSampleChannel(sc, t, k, q, 0); r.y = texSampleChannel(sc, t, k, q, 1);
  r.z = texSampleChannel(sc, t, k, q, 2); r.w = texSampleChannel(sc, t, k, q, 3);
  return r;
}
YART_HD f2 texSample2(const SceneDev& sc, int32_t tex, f2 uv) {
  const TexDev t = sc.textures[tex];
  texTally(t);
  const TexTaps k = texTaps(t, uv);
  const TexQuad q = texQuadOrZero(sc, t, k);
  return mk2(texSampleChannel(sc, t, k, q, 0), texSampleChannel(sc, t, k, q, 1));
}
YART_HD float texSample1(const SceneDev& sc, int32_t tex, f2 uv) {
  const TexDev t = sc.textures[tex];
  texTally(t);
  const TexTaps k = texTaps(t, uv);
  const TexQuad q = texQuadOrZero(sc, t, k);
  return texSampleChannel(sc, t, k, q, 0);
}

// ---------------------------------------------------------------------------
// LUT lookups (bsdf/luts.hpp)
// ---------------------------------------------------------------------------
YART_HD float trilerp8(const float* x, float u, float v, float w) {    // math_base.hpp:61-79
  float up = 1.0f - u, vp = 1.0f - v, wp = 1.0f - w;
  return ((((((x[0] * up * vp * wp + x[1] * up * vp * w) + x[2] * up * v * wp) + x[3] * up * v * w) +
            x[4] * u * vp * wp) + x[5] * u * vp * w) + x[6] * u * v * wp) + x[7] * u * v * w;
}
// The interpolated lookups read the footprint copies of the tables (LutDev::fp*: the values of one lookup side by side):
// the same floats in the same formula, one or two 16-byte loads instead of 2-8 scattered ones.
// A kernel may hand these functions a copy of the PLAIN tables of the glossy lobes (E, Eavg, baseE, baseEavg: the first
// LutDev::glassE floats) in LDS instead (k_wf_shade): a lookup is then 2-8 four-byte LDS reads — a fifth of the latency of an L2
// hit, and these lookups sit in the chain of dependent reads that kernel waits on. The pointer says which it is; behind the LDS
// copy lies the address of the tables in memory, for the glass lobes' lookups.
YART_HD bool lutInLds(const float* lut) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_is_shared(lut);
#else
  (void)lut; return false;
#endif
}
YART_HD float ggxE(const float* lut, float cosTheta, float r) {        // luts.hpp:33-44
  float ro = r * 31.0f, co = cosTheta * 31.0f;
  uint32_t ri = sizeTClamp(ro, 30), ci = sizeTClamp(co, 30);
  ro -= float(ri); co -= float(ci);
  if (lutInLds(lut)) {
    const float* E = lut + LutDev::E + ri * 32 + ci;
    return bilerp1(E[0], E[1], E[32], E[33], ro, co);
  }
  const f4 e = *reinterpret_cast<const f4*>(lut + LutDev::fpE + (ri * 32 + ci) * 4);   // d00, d01, d10, d11
  return bilerp1(e.x, e.y, e.z, e.w, ro, co);
}
YART_HD float ggxEavg(const float* lut, float r) {                     // luts.hpp:52-57
  uint32_t ri = sizeTClamp(r * 31.0f, 30);
  float ro = r * 31.0f - float(ri);
  if (lutInLds(lut)) return lerpf(lut[LutDev::Eavg + ri], lut[LutDev::Eavg + ri + 1], ro);
  const f2 t = *reinterpret_cast<const f2*>(lut + LutDev::fpEavg + ri * 4);
  return lerpf(t.x, t.y, ro);
}
YART_HD float ggxBaseE(const float* lut, float f0, float r, float cosTheta) {   // luts.hpp:68-97
  float f0o = f0 * 15.0f, ro = r * 15.0f, co = cosTheta * 15.0f;
  uint32_t f0i = sizeTClamp(f0o, 14), ri = sizeTClamp(ro, 14), ci = sizeTClamp(co, 14);
  f0o -= float(f0i); ro -= float(ri); co -= float(ci);
  if (lutInLds(lut)) {
    const float* B = lut + LutDev::baseE + (f0i * 16 + ri) * 16 + ci;      // [a][b][c] at a * 256 + b * 16 + c
    const float vals[8] = {B[0], B[1], B[16], B[17], B[256], B[257], B[272], B[273]};
    return trilerp8(vals, f0o, ro, co);
  }
  const f4* T = reinterpret_cast<const f4*>(lut + LutDev::fpBaseE + ((f0i * 16 + ri) * 16 + ci) * 8);
  const f4 lo = T[0], hi = T[1];
  const float vals[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return trilerp8(vals, f0o, ro, co);
}
YART_HD float ggxBaseEavg(const float* lut, float f0, float r) {       // luts.hpp:106-116
  uint32_t f0i = sizeTClamp(f0 * 15.0f, 14), ri = sizeTClamp(r * 15.0f, 14);
  float f0o = f0 * 15.0f - float(f0i), ro = r * 15.0f - float(ri);
  if (lutInLds(lut)) {
    const float* A = lut + LutDev::baseEavg + f0i * 16 + ri;
    return bilerp1(A[0], A[1], A[16], A[17], f0o, ro);
  }
  const f4 t = *reinterpret_cast<const f4*>(lut + LutDev::fpBaseEavg + (f0i * 16 + ri) * 4);
  return bilerp1(t.x, t.y, t.z, t.w, f0o, ro);
}
YART_HD float ggxGlassE(const float* lut, float ior, float r, float cosTheta) {   // luts.hpp:126-158
  if (lutInLds(lut)) {                                          // the glass tables are not in the LDS copy: their address lies behind it
    uint64_t p; const uint32_t* w = reinterpret_cast<const uint32_t*>(lut + LutDev::glassE);
    p = uint64_t(w[0]) | (uint64_t(w[1]) << 32);
    lut = reinterpret_cast<const float*>(p);
  }
  bool inv = ior < 1.0f;
  if (inv) ior = 1.0f / ior;
  float f0 = sqrtf(fabsf((1.0f - ior) / (1.0f + ior)));
  uint32_t f0i = sizeTClamp(f0 * 15.0f, 14), ri = sizeTClamp(r * 15.0f, 14),
           ci = sizeTClamp(cosTheta * 15.0f, 14);
  float f0o = f0 * 15.0f - float(f0i), ro = r * 15.0f - float(ri), co = cosTheta * 15.0f - float(ci);
  const f4* T = reinterpret_cast<const f4*>(lut + (inv ? LutDev::fpGlassInvE : LutDev::fpGlassE) + ((f0i * 16 + ci) * 16 + ri) * 8);
  const f4 lo = T[0], hi = T[1];
  const float vals[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return trilerp8(vals, f0o, co, ro);
}

// ---------------------------------------------------------------------------
// GGX microfacet distribution (core/bsdf.hpp:175-291)
// ---------------------------------------------------------------------------
struct GGX { float ax, ay, r; };

YART_HD GGX makeGGX(float roughness) {                                  // bsdf.hpp:177-179
  GGX g; g.r = roughness; g.ax = g.ay = roughness * roughness; return g;
}
YART_HD GGX makeGGX(float roughness, float anisotropic) {               // bsdf.hpp:181-187
  GGX g; g.r = roughness;
  float alpha = roughness * roughness;
  float aspect = sqrtf(1.0f - 0.9f * anisotropic);
  g.ax = alpha / aspect;
  g.ay = alpha * aspect;
  return g;
}
YART_HD bool ggxSmooth(const GGX& g) { return g.ax < 1e-3f && g.ay < 1e-3f; }   // :239-241
YART_HD float ggxMdf(const GGX& g, f3 w) {                              // bsdf.hpp:194-217
  const float cos2Theta = w.z * w.z;
  const float sin2Theta = stdmax(0.0f, 1.0f - cos2Theta);
  const float tan2Theta = sin2Theta / cos2Theta;
  const float cos4Theta = cos2Theta * cos2Theta;
  float k = tan2Theta;
  if (g.ax != g.ay) {
    float cos2Phi = sin2Theta == 0.0f ? 1.0f : w.x * w.x / sin2Theta;
    float sin2Phi = sin2Theta == 0.0f ? 1.0f : w.y * w.y / sin2Theta;
    k *= (cos2Phi / (g.ax * g.ax) + sin2Phi / (g.ay * g.ay));
  } else {
    k /= (g.ax * g.ax);
  }
  const float k2 = (1.0f + k) * (1.0f + k);
  return 1.0f / (kPi * g.ax * g.ay * cos4Theta * k2);
}
YART_HD float ggxLambda(const GGX& g, f3 w) {                           // bsdf.hpp:276-290
  const float cos2Theta = w.z * w.z;
  const float sin2Theta = 1.0f - cos2Theta;
  const float tan2Theta = sin2Theta / cos2Theta;
  float alpha2 = g.ax * g.ax;
  if (g.ax != g.ay) {
    float cos2Phi = sin2Theta == 0.0f ? 1.0f : w.x * w.x / sin2Theta;
    float sin2Phi = sin2Theta == 0.0f ? 0.0f : w.y * w.y / sin2Theta;
    alpha2 = alpha2 * cos2Phi + g.ay * g.ay * sin2Phi;
  }
  return (sqrtf(1.0f + alpha2 * tan2Theta) - 1.0f) * 0.5f;
}
YART_HD float ggxG1(const GGX& g, f3 w) { return 1.0f / (1.0f + ggxLambda(g, w)); }
YART_HD float ggxG(const GGX& g, f3 wo, f3 wi) { return 1.0f / (1.0f + ggxLambda(g, wo) + ggxLambda(g, wi)); }
YART_HD float ggxVmdf(const GGX& g, f3 w, f3 wm) {                      // bsdf.hpp:232-237
  return ggxG1(g, w) / fabsf(w.z) * ggxMdf(g, wm) * absDot(w, wm);
}
YART_HD f2 sampleDiskUniform(f2 u) {                                    // math/sampling.hpp:40-45
  const float r = sqrtf(u.x);
  const float theta = 2.0f * kPi * u.y;                       // u.y in [0, 1): theta in [0, 2 pi]
  return mk2(r * ycosf2pi(theta), r * ysinf2pi(theta));
}
YART_HD f3 ggxSampleVisibleMicrofacet(const GGX& g, f3 w, f2 u) {       // bsdf.hpp:243-271
  f3 wh = normalized(mk3(g.ax * w.x, g.ay * w.y, w.z));
  if (wh.z < 0) wh = wh * -1.0f;
  const f3 b = (wh.z < 0.9999f) ? normalized(cross(mk3(0, 0, 1), wh)) : mk3(1, 0, 0);
  const f3 t = cross(wh, b);
  f2 p = sampleDiskUniform(u);
  const float h = sqrtf(1.0f - p.x * p.x);
  p.y = lerpf(h, p.y, 0.5f * wh.z + 0.5f);
  const float pz = sqrtf(stdmax(0.0f, 1.0f - ((0.0f + p.x * p.x) + p.y * p.y)));
  f3 nh = (p.x * b + p.y * t) + pz * wh;
  return normalized(mk3(g.ax * nh.x, g.ay * nh.y, stdmax(1e-6f, nh.z)));
}

YART_HD float roughen(float roughness) {                                // bsdf.hpp:16-18
  return ymax(roughness, stdclamp(roughness * 2.0f, 0.1f, 0.3f));
}
YART_HD float FavgFit(float ior) { return (ior - 1.0f) / (4.08567f + 1.00071f * ior); }   // parametric.cpp:7-9

// ---------------------------------------------------------------------------
// BSDFSample (core/bsdf.hpp:20-41)
// ---------------------------------------------------------------------------
enum : int {
  SC_ABSORBED = 0, SC_EMITTED = 1, SC_REFLECTED = 2, SC_TRANSMITTED = 4, SC_DIFFUSE = 8,
  SC_GLOSSY = 16, SC_SPECULAR = 32
};
struct BsdfSample {
  int scatter;
  f3 f, Le, wi;
  float pdf, roughness;
};
YART_HD BsdfSample mkSample(int sc, f3 f, f3 Le, f3 wi, float pdf, float rough) {
  BsdfSample s; s.scatter = sc; s.f = f; s.Le = Le; s.wi = wi; s.pdf = pdf; s.roughness = rough;
  return s;
}
YART_HD BsdfSample absorbed() { return mkSample(SC_ABSORBED, mk3(0), mk3(0), mk3(0), 0.0f, 0.0f); }

// Per-hit material parameters after the texture fetches every entry point starts with
// (parametric.cpp:89-104, 140-154, 187-203).
struct MatEval {
  f3 base;
  float r, m, t, c, cr;
};
YART_HD void matFetchScalars(const SceneDev& sc, const MaterialDev& mt, f2 uv, MatEval& e) {
  e.r = mt.roughness; e.m = mt.cMetallic; e.t = mt.cTrans;
  e.c = mt.clearcoat; e.cr = mt.clearcoatRoughness;
  if (mt.texMR >= 0) {
    f2 mr = texSample2(sc, mt.texMR, uv);
    e.r *= mr.x; e.m *= mr.y;
  }
  if (mt.texTransmission >= 0) e.t *= texSample1(sc, mt.texTransmission, uv);
  if (mt.texClearcoat >= 0) {
    float s = texSample1(sc, mt.texClearcoat, uv);      // float2(mono sample): both lanes equal
    e.c *= s; e.cr *= s;
  }
}
YART_HD f3 matBase(const SceneDev& sc, const MaterialDev& mt, f2 uv) {   // parametric.cpp:75-78
  if (mt.texBase >= 0) {
    f4 s = texSample4(sc, mt.texBase, uv);
    return mt.base * mk3(s.x, s.y, s.z);
  }
  return mt.base;
}
YART_HD float matAlpha(const SceneDev& sc, const MaterialDev& mt, f2 uv) {   // parametric.cpp:69-73
  if ((mt.flags & MAT_HAS_ALPHA) && mt.texBase >= 0) return texSample4(sc, mt.texBase, uv).w;
  return 1.0f;
}
YART_HD f3 matAttenuation(const MaterialDev& mt, float d) {             // parametric.cpp:834-838
  if (mt.flags & MAT_THIN) return mk3(1.0f);
  f3 e = ((mt.volumeColor - 1.0f) * d) * mt.volumeDensity;
  return mk3(yexpf(e.x), yexpf(e.y), yexpf(e.z));
}

// ---- metallic lobe (parametric.cpp:260-352) ----
YART_HD f3 fMetallic(const float* lut, f3 wo, f3 wi, f3 base, const GGX& mf) {
  if (ggxSmooth(mf)) return mk3(0);
  const float cosTheta_o = fabsf(wo.z), cosTheta_i = fabsf(wi.z);
  if (cosTheta_i == 0 || cosTheta_o == 0) return mk3(0);
  f3 wm = wo + wi;
  if (length2(wm) == 0.0f) return mk3(0);
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  const f3 Fss = fresnelSchlick(base, absDot(wo, wm));
  const f3 Mss = Fss * ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
  const float Ess = ggxE(lut, cosTheta_o, mf.r);
  const f3 Mms = Mss * base * (1.0f - Ess) / Ess;
  return Mss + Mms;
}
YART_HD float pdfMetallic(f3 wo, f3 wi, const GGX& mf) {
  if (ggxSmooth(mf)) return 0;
  f3 wm = wo + wi;
  if (length2(wm) == 0.0f) return 0;
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  return ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm));
}
YART_HD BsdfSample sampleMetallic(const float* lut, const MaterialDev& mt, f3 wo, f3 base,
                                  const GGX& mf, f2 u) {
  if (ggxSmooth(mf)) {
    const f3 F = fresnelSchlick(base, wo.z);
    return mkSample(SC_REFLECTED | SC_SPECULAR, F / fabsf(wo.z), mk3(0), mk3(-wo.x, -wo.y, wo.z), 1.0f, 0.0f);
  }
  f3 wm = ggxSampleVisibleMicrofacet(mf, wo, u);
  f3 wi = reflect(wo, wm);
  if (wo.z * wi.z < 0.0f) return absorbed();
  const float pdf = ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm));
  const float cosTheta_o = fabsf(wo.z), cosTheta_i = fabsf(wi.z);
  const f3 Fss = fresnelSchlick(base, absDot(wo, wm));
  const f3 Mss = Fss * ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
  const float Ess = ggxE(lut, cosTheta_o, mf.r);
  const f3 Mms = Mss * base * (1.0f - Ess) / Ess;
  return mkSample(SC_REFLECTED | SC_GLOSSY, Mss + Mms, mk3(0), wi, pdf, mt.roughness);   // :350 untextured
}

// ---- dielectric lobe (parametric.cpp:354-575) ----
YART_HD f3 fDielectric(const float* lut, const MaterialDev& mt, f3 wo, f3 wi, f3 base, const GGX& mf) {
  if (ggxSmooth(mf)) return mk3(0);
  const float cosTheta_o = wo.z, cosTheta_i = wi.z;
  const bool isReflection = cosTheta_o * cosTheta_i > 0.0f;
  float ior = 1.0f;
  if (!isReflection) ior = cosTheta_o > 0.0f ? mt.ior : 1.0f / mt.ior;
  f3 wm = ior * wi + wo;
  if (cosTheta_i == 0.0f || cosTheta_o == 0.0f || length2(wm) == 0.0f) return mk3(0);
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  if (dot(wm, wi) * cosTheta_i < 0.0f || dot(wm, wo) * cosTheta_o < 0.0f) return mk3(0);
  const float Fss = fresnelDielectric(absDot(wo, wm), ior);
  const float T = 1.0f - Fss;
  const float E_o = ggxGlassE(lut, ior, mf.r, fabsf(cosTheta_o));
  if (isReflection) {
    const float Mss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
    return mk3(Fss * Mss / E_o);
  } else if (mt.flags & MAT_THIN) {
    f3 wip = reflect(-wi, mk3(0, 0, 1));
    wm = normalized(wip + wo);
    const float cosTheta_ip = fabsf(wip.z);
    const float Tss = ggxMdf(mf, wm) * ggxG(mf, wo, wip) / (4 * cosTheta_o * cosTheta_ip);
    return T * base * Tss / E_o;
  } else {
    const float temp = dot(wi, wm) * ior + dot(wo, wm);
    const float dwm_dwi = absDot(wi, wm) * absDot(wo, wm) / (temp * temp);
    const float Tss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) * dwm_dwi / (fabsf(cosTheta_i * cosTheta_o));
    return T * base * Tss / E_o;
  }
}
YART_HD float pdfDielectric(const MaterialDev& mt, f3 wo, f3 wi, const GGX& mf) {
  if (ggxSmooth(mf)) return 0;
  const float cosTheta_o = wo.z, cosTheta_i = wi.z;
  const bool isReflection = cosTheta_o * cosTheta_i > 0.0f;
  float ior = 1.0f;
  if (!isReflection) ior = cosTheta_o > 0.0f ? mt.ior : 1.0f / mt.ior;
  f3 wm = ior * wi + wo;
  if (cosTheta_i == 0.0f || cosTheta_o == 0.0f || length2(wm) == 0.0f) return 0;
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  if (dot(wm, wi) * cosTheta_i < 0.0f || dot(wm, wo) * cosTheta_o < 0.0f) return 0;
  const float F = fresnelDielectric(dot(wo, wm), mt.ior);
  const float T = 1.0f - F;
  float pdf;
  if (isReflection) {
    pdf = ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm)) * F;
  } else if (mt.flags & MAT_THIN) {
    f3 wip = reflect(-wi, mk3(0, 0, 1));
    wm = normalized(wip + wo);
    pdf = ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm)) * T;
  } else {
    const float temp = dot(wi, wm) + dot(wo, wm) / ior;
    const float dwm_dwi = absDot(wo, wm) / (temp * temp);
    pdf = ggxVmdf(mf, wo, wm) * dwm_dwi * T;
  }
  return pdf;
}
YART_HD BsdfSample sampleDielectric(const float* lut, const MaterialDev& mt, f3 wo, f3 base,
                                    const GGX& mf, f2 u, float uc) {
  const bool thin = mt.flags & MAT_THIN;
  const float ior = (thin || wo.z > 0.0f) ? mt.ior : 1.0f / mt.ior;
  if (ggxSmooth(mf)) {
    float F = fresnelDielectric(fabsf(wo.z), ior);
    float T = 1.0f - F;
    if (uc < F) {
      f3 wi = mk3(-wo.x, -wo.y, wo.z);
      return mkSample(SC_REFLECTED | SC_SPECULAR, mk3(F / fabsf(wi.z)), mk3(0), wi, F, 0.0f);
    } else {
      f3 wi = mk3(0);
      if (thin) wi = -wo;
      else if (!refract(wo, mk3(0, 0, 1), mt.ior, wi)) return absorbed();
      return mkSample(SC_TRANSMITTED | SC_SPECULAR, T * base / fabsf(wi.z), mk3(0), wi, T, 0.0f);
    }
  }
  f3 wm = ggxSampleVisibleMicrofacet(mf, wo, u);
  const float Fss = fresnelDielectric(absDot(wo, wm), ior);
  const float cosTheta_o = fabsf(wo.z);
  const float E_o = ggxGlassE(lut, ior, mf.r, cosTheta_o);
  if (uc < Fss) {
    const f3 wi = reflect(wo, wm);
    if (wo.z * wi.z < 0.0f) return absorbed();
    const float cosTheta_i = fabsf(wi.z);
    const float Mss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
    const float pdf = ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm)) * Fss;
    return mkSample(SC_REFLECTED | SC_GLOSSY, mk3(Fss * Mss / E_o), mk3(0), wi, pdf, mf.r);
  } else if (thin) {
    const f3 wi = reflect(wo, wm) * mk3(1, 1, -1);
    const float cosTheta_i = fabsf(wi.z);
    const float Tss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
    const float pdf = ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm)) * (1.0f - Fss);
    return mkSample(SC_TRANSMITTED | SC_GLOSSY, (1.0f - Fss) * Tss * base / E_o, mk3(0), wi, pdf, mf.r);
  } else {
    f3 wi = mk3(0);
    const bool tir = !refract(wo, wm, mt.ior, wi);
    if (tir || wo.z * wi.z > 0.0f || wi.z == 0.0f) return absorbed();
    const float temp = dot(wi, wm) * ior + dot(wo, wm);
    const float dwm_dwi = absDot(wi, wm) / (temp * temp);
    const float pdf = ggxVmdf(mf, wo, wm) * dwm_dwi * (1.0f - Fss);
    const float Tss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) *
                      fabsf(dot(wi, wm) * dot(wo, wm) / (wi.z * wo.z * temp * temp));
    return mkSample(SC_TRANSMITTED | SC_GLOSSY, (1.0f - Fss) * Tss * base / E_o, mk3(0), wi, pdf, mf.r);
  }
}

// ---- glossy-diffuse lobe (parametric.cpp:577-725) ----
YART_HD f3 fGlossy(const float* lut, const MaterialDev& mt, f3 wo, f3 wi, f3 base, const GGX& mf) {
  if (ggxSmooth(mf)) return mk3(0);
  const float cosTheta_o = fabsf(wo.z), cosTheta_i = fabsf(wi.z);
  if (cosTheta_i == 0 || cosTheta_o == 0) return mk3(0);
  f3 wm = wo + wi;
  if (length2(wm) == 0.0f) return mk3(0);
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  const float Fss = fresnelDielectric(dot(wo, wm), mt.ior);
  const float Mss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
  const float Favg = FavgFit(mt.ior);
  const float Eavg = ggxEavg(lut, mf.r);
  const float Mms = (1.0f - ggxE(lut, cosTheta_o, mf.r)) * (1.0f - ggxE(lut, cosTheta_i, mf.r)) /
                    (kPi * (1.0f - Eavg));
  const float Fms = Favg * Favg * Eavg / (1.0f - Favg * (1.0f - Eavg));
  const float r = (1.0f - mt.ior) / (1.0f + mt.ior);
  const float F0 = r * r;
  const float cDiffuse = (1.0f - ggxBaseE(lut, F0, mf.r, cosTheta_o)) *
                         (1.0f - ggxBaseE(lut, F0, mf.r, cosTheta_i)) /
                         (kPi * (1.0f - ggxBaseEavg(lut, F0, mf.r)));
  const f3 diffuse = base * cDiffuse;
  return mk3(Fss * Mss + Mms * Fms) + diffuse;
}
YART_HD float pdfGlossy(const float* lut, const MaterialDev& mt, f3 wo, f3 wi, const GGX& mf) {
  if (ggxSmooth(mf)) return 0;
  const float cosTheta_o = fabsf(wo.z), cosTheta_i = fabsf(wi.z);
  f3 wm = wo + wi;
  if (length2(wm) == 0.0f) return 0;
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  const float Fss = fresnelDielectric(dot(wo, wm), mt.ior);
  const float Favg = FavgFit(mt.ior);
  const float EmsAvg = ggxEavg(lut, mf.r);
  const float Fms = Favg * Favg * EmsAvg / (1.0f - Favg * (1.0f - EmsAvg));
  const float Ems_o = ggxE(lut, cosTheta_o, mf.r);
  const float kappa = 1.0f - (Favg * Ems_o + Fms * (1.0f - Ems_o));
  return (Fss + Fms) * ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm)) + cosTheta_i * kappa;
}
YART_HD f3 sampleCosineHemisphere(f2 u) {                               // math/sampling.hpp:30-38
  const float phi = u.x * 2.0f * kPi;
  const float sqrtr2 = sqrtf(u.y);
  const float x = ycosf2pi(phi) * sqrtr2;                     // phi = 2 pi u.x in [0, 2 pi]
  const float y = ysinf2pi(phi) * sqrtr2;
  const float z = sqrtf(1.0f - u.y);
  return mk3(x, y, z);
}
YART_HD BsdfSample sampleGlossy(const float* lut, const MaterialDev& mt, f3 wo, f3 base, f3 emission,
                                const GGX& mf, f2 u, float uc) {
  const float cosTheta_o = wo.z;                       // signed on purpose (Appendix A.6)
  const float Favg = FavgFit(mt.ior);
  const float Eavg = ggxEavg(lut, mf.r);
  const float Fms = Favg * Favg * Eavg / (1.0f - Favg * (1.0f - Eavg));
  const float E_o = ggxE(lut, cosTheta_o, mf.r);
  const float kappa = 1.0f - (Favg * E_o + Fms * (1.0f - E_o));
  if (uc < kappa) {
    f3 wi = sampleCosineHemisphere(u);
    if (wo.z < 0) wi = wi * -1.0f;
    const float cosTheta_i = wi.z;
    const float r = (1.0f - mt.ior) / (1.0f + mt.ior);
    const float F0 = r * r;
    const float cDiffuse = (1.0f - ggxBaseE(lut, F0, mf.r, cosTheta_o)) *
                           (1.0f - ggxBaseE(lut, F0, mf.r, cosTheta_i)) /
                           (kPi * (1.0f - ggxBaseEavg(lut, F0, mf.r)));
    return mkSample(SC_REFLECTED | SC_DIFFUSE | (length2(emission) > 0.0f ? SC_EMITTED : 0),
                    base * cDiffuse, emission, wi, fabsf(wi.z) * cDiffuse, 1.0f);
  }
  if (ggxSmooth(mf)) {
    const float F = fresnelDielectric(wo.z, mt.ior);
    f3 wi = mk3(-wo.x, -wo.y, wo.z);
    return mkSample(SC_REFLECTED | SC_SPECULAR, mk3(F / fabsf(wi.z)), mk3(0), wi, F, 0.0f);
  }
  f3 wm = ggxSampleVisibleMicrofacet(mf, wo, u);
  const f3 wi = reflect(wo, wm);
  const float cosTheta_i = wi.z;
  if (wo.z * wi.z < 0.0f) return absorbed();
  const float Fss = fresnelDielectric(dot(wo, wm), mt.ior);
  const float Mss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
  const float Mms = (1.0f - E_o) * (1.0f - ggxE(lut, cosTheta_i, mf.r)) / (kPi * (1.0f - Eavg));
  const float pdf = ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm)) * Fss;
  return mkSample(SC_REFLECTED | SC_GLOSSY, mk3(Fss * Mss + Fms * Mms), mk3(0), wi, pdf, mf.r);
}

// ---- clearcoat lobe (parametric.cpp:727-832) ----
YART_HD f3 fClearcoat(f3 wo, f3 wi, const GGX& mf, float* Fc) {
  if (ggxSmooth(mf)) return mk3(0);
  const float cosTheta_o = fabsf(wo.z), cosTheta_i = fabsf(wi.z);
  if (cosTheta_i == 0 || cosTheta_o == 0) return mk3(0);
  f3 wm = wo + wi;
  if (length2(wm) == 0.0f) return mk3(0);
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  const float Fss = fresnelDielectric(dot(wo, wm), 1.5f);
  const float Mss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
  *Fc = ymax(fresnelDielectric(cosTheta_o, 1.5f), fresnelDielectric(cosTheta_i, 1.5f));
  return mk3(Fss * Mss);
}
YART_HD float pdfClearcoat(f3 wo, f3 wi, const GGX& mf, float* Fc) {
  if (ggxSmooth(mf)) return 0;
  f3 wm = wo + wi;
  if (length2(wm) == 0.0f) return 0;
  wm = normalized(wm.z < 0.0f ? -wm : wm);
  const float Fss = fresnelDielectric(dot(wo, wm), 1.5f);
  *Fc = ymax(fresnelDielectric(fabsf(wo.z), 1.5f), fresnelDielectric(fabsf(wi.z), 1.5f));
  return Fss * ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm));
}
YART_HD BsdfSample sampleClearcoat(const MaterialDev& mt, f3 wo, const GGX& mf, f2 u) {
  const float cosTheta_o = wo.z;
  if (ggxSmooth(mf)) {
    const float F = fresnelDielectric(wo.z, mt.ior);            // :798 uses m_ior (Appendix A.20)
    f3 wi = mk3(-wo.x, -wo.y, wo.z);
    return mkSample(SC_REFLECTED | SC_SPECULAR, mk3(F / fabsf(wi.z)), mk3(0), wi, F, 0.0f);
  }
  f3 wm = ggxSampleVisibleMicrofacet(mf, wo, u);
  const f3 wi = reflect(wo, wm);
  const float cosTheta_i = wi.z;
  if (wo.z * wi.z < 0.0f) return absorbed();
  const float Fss = fresnelDielectric(dot(wo, wm), 1.5f);
  const float Mss = ggxMdf(mf, wm) * ggxG(mf, wo, wi) / (4 * cosTheta_o * cosTheta_i);
  const float pdf = ggxVmdf(mf, wo, wm) / (4 * absDot(wo, wm)) * Fss;
  return mkSample(SC_REFLECTED | SC_GLOSSY, mk3(Fss * Mss), mk3(0), wi, pdf, mt.clearcoatRoughness);
}

// ---- lobe mixture in the local frame (parametric.cpp:84-258) ----
// The *E variants take the material parameters already fetched (matEvaluate): the wavefront shade
// stage evaluates sample / f / pdf at the same hit and fetches the textures once.
YART_HD MatEval matEvaluate(const SceneDev& sc, const MaterialDev& mt, f2 uv) {
  MatEval e;
  e.base = matBase(sc, mt, uv);
  matFetchScalars(sc, mt, uv, e);
  return e;
}
YART_HD f3 bsdfFImplE(const SceneDev& sc, const MaterialDev& mt, const MatEval& e, f3 _wo, f3 _wi) {
  GGX mf = makeGGX(e.r, mt.anisotropic);
  const float cMetallic = e.m;
  const float cDielectric = (1.0f - e.m) * e.t;
  const float cGlossy = (1.0f - e.m) * (1.0f - e.t);
  f3 wo = mul3x3(mt.localRot, _wo), wi = mul3x3(mt.localRot, _wi);
  f3 val = mk3(0);
  if (cMetallic > 0.0f) val += cMetallic * fMetallic(sc.lut, wo, wi, e.base, mf);
  if (cDielectric > 0.0f) val += cDielectric * fDielectric(sc.lut, mt, wo, wi, e.base, mf);
  if (cGlossy > 0.0f) val += cGlossy * fGlossy(sc.lut, mt, wo, wi, e.base, mf);
  if (e.c > 0.0f) {
    GGX mfc = makeGGX(e.cr);
    float Fc = 0.0f;
    f3 valClear = fClearcoat(wo, wi, mfc, &Fc);
    val = (1.0f - e.c * Fc) * val + e.c * valClear;
  }
  return val;
}
YART_HD f3 bsdfFImpl(const SceneDev& sc, const MaterialDev& mt, f3 _wo, f3 _wi, f2 uv) {
  return bsdfFImplE(sc, mt, matEvaluate(sc, mt, uv), _wo, _wi);
}
YART_HD float bsdfPdfImplE(const SceneDev& sc, const MaterialDev& mt, const MatEval& e, f3 wo, f3 wi) {
  GGX mf = makeGGX(e.r, mt.anisotropic);
  const float pMetallic = e.m;
  const float pDielectric = (1.0f - e.m) * e.t;
  const float pGlossy = (1.0f - e.m) * (1.0f - e.t);
  float pdf = 0.0f;     // note: pdfImpl does NOT apply the anisotropy rotation (parametric.cpp:135-177)
  if (pMetallic > 0.0f) pdf += pMetallic * pdfMetallic(wo, wi, mf);
  if (pDielectric > 0.0f) pdf += pDielectric * pdfDielectric(mt, wo, wi, mf);
  if (pGlossy > 0.0f) pdf += pGlossy * pdfGlossy(sc.lut, mt, wo, wi, mf);
  if (e.c > 0.0f) {
    GGX mfc = makeGGX(e.cr);
    float Fc = 0.0f;
    float pdfClear = pdfClearcoat(wo, wi, mfc, &Fc);
    pdf = (1.0f - e.c * Fc) * pdf + e.c * pdfClear;
  }
  return pdf;
}
YART_HD float bsdfPdfImpl(const SceneDev& sc, const MaterialDev& mt, f3 wo, f3 wi, f2 uv) {
  MatEval e;
  e.base = mk3(0);
  matFetchScalars(sc, mt, uv, e);
  return bsdfPdfImplE(sc, mt, e, wo, wi);
}
YART_HD BsdfSample bsdfSampleImplE(const SceneDev& sc, const MaterialDev& mt, MatEval e, f3 _wo, f2 uv, f2 u,
                                   float uc, float uc2, bool regularized) {
  if (regularized) {
    e.r = roughen(e.r);
    e.cr = roughen(e.cr);
  }
  GGX mfCoat = makeGGX(e.cr);
  // The reference computes the clearcoat selection probability for every hit (parametric.cpp:205-224: a visible-normal
  // sample, two LUT lookups, kappa). With c = m = t = 0 — a material without clearcoat, metal or transmission, the bulk of
  // most scenes — it cannot matter: pClearcoat = 0 * (1 - kappa) is +-0 (or NaN), pMetallic and pDielectric are that times 0,
  // and `uc2 < p` is false for all three whatever kappa was; the glossy-diffuse lobe is sampled. Skipping the computation
  // there leaves every result as it is (the shade kernel spends ~a fifth of its instructions on it otherwise).
  float pClearcoat = 0.0f, pMetallic = 0.0f, pDielectric = 0.0f;
  if (!(e.c == 0.0f && e.m == 0.0f && e.t == 0.0f)) {
    f3 wmCoat = ggxSampleVisibleMicrofacet(mfCoat, _wo, u);
    const float Favg = FavgFit(1.5f);
    const float Eavg = ggxEavg(sc.lut, e.cr);
    const float Fms = Favg * Favg * Eavg / (1.0f - Favg * (1.0f - Eavg));
    const float E_o = ggxE(sc.lut, absDot(_wo, wmCoat), e.cr);
    const float kappa = 1.0f - (Favg * E_o + Fms * (1.0f - E_o));
    // "c * (1.0 - kappa)" is evaluated in double (parametric.cpp:221)
    pClearcoat = float(double(e.c) * (1.0 - double(kappa)));
    pMetallic = (1.0f - pClearcoat) * e.m;
    pDielectric = (1.0f - pClearcoat) * (e.m + (1.0f - e.m) * e.t);
  }
  BsdfSample s;
  if (uc2 < pClearcoat) {
    s = sampleClearcoat(mt, _wo, mfCoat, u);
  } else {
    GGX mf = makeGGX(e.r, mt.anisotropic);
    f3 wo = mul3x3(mt.localRot, _wo);
    if (uc2 < pMetallic) {
      s = sampleMetallic(sc.lut, mt, wo, e.base, mf, u);
    } else if (uc2 < pDielectric) {
      s = sampleDielectric(sc.lut, mt, wo, e.base, mf, u, uc);
    } else {
      f3 emission = mt.emission;
      if ((mt.flags & MAT_HAS_EMISSION) && mt.texEmission >= 0)
        emission *= texSample3(sc, mt.texEmission, uv);
      s = sampleGlossy(sc.lut, mt, wo, e.base, emission, mf, u, uc);
    }
    s.wi = mul3x3(mt.invRot, s.wi);
  }
  return s;
}
YART_HD BsdfSample bsdfSampleImpl(const SceneDev& sc, const MaterialDev& mt, f3 _wo, f2 uv, f2 u,
                                  float uc, float uc2, bool regularized) {
  return bsdfSampleImplE(sc, mt, matEvaluate(sc, mt, uv), _wo, uv, u, uc, uc2, regularized);
}

// ---- world-space wrappers (core/bsdf.cpp:5-58) ----
YART_HD Frame shadingFrame(f3 n, f3 t) {
  return length2(t) > 0 ? frameFromNormalTangent(n, t, 1.0f) : frameFromNormal(n);
}
YART_HD f3 bsdfF(const SceneDev& sc, const MaterialDev& mt, f3 wo, f3 wi, f3 n, f3 t, f2 uv) {
  Frame fr = shadingFrame(n, t);
  return bsdfFImpl(sc, mt, wtl(fr, wo), wtl(fr, wi), uv);
}
YART_HD float bsdfPdf(const SceneDev& sc, const MaterialDev& mt, f3 wo, f3 wi, f3 n, f3 t, f2 uv) {
  Frame fr = shadingFrame(n, t);
  return bsdfPdfImpl(sc, mt, wtl(fr, wo), wtl(fr, wi), uv);
}
YART_HD BsdfSample bsdfSample(const SceneDev& sc, const MaterialDev& mt, f3 wo, f3 n, f3 t, f2 uv,
                              f2 u, float uc, float uc2, bool regularized) {
  Frame fr = shadingFrame(n, t);
  BsdfSample s = bsdfSampleImpl(sc, mt, wtl(fr, wo), uv, u, uc, uc2, regularized);
  s.wi = ltw(fr, s.wi);
  return s;
}
YART_HD f3 bsdfNormal(const SceneDev& sc, const MaterialDev& mt, f3 n, f4 t, f2 uv) {   // bsdf.cpp:44-58
  f3 sn = n;
  if (mt.texNormal >= 0) {
    f3 sampled = texSample3(sc, mt.texNormal, uv) * 2.0f - 1.0f;
    Frame fr = frameFromNormalTangent(n, mk3(t.x, t.y, t.z), t.w);
    sn = normalized(ltw(fr, sampled));
  }
  return sn;
}

}  // namespace yart_hip
