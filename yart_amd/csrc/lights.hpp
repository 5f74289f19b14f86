// lights.hpp — area / infinite lights, environment-map importance sampling and the
// power light sampler. Restates reference core/light.cpp:16-243,
// core/light-sampler.cpp:32-93, math/sampling.hpp:54-64 + 118-196,
// math/sampling.cpp:5-60 and math/math_base.hpp:106-119 (findFirst).
#pragma once
#include "bsdf.hpp"

namespace yart_hip {

struct LightSample {         // core/light.hpp:10-15
  f3 Li, wi, p, n;
  float pdf;
};
YART_HD LightSample emptyLightSample() {
  LightSample s; s.Li = mk3(0); s.wi = mk3(0); s.p = mk3(0); s.n = mk3(0); s.pdf = 0; return s;
}

YART_HD f3 sampleTriUniform(f2 u) {                       // sampling.hpp:54-64
  float b0, b1;
  if (u.x < u.y) { b0 = u.x * 0.5f; b1 = u.y - b0; }
  else { b1 = u.y * 0.5f; b0 = u.x - b1; }
  return mk3(b0, b1, 1.0f - b0 - b1);
}

// PiecewiseConstant1D::sample (sampling.cpp:5-34) over cdf[0..n], func[0..n-1].
// The reference's search is a lower bound over cdf[1..n-1]: A(u) = first index whose cdf is >= u
// (n if none), which is monotone in u. With a guide table G[j] = A(j / K), K a power of two (so
// that floor(u * K) is exact), A(u) lies in [G[j], G[j+1]] for j = floor(u * K): the same index is
// found with ~1-2 dependent loads instead of log2(n).
YART_HD float pc1dSample(const float* func, const float* cdf, uint32_t n, float integral, float mn,
                         float mx, float u, float& pdf, uint32_t& offset, const uint32_t* guide = nullptr,
                         uint32_t K = 0) {
  uint32_t first = 1, size = n - 1;                 // n >= 1
  if (guide != nullptr) {
    uint32_t j = u >= 0.0f ? uint32_t(u * float(K)) : 0u;     // NaN -> 0
    if (j > K - 1u) j = K - 1u;                               // u < 1 for every sampler value
    first = guide[j];
    size = guide[j + 1u] - first;
  }
  while (size > 0) {
    const uint32_t half = size >> 1, middle = first + half;
    if (cdf[middle] < u) { first = middle + 1; size -= half + 1; }
    else size = half;
  }
  uint32_t o = first - 1;                           // first >= 1
  if (o > n - 1) o = n - 1;
  offset = o;
  float du = u - cdf[o];
  // reference typo kept: normalises by cdf[0+1]-cdf[o]
  if (cdf[1] - cdf[o] > 0) du /= cdf[1] - cdf[o];
  pdf = (integral > 0) ? func[o] / integral : 0.0f;
  return lerpf(mn, mx, (float(o) + du) / float(n));
}

YART_HD f3 envLe(const SceneDev& sc, const LightDev& l, f2 uv) {      // light.cpp:199-204
  // bounds are the full [0,1]^2 domain (ImageInfiniteLight default, light.hpp:150-154)
  if (uv.x < 0.0f || uv.x > 1.0f || uv.y < 0.0f || uv.y > 1.0f) return mk3(0);
  return texSample3(sc, l.texture, uv);
}
YART_HD float envPdf(const SceneDev& sc, const LightDev& l, f3 wi) {  // light.cpp:211-217
  f2 uv = octahedralUV(mulVector(l.xf.inv, wi));
  if (uv.x < 0.0f || uv.x > 1.0f || uv.y < 0.0f || uv.y > 1.0f) return 0;
  const EnvDev& e = sc.envs[l.envOffset];
  // PiecewiseConstant2D::pdf (sampling.cpp:45-58); domain (0,0)-(1,1): p = (uv - 0) / 1
  f2 p = mk2((uv.x - 0.0f) / 1.0f, (uv.y - 0.0f) / 1.0f);
  uint32_t iu = uint32_t(p.x * float(e.w)); if (iu > e.w - 1) iu = e.w - 1;
  uint32_t iv = uint32_t(p.y * float(e.h)); if (iv > e.h - 1) iv = e.h - 1;
  float pdf = sc.envData[e.funcOffset + iv * e.w + iu] / e.margIntegral;
  return pdf / (4.0f * kPi);
}
// PiecewiseConstant1D::sample (sampling.cpp:5-34, as pc1dSample above) over interleaved records: record k = STRIDE floats,
// [0] = cdf[k], [1] = func[k]. cdf1 = cdf[1] (the reference's normalisation typo, Appendix A.7). Returns the record index.
template <uint32_t STRIDE>
YART_HD float pc1dSampleRecords(const float* rec, uint32_t n, float integral, float cdf1, float u, float& pdf, uint32_t& offset,
                                const uint32_t* guide, uint32_t K) {
  uint32_t first = 1, size = n - 1;
  if (guide != nullptr) {
    uint32_t j = u >= 0.0f ? uint32_t(u * float(K)) : 0u;
    if (j > K - 1u) j = K - 1u;
    first = guide[j];
    size = guide[j + 1u] - first;
  }
  while (size > 0) {
    const uint32_t half = size >> 1, middle = first + half;
    if (rec[size_t(middle) * STRIDE] < u) { first = middle + 1; size -= half + 1; }
    else size = half;
  }
  uint32_t o = first - 1;
  if (o > n - 1) o = n - 1;
  offset = o;
  const float cdfO = rec[size_t(o) * STRIDE], funcO = rec[size_t(o) * STRIDE + 1u];
  float du = u - cdfO;
  if (cdf1 - cdfO > 0) du /= cdf1 - cdfO;
  pdf = (integral > 0) ? funcO / integral : 0.0f;
  return lerpf(0.0f, 1.0f, (float(o) + du) / float(n));
}
YART_HD LightSample envSample(const SceneDev& sc, const LightDev& l, f2 u) {   // light.cpp:219-238
  const EnvDev& e = sc.envs[l.envOffset];
  float pdf1, pdf0;
  uint32_t ov, ou;
  // marginal over v with u.y, then conditional row with u.x (sampling.cpp:36-43), on the interleaved copies of the tables
  // (EnvDev::pairOffset): the marginal record of the chosen row brings the row's integral and cdf[1] with it
  const uint32_t* gm = e.guideKh ? sc.envGuide + e.guideOffset : nullptr;
  const float* marg = sc.envData + e.pairOffset;
  float d1 = pc1dSampleRecords<4>(marg, e.h, e.margIntegral, marg[4], u.y, pdf1, ov, gm, e.guideKh);
  const float rowInt = marg[size_t(ov) * 4u + 1u], rowCdf1 = marg[size_t(ov) * 4u + 2u];
  const uint32_t* gr = e.guideKw ? sc.envGuide + e.guideOffset + (e.guideKh + 1u) + size_t(ov) * (e.guideKw + 1u) : nullptr;
  const float* row = marg + size_t(e.h + 1u) * 4u + size_t(ov) * (e.w + 1u) * 2u;
  float d0 = pc1dSampleRecords<2>(row, e.w, rowInt, rowCdf1, u.x, pdf0, ou, gr, e.guideKw);
  float pdf = pdf0 * pdf1;
  if (pdf == 0.0f) return emptyLightSample();
  f2 uv = mk2(d0, d1);
  f3 wi = mulVector(l.xf.fwd, invOctahedralUV(uv));
  pdf /= e.surfaceArea;
  LightSample s;
  s.Li = envLe(sc, l, uv);
  s.wi = wi;
  s.p = (wi * 2.0f) * l.radius;
  s.n = -wi;
  s.pdf = pdf;
  return s;
}

YART_HD LightSample areaSample(const SceneDev& sc, const LightDev& l, f3 p, f2 u) {   // light.cpp:44-73
  const MeshDev& mesh = sc.meshes[l.mesh];
  const u4 tv = sc.triVerts[mesh.triOffset + l.tri];
  const f3 b = sampleTriUniform(u);
  const f4 p0 = sc.vPos[mesh.vertOffset + tv.x], p1 = sc.vPos[mesh.vertOffset + tv.y],
           p2 = sc.vPos[mesh.vertOffset + tv.z];
  const f4 n0 = sc.vNormal[mesh.vertOffset + tv.x], n1 = sc.vNormal[mesh.vertOffset + tv.y],
           n2 = sc.vNormal[mesh.vertOffset + tv.z];
  f3 pos = (b.x * mk3(p0.x, p0.y, p0.z) + b.y * mk3(p1.x, p1.y, p1.z)) + b.z * mk3(p2.x, p2.y, p2.z);
  f3 nrm = (b.x * mk3(n0.x, n0.y, n0.z) + b.y * mk3(n1.x, n1.y, n1.z)) + b.z * mk3(n2.x, n2.y, n2.z);
  pos = mulPoint(l.xf.fwd, pos);
  nrm = mulNormalT(l.xf.inv, nrm);
  LightSample s;
  s.Li = l.emission;
  s.wi = normalized(pos - p);
  s.p = pos;
  s.n = nrm;
  s.pdf = 1.0f / l.area;
  return s;
}

YART_HD LightSample lightSample(const SceneDev& sc, const LightDev& l, f3 p, f2 u) {
  if (l.type == LIGHT_AREA) return areaSample(sc, l, p, u);
  if (l.type == LIGHT_IMAGE_INF) return envSample(sc, l, u);
  return emptyLightSample();                               // UniformInfiniteLight::sample, light.cpp:112-131
}
YART_HD float lightPdf(const SceneDev& sc, const LightDev& l, f3 wi) {
  if (l.type == LIGHT_AREA) return 1.0f / l.area;          // light.cpp:40-42
  if (l.type == LIGHT_IMAGE_INF) return envPdf(sc, l, wi);
  return 0;                                                // light.cpp:106-110
}
YART_HD f3 lightLe(const SceneDev& sc, const LightDev& l, f2 uv) {
  if (l.type == LIGHT_IMAGE_INF) return envLe(sc, l, uv);
  return l.emission;
}

// PowerLightSampler (light-sampler.cpp:52-93)
YART_HD float pInfinite(const SceneDev& sc) {
  return sc.nArea == 0 ? 1.0f : float(sc.nInfinite) / float(sc.nInfinite + 1);
}
YART_HD uint32_t lightSamplerSample(const SceneDev& sc, float u, float& pl) {
  const uint32_t infCount = sc.nInfinite;
  const float pInf = pInfinite(sc);
  if (u < pInf) {
    u /= pInf;
    // min(infCount - 1, size_t(u * infCount)): u in [0,1) so no negative conversion
    uint32_t idx = uint32_t(u * float(infCount));
    if (infCount - 1 < idx) idx = infCount - 1;
    pl = pInf / float(infCount);
    return sc.infiniteLights[idx];
  }
  u = (u - pInf) / (1.0f - pInf);
  u *= sc.totalPower;
  // findFirst(size, i -> powers[i] < u), math_base.hpp:106-119
  int64_t size = int64_t(sc.nArea);
  int64_t sz = size - 1, first = 0;
  while (sz > 0) {
    int64_t half = sz >> 1, middle = first + half;
    bool res = sc.areaPowerCdf[middle] < u;
    first = res ? (middle + 1) : first;
    sz = res ? sz - (half + 1) : half;
  }
  if (first < 0) first = 0;
  if (first > size - 1) first = size - 1;
  const uint32_t li = sc.areaLights[first];
  pl = sc.lights[li].power / sc.totalPower * (1.0f - pInf);
  return li;
}
YART_HD float lightSamplerP(const SceneDev& sc, uint32_t lightIdx) {   // light-sampler.cpp:80-93
  const float pInf = pInfinite(sc);
  const LightDev& l = sc.lights[lightIdx];
  if (l.type != LIGHT_AREA) return pInf / float(sc.nInfinite);
  return l.power / sc.totalPower * (1.0f - pInf);
}

}  // namespace yart_hip
