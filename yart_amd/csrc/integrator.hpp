// integrator.hpp — camera ray generation and the MIS + NEE path loop for one
// (pixel, sample) in a single flow of control (the "megakernel" form; the
// wavefront kernels in wavefront.hip split the same loop at the trace calls).
//
// Restates reference core/camera.hpp:138-164 + math/sampling.hpp:20-28, 72-89,
// cpu/ray-integrator.cpp:11-18 and cpu/mis-integrator.cpp:13-148.
// Sampler draw order: film 2D, lens 2D; per bounce:
// traversal alpha draws, 2D u, 1D uc, 1D uc2, [NEE: 1D, 2D, shadow alpha draws],
// [Russian roulette 1D].
#pragma once
#include "lights.hpp"
#include "traverse.hpp"

namespace yart_hip {

struct RenderConst {
  SamplerConfig sampler;
  uint32_t maxDepth;
  f3 background;
};

YART_HD f2 pixelJitterGaussian(f2 u, float stdDev) {          // sampling.hpp:20-28
  float a = sqrtf(-2.0f * ylogf(u.x)) * stdDev;
  float b = 2.0f * kPi * u.y;
  return mk2(a * ycosf(b), a * ysinf(b));
}
YART_HD f2 samplePolyUniform(f2 u, uint32_t sides) {          // sampling.hpp:72-89
  u.x *= float(sides);
  uint32_t side = uint32_t(u.x);
  if (sides - 1 < side) side = sides - 1;
  u.x -= float(side);
  f3 b = sampleTriUniform(u);
  float theta1 = float(side) / float(sides) * 2.0f * kPi;
  float theta2 = float(side + 1) / float(sides) * 2.0f * kPi;
  float c1 = ycosf(theta1), s1 = ysinf(theta1);
  float c2 = ycosf(theta2), s2 = ysinf(theta2);
  f2 r = (mk2(0, 0) * b.x + mk2(-s1, c1) * b.y) + mk2(-s2, c2) * b.z;
  return r;
}
YART_HD void cameraRay(const CameraDev& cam, uint32_t px, uint32_t py, f2 uvFilm, f2 uvLens, f3& o,
                       f3& d) {                               // camera.hpp:138-164
  f2 j = pixelJitterGaussian(uvFilm, 0.3f);
  f2 jitter = mk2(j.x + float(px), j.y + float(py));
  f3 pixel = (cam.topLeftPixel + cam.pixelDeltaU * jitter.x) + cam.pixelDeltaV * jitter.y;
  f3 origin = cam.position;
  if (cam.apertureRadius > 0.0f) {
    f2 a = cam.apertureSides == 0 ? sampleDiskUniform(uvLens) : samplePolyUniform(uvLens, cam.apertureSides);
    f3 lensPos = mk3(a.x, a.y, 0.0f);
    lensPos *= cam.apertureRadius;
    Frame fr; fr.x = cam.frameX; fr.y = cam.frameY; fr.z = cam.frameZ;
    origin += ltw(fr, lensPos);
  }
  o = origin;
  d = normalized(pixel - origin);
}

struct PathCtx {
  const SceneDev* sc;
  const uint32_t* sobol;       // 52-entry dimension-1 generator matrix
  TravStack stk;
  RenderConst rc;
#if defined(YART_COUNT_TRAVERSAL)
  mutable uint32_t nBox = 0, nTri = 0, nTrav = 0, nShade = 0;
#endif
};
#if defined(YART_COUNT_TRAVERSAL)
#define YART_FOLD_COUNTS(cx, ac) ((cx).nBox += (ac).nBox, (cx).nTri += (ac).nTri, (cx).nTrav += (ac).nTrav)
#define YART_COUNT_SHADE(cx) ((cx).nShade++)
#else
#define YART_FOLD_COUNTS(cx, ac) ((void)0)
#define YART_COUNT_SHADE(cx) ((void)0)
#endif

// MISIntegrator::unoccluded (mis-integrator.cpp:135-148)
YART_HD bool unoccluded(const PathCtx& cx, Sampler& smp, f3 from, f3 to, f3& attenuation) {
  f3 dir = normalized(to - from);
  HitRec hr;
  hr.t = length(to - from) - 0.001f;
  hr.u = hr.v = 0; hr.tri = 0; hr.node = 0; hr.backSide = 0;
  attenuation = mk3(1.0f);
  AlphaCtx ac; ac.sampler = &smp; ac.cfg = cx.rc.sampler;
#if defined(YART_EMULATE_LEAN_HANDOVER)
  {   // host tests only (tests/hostsim): the lean kernels' walk first; a ray it hands over is traced again by the general walk
    HitRec hl = hr; f3 al = mk3(1.0f);
    AlphaCtx af; af.sampler = nullptr; af.cfg = SamplerConfig{};
    const bool occl = traverseScene<true, TRAV_FAST>(*cx.sc, from, dir, 0.001f, hl, al, cx.stk, af);
#if defined(YART_EMULATE_LEAN_DEBUG)
    if (!af.deferred) {
      Sampler sc2 = smp; HitRec hg = hr; f3 ag = mk3(1.0f);
      AlphaCtx ax; ax.sampler = &sc2; ax.cfg = cx.rc.sampler;
      const bool og = traverseScene<true>(*cx.sc, from, dir, 0.001f, hg, ag, cx.stk, ax);
      if (og != occl || sc2.dim != smp.dim || (!og && (ag.x != al.x || ag.y != al.y || ag.z != al.z)))
        std::fprintf(stderr, "SHADOW differs: lean occl %d t %.9g tri %u node %u | general occl %d t %.9g tri %u node %u draws %u att %g %g %g | from %.9g %.9g %.9g dir %.9g %.9g %.9g tmax %.9g\n",
                     int(occl), hl.t, hl.tri, hl.node, int(og), hg.t, hg.tri, hg.node, sc2.dim - smp.dim, ag.x, ag.y, ag.z, from.x, from.y, from.z, dir.x, dir.y, dir.z, hr.t);
    }
#endif
    if (!af.deferred) { attenuation = al; return !occl; }
  }
#endif
  bool occluded = traverseScene<true>(*cx.sc, from, dir, 0.001f, hr, attenuation, cx.stk, ac);
  YART_FOLD_COUNTS(cx, ac);
  return !occluded;
}

// MISIntegrator::Ld (mis-integrator.cpp:111-133)
YART_HD f3 directLight(const PathCtx& cx, Sampler& smp, f3 wo, const Hit& hit, uint32_t& rays) {
  const SceneDev& sc = *cx.sc;
  if (sc.nLights == 0) return mk3(0);
  float uc = get1D(smp, cx.rc.sampler);
  f2 u = get2D(smp, cx.rc.sampler, cx.sobol);
  float pl;
  uint32_t li = lightSamplerSample(sc, uc, pl);
  const LightDev& l = sc.lights[li];
  LightSample ls = lightSample(sc, l, hit.p, u);
  const MaterialDev& mt = sc.materials[hit.material];
  f3 f = bsdfF(sc, mt, wo, ls.wi, hit.n, hit.tg, hit.uv);
  f3 att = mk3(1.0f);
  if (length2(f) == 0.0f || !unoccluded(cx, smp, hit.p, ls.p, att)) return mk3(0);
  rays++;
  float pdfBSDF = bsdfPdf(sc, mt, wo, ls.wi, hit.n, hit.tg, hit.uv);
  float pdfLight = pl * ls.pdf / absDot(ls.n, ls.wi);
  if (l.type == LIGHT_AREA) pdfLight *= length2(hit.p - ls.p);
  return ls.Li * f * att * absDot(ls.wi, hit.n) / (pdfBSDF + pdfLight);
}

// MISIntegrator::Li (mis-integrator.cpp:13-106)
YART_HD f3 pathRadiance(const PathCtx& cx, Sampler& smp, f3 ro, f3 rd, uint32_t& rays) {
  const SceneDev& sc = *cx.sc;
  f3 lastP = mk3(0);
  f3 L = mk3(0.0f), attenuation = mk3(1.0f);
  uint32_t depth = 0;
  bool specularBounce = false, regularized = false;
  float lastPdf = 0.0f, accRoughness = 0.0f;
  while (depth < cx.rc.maxDepth) {
    rays++;
    HitRec hr;
    hr.t = kInf; hr.u = hr.v = 0; hr.tri = 0; hr.node = 0; hr.backSide = 0;
    f3 dummy = mk3(1.0f);
    AlphaCtx ac; ac.sampler = &smp; ac.cfg = cx.rc.sampler;
    bool didHit;
#if defined(YART_EMULATE_LEAN_HANDOVER)
    {
      HitRec hl = hr; f3 al = mk3(1.0f);
      AlphaCtx af; af.sampler = nullptr; af.cfg = SamplerConfig{};
      const bool h = traverseScene<false, TRAV_FAST>(sc, ro, rd, 0.001f, hl, al, cx.stk, af);
#if defined(YART_EMULATE_LEAN_DEBUG)
      if (!af.deferred) {
        Sampler sc2 = smp; HitRec hg = hr; f3 ag = mk3(1.0f);
        AlphaCtx ax; ax.sampler = &sc2; ax.cfg = cx.rc.sampler;
        const bool g = traverseScene<false>(sc, ro, rd, 0.001f, hg, ag, cx.stk, ax);
        if (g != h || sc2.dim != smp.dim || (g && (hg.t != hl.t || hg.tri != hl.tri || hg.node != hl.node || hg.u != hl.u)))
          std::fprintf(stderr, "EXTEND differs: lean hit %d t %.9g tri %u node %u | general hit %d t %.9g tri %u node %u draws %u | o %.9g %.9g %.9g d %.9g %.9g %.9g\n",
                       int(h), hl.t, hl.tri, hl.node, int(g), hg.t, hg.tri, hg.node, sc2.dim - smp.dim, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
      }
#endif
      if (!af.deferred) { hr = hl; didHit = h; }
      else didHit = traverseScene<false>(sc, ro, rd, 0.001f, hr, dummy, cx.stk, ac);
    }
#else
    didHit = traverseScene<false>(sc, ro, rd, 0.001f, hr, dummy, cx.stk, ac);
#endif
    YART_FOLD_COUNTS(cx, ac);
    if (!didHit) {
      for (uint32_t k = 0; k < sc.nInfinite; k++) {
        const LightDev& l = sc.lights[sc.infiniteLights[k]];
        f3 Le = lightLe(sc, l, octahedralUV(rd));      // ignores the light's transform (Appendix A.8)
        if (depth == 0 || specularBounce) {
          L += attenuation * Le;
        } else {
          float pdfLight = lightPdf(sc, l, rd);
          float wBSDF = lastPdf / (lastPdf + pdfLight);
          L += attenuation * wBSDF * Le;
        }
      }
      L += attenuation * cx.rc.background;
      break;
    }
    Hit hit = finalizeHit(sc, hr, ro, rd);
    YART_COUNT_SHADE(cx);
    const MaterialDev& mt = sc.materials[hit.material];

    f2 u = get2D(smp, cx.rc.sampler, cx.sobol);
    float uc = get1D(smp, cx.rc.sampler);
    float uc2 = get1D(smp, cx.rc.sampler);
    BsdfSample res = bsdfSample(sc, mt, -rd, hit.n, hit.tg, hit.uv, u, uc, uc2, regularized);

    if (res.scatter & SC_EMITTED) {
      if (depth == 0 || specularBounce) {
        L += attenuation * res.Le;
      } else if (hit.lightIdx != -1) {
        const LightDev& l = sc.lights[hit.lightIdx];
        float pdfLight = lightPdf(sc, l, -rd) * length2(lastP - hit.p) *
                         lightSamplerP(sc, uint32_t(hit.lightIdx)) / absDot(-rd, hit.n);
        float wBSDF = lastPdf / (lastPdf + pdfLight);
        L += attenuation * wBSDF * res.Le;
      }
    }
    if (!(res.scatter & (SC_REFLECTED | SC_TRANSMITTED))) break;

    if (!(res.scatter & (SC_EMITTED | SC_SPECULAR)))
      L += attenuation * directLight(cx, smp, -rd, hit, rays);

    f3 fcos = res.f * absDot(res.wi, hit.n);
    attenuation *= fcos / res.pdf;
    if (hit.backSide) attenuation *= matAttenuation(mt, hit.t);
    ro = hit.p; rd = res.wi;

    specularBounce = (res.scatter & SC_SPECULAR) != 0;
    accRoughness += res.roughness;
    regularized = accRoughness > 0.5f;
    lastPdf = res.pdf;
    lastP = hit.p;
    depth++;

    if (depth > 1 && maxComponent(attenuation) < 1.0f) {
      float q = stdmax(0.0f, 1.0f - maxComponent(attenuation));
      if (get1D(smp, cx.rc.sampler) < q) break;
      attenuation /= 1.0f - q;
    }
  }
  return L;
}

// RayIntegrator::sample (ray-integrator.cpp:11-18): one radiance sample of pixel (px,py)
YART_HD f3 samplePixel(const PathCtx& cx, const CameraDev& cam, uint32_t px, uint32_t py, uint32_t s,
                       uint32_t& rays) {
  Sampler smp;
  startPixelSample(smp, cx.rc.sampler, px, py, s);
  f2 uvFilm = get2D(smp, cx.rc.sampler, cx.sobol);      // getPixel2D(); evaluated first (Appendix A.1)
  f2 uvLens = get2D(smp, cx.rc.sampler, cx.sobol);
  f3 o, d;
  cameraRay(cam, px, py, uvFilm, uvLens, o, d);
  return pathRadiance(cx, smp, o, d, rays);
}

}  // namespace yart_hip
