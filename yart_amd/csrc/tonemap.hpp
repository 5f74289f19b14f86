// tonemap.hpp — AgX tonemap and 8-bit output encoding (the step right after
// the path). Restates reference core/tonemapping.hpp:14-92 (AgX::start / applyLook / end with the
// looks none / golden / punchy) and the per-channel body of output/ppm.cpp:7-21, expression by
// expression. log2 / pow are glibc's log2f / powf in the reference; libm_pow.hpp evaluates the same
// algorithms on the device, so the tonemapped frame and the bytes are the reference's bit for bit.
#pragma once
#include "ymath.hpp"
#include "libm_pow.hpp"

namespace yart_hip {

struct AgxLook { f3 offset, slope, power; float sat; };

YART_HD AgxLook agxLook(int which) {           // tonemapping.hpp:21-34
  AgxLook l;
  l.offset = mk3(0.0f); l.slope = mk3(1.0f); l.power = mk3(1.0f); l.sat = 1.0f;                      // none
  if (which == 1) { l.slope = mk3(1.0f, 0.9f, 0.5f); l.power = mk3(0.8f); l.sat = 0.8f; }          // golden
  if (which == 2) { l.power = mk3(1.35f); l.sat = 1.4f; }                                          // punchy
  return l;
}

// log2 / pow with glibc's values (libm_pow.hpp: its algorithms and tables, checked against libm over every
// float for log2f and for the exponents used here), on the device and on the host alike
YART_HD float ylog2f(float x) { return libm_pow::log2f_(x); }
YART_HD float ypowf(float x, float y) { return libm_pow::powf_(x, y); }

YART_HD f3 agxContrast(f3 x) {                 // tonemapping.hpp:43-54 (double literals narrow to float)
  const f3 x2 = x * x;
  const f3 x4 = x2 * x2;
  return (((((((x4 * 15.5f) * x2) - ((x4 * 40.14f) * x)) + (x4 * 31.96f)) - ((x2 * 6.868f) * x)) + (x2 * 0.4298f)) +
          (x * 0.1191f)) - 0.00232f;
}

YART_HD f3 agxTonemap(f3 hdr, const AgxLook& look) {
  // start (:56-71)
  const float agx[9] = {float(0.842479062253094), float(0.0784335999999992), float(0.0792237451477643),
                        float(0.0423282422610123), float(0.878468636469772), float(0.0791661274605434),
                        float(0.0423756549057051), float(0.0784336), float(0.879142973793104)};
  const float minEv = -12.47393f, maxEv = 4.026069f;
  f3 val = mul3x3(agx, hdr);
  val = mk3(ymin(maxEv, ymax(minEv, ylog2f(val.x))), ymin(maxEv, ymax(minEv, ylog2f(val.y))),
            ymin(maxEv, ymax(minEv, ylog2f(val.z))));
  val = (val - minEv) / (maxEv - minEv);
  val = agxContrast(val);
  // applyLook (:73-79)
  const float luma = dot(val, mk3(0.2126f, 0.7152f, 0.0722f));
  const f3 b = val * look.slope + look.offset;
  val = mk3(ypowf(b.x, look.power.x), ypowf(b.y, look.power.y), ypowf(b.z, look.power.z));
  val = mk3(luma) + look.sat * (val - luma);
  // end (:81-91)
  const float inv[9] = {float(1.19687900512017), float(-0.0980208811401368), float(-0.0990297440797205),
                        float(-0.0528968517574562), float(1.15190312990417), float(-0.0989611768448433),
                        float(-0.0529716355144438), float(-0.0980434501171241), float(1.15107367264116)};
  val = mul3x3(inv, val);
  val = mk3(ymin(1.0f, ymax(0.0f, val.x)), ymin(1.0f, ymax(0.0f, val.y)), ymin(1.0f, ymax(0.0f, val.z)));
  return mk3(ypowf(val.x, 2.2f), ypowf(val.y, 2.2f), ypowf(val.z, 2.2f));
}

// output/ppm.cpp:15-17: clamp(pow(v, 1/2.2), 0, 1) * 255.999 truncated to a byte. std::clamp leaves a
// NaN unchanged and the x86 float -> uint8 conversion of NaN yields 0: stated explicitly.
YART_HD uint8_t ppmByte(float v) {
  const float gamma = 1.0f / 2.2f;
  float mapped = ypowf(v, gamma);
  mapped = mapped < 0.0f ? 0.0f : (1.0f < mapped ? 1.0f : mapped);      // std::clamp(mapped, 0, 1)
  if (!(mapped == mapped)) return 0;
  return uint8_t(mapped * 255.999f);
}

}  // namespace yart_hip
