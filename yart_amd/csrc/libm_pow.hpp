// libm_pow.hpp — glibc 2.35 log2f / powf evaluated by their own algorithms (sysdeps/ieee754/flt-32/
// e_log2f.c, e_powf.c: table-driven log2 in double precision, 2^x by the exp2f table and a cubic),
// with the multiply-adds contracted to FMAs as the x86-64 multiarch variants (__log2f_fma, __powf_fma)
// are built. Host + device: the tonemap stage (tonemap.hpp) needs the reference's exact values, and
// tests/hostsim `libmcheck` compares these functions with the box's libm over every float / a dense
// sweep. Tables: __log2f_data, __powf_log2_data, __exp2f_data as found in libm.so.6.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define YART_LIBM_HD __host__ __device__ __forceinline__
#else
#define YART_LIBM_HD inline
#endif

namespace yart_hip {
namespace libm_pow {

YART_LIBM_HD uint32_t asU32(float f) { return __builtin_bit_cast(uint32_t, f); }
YART_LIBM_HD float asF32(uint32_t u) { return __builtin_bit_cast(float, u); }
YART_LIBM_HD uint64_t asU64(double d) { return __builtin_bit_cast(uint64_t, d); }
YART_LIBM_HD double asF64(uint64_t u) { return __builtin_bit_cast(double, u); }

// {invc, logc}: 1/c and log2(c) for the 16 subintervals of [0x1.66p-1, 0x1.66p0)
YART_LIBM_HD void log2Entry(int i, double& invc, double& logc) {
  const double T[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
  invc = T[i][0]; logc = T[i][1];
}

YART_LIBM_HD float log2f_(float x) {                        // e_log2f.c
  uint32_t ix = asU32(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return -__builtin_huge_valf();         // __math_divzerof(1)
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);   // __math_invalidf: NaN (a NaN x stays itself)
    ix = asU32(x * 0x1p23f);
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = int((tmp >> 19) % 16u);
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = int32_t(tmp) >> 23;
  double invc, logc;
  log2Entry(i, invc, logc);
  const double z = double(asF32(iz));
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = logc + double(k);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.ecabf496832ep-2, r, -0x1.715479ffae3dep-1);       // A[1]*r + A[2]
  y = __builtin_fma(-0x1.712b6f70a7e4dp-2, r2, y);                                // A[0]*r2 + y
  const double p = __builtin_fma(0x1.715475f35c8b8p+0, r, y0);                    // A[3]*r + y0
  y = __builtin_fma(y, r2, p);
  return float(y);
}

YART_LIBM_HD uint64_t exp2Tab(uint32_t i) {                  // __exp2f_data.tab
  const uint64_t T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  return T[i];
}

// checkint (e_powf.c): 0 not an integer, 1 odd, 2 even
YART_LIBM_HD int checkint(uint32_t iy) {
  const int e = int(iy >> 23 & 0xff);
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
YART_LIBM_HD bool zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

YART_LIBM_HD float powf_(float x, float y) {                 // e_powf.c
  uint32_t signBias = 0;
  uint32_t ix = asU32(x);
  const uint32_t iy = asU32(y);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || zeroinfnan(iy)) {
    if (zeroinfnan(iy)) {
      if (2 * iy == 0) return 1.0f;                          // (signalling NaNs are not distinguished here)
      if (ix == 0x3f800000u) return 1.0f;
      if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
      if (2 * ix == 2 * 0x3f800000u) return 1.0f;
      if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
      return y * y;
    }
    if (zeroinfnan(ix)) {
      float x2 = x * x;
      if ((ix & 0x80000000u) && checkint(iy) == 1) x2 = -x2;
      return (iy & 0x80000000u) ? 1 / x2 : x2;
    }
    if (ix & 0x80000000u) {                                  // finite x < 0
      const int yint = checkint(iy);
      if (yint == 0) return (x - x) / (x - x);               // __math_invalidf
      if (yint == 1) signBias = 1u << (5 + 11);              // SIGN_BIAS
      ix &= 0x7fffffffu;
    }
    if (ix < 0x00800000u) {                                  // subnormal x
      ix = asU32(x * 0x1p23f);
      ix &= 0x7fffffffu;
      ix -= 23u << 23;
    }
  }
  // log2_inline
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = int((tmp >> 19) % 16u);
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = int32_t(top) >> 23;
  double invc, logc;
  log2Entry(i, invc, logc);
  const double z = double(asF32(iz));
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = logc + double(k);
  const double r2 = r * r;
  double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);    // A[0]*r + A[1]
  const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);   // A[2]*r + A[3]
  const double r4 = r2 * r2;
  double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);                            // A[4]*r + y0
  q = __builtin_fma(p, r2, q);
  const double logx = __builtin_fma(yy, r4, q);
  const double ylogx = double(y) * logx;
  if ((asU64(ylogx) >> 47 & 0xffff) >= (asU64(126.0) >> 47)) {
    if (ylogx > 0x1.fffffffd1d571p+6) {                      // __math_oflowf
      const float huge = 0x1p97f;
      return (signBias ? -huge : huge) * huge;
    }
    if (ylogx <= -150.0) {                                   // __math_uflowf
      const float tiny = 0x1p-95f;
      return (signBias ? -tiny : tiny) * tiny;
    }
  }
  // exp2_inline
  const double SHIFT = 0x1.8p+52 / 32;
  double kd = ylogx + SHIFT;
  const uint64_t ki = asU64(kd);
  kd -= SHIFT;
  const double rr = ylogx - kd;
  uint64_t t = exp2Tab(uint32_t(ki % 32));
  const uint64_t ski = ki + signBias;
  t += ski << (52 - 5);
  const double s = asF64(t);
  const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
  const double rr2 = rr * rr;
  double out = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
  out = __builtin_fma(zz, rr2, out);
  out = out * s;
  return float(out);
}

}  // namespace libm_pow
}  // namespace yart_hip
