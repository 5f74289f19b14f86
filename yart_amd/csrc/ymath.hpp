// ymath.hpp — scalar/vector arithmetic for the path-tracing kernels.
//
// Parity contract: the reference (teofum/yart) is compiled for baseline x86-64,
// i.e. every float operation is an individually rounded IEEE-754 binary32
// operation in source order (no FMA contraction, no reassociation). This header
// states each vector operation with the same association the reference's
// templates expand to (file:line given per function), and the build uses
// -ffp-contract=off and correctly rounded division / sqrt, so that device
// results are bit-identical wherever no transcendental function is involved.
//
// The same header compiles for gfx950 device code (hipcc) and for the host side
// of the library (scene preparation runs the identical arithmetic on the CPU).
#pragma once
#include <cstdint>
#include <cmath>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define YART_HD __host__ __device__ __forceinline__
#else
#define YART_HD inline
#endif

namespace yart_hip {

constexpr float kPi = 3.14159274101257324f;          // float(M_PI), math_base.hpp:12
constexpr float kOneMinusEpsilon = 0x1.fffffep-1f;   // math_base.hpp:15
constexpr float kInf = __builtin_huge_valf();

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct u4 { uint32_t x, y, z, w; };

YART_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
YART_HD f3 mk3(float s) { return mk3(s, s, s); }
YART_HD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }

// vec.hpp:156-237 — component-wise operators
YART_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
YART_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
YART_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
YART_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
YART_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
YART_HD f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }   // vec.hpp:286-292
YART_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
YART_HD f3 operator+(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
YART_HD f3 operator-(f3 a, float s) { return mk3(a.x - s, a.y - s, a.z - s); }
YART_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
YART_HD f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
YART_HD f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
YART_HD f3& operator*=(f3& a, float s) { a = a * s; return a; }
YART_HD f3& operator/=(f3& a, float s) { a = a / s; return a; }

YART_HD f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
YART_HD f2 operator*(f2 a, float s) { return mk2(a.x * s, a.y * s); }
YART_HD f2 operator*(float s, f2 a) { return mk2(a.x * s, a.y * s); }

// vec.hpp:385-388 / 336-345 / 347-354
YART_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
YART_HD float length2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
YART_HD float length(f3 a) { return sqrtf(length2(a)); }
YART_HD f3 normalized(f3 a) { return a / length(a); }
YART_HD float absDot(f3 a, f3 b) { return fabsf(dot(a, b)); }     // vec.hpp:391-396
YART_HD f3 cross(f3 a, f3 b) {                                     // vec.hpp:399-408
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vec.hpp:377-383 — note: seeded with FLT_MIN, not lowest
YART_HD float maxComponent(f3 a) {
  float m = 1.17549435e-38f;
  if (a.x > m) m = a.x;
  if (a.y > m) m = a.y;
  if (a.z > m) m = a.z;
  return m;
}
// math_base.hpp:86-94: min/max as "m < n ? m : n" (NaN falls through to n)
YART_HD float ymin(float m, float n) { return m < n ? m : n; }
YART_HD float ymax(float m, float n) { return m > n ? m : n; }
// std::min / std::max / std::clamp semantics (first argument wins on NaN/equal)
YART_HD float stdmin(float a, float b) { return b < a ? b : a; }
YART_HD float stdmax(float a, float b) { return a < b ? b : a; }
YART_HD float stdclamp(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
YART_HD float lerpf(float a, float b, float t) { return (1.0f - t) * a + t * b; }   // math_base.hpp:36-39

// ---------------------------------------------------------------------------
// Transform (math/transform.hpp:11-111, math/mat.hpp:561-573): row-major 4x4
// forward and inverse matrices; normal matrices are transposes of the other's
// upper 3x3. A matrix-vector product accumulates "res += m(i,j)*v[j]" starting
// from 0, in j order, with the homogeneous coordinate included.
// ---------------------------------------------------------------------------
struct Xform {
  float fwd[16];
  float inv[16];
};

YART_HD f3 mulPoint(const float* m, f3 v) {       // float3(m * float4(v, 1))
  return mk3((((0.0f + m[0] * v.x) + m[1] * v.y) + m[2] * v.z) + m[3] * 1.0f,
             (((0.0f + m[4] * v.x) + m[5] * v.y) + m[6] * v.z) + m[7] * 1.0f,
             (((0.0f + m[8] * v.x) + m[9] * v.y) + m[10] * v.z) + m[11] * 1.0f);
}
YART_HD f3 mulVector(const float* m, f3 v) {      // float3(m * float4(v, 0))
  return mk3((((0.0f + m[0] * v.x) + m[1] * v.y) + m[2] * v.z) + m[3] * 0.0f,
             (((0.0f + m[4] * v.x) + m[5] * v.y) + m[6] * v.z) + m[7] * 0.0f,
             (((0.0f + m[8] * v.x) + m[9] * v.y) + m[10] * v.z) + m[11] * 0.0f);
}
// normalized(transpose(float3x3(other)) * v): transform.hpp:50-58, 67-73
YART_HD f3 mulNormalT(const float* other, f3 v) {
  f3 r = mk3(((0.0f + other[0] * v.x) + other[4] * v.y) + other[8] * v.z,
             ((0.0f + other[1] * v.x) + other[5] * v.y) + other[9] * v.z,
             ((0.0f + other[2] * v.x) + other[6] * v.y) + other[10] * v.z);
  return normalized(r);
}
YART_HD f3 mul3x3(const float* m9, f3 v) {        // float3x3 * float3 (row-major 3x3)
  return mk3(((0.0f + m9[0] * v.x) + m9[1] * v.y) + m9[2] * v.z,
             ((0.0f + m9[3] * v.x) + m9[4] * v.y) + m9[5] * v.z,
             ((0.0f + m9[6] * v.x) + m9[7] * v.y) + m9[8] * v.z);
}

// ---------------------------------------------------------------------------
// Frame (math/frame.hpp:21-60)
// ---------------------------------------------------------------------------
struct Frame { f3 x, y, z; };

YART_HD Frame frameFromNormal(f3 n) {                       // frame.hpp:27-32
  Frame f; f.z = n;
  f3 a = fabsf(n.x) > 0.5f ? mk3(0, 1, 0) : mk3(1, 0, 0);
  f.y = normalized(cross(n, a));
  f.x = cross(n, f.y);
  return f;
}
YART_HD Frame frameFromNormalTangent(f3 n, f3 t, float handedness) {   // frame.hpp:34-50
  Frame f; f.z = n;
  if (absDot(t, n) > 0.9f) {
    f3 a = fabsf(n.x) > 0.5f ? mk3(0, 1, 0) : mk3(1, 0, 0);
    f.y = normalized(cross(n, a));
    f.x = cross(n, f.y);
  } else {
    f.y = normalized(cross(n, t)) * handedness;
    f.x = cross(f.y, f.z);
  }
  return f;
}
YART_HD f3 wtl(const Frame& f, f3 w) { return mk3(dot(w, f.x), dot(w, f.y), dot(w, f.z)); }
YART_HD f3 ltw(const Frame& f, f3 l) { return (l.x * f.x + l.y * f.y) + l.z * f.z; }

// ---------------------------------------------------------------------------
// math/math.hpp
// ---------------------------------------------------------------------------
YART_HD f3 reflect(f3 wo, f3 n) {                 // math.hpp:15-20: -wo + (n*2)*dot(wo,n)
  return (-wo) + (n * 2.0f) * dot(wo, n);
}
YART_HD bool refract(f3 wi, f3 n, float ior, f3& wt) {   // math.hpp:22-41
  float cosTheta = dot(wi, n);
  if (cosTheta < 0.0f) {
    ior = 1.0f / ior;
    cosTheta *= -1.0f;
    n = n * -1.0f;
  }
  float sin2Theta = 1.0f - cosTheta * cosTheta;
  float sin2Theta_t = sin2Theta / (ior * ior);
  if (sin2Theta_t >= 1.0f) return false;
  float cosTheta_t = sqrtf(1.0f - sin2Theta_t);
  wt = ((-wi) / ior) + (cosTheta / ior - cosTheta_t) * n;
  return true;
}
YART_HD float fresnelDielectric(float cosTheta, float ior) {   // math.hpp:43-61
  cosTheta = stdclamp(cosTheta, -1.0f, 1.0f);
  if (cosTheta < 0.0f) {
    ior = 1.0f / ior;
    cosTheta = -cosTheta;
  }
  float sin2Theta = 1.0f - cosTheta * cosTheta;
  float sin2Theta_t = sin2Theta / (ior * ior);
  if (sin2Theta_t >= 1.0f) return 1.0f;
  float cosTheta_t = sqrtf(1.0f - sin2Theta_t);
  float r_prl = (ior * cosTheta - cosTheta_t) / (ior * cosTheta + cosTheta_t);
  float r_per = (cosTheta - ior * cosTheta_t) / (cosTheta + ior * cosTheta_t);
  return (r_prl * r_prl + r_per * r_per) * 0.5f;
}
YART_HD f3 fresnelSchlick(f3 r, float cosTheta) {              // math.hpp:80-88
  const float k = 1.0f - cosTheta;
  const float k2 = k * k;
  return r + (mk3(1.0f) - r) * (k2 * k2 * k);
}
YART_HD uint32_t reverseBits32(uint32_t n) {                   // math.hpp:102-109
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(n);
#else
  n = (n << 16) | (n >> 16);
  n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
  n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
  n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
  n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
  return n;
#endif
}
YART_HD uint64_t leftShift2(uint64_t x) {                      // math.hpp:122-130
  x &= 0xffffffffull;
  x = (x ^ (x << 16)) & 0x0000ffff0000ffffull;
  x = (x ^ (x << 8)) & 0x00ff00ff00ff00ffull;
  x = (x ^ (x << 4)) & 0x0f0f0f0f0f0f0f0full;
  x = (x ^ (x << 2)) & 0x3333333333333333ull;
  x = (x ^ (x << 1)) & 0x5555555555555555ull;
  return x;
}
YART_HD uint64_t encodeMorton2(uint32_t x, uint32_t y) {       // math.hpp:132-134
  return (leftShift2(y) << 1) | leftShift2(x);
}
YART_HD float copysign1(float s) { return copysignf(1.0f, s); }
YART_HD f2 octahedralUV(f3 v) {                                // math.hpp:151-166
  f3 vAbs = mk3(fabsf(v.x), fabsf(v.y), fabsf(v.z));
  float s = ((0.0f + vAbs.x) + vAbs.y) + vAbs.z;
  v = v / s;
  // reference: "vAbs /= sum(vAbs)" evaluated after v was divided; sum(vAbs) is unchanged
  vAbs = vAbs / s;
  f2 res;
  if (v.y >= 0) {
    res = mk2(v.x, v.z);
  } else {
    res = mk2((1.0f - vAbs.z) * copysign1(v.x), (1.0f - vAbs.x) * copysign1(v.z));
  }
  return mk2((res.x + 1.0f) * 0.5f, (res.y + 1.0f) * 0.5f);
}
YART_HD f3 invOctahedralUV(f2 uv) {                            // math.hpp:168-179
  f3 res;
  res.x = 2.0f * uv.x - 1.0f;
  res.z = 2.0f * uv.y - 1.0f;
  res.y = 1.0f - (fabsf(res.x) + fabsf(res.z));
  if (res.y < 0.0f) {
    float xo = res.x;
    res.x = (1.0f - fabsf(res.z)) * copysign1(res.x);
    res.z = (1.0f - fabsf(xo)) * copysign1(res.z);
  }
  return normalized(res);
}

// ---------------------------------------------------------------------------
// Transcendentals. The reference calls libm's sinf / cosf / logf / expf (pixel
// jitter, disk / hemisphere / VNDF sampling, volume attenuation). On the host
// the library calls the same libm. On the device the same results are produced
// by evaluating glibc 2.35's algorithms (the ARM "optimized routines" float
// functions: a double-precision polynomial after a table / quadrant reduction,
// in the FMA-contracted form x86-64 glibc selects on FMA-capable CPUs), which
// were checked here against libm.so.6 over every float in [0,100] (sin, cos:
// 1.12e9 inputs, 0 mismatches), [1e-30,1e30] (log: 1.67e9, 0 mismatches) and
// |x| in [1e-10,80] (exp: 6.6e8, 2 mismatches). ocml's own sinf/cosf differ
// from glibc's in the last bit for roughly a quarter of the inputs, which is
// what made rare paths diverge; these do not.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
namespace libm_emul {
struct SinCosTab { double c0, c1, c2, c3, c4, s1, s2, s3; };
__device__ __forceinline__ uint32_t top12(float x) { return (__builtin_bit_cast(uint32_t, x) >> 20) & 0x7ffu; }
__device__ __forceinline__ float poly(double x, double x2, bool neg, int n) {
  // __sincosf_table[neg]: the second table negates the cosine coefficients
  const double sg = neg ? -1.0 : 1.0;
  if ((n & 1) == 0) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    double x3 = x * x2;
    double s1 = __builtin_fma(x2, s3c, s2c);
    double x7 = x3 * x2;
    double s = __builtin_fma(x3, s1c, x);
    return float(__builtin_fma(x7, s1, s));
  } else {
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
                 c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double cc2 = __builtin_fma(x2, c4, c3);
    double cc1 = __builtin_fma(x2, c1, c0);
    double x6 = x4 * x2;
    double c = __builtin_fma(x4, c2, cc1);
    return float(__builtin_fma(x6, cc2, c));
  }
}
__device__ __forceinline__ double reduceFast(double x, int& n) {
  double r = x * 0x1.45F306DC9C883p+23;
  n = (int32_t(r) + 0x800000) >> 24;
  return __builtin_fma(-double(n), 0x1.921FB54442D18p0, x);
}
__device__ __forceinline__ double quadSign(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; }
__device__ __forceinline__ float sinf_(float y) {
  double x = y;
  if (top12(y) < top12(0x1.921FB6p-1f)) {
    if (top12(y) < top12(0x1p-12f)) return y;
    return poly(x, x * x, false, 0);
  } else if (top12(y) < top12(120.0f)) {
    int n;
    x = reduceFast(x, n);
    return poly(x * quadSign(n), x * x, (n & 2) != 0, n);
  }
  return float(sin(double(y)));
}
__device__ __forceinline__ float cosf_(float y) {
  double x = y;
  if (top12(y) < top12(0x1.921FB6p-1f)) {
    if (top12(y) < top12(0x1p-12f)) return 1.0f;
    return poly(x, x * x, false, 1);
  } else if (top12(y) < top12(120.0f)) {
    int n;
    x = reduceFast(x, n);
    return poly(x * quadSign(n + 1), x * x, ((n + 1) & 2) != 0, n ^ 1);
  }
  return float(cos(double(y)));
}
__device__ __forceinline__ float logf_(float x) {
  const double T[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
  uint32_t ix = __builtin_bit_cast(uint32_t, x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return -kInf;
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
    ix = __builtin_bit_cast(uint32_t, x * 0x1p23f);
    ix -= 23u << 23;
  }
  uint32_t tmp = ix - 0x3f330000u;
  int i = int((tmp >> 19) % 16u);
  int k = int32_t(tmp) >> 23;
  uint32_t iz = ix - (tmp & 0xff800000u);
  double z = double(__builtin_bit_cast(float, iz));
  double r = __builtin_fma(z, T[i][0], -1.0);
  double y0 = __builtin_fma(double(k), 0x1.62e42fefa39efp-1, T[i][1]);
  double r2 = r * r;
  double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
  y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
  y = __builtin_fma(y, r2, y0 + r);
  return float(y);
}
__device__ __forceinline__ float expf_(float x) {
  const uint64_t T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  if (!(fabsf(x) < 80.0f)) return float(exp(double(x)));        // overflow / underflow / NaN tails
  const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32,
               C2 = 0x1.62e42ff0c52d6p-1 / 32;
  double z = InvLn2N * double(x);
  double kd = z + SHIFT;
  uint64_t ki = __builtin_bit_cast(uint64_t, kd);
  kd -= SHIFT;
  double r = z - kd;
  uint64_t t = T[ki % 32] + (ki << 47);
  double s = __builtin_bit_cast(double, t);
  z = __builtin_fma(C0, r, C1);
  double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  return float(y * s);
}
}  // namespace libm_emul
// Arguments known to lie in [0, 2 pi] (2 pi u with a sampler value u in [0, 1)): the same two branches without the
// |x| >= 120 tail, whose double-precision sin / cos (Payne-Hanek reduction, ~700 fp64 instructions inlined per call site)
// can never run for them. NOT valid outside [0, 120).
__device__ __forceinline__ float ysinf2pi(float y) {
  using namespace libm_emul;
  double x = y;
  if (top12(y) < top12(0x1.921FB6p-1f)) {
    if (top12(y) < top12(0x1p-12f)) return y;
    return poly(x, x * x, false, 0);
  }
  int n;
  x = reduceFast(x, n);
  return poly(x * quadSign(n), x * x, (n & 2) != 0, n);
}
__device__ __forceinline__ float ycosf2pi(float y) {
  using namespace libm_emul;
  double x = y;
  if (top12(y) < top12(0x1.921FB6p-1f)) {
    if (top12(y) < top12(0x1p-12f)) return 1.0f;
    return poly(x, x * x, false, 1);
  }
  int n;
  x = reduceFast(x, n);
  return poly(x * quadSign(n + 1), x * x, ((n + 1) & 2) != 0, n ^ 1);
}
__device__ __forceinline__ float ysinf(float x) { return libm_emul::sinf_(x); }
__device__ __forceinline__ float ycosf(float x) { return libm_emul::cosf_(x); }
__device__ __forceinline__ float ylogf(float x) { return libm_emul::logf_(x); }
__device__ __forceinline__ float yexpf(float x) { return libm_emul::expf_(x); }
#else
inline float ysinf2pi(float x) { return sinf(x); }
inline float ycosf2pi(float x) { return cosf(x); }
inline float ysinf(float x) { return sinf(x); }
inline float ycosf(float x) { return cosf(x); }
inline float ylogf(float x) { return logf(x); }
inline float yexpf(float x) { return expf(x); }
#endif

// Float -> size_t conversion as x86-64/clang performs it for the reference's
// "size_t(x)" on possibly negative x (UB in C++, but the goldens pin this
// outcome): trunc toward zero as a signed 64-bit value,
// reinterpreted as unsigned. Returns min(that, cap).
YART_HD uint32_t sizeTClamp(float x, uint32_t cap) {
  if (!(x > -1.0f)) return cap;       // x <= -1 or NaN -> huge unsigned -> clamps to cap
  if (x < 0.0f) return 0;             // (-1, 0) truncates to 0
  if (x >= float(cap)) return cap;
  return uint32_t(x);
}

}  // namespace yart_hip
