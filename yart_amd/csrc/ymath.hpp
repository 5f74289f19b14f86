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
