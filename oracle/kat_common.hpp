// kat_common.hpp — inputs and JSON writer of the known-answer-test "program".
// TEST INFRASTRUCTURE shared by oracle/ref_driver.cpp (compiled reference) and
// oracle/oracle_main.cpp (CPU restatement): both evaluate the same inputs, and
// tests/ compares the two JSON files section by section (floats are written as
// their IEEE-754 bit patterns so that equality is exact). tests/kat_inputs.py
// restates the same inputs for the GPU probes.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace kat {

// Numerical-Recipes LCG; float in [0,1) from the top 24 bits.
struct Lcg {
  uint32_t x;
  explicit Lcg(uint32_t seed) : x(seed) {}
  float next() { x = x * 1664525u + 1013904223u; return float(x >> 8) * (1.0f / 16777216.0f); }
  float sym() { return next() * 2.0f - 1.0f; }
};

inline uint64_t fnv1a(const void* data, size_t bytes) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

inline std::vector<uint64_t> mixInputs() {
  return {0ull, 1ull, 2ull, 0x55555555ull, 0xdeadbeefcafef00dull, 0xffffffffffffffffull,
          0x123456789abcdefull, 1ull << 40};
}
inline std::vector<std::pair<uint32_t, uint32_t>> mortonInputs() {
  return {{0, 0}, {1, 0}, {0, 1}, {3, 5}, {1000, 700}, {1919, 1079}, {65535, 65535}, {3839, 2159}};
}
inline std::vector<float> log2Inputs() {
  return {1.f, 2.f, 3.f, 16.f, 48.f, 64.f, 256.f, 1000.f, 1024.f, 0.25f, 5.f, 6.f, 11.f, 12.f};
}

struct SamplerCase { uint32_t spp, tile, px, py, sample; };
inline std::vector<SamplerCase> samplerCases() {
  return {{16, 64, 3, 5, 7},      {16, 64, 0, 0, 0},        {16, 64, 255, 255, 15},
          {64, 64, 100, 37, 63},  {256, 64, 1000, 700, 200}, {256, 64, 1919, 1079, 255},
          {1024, 64, 640, 360, 1023}, {512, 64, 3839, 2159, 300}, {8, 64, 17, 9, 5},
          {32, 64, 77, 200, 31},  {48, 64, 5, 6, 40},        {16, 32, 40, 41, 3}};
}
// 2 = get2D, 1 = get1D — the mix a path consumes (camera 2+2, bounce 2+1+1, NEE 1+2, RR 1)
inline std::vector<int> samplerPattern() {
  return {2, 2, 2, 1, 1, 1, 2, 1, 2, 1, 1, 1, 2, 1, 1, 2};
}

struct LutInput { float c, r, f0, ior; };
inline std::vector<LutInput> lutInputs() {
  std::vector<LutInput> v;
  const float cs[] = {-1.0f, -0.73f, -0.4f, -0.01f, 0.0f, 0.013f, 0.25f, 0.5f, 0.77f, 0.999f, 1.0f};
  const float rs[] = {0.0f, 0.03f, 0.2f, 0.5f, 0.81f, 1.0f};
  const float fs[] = {0.0f, 0.04f, 0.35f, 1.0f};
  const float is[] = {1.5f, 1.0f / 1.5f, 1.33f, 2.4f};
  for (float c : cs) for (float r : rs) for (int k = 0; k < 4; k++) v.push_back({c, r, fs[k], is[k]});
  return v;
}

constexpr int bsdfCasesPerMaterial = 48;

inline std::vector<size_t> lightSubset(size_t n) {
  std::vector<size_t> v;
  for (size_t i = 0; i < n; i++)
    if (i < 6 || i + 3 >= n || (i % 997) == 0) v.push_back(i);
  return v;
}

class Writer {
 public:
  explicit Writer(const std::string& path) : f_(std::fopen(path.c_str(), "w")) {
    if (!f_) throw std::runtime_error("kat: cannot write " + path);
    std::fputs("{\n", f_);
  }
  ~Writer() { if (f_) close(); }
  void u64(const char* name, const std::vector<uint64_t>& v) {
    head(name);
    for (size_t i = 0; i < v.size(); i++) std::fprintf(f_, "%s%llu", i ? "," : "", (unsigned long long) v[i]);
    std::fputs("]", f_);
  }
  void i64(const char* name, const std::vector<int64_t>& v) {
    head(name);
    for (size_t i = 0; i < v.size(); i++) std::fprintf(f_, "%s%lld", i ? "," : "", (long long) v[i]);
    std::fputs("]", f_);
  }
  // floats as uint32 bit patterns
  void f32(const char* name, const std::vector<float>& v) {
    head(name);
    for (size_t i = 0; i < v.size(); i++) {
      uint32_t b; std::memcpy(&b, &v[i], 4);
      std::fprintf(f_, "%s%u", i ? "," : "", b);
    }
    std::fputs("]", f_);
  }
  void close() { std::fputs("\n}\n", f_); std::fclose(f_); f_ = nullptr; }

 private:
  void head(const char* name) { std::fprintf(f_, "%s\"%s\": [", first_ ? "" : ",\n", name); first_ = false; }
  FILE* f_;
  bool first_ = true;
};

}  // namespace kat
