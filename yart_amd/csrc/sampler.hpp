// sampler.hpp — ZSobol sampler with FastOwen scrambling (integer path; bit-exact).
//
// Restates reference core/sampler.hpp:72-174 (SobolSampler), core/scrambler.hpp:53-69
// (FastOwenScrambler), core/rng.hpp:25-100 (murmurHash64A on a 4-byte key, mixBits)
// and math/math.hpp:102-134 (bit reversal, Morton code) as a register-resident
// state machine: {dim, mortonIndex} per path, {log2spp, nBase4Digits} per render.
#pragma once
#include "ymath.hpp"

namespace yart_hip {

// Per-render tables that hoist the sample-independent part of the ZSobol index permutation
// out of the per-sample work (sampler.hpp:155-173: the permutation of a base-4 digit depends
// on the dimension and on the HIGHER digits only, so the digits that lie in the pixel bits of
// the Morton index — and the permutation rows of the first two sample digits — are the same
// for all samples of a pixel). One 64-bit entry per (dimension, pixel of this rank's list):
//   bits  0..23  permuted pixel digits, as (index >> log2spp)
//   bits 24..31  permutation row of the top sample digit       (higher digits = pixel)
//   bits 32..63  rows of the second sample digit, one byte per value of the (unpermuted) top one
// plus hashDim(d) for d <= dims, and byte-wise XOR tables of the Sobol' dimension-1 matrix.
// A draw then costs 2 table reads + the remaining (spp-dependent) low digits instead of
// nBase4Digits 64-bit hash evaluations. Dimensions >= dims fall back to the direct evaluation.
// (Tables for the low digits as well — 80 B instead of 8 B per dimension and pixel — were measured in round 2 and
// lose: the gathers cost more than the hashes, profiles/r2_sampler_rows_kernel_stats.txt; removed.)
struct SamplerTables {
  const uint64_t* entries = nullptr;   // [stride = pixels][dims]: the draws of one shading step (dimensions d .. d + 7) of a path lie in one or two sectors
  const uint64_t* hash = nullptr;      // [dims + 3]
  const uint32_t* sobol1 = nullptr;    // [8][256]: XOR of the matrix columns selected by byte b of the index
  uint32_t dims = 0, stride = 0;
};

struct SamplerConfig {
  uint32_t log2spp;       // log2Int(float(spp)), math_base.hpp:156-160
  uint32_t nBase4Digits;  // log2Int(roundUpPow2(tile)) + (log2spp+1)/2, sampler.hpp:74-82
  SamplerTables tab;      // optional (wavefront pipeline)
};

// math_base.hpp:156-160 — rounds to nearest in log space (48 -> 6)
inline int32_t log2IntHost(float v) {
  if (v < 1) return -log2IntHost(1 / v);
  uint32_t b; std::memcpy(&b, &v, 4);
  const uint32_t midsignif = 0x3504f3u;
  return int32_t(b >> 23) - 127 + (((b & 0x7fffffu) >= midsignif) ? 1 : 0);
}
inline int32_t roundUpPow2Host(int32_t v) {   // math_base.hpp:162-170
  v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
  return v + 1;
}
inline SamplerConfig makeSamplerConfig(uint32_t spp, uint32_t tileSize) {
  SamplerConfig c;
  c.log2spp = uint32_t(log2IntHost(float(spp)));
  // maxComponent(uint2{tile,tile}) starts from numeric_limits<unsigned>::min() = 0
  uint32_t res = uint32_t(roundUpPow2Host(int32_t(tileSize)));
  uint32_t log4spp = (c.log2spp + 1) / 2;
  c.nBase4Digits = uint32_t(log2IntHost(float(res))) + log4spp;
  c.tab = SamplerTables{};
  return c;
}

YART_HD uint64_t mixBits(uint64_t v) {     // rng.hpp:93-100
  v ^= (v >> 31);
  v *= 0x7fb5d329728ea185ull;
  v ^= (v >> 27);
  v *= 0x81dadef4bc2dd44dull;
  v ^= (v >> 33);
  return v;
}

// hash(uint32_t) = murmurHash64A over the 4 key bytes, seed 0 (rng.hpp:25-91):
// len=4 -> no 8-byte blocks; tail switch cases 4..1 xor the little-endian value.
YART_HD uint64_t hashDim(uint32_t dim) {
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  uint64_t h = 0ull ^ (4ull * m);
  h ^= uint64_t(dim);
  h *= m;
  h ^= h >> 47;
  h *= m;
  h ^= h >> 47;
  return h;
}

YART_HD uint32_t fastOwen(uint32_t v, uint32_t seed) {   // scrambler.hpp:57-65
  v = reverseBits32(v);
  v ^= v * 0x3d20adeau;
  v += seed;
  v *= (seed >> 16) | 1u;
  v ^= v * 0x05526c56u;
  v ^= v * 0x53a22864u;
  return reverseBits32(v);
}

// Generator matrix of Sobol' dimension 1 as the reference's table holds it
// (sobol.tables entries 52..103): column k<32 is row k of Pascal's triangle mod 2,
// v[k] = v[k-1] ^ (v[k-1] >> 1), v[0] = 1<<31; the 52-column table repeats the
// first 20 columns for k = 32..51. Checked against the reference table in
// tests/test_oracle_kat.py (dump: oracle/_ref/yart_ref luts).
YART_HD uint32_t sobolDim1Column(uint32_t k) {
  // closed form: bit (31-j) of column k is C(k, j) mod 2 = ((k & j) == j) (Lucas)
  k &= 31u;   // columns 32..51 repeat 0..19 in the reference table
  uint32_t v = 0;
  for (uint32_t j = 0; j <= k; j++) v |= (((k & j) == j) ? 1u : 0u) << (31u - j);
  return v;
}

struct Sampler {
  uint64_t morton;
  uint32_t dim;
  uint32_t pix = 0;          // SamplerTables column of this path's pixel
};

// permutations[24][4] of sampler.hpp:115-140, 2 bits per entry, one byte per row,
// eight rows per 64-bit word (register-resident: no table memory on the GPU)
YART_HD uint32_t permutationRow(uint32_t p) {
  // row bytes: E4 B4 D8 78 6C 9C E1 B1 | C9 39 2D 8D C6 36 D2 72 | 4E 1E 27 87 1B 4B 63 93
  const uint64_t t0 = 0xB1E19C6C78D8B4E4ull, t1 = 0x72D236C68D2D39C9ull, t2 = 0x93634B1B87271E4Eull;
  const uint32_t sel = p >> 3;
  const uint64_t t = sel == 0 ? t0 : (sel == 1 ? t1 : t2);
  return uint32_t(t >> ((p & 7u) * 8u)) & 0xffu;   // digit d -> (row >> (2*d)) & 3
}

YART_HD void startPixelSample(Sampler& s, const SamplerConfig& c, uint32_t px, uint32_t py,
                              uint32_t sample) {         // sampler.hpp:84-87
  s.dim = 0;
  s.morton = (encodeMorton2(px, py) << c.log2spp) | uint64_t(sample);
}

// v % 24 for a 40-bit v (mixBits(...) >> 24) without a 64-bit magic-number multiply (four quarter-rate 32-bit multiplies on
// CDNA): v = 8 q + (v & 7) and 24 = 8 * 3, so v % 24 = 8 (q % 3) + (v & 7); 2^24 = 1 (mod 3) folds the 37-bit q to 25 bits,
// whose remainder by 3 is one 32-bit multiply-high. Checked against % 24 by hostsim selftest.
YART_HD uint32_t mod24of40(uint64_t v) {
  const uint64_t q = v >> 3;
  const uint32_t a = uint32_t(q & 0xffffffull) + uint32_t(q >> 24);       // < 2^24 + 2^13, same residue mod 3
  const uint32_t t = uint32_t((uint64_t(a) * 0xAAAAAAABull) >> 33);      // a / 3
  return 8u * (a - 3u * t) + (uint32_t(v) & 7u);
}
// one digit of the permutation: row of permutations[][] for the digit whose higher digits are `higher`
YART_HD uint32_t permutationRowFor(uint64_t higher, uint64_t dimMix) {
  return permutationRow(mod24of40(mixBits(higher ^ dimMix) >> 24));
}

YART_HD uint64_t getSampleIndexDirect(const Sampler& s, const SamplerConfig& c) {   // sampler.hpp:155-173
  uint64_t index = 0;
  const bool pow2Samples = c.log2spp & 1u;
  const int lastDigit = pow2Samples ? 1 : 0;
  const uint64_t dimMix = uint64_t(0x55555555u * s.dim);      // 32-bit wrapping multiply
  for (int i = int(c.nBase4Digits) - 1; i >= lastDigit; i--) {
    uint32_t digitShift = uint32_t(2 * i - lastDigit);
    uint32_t digit = uint32_t(s.morton >> digitShift) & 3u;
    uint64_t higherDigits = s.morton >> (digitShift + 2);
    digit = (permutationRowFor(higherDigits, dimMix) >> (2u * digit)) & 3u;
    index |= uint64_t(digit) << digitShift;
  }
  if (pow2Samples) {
    uint32_t digit = uint32_t(s.morton & 1ull);
    index |= uint64_t(digit ^ uint32_t(mixBits((s.morton >> 1) ^ dimMix) & 1ull));
  }
  return index;
}

// Digit bookkeeping shared by the table builder and the table reader. Digits i >= firstPixelDigit
// lie entirely in the pixel bits (a digit never straddles: digitShift has the parity of log2spp).
YART_HD int samplerFirstPixelDigit(const SamplerConfig& c) { return int((c.log2spp + 1u) / 2u); }

YART_HD uint64_t samplerTableEntry(const SamplerConfig& c, uint64_t pixelMorton, uint32_t dim) {
  const int lastDigit = int(c.log2spp & 1u);
  const uint64_t dimMix = uint64_t(0x55555555u * dim);
  const uint64_t morton = pixelMorton << c.log2spp;
  const int firstPix = samplerFirstPixelDigit(c);
  uint64_t index = 0;
  for (int i = int(c.nBase4Digits) - 1; i >= firstPix; i--) {
    const uint32_t digitShift = uint32_t(2 * i - lastDigit);
    const uint32_t digit = uint32_t(morton >> digitShift) & 3u;
    index |= uint64_t((permutationRowFor(morton >> (digitShift + 2), dimMix) >> (2u * digit)) & 3u) << digitShift;
  }
  uint64_t e = (index >> c.log2spp) & 0xffffffull;
  e |= uint64_t(permutationRowFor(pixelMorton, dimMix)) << 24;
  for (uint32_t k = 0; k < 4; k++) e |= uint64_t(permutationRowFor((pixelMorton << 2) | k, dimMix)) << (32 + 8 * k);
  return e;
}

YART_HD uint64_t getSampleIndex(const Sampler& s, const SamplerConfig& c) {
  if (c.tab.entries == nullptr || s.dim >= c.tab.dims) return getSampleIndexDirect(s, c);
  const uint64_t e = c.tab.entries[size_t(s.pix) * c.tab.dims + s.dim];
  const int lastDigit = int(c.log2spp & 1u);
  const uint64_t dimMix = uint64_t(0x55555555u * s.dim);
  uint64_t index = (e & 0xffffffull) << c.log2spp;
  int i = samplerFirstPixelDigit(c) - 1;                        // top sample digit
  if (i >= lastDigit) {
    const uint32_t shift = uint32_t(2 * i - lastDigit);
    const uint32_t top = uint32_t(s.morton >> shift) & 3u;
    index |= uint64_t((uint32_t(e >> 24) >> (2u * top)) & 3u) << shift;
    if (--i >= lastDigit) {
      const uint32_t shift2 = shift - 2u;
      const uint32_t digit = uint32_t(s.morton >> shift2) & 3u;
      index |= uint64_t((uint32_t(e >> (32u + 8u * top)) >> (2u * digit)) & 3u) << shift2;
      for (--i; i >= lastDigit; i--) {                          // remaining (spp-dependent) digits: hashed per draw
        const uint32_t digitShift = uint32_t(2 * i - lastDigit);
        const uint32_t dg = uint32_t(s.morton >> digitShift) & 3u;
        const uint32_t row = permutationRowFor(s.morton >> (digitShift + 2), dimMix);
        index |= uint64_t((row >> (2u * dg)) & 3u) << digitShift;
      }
    }
  }
  if (lastDigit) {
    const uint32_t digit = uint32_t(s.morton & 1ull);
    const uint32_t mix = uint32_t(mixBits((s.morton >> 1) ^ dimMix) & 1ull);
    index |= uint64_t(digit ^ mix);
  }
  return index;
}

YART_HD float sobolToFloat(uint32_t v) {                  // sampler.hpp:152
  return stdmin(float(v) * 0x1p-32f, kOneMinusEpsilon);
}

YART_HD float sobolDim0(uint64_t idx, uint32_t seed) {    // sampler.hpp:144-145
  return sobolToFloat(fastOwen(reverseBits32(uint32_t(idx)), seed));
}
YART_HD float sobolDim1(uint64_t idx, uint32_t seed, const uint32_t* __restrict__ matrix52) {
  uint32_t v = 0;
  uint64_t d = idx;
  for (uint32_t i = 0; d != 0; d >>= 1, i++) v ^= uint32_t(d & 1ull) * matrix52[i];   // :147-149
  return sobolToFloat(fastOwen(v, seed));
}

YART_HD uint64_t samplerHash(const SamplerConfig& c, uint32_t dim) {
  return (c.tab.hash != nullptr && dim < c.tab.dims + 3u) ? c.tab.hash[dim] : hashDim(dim);
}
YART_HD float sobolDim1Tab(uint64_t idx, uint32_t seed, const uint32_t* __restrict__ byteTab) {
  uint32_t v = 0;
  for (uint32_t b = 0; idx != 0; idx >>= 8, b++) v ^= byteTab[b * 256u + uint32_t(idx & 0xffull)];
  return sobolToFloat(fastOwen(v, seed));
}

YART_HD float get1D(Sampler& s, const SamplerConfig& c) {           // sampler.hpp:89-94
  uint64_t idx = getSampleIndex(s, c);
  s.dim++;
  uint32_t h = uint32_t(samplerHash(c, s.dim));
  return sobolDim0(idx, h);
}
YART_HD f2 get2D(Sampler& s, const SamplerConfig& c, const uint32_t* __restrict__ matrix52) {   // :96-107
  uint64_t idx = getSampleIndex(s, c);
  s.dim += 2;
  uint64_t hb = samplerHash(c, s.dim);
  const float y = c.tab.sobol1 != nullptr ? sobolDim1Tab(idx, uint32_t(hb >> 32), c.tab.sobol1)
                                          : sobolDim1(idx, uint32_t(hb >> 32), matrix52);
  return mk2(sobolDim0(idx, uint32_t(hb)), y);
}

}  // namespace yart_hip
