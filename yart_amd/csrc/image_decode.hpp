// image_decode.hpp — PNG and Radiance .hdr decoders for the glTF importer. Host only.
//
// The reference decodes embedded images with stb_image asking for 4 channels
// (core/texture.hpp:62-70 `stbi_load_from_memory(.., 4)`, core/texture.cpp:5-20 `stbi_loadf(.., 4)`).
// These decoders are written from the PNG (ISO/IEC 15948) and Radiance RGBE format descriptions
// and reproduce the conventions of that call which reach the pixels: 16-bit samples keep their
// high byte, 1/2/4-bit grey is scaled to 0..255, grey / RGB gain alpha 255, a tRNS colour key
// gives alpha 0, RGBE mantissas are scaled by 2^(e-136) with alpha 1. PNG is lossless, so the
// bytes equal the reference's (pinned by tests/golden/gltf/*.tex against oracle/_ref `texture`).
// Baseline and progressive Huffman JPEG are decoded with the integer arithmetic stb_image documents for the
// parts the standard leaves open (see decodeJpeg); arithmetic-coded / CMYK files are refused with an error.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace yart_hip {
namespace image {

struct Image8 { uint32_t width = 0, height = 0; std::vector<uint8_t> rgba; };      // 4 bytes per pixel
struct ImageF { uint32_t width = 0, height = 0; std::vector<float> rgba; };        // 4 floats per pixel

inline bool isPng(const uint8_t* d, size_t n) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  return n >= 8 && std::memcmp(d, sig, 8) == 0;
}
inline bool isJpeg(const uint8_t* d, size_t n) { return n >= 3 && d[0] == 0xFF && d[1] == 0xD8 && d[2] == 0xFF; }

namespace detail {
inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}
// reverses the scanline filters of one (sub-)image in place; `raw` holds h rows of 1 + rowBytes
inline void unfilter(uint8_t* raw, uint32_t h, size_t rowBytes, size_t bpp) {
  std::vector<uint8_t> zero(rowBytes, 0);
  const uint8_t* prior = zero.data();
  for (uint32_t y = 0; y < h; y++) {
    uint8_t* row = raw + size_t(y) * (rowBytes + 1);
    const uint8_t ft = row[0];
    uint8_t* cur = row + 1;
    for (size_t i = 0; i < rowBytes; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prior[i], c = i >= bpp ? prior[i - bpp] : 0;
      int v = cur[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: throw std::runtime_error("png: unknown scanline filter");
      }
      cur[i] = uint8_t(v);
    }
    prior = cur;
  }
}
}  // namespace detail

inline Image8 decodePng(const uint8_t* data, size_t len) {
  using namespace detail;
  if (!isPng(data, len)) throw std::runtime_error("png: bad signature");
  size_t pos = 8;
  uint32_t w = 0, h = 0, depth = 0, color = 0, interlace = 0;
  bool haveHdr = false, haveKey = false;
  uint8_t palette[256][4];
  for (auto& e : palette) { e[0] = e[1] = e[2] = 0; e[3] = 255; }
  uint16_t key[3] = {0, 0, 0};
  std::vector<uint8_t> idat;
  for (;;) {
    if (pos + 8 > len) throw std::runtime_error("png: truncated chunk header");
    const uint32_t clen = be32(data + pos);
    const uint8_t* type = data + pos + 4;
    const uint8_t* body = data + pos + 8;
    if (size_t(clen) + 12 > len - pos) throw std::runtime_error("png: truncated chunk");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (clen != 13) throw std::runtime_error("png: bad IHDR");
      w = be32(body); h = be32(body + 4); depth = body[8]; color = body[9]; interlace = body[12];
      if (body[10] != 0 || body[11] != 0 || interlace > 1) throw std::runtime_error("png: unsupported compression / filter / interlace method");
      if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) throw std::runtime_error("png: bad dimensions");
      const bool okDepth = (color == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                           (color == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                           ((color == 2 || color == 4 || color == 6) && (depth == 8 || depth == 16));
      if (!okDepth) throw std::runtime_error("png: bad colour type / bit depth");
      haveHdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      if (clen % 3 != 0 || clen > 768) throw std::runtime_error("png: bad PLTE");
      for (uint32_t i = 0; i < clen / 3; i++) { palette[i][0] = body[3 * i]; palette[i][1] = body[3 * i + 1]; palette[i][2] = body[3 * i + 2]; }
    } else if (!std::memcmp(type, "tRNS", 4)) {
      if (!haveHdr) throw std::runtime_error("png: tRNS before IHDR");
      if (color == 3) { for (uint32_t i = 0; i < clen && i < 256; i++) palette[i][3] = body[i]; }
      else if (color == 0 && clen >= 2) { key[0] = uint16_t((body[0] << 8) | body[1]); haveKey = true; }
      else if (color == 2 && clen >= 6) { for (int k = 0; k < 3; k++) key[k] = uint16_t((body[2 * k] << 8) | body[2 * k + 1]); haveKey = true; }
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + clen);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += size_t(clen) + 12;
  }
  if (!haveHdr || idat.empty()) throw std::runtime_error("png: missing IHDR / IDAT");

  const uint32_t channels = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4;
  const size_t bitsPerPixel = size_t(channels) * depth;
  const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
  struct Pass { uint32_t x0, y0, dx, dy, w, h; };
  std::vector<Pass> passes;
  if (!interlace) passes.push_back({0, 0, 1, 1, w, h});
  else {
    static const uint32_t xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1};
    static const uint32_t xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
    for (int p = 0; p < 7; p++) {
      const uint32_t pw = (w + xs[p] - 1 - xo[p]) / xs[p], ph = (h + ys[p] - 1 - yo[p]) / ys[p];
      if (w > xo[p] && h > yo[p] && pw && ph) passes.push_back({xo[p], yo[p], xs[p], ys[p], pw, ph});
    }
  }
  size_t rawSize = 0;
  for (const Pass& p : passes) rawSize += (size_t(p.w) * bitsPerPixel + 7) / 8 * p.h + p.h;
  std::vector<uint8_t> raw(rawSize);
  {
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) throw std::runtime_error("png: zlib init failed");
    zs.next_in = idat.data(); zs.avail_in = uInt(idat.size());
    zs.next_out = raw.data(); zs.avail_out = uInt(raw.size());
    const int rc = inflate(&zs, Z_FINISH);
    const size_t got = raw.size() - zs.avail_out;
    inflateEnd(&zs);
    if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got != raw.size()) throw std::runtime_error("png: corrupt or short image data");
  }

  Image8 img;
  img.width = w; img.height = h;
  img.rgba.assign(size_t(w) * h * 4, 0);
  const uint32_t scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;      // grey samples below 8 bits
  size_t off = 0;
  for (const Pass& p : passes) {
    const size_t rowBytes = (size_t(p.w) * bitsPerPixel + 7) / 8;
    uint8_t* sub = raw.data() + off;
    unfilter(sub, p.h, rowBytes, bpp);
    for (uint32_t y = 0; y < p.h; y++) {
      const uint8_t* row = sub + size_t(y) * (rowBytes + 1) + 1;
      for (uint32_t x = 0; x < p.w; x++) {
        uint16_t s[4] = {0, 0, 0, 0};                      // samples at their native depth
        if (depth == 16) for (uint32_t c = 0; c < channels; c++) s[c] = uint16_t((row[(size_t(x) * channels + c) * 2] << 8) | row[(size_t(x) * channels + c) * 2 + 1]);
        else if (depth == 8) for (uint32_t c = 0; c < channels; c++) s[c] = row[size_t(x) * channels + c];
        else { const size_t bit = size_t(x) * depth; s[0] = uint16_t((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u)); }
        auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? uint8_t(v >> 8) : uint8_t(v); };
        uint8_t* out = &img.rgba[(size_t(p.y0 + y * p.dy) * w + (p.x0 + x * p.dx)) * 4];
        switch (color) {
          case 0: {
            const uint8_t g = depth < 8 ? uint8_t(s[0] * scale) : to8(s[0]);
            // the key of a sub-byte image is compared after scaling (low byte of the tRNS entry)
            const bool keyed = haveKey && (depth == 16 ? s[0] == key[0] : g == uint8_t((key[0] & 255u) * scale));
            out[0] = out[1] = out[2] = g; out[3] = keyed ? 0 : 255;
            break;
          }
          case 2: {
            const bool keyed = haveKey && (depth == 16 ? (s[0] == key[0] && s[1] == key[1] && s[2] == key[2])
                                                       : (s[0] == (key[0] & 255u) && s[1] == (key[1] & 255u) && s[2] == (key[2] & 255u)));
            out[0] = to8(s[0]); out[1] = to8(s[1]); out[2] = to8(s[2]); out[3] = keyed ? 0 : 255;
            break;
          }
          case 3: std::memcpy(out, palette[s[0] & 255u], 4); break;
          case 4: out[0] = out[1] = out[2] = to8(s[0]); out[3] = to8(s[1]); break;
          default: out[0] = to8(s[0]); out[1] = to8(s[1]); out[2] = to8(s[2]); out[3] = to8(s[3]); break;
        }
      }
    }
    off += (rowBytes + 1) * p.h;
  }
  return img;
}

// ---- JPEG (ITU-T T.81 baseline / extended sequential Huffman, 8-bit, 1 or 3 components) -----------------------
// The entropy decoding is the standard's; what is implementation defined — the inverse DCT, the chroma
// upsampling filters and the YCbCr -> RGB arithmetic — follows stb_image's published integer arithmetic, because
// the reference's texture bytes are whatever that decoder yields (core/texture.hpp:69):
//   * IDCT: the IJG "islow" factorisation with 12-bit constants; column pass keeps 2 extra bits
//     ((x + 512) >> 10), row pass removes 17 with the +128 level shift folded in, clamped to 0..255;
//   * upsampling, chosen per component from (h_max / h, v_max / v): 1x1 none; 1x2 (3 near + far + 2) >> 2;
//     2x1 the 3:1 horizontal filter with replicated ends; 2x2 the separable 3:1 filter ((3 t0 + t1 + 8) >> 4 on
//     t = 3 near + far); anything else sample replication; "near" alternates between the two source rows;
//   * colour: 20-bit fixed point, the Cb term of green masked to its high 16 bits; a 3-component frame is taken
//     as RGB when its component ids are 'R','G','B' or when an Adobe APP14 marker says transform 0 and there
//     is no JFIF marker.
// Progressive frames (SOF2: spectral selection and successive approximation, T.81 Annex G) accumulate their
// coefficients over the scans and are dequantised (16-bit wrap-around as in stb_image) and transformed at the end.
// Pinned against the reference's loader on files from two independent encoders (tests/golden/gltf/jpg_*).
// Arithmetic-coded, lossless, hierarchical, 12-bit and 4-component (CMYK / YCCK) files are refused.
namespace detail {
struct JpegHuff {
  uint8_t bits[17] = {0};          // number of codes of each length 1..16
  uint8_t vals[256] = {0};
  int32_t maxcode[18] = {0};       // largest code of length l, left-aligned compare value + 1 (or -1)
  int32_t valptr[17] = {0};        // index of the first value of length l
  int32_t mincode[17] = {0};
  bool defined = false;
  void build() {
    int32_t code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k; mincode[l] = code;
      code += bits[l]; k += bits[l];
      if (code > (1 << l)) throw std::runtime_error("jpeg: over-subscribed Huffman table");
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    defined = true;
  }
};
struct JpegBits {
  const uint8_t* p; const uint8_t* end;
  uint32_t acc = 0; int n = 0;
  int marker = -1;                 // marker met inside the entropy-coded segment (then zeros are fed)
  void fill() {
    while (n <= 24) {
      uint32_t b = 0;
      if (marker < 0 && p < end) {
        b = *p++;
        if (b == 0xFF) {
          uint8_t c = p < end ? *p++ : 0xD9;
          while (c == 0xFF && p < end) c = *p++;
          if (c != 0) { marker = c; b = 0; }
        }
      }
      acc |= b << (24 - n);
      n += 8;
    }
  }
  int bit() { if (n < 1) fill(); const int b = int(acc >> 31); acc <<= 1; n--; return b; }
  int get(int s) {                 // s <= 16
    if (s == 0) return 0;
    if (n < s) fill();
    const int v = int(acc >> (32 - s)); acc <<= s; n -= s; return v;
  }
  void reset() { acc = 0; n = 0; marker = -1; }
};
inline int jpegDecodeSymbol(JpegBits& br, const JpegHuff& h) {
  int32_t code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + (code - h.mincode[l])];
  }
  throw std::runtime_error("jpeg: bad Huffman code");
}
inline int jpegExtend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
inline uint8_t clamp255(int x) { return uint8_t(x < 0 ? 0 : x > 255 ? 255 : x); }
inline int f2f(double x) { return int(x * 4096 + 0.5); }
// one 1-D pass of the "islow" inverse DCT on 8 values; results in x0..x3 / t0..t3 as the two passes combine them
struct Idct1D { int x0, x1, x2, x3, t0, t1, t2, t3; };
inline Idct1D idct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
  Idct1D r;
  int p2 = s2, p3 = s6;
  int p1 = (p2 + p3) * f2f(0.5411961f);
  int t2 = p1 + p3 * f2f(-1.847759065f);
  int t3 = p1 + p2 * f2f(0.765366865f);
  p2 = s0; p3 = s4;
  int t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
  r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
  t0 = s7; t1 = s5; t2 = s3; t3 = s1;
  p3 = t0 + t2;
  int p4 = t1 + t3;
  p1 = t0 + t3; p2 = t1 + t2;
  const int p5 = (p3 + p4) * f2f(1.175875602f);
  t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f); t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f);
  p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f);
  p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);
  r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
  return r;
}
inline void jpegIdct(uint8_t* out, size_t stride, const int16_t d[64]) {
  int v[64];
  for (int i = 0; i < 8; i++) {
    const Idct1D r = idct1d(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
    const int x0 = r.x0 + 512, x1 = r.x1 + 512, x2 = r.x2 + 512, x3 = r.x3 + 512;
    v[i] = (x0 + r.t3) >> 10; v[56 + i] = (x0 - r.t3) >> 10;
    v[8 + i] = (x1 + r.t2) >> 10; v[48 + i] = (x1 - r.t2) >> 10;
    v[16 + i] = (x2 + r.t1) >> 10; v[40 + i] = (x2 - r.t1) >> 10;
    v[24 + i] = (x3 + r.t0) >> 10; v[32 + i] = (x3 - r.t0) >> 10;
  }
  for (int i = 0; i < 8; i++) {
    const int* w = v + 8 * i;
    uint8_t* o = out + stride * size_t(i);
    const Idct1D r = idct1d(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    const int bias = 65536 + (128 << 17);
    const int x0 = r.x0 + bias, x1 = r.x1 + bias, x2 = r.x2 + bias, x3 = r.x3 + bias;
    o[0] = clamp255((x0 + r.t3) >> 17); o[7] = clamp255((x0 - r.t3) >> 17);
    o[1] = clamp255((x1 + r.t2) >> 17); o[6] = clamp255((x1 - r.t2) >> 17);
    o[2] = clamp255((x2 + r.t1) >> 17); o[5] = clamp255((x2 - r.t1) >> 17);
    o[3] = clamp255((x3 + r.t0) >> 17); o[4] = clamp255((x3 - r.t0) >> 17);
  }
}
static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                                    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                    47, 55, 62, 63};
}  // namespace detail

inline Image8 decodeJpeg(const uint8_t* data, size_t len) {
  using namespace detail;
  if (!isJpeg(data, len)) throw std::runtime_error("jpeg: no SOI marker");
  struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dcPred = 0;
    uint32_t x = 0, y = 0, w2 = 0, h2 = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coef;       // progressive: 64 coefficients per block, (w2 / 8) x (h2 / 8) blocks
  };
  Comp comp[3];
  int nComp = 0, hMax = 1, vMax = 1, restartInterval = 0, adobe = -1, rgbIds = 0;
  bool jfif = false, haveFrame = false, sawScan = false, progressive = false;
  uint32_t W = 0, H = 0, mcuX = 0, mcuY = 0;
  uint16_t dequant[4][64] = {};
  JpegHuff hdc[4], hac[4];
  size_t pos = 2;
  auto u8 = [&]() -> int { if (pos >= len) throw std::runtime_error("jpeg: truncated"); return data[pos++]; };
  auto u16 = [&]() -> int { const int a = u8(); return (a << 8) | u8(); };
  auto nextMarker = [&]() -> int {
    int x = u8();
    if (x != 0xFF) throw std::runtime_error("jpeg: marker expected");
    while (x == 0xFF) x = u8();
    return x;
  };
  int pending = -1;
  for (;;) {
    const int m = pending >= 0 ? pending : nextMarker();
    pending = -1;
    if (m == 0xD9) break;                                              // EOI
    if (m >= 0xD0 && m <= 0xD7) continue;                              // stray RSTn between segments
    if (m == 0xC0 || m == 0xC1 || m == 0xC2) {                         // SOF0 / SOF1 / SOF2
      progressive = m == 0xC2;
      if (haveFrame) throw std::runtime_error("jpeg: more than one frame");
      const int Lf = u16();
      if (u8() != 8) throw std::runtime_error("jpeg: only 8-bit samples are supported");
      H = uint32_t(u16()); W = uint32_t(u16());
      nComp = u8();
      if (H == 0 || W == 0) throw std::runtime_error("jpeg: zero image dimension");
      if (nComp == 4) throw std::runtime_error("jpeg: 4-component (CMYK / YCCK) files are not supported");
      if (nComp != 1 && nComp != 3) throw std::runtime_error("jpeg: bad component count");
      if (Lf != 8 + 3 * nComp) throw std::runtime_error("jpeg: bad SOF length");
      for (int i = 0; i < nComp; i++) {
        comp[i].id = u8();
        if (nComp == 3 && comp[i].id == "RGB"[i]) rgbIds++;
        const int q = u8();
        comp[i].h = q >> 4; comp[i].v = q & 15; comp[i].tq = u8();
        if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) throw std::runtime_error("jpeg: bad component parameters");
        hMax = std::max(hMax, comp[i].h); vMax = std::max(vMax, comp[i].v);
      }
      for (int i = 0; i < nComp; i++) if (hMax % comp[i].h || vMax % comp[i].v) throw std::runtime_error("jpeg: fractional sampling ratios are not supported");
      mcuX = (W + uint32_t(hMax) * 8 - 1) / (uint32_t(hMax) * 8); mcuY = (H + uint32_t(vMax) * 8 - 1) / (uint32_t(vMax) * 8);
      for (int i = 0; i < nComp; i++) {
        Comp& c = comp[i];
        c.x = (W * uint32_t(c.h) + uint32_t(hMax) - 1) / uint32_t(hMax); c.y = (H * uint32_t(c.v) + uint32_t(vMax) - 1) / uint32_t(vMax);
        c.w2 = mcuX * uint32_t(c.h) * 8; c.h2 = mcuY * uint32_t(c.v) * 8;
        c.plane.assign(size_t(c.w2) * c.h2, 0);
        if (progressive) c.coef.assign(size_t(c.w2) * c.h2, 0);
      }
      haveFrame = true;
    } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      throw std::runtime_error("jpeg: lossless / hierarchical / arithmetic-coded files are not supported");
    } else if (m == 0xC4) {                                            // DHT
      int L = u16() - 2;
      while (L > 0) {
        const int q = u8(), tc = q >> 4, th = q & 15;
        if (tc > 1 || th > 3) throw std::runtime_error("jpeg: bad DHT header");
        JpegHuff& h = tc ? hac[th] : hdc[th];
        int n = 0;
        for (int l = 1; l <= 16; l++) { h.bits[l] = uint8_t(u8()); n += h.bits[l]; }
        if (n > 256) throw std::runtime_error("jpeg: bad DHT header");
        for (int i = 0; i < n; i++) h.vals[i] = uint8_t(u8());
        h.build();
        L -= 17 + n;
      }
      if (L != 0) throw std::runtime_error("jpeg: bad DHT length");
    } else if (m == 0xDB) {                                            // DQT
      int L = u16() - 2;
      while (L > 0) {
        const int q = u8(), p = q >> 4, t = q & 15;
        if (p > 1 || t > 3) throw std::runtime_error("jpeg: bad DQT header");
        for (int i = 0; i < 64; i++) dequant[t][kZigzag[i]] = uint16_t(p ? u16() : u8());
        L -= p ? 129 : 65;
      }
      if (L != 0) throw std::runtime_error("jpeg: bad DQT length");
    } else if (m == 0xDD) {                                            // DRI
      if (u16() != 4) throw std::runtime_error("jpeg: bad DRI length");
      restartInterval = u16();
    } else if (m == 0xDC) {                                            // DNL
      if (u16() != 4 || uint32_t(u16()) != H) throw std::runtime_error("jpeg: bad DNL segment");
    } else if (m == 0xDA) {                                            // SOS
      if (!haveFrame) throw std::runtime_error("jpeg: scan before frame header");
      const int Ls = u16(), ns = u8();
      if (ns < 1 || ns > nComp || Ls != 6 + 2 * ns) throw std::runtime_error("jpeg: bad SOS header");
      int order[3] = {0, 0, 0};
      for (int i = 0; i < ns; i++) {
        const int id = u8(), q = u8();
        int which = 0;
        while (which < nComp && comp[which].id != id) which++;
        if (which == nComp) throw std::runtime_error("jpeg: scan names an unknown component");
        comp[which].hd = q >> 4; comp[which].ha = q & 15;
        if (comp[which].hd > 3 || comp[which].ha > 3) throw std::runtime_error("jpeg: bad Huffman table index");
        order[i] = which;
      }
      const int ss = u8(), se = u8(), a = u8(), ah = a >> 4, al = a & 15;
      if (!progressive) { if (ss != 0 || a != 0) throw std::runtime_error("jpeg: bad SOS parameters for a sequential frame"); }
      else {
        if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) throw std::runtime_error("jpeg: bad SOS parameters");
        if (ss == 0 && se != 0) throw std::runtime_error("jpeg: a progressive scan cannot mix DC and AC coefficients");
        if (ss != 0 && ns != 1) throw std::runtime_error("jpeg: progressive AC scans hold one component");
      }
      JpegBits br{data + pos, data + len};
      int todo = restartInterval ? restartInterval : 0x7fffffff;
      int eobRun = 0;
      for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
      // progressive scans (T.81 G.1.2): the block's 64 coefficients live in c.coef across scans
      auto progBlock = [&](Comp& c, size_t bx, size_t by) {
        int16_t* d = c.coef.data() + 64 * (bx + by * size_t(c.w2 / 8));
        if (ss == 0) {                                                 // DC: first pass or one more bit
          if (ah == 0) {
            const JpegHuff& dc = hdc[c.hd];
            if (!dc.defined) throw std::runtime_error("jpeg: scan uses an undefined Huffman table");
            const int t = jpegDecodeSymbol(br, dc);
            if (t > 15) throw std::runtime_error("jpeg: bad DC category");
            c.dcPred += jpegExtend(br.get(t), t);
            std::memset(d, 0, 64 * sizeof(int16_t));
            d[0] = int16_t(c.dcPred * (1 << al));
          } else if (br.bit()) d[0] = int16_t(d[0] + int16_t(1 << al));
          return;
        }
        const JpegHuff& ac = hac[c.ha];
        if (!ac.defined) throw std::runtime_error("jpeg: scan uses an undefined Huffman table");
        if (ah == 0) {                                                 // AC first pass
          if (eobRun) { eobRun--; return; }
          int k = ss;
          do {
            const int rs = jpegDecodeSymbol(br, ac), sz = rs & 15, r = rs >> 4;
            if (sz == 0) {
              if (r < 15) { eobRun = 1 << r; if (r) eobRun += br.get(r); eobRun--; break; }
              k += 16;
            } else {
              k += r;
              if (k > 63) throw std::runtime_error("jpeg: AC run past the end of the block");
              d[kZigzag[k++]] = int16_t(jpegExtend(br.get(sz), sz) * (1 << al));
            }
          } while (k <= se);
          return;
        }
        const int16_t bit = int16_t(1 << al);                          // AC refinement
        auto refine = [&](int16_t& p) {
          if (br.bit() && (p & bit) == 0) p = int16_t(p > 0 ? p + bit : p - bit);
        };
        if (eobRun) {
          eobRun--;
          for (int k = ss; k <= se; k++) { int16_t& p = d[kZigzag[k]]; if (p != 0) refine(p); }
          return;
        }
        int k = ss;
        do {
          const int rs = jpegDecodeSymbol(br, ac);
          int sz = rs & 15, r = rs >> 4;
          if (sz == 0) {
            if (r < 15) { eobRun = (1 << r) - 1; if (r) eobRun += br.get(r); r = 64; }
          } else {
            if (sz != 1) throw std::runtime_error("jpeg: bad refinement code");
            sz = br.bit() ? bit : -bit;
          }
          while (k <= se) {
            int16_t& p = d[kZigzag[k++]];
            if (p != 0) refine(p);
            else { if (r == 0) { p = int16_t(sz); break; } r--; }
          }
        } while (k <= se);
      };
      auto block = [&](Comp& c, size_t bx, size_t by) {
        if (progressive) { progBlock(c, bx, by); return; }
        const JpegHuff& dc = hdc[c.hd]; const JpegHuff& ac = hac[c.ha];
        if (!dc.defined || !ac.defined) throw std::runtime_error("jpeg: scan uses an undefined Huffman table");
        const uint16_t* dq = dequant[c.tq];
        int16_t coef[64] = {0};
        const int t = jpegDecodeSymbol(br, dc);
        if (t > 15) throw std::runtime_error("jpeg: bad DC category");
        c.dcPred += jpegExtend(br.get(t), t);
        coef[0] = int16_t(c.dcPred * dq[0]);
        for (int k = 1; k < 64;) {
          const int rs = jpegDecodeSymbol(br, ac), s = rs & 15, r = rs >> 4;
          if (s == 0) { if (rs != 0xF0) break; k += 16; continue; }
          k += r;
          if (k > 63) throw std::runtime_error("jpeg: AC run past the end of the block");
          const uint8_t z = kZigzag[k++];
          coef[z] = int16_t(jpegExtend(br.get(s), s) * dq[z]);
        }
        jpegIdct(c.plane.data() + size_t(c.w2) * by * 8 + bx * 8, c.w2, coef);
      };
      auto restartCheck = [&](bool last) {
        if (--todo > 0 || last) return;
        br.fill();                                                     // runs into the RSTn marker
        if (br.marker < 0xD0 || br.marker > 0xD7) throw std::runtime_error("jpeg: restart marker expected");
        br.reset();
        for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
        eobRun = 0;
        todo = restartInterval;
      };
      if (ns == 1) {                                                   // non-interleaved: the component's own block grid
        Comp& c = comp[order[0]];
        const size_t bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
        for (size_t j = 0; j < bh; j++) for (size_t i = 0; i < bw; i++) { block(c, i, j); restartCheck(j + 1 == bh && i + 1 == bw); }
      } else {
        for (size_t j = 0; j < mcuY; j++)
          for (size_t i = 0; i < mcuX; i++) {
            for (int k = 0; k < ns; k++) {
              Comp& c = comp[order[k]];
              for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++) block(c, i * size_t(c.h) + size_t(x), j * size_t(c.v) + size_t(y));
            }
            restartCheck(j + 1 == mcuY && i + 1 == mcuX);
          }
      }
      sawScan = true;
      // continue after the entropy-coded segment: at the marker the bit reader ran into, else search for one
      pos = size_t(br.p - data);
      if (br.marker >= 0) pending = br.marker;
      else {
        while (pos + 1 < len && !(data[pos] == 0xFF && data[pos + 1] != 0x00 && data[pos + 1] != 0xFF)) pos++;
        if (pos + 1 >= len) break;                                     // no EOI: accept what was decoded
      }
    } else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {                // APPn / COM
      int L = u16();
      if (L < 2) throw std::runtime_error("jpeg: bad segment length");
      L -= 2;
      if (pos + size_t(L) > len) throw std::runtime_error("jpeg: truncated segment");
      if (m == 0xE0 && L >= 5 && std::memcmp(data + pos, "JFIF\0", 5) == 0) jfif = true;
      if (m == 0xEE && L >= 12 && std::memcmp(data + pos, "Adobe\0", 6) == 0) adobe = data[pos + 11];
      pos += size_t(L);
    } else {
      throw std::runtime_error("jpeg: unknown marker");
    }
  }
  if (!haveFrame || !sawScan) throw std::runtime_error("jpeg: no image data");
  if (progressive)
    for (int i = 0; i < nComp; i++) {
      Comp& c = comp[i];
      const size_t bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
      for (size_t by = 0; by < bh; by++)
        for (size_t bx = 0; bx < bw; bx++) {
          int16_t* d = c.coef.data() + 64 * (bx + by * size_t(c.w2 / 8));
          for (int k = 0; k < 64; k++) d[k] = int16_t(d[k] * dequant[c.tq][k]);
          jpegIdct(c.plane.data() + size_t(c.w2) * by * 8 + bx * 8, c.w2, d);
        }
    }

  Image8 img;
  img.width = W; img.height = H;
  img.rgba.assign(size_t(W) * H * 4, 255);
  const bool isRgb = nComp == 3 && (rgbIds == 3 || (adobe == 0 && !jfif));
  struct Resample { int hs, vs, ystep; uint32_t wLo, ypos; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; };
  Resample rs[3];
  for (int k = 0; k < nComp; k++) {
    Resample& r = rs[k];
    r.hs = hMax / comp[k].h; r.vs = vMax / comp[k].v; r.ystep = r.vs >> 1;
    r.wLo = (W + uint32_t(r.hs) - 1) / uint32_t(r.hs); r.ypos = 0;
    r.line0 = r.line1 = comp[k].plane.data();
    r.buf.assign(size_t(r.wLo) * size_t(r.hs) + 8, 0);
  }
  const int kCr = f2f(1.40200f) << 8, kCrG = -(f2f(0.71414f) << 8), kCbG = -(f2f(0.34414f) << 8), kCb = f2f(1.77200f) << 8;
  for (uint32_t j = 0; j < H; j++) {
    const uint8_t* row[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < nComp; k++) {
      Resample& r = rs[k];
      const bool bot = r.ystep >= (r.vs >> 1);
      const uint8_t* nearRow = bot ? r.line1 : r.line0;
      const uint8_t* farRow = bot ? r.line0 : r.line1;
      uint8_t* o = r.buf.data();
      const int w = int(r.wLo);
      if (r.hs == 1 && r.vs == 1) row[k] = nearRow;
      else {
        if (r.hs == 1 && r.vs == 2) { for (int i = 0; i < w; i++) o[i] = uint8_t((3 * nearRow[i] + farRow[i] + 2) >> 2); }
        else if (r.hs == 2 && r.vs == 1) {
          if (w == 1) o[0] = o[1] = nearRow[0];
          else {
            o[0] = nearRow[0]; o[1] = uint8_t((nearRow[0] * 3 + nearRow[1] + 2) >> 2);
            int i = 1;
            for (; i < w - 1; i++) { const int n = 3 * nearRow[i] + 2; o[i * 2] = uint8_t((n + nearRow[i - 1]) >> 2); o[i * 2 + 1] = uint8_t((n + nearRow[i + 1]) >> 2); }
            o[i * 2] = uint8_t((nearRow[w - 2] * 3 + nearRow[w - 1] + 2) >> 2); o[i * 2 + 1] = nearRow[w - 1];
          }
        } else if (r.hs == 2 && r.vs == 2) {
          if (w == 1) o[0] = o[1] = uint8_t((3 * nearRow[0] + farRow[0] + 2) >> 2);
          else {
            int t1 = 3 * nearRow[0] + farRow[0];
            o[0] = uint8_t((t1 + 2) >> 2);
            for (int i = 1; i < w; i++) {
              const int t0 = t1;
              t1 = 3 * nearRow[i] + farRow[i];
              o[i * 2 - 1] = uint8_t((3 * t0 + t1 + 8) >> 4); o[i * 2] = uint8_t((3 * t1 + t0 + 8) >> 4);
            }
            o[w * 2 - 1] = uint8_t((t1 + 2) >> 2);
          }
        } else { for (int i = 0; i < w; i++) for (int q = 0; q < r.hs; q++) o[i * r.hs + q] = nearRow[i]; }
        row[k] = o;
      }
      if (++r.ystep >= r.vs) {
        r.ystep = 0;
        r.line0 = r.line1;
        if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
      }
    }
    uint8_t* out = &img.rgba[size_t(j) * W * 4];
    if (nComp == 1) { for (uint32_t i = 0; i < W; i++) { out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = row[0][i]; } }
    else if (isRgb) { for (uint32_t i = 0; i < W; i++) { out[4 * i] = row[0][i]; out[4 * i + 1] = row[1][i]; out[4 * i + 2] = row[2][i]; } }
    else {
      for (uint32_t i = 0; i < W; i++) {
        const int yf = (int(row[0][i]) << 20) + (1 << 19), cr = int(row[2][i]) - 128, cb = int(row[1][i]) - 128;
        const int r = yf + cr * kCr;
        const int g = yf + cr * kCrG + int(uint32_t(cb * kCbG) & 0xffff0000u);
        const int b = yf + cb * kCb;
        out[4 * i] = clamp255(r >> 20); out[4 * i + 1] = clamp255(g >> 20); out[4 * i + 2] = clamp255(b >> 20);
      }
    }
  }
  return img;
}

// 8-bit RGBA of an embedded glTF image (PNG or baseline JPEG, the two formats glTF 2.0 allows)
inline Image8 decodeImage8(const uint8_t* data, size_t len) {
  if (isPng(data, len)) return decodePng(data, len);
  if (isJpeg(data, len)) return decodeJpeg(data, len);
  throw std::runtime_error("unknown embedded image format (PNG or JPEG expected)");
}

// Radiance RGBE (.hdr), "-Y h +X w" orientation, flat or new-style RLE scanlines.
inline ImageF decodeHdr(const uint8_t* data, size_t len) {
  size_t pos = 0;
  auto line = [&]() {
    std::string s;
    while (pos < len && data[pos] != '\n') s.push_back(char(data[pos++]));
    if (pos < len) pos++;
    return s;
  };
  const std::string magic = line();
  if (magic != "#?RADIANCE" && magic != "#?RGBE") throw std::runtime_error("hdr: not a Radiance file");
  bool okFormat = false;
  for (;;) {
    if (pos >= len) throw std::runtime_error("hdr: truncated header");
    const std::string s = line();
    if (s.empty()) break;
    if (s == "FORMAT=32-bit_rle_rgbe") okFormat = true;
  }
  if (!okFormat) throw std::runtime_error("hdr: unsupported pixel format");
  const std::string res = line();
  long hh = 0, ww = 0;
  if (std::sscanf(res.c_str(), "-Y %ld +X %ld", &hh, &ww) != 2 || hh <= 0 || ww <= 0 || hh > (1 << 24) || ww > (1 << 24))
    throw std::runtime_error("hdr: unsupported resolution line");
  const uint32_t w = uint32_t(ww), h = uint32_t(hh);
  ImageF img;
  img.width = w; img.height = h;
  img.rgba.assign(size_t(w) * h * 4, 0.0f);
  auto get8 = [&]() -> uint8_t { return pos < len ? data[pos++] : uint8_t(0); };     // reads past the end yield 0
  auto convert = [&](float* out, const uint8_t* in) {
    if (in[3] != 0) {
      const float f1 = float(std::ldexp(1.0f, int(in[3]) - (128 + 8)));
      out[0] = in[0] * f1; out[1] = in[1] * f1; out[2] = in[2] * f1;
    } else out[0] = out[1] = out[2] = 0.0f;
    out[3] = 1.0f;
  };
  auto flat = [&](uint32_t firstI) {               // uncompressed pixels from (firstI, 0) on
    for (uint32_t j = 0; j < h; j++)
      for (uint32_t i = j == 0 ? firstI : 0; i < w; i++) {
        uint8_t rgbe[4];
        for (auto& b : rgbe) b = get8();
        convert(&img.rgba[(size_t(j) * w + i) * 4], rgbe);
      }
  };
  if (w < 8 || w >= 32768) { flat(0); return img; }
  std::vector<uint8_t> scan(size_t(w) * 4);
  for (uint32_t j = 0; j < h; j++) {
    const uint8_t c1 = get8(), c2 = get8(), l1 = get8();
    if (c1 != 2 || c2 != 2 || (l1 & 0x80)) {
      // not run-length encoded: these bytes are the first pixel of a flat image
      const uint8_t rgbe[4] = {c1, c2, l1, get8()};
      convert(&img.rgba[0], rgbe);
      flat(1);
      return img;
    }
    const uint32_t n = (uint32_t(l1) << 8) | get8();
    if (n != w) throw std::runtime_error("hdr: corrupt scanline length");
    for (uint32_t k = 0; k < 4; k++) {
      uint32_t i = 0;
      while (i < w) {
        uint32_t count = get8();
        const uint32_t left = w - i;
        if (count > 128) {
          const uint8_t v = get8();
          count -= 128;
          if (count == 0 || count > left) throw std::runtime_error("hdr: corrupt run");
          for (uint32_t z = 0; z < count; z++) scan[size_t(i++) * 4 + k] = v;
        } else {
          if (count == 0 || count > left) throw std::runtime_error("hdr: corrupt run");
          for (uint32_t z = 0; z < count; z++) scan[size_t(i++) * 4 + k] = get8();
        }
      }
    }
    for (uint32_t i = 0; i < w; i++) convert(&img.rgba[(size_t(j) * w + i) * 4], &scan[size_t(i) * 4]);
  }
  return img;
}

}  // namespace image
}  // namespace yart_hip
