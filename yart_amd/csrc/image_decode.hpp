// image_decode.hpp — PNG and Radiance .hdr decoders for the glTF importer. Host only.
//
// The reference decodes embedded images with stb_image asking for 4 channels
// (core/texture.hpp:62-70 `stbi_load_from_memory(.., 4)`, core/texture.cpp:5-20 `stbi_loadf(.., 4)`).
// These decoders are written from the PNG (ISO/IEC 15948) and Radiance RGBE format descriptions
// and reproduce the conventions of that call which reach the pixels: 16-bit samples keep their
// high byte, 1/2/4-bit grey is scaled to 0..255, grey / RGB gain alpha 255, a tRNS colour key
// gives alpha 0, RGBE mantissas are scaled by 2^(e-136) with alpha 1. PNG is lossless, so the
// bytes equal the reference's (pinned by tests/golden/gltf/*.tex against oracle/_ref `texture`).
// JPEG is not decoded: its IDCT / chroma upsampling are implementation defined and a texture
// that differs from the reference's by a bit would void the parity contract; such images are
// refused with an error instead.
#pragma once
#include <zlib.h>

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

// 8-bit image of an embedded glTF image; PNG only (see the header comment)
inline Image8 decodeImage8(const uint8_t* data, size_t len) {
  if (isPng(data, len)) return decodePng(data, len);
  if (isJpeg(data, len)) throw std::runtime_error("JPEG textures are not supported (decode is implementation defined; re-encode the asset's images as PNG)");
  throw std::runtime_error("unknown embedded image format (PNG expected)");
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
