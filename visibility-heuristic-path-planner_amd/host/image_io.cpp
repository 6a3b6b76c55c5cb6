#include "image_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace vbs {
namespace {

constexpr uint32_t kMaxImageSide = 8192;  // VHP_MAX_SIDE: nothing larger can become a grid

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool decodePng(const std::vector<uint8_t>& f, Image& out, std::string* err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) { *err = "not a PNG file"; return false; }
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  for (size_t p = 8; p + 12 <= f.size();) {
    const uint32_t len = be32(&f[p]);
    const char* tag = reinterpret_cast<const char*>(&f[p + 4]);
    const uint8_t* body = &f[p + 8];
    if ((size_t)len > f.size() || p + 12 + (size_t)len > f.size()) { *err = "truncated PNG chunk"; return false; }
    if (be32(body + len) != (uint32_t)crc32(0, &f[p + 4], (uInt)(len + 4))) { *err = "PNG chunk CRC mismatch"; return false; }
    if (!std::memcmp(tag, "IHDR", 4)) {
      if (len != 13) { *err = "bad PNG header"; return false; }
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
      if (w > kMaxImageSide || h > kMaxImageSide) { *err = "PNG larger than the solver's maximum grid side"; return false; }
    } else if (!std::memcmp(tag, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!std::memcmp(tag, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(tag, "IEND", 4)) {
      break;
    }
    p += 12 + len;
  }
  if (!w || !h || depth != 8 || interlace) { *err = "unsupported PNG (need 8-bit, non-interlaced)"; return false; }
  const int ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : ctype == 3 ? 1 : 0;
  if (!ch) { *err = "unsupported PNG colour type"; return false; }
  const size_t stride = (size_t)w * ch;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf n = raw.size();
  if (uncompress(raw.data(), &n, idat.data(), idat.size()) != Z_OK || n != raw.size()) { *err = "PNG inflate failed"; return false; }
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  out.create(w, h, Rgba{});
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* line = &raw[y * (stride + 1)];
    const int ft = line[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
      int v = line[1 + i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: v += paeth(a, b, c); break;
        default: *err = "bad PNG filter"; return false;
      }
      cur[i] = (uint8_t)v;
    }
    for (uint32_t x = 0; x < w; ++x) {
      Rgba q;
      const uint8_t* s = &cur[(size_t)x * ch];
      switch (ctype) {
        case 0: q = {s[0], s[0], s[0], 255}; break;
        case 4: q = {s[0], s[0], s[0], s[1]}; break;
        case 2: q = {s[0], s[1], s[2], 255}; break;
        case 6: q = {s[0], s[1], s[2], s[3]}; break;
        case 3:
          if ((size_t)s[0] * 3 + 2 < plte.size()) q = {plte[s[0] * 3], plte[s[0] * 3 + 1], plte[s[0] * 3 + 2], 255};
          break;
      }
      out.at(x, y) = q;
    }
    prev.swap(cur);
  }
  return true;
}

bool decodePgm(const std::vector<uint8_t>& f, Image& out, std::string* err) {
  unsigned w = 0, h = 0, maxv = 0;
  int used = 0;
  // the header is text: parse a NUL-terminated copy of (at most) its first bytes, never the raw buffer
  const std::string head(reinterpret_cast<const char*>(f.data()), std::min<size_t>(f.size(), 64));
  if (std::sscanf(head.c_str(), "P5 %u %u %u%n", &w, &h, &maxv, &used) != 3 || maxv > 255 || !w || !h ||
      w > kMaxImageSide || h > kMaxImageSide) {
    *err = "unsupported PGM";
    return false;
  }
  const size_t off = (size_t)used + 1;
  if (f.size() < off + (size_t)w * h) { *err = "truncated PGM"; return false; }
  out.create(w, h, Rgba{});
  for (size_t k = 0; k < (size_t)w * h; ++k) out.px[k] = {f[off + k], f[off + k], f[off + k], 255};
  return true;
}

}  // namespace

bool loadImage(const std::string& path, Image& out, std::string* err) {
  std::ifstream in(path, std::ios::binary);
  if (!in) { *err = "cannot open " + path; return false; }
  std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (f.size() >= 2 && f[0] == 'P' && f[1] == '5') return decodePgm(f, out, err);
  return decodePng(f, out, err);
}

bool savePng(const std::string& path, const Image& img, std::string* err) {
  const size_t stride = (size_t)img.width * 4;
  std::vector<uint8_t> raw((stride + 1) * img.height);
  for (unsigned y = 0; y < img.height; ++y) {
    uint8_t* line = &raw[y * (stride + 1)];
    line[0] = 0;
    std::memcpy(line + 1, &img.px[(size_t)y * img.width], stride);
  }
  uLongf n = compressBound(raw.size());
  std::vector<uint8_t> z(n);
  if (compress2(z.data(), &n, raw.data(), raw.size(), 6) != Z_OK) { *err = "deflate failed"; return false; }
  z.resize(n);
  std::vector<uint8_t> f = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  auto chunk = [&](const char* tag, const std::vector<uint8_t>& body) {
    put32(f, (uint32_t)body.size());
    const size_t at = f.size();
    f.insert(f.end(), tag, tag + 4);
    f.insert(f.end(), body.begin(), body.end());
    put32(f, (uint32_t)crc32(0, &f[at], (uInt)(f.size() - at)));
  };
  std::vector<uint8_t> ihdr;
  put32(ihdr, img.width);
  put32(ihdr, img.height);
  ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});
  chunk("IHDR", ihdr);
  chunk("IDAT", z);
  chunk("IEND", {});
  std::ofstream o(path, std::ios::binary);
  if (!o) { *err = "cannot write " + path; return false; }
  o.write(reinterpret_cast<const char*>(f.data()), (std::streamsize)f.size());
  return true;
}

}  // namespace vbs
