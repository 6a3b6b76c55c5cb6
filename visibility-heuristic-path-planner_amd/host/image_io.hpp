// image_io.hpp -- minimal PNG / PGM I/O for the image surface (mode 2 maps, result images).
// The reference uses SFML's sf::Image (environment.cpp:183-214, solver.cpp:898-1017);
// SFML is not available here, so 8-bit PNG decode/encode is done on zlib directly.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace vbs {

struct Rgba { uint8_t r = 0, g = 0, b = 0, a = 255; };

struct Image {
  unsigned width = 0, height = 0;
  std::vector<Rgba> px;  // row-major, y = 0 at the top
  Rgba& at(unsigned x, unsigned y) { return px[(size_t)y * width + x]; }
  const Rgba& at(unsigned x, unsigned y) const { return px[(size_t)y * width + x]; }
  void create(unsigned w, unsigned h, Rgba fill) { width = w; height = h; px.assign((size_t)w * h, fill); }
};

// PNG (8-bit gray / gray+alpha / RGB / RGBA / palette, non-interlaced) or binary PGM (P5)
bool loadImage(const std::string& path, Image& out, std::string* err);
bool savePng(const std::string& path, const Image& img, std::string* err);

}  // namespace vbs
