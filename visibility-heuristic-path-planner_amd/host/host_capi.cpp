// host_capi.cpp -- C entry points over the host-side surface (settings.config parser,
// environment generator / image loader, text writers) so the CPU test-suite can exercise
// them without a GPU.  Part of libvhp_host.so; no HIP dependency.
#include <cstdint>
#include <cstring>
#include <iostream>
#include <sstream>

#include "config.hpp"
#include "environment.hpp"
#include "image_io.hpp"

extern "C" {

struct vhp_host_config {
  int32_t mode;
  uint64_t ncols, nrows, nb_of_obstacles, minWidth, maxWidth, minHeight, maxHeight;
  int32_t randomSeed, seedValue;
  char imagePath[1024];
  int32_t start_x, start_y, end_x, end_y;
  uint64_t max_iter;
  double visibilityThreshold;
  float lightStrength;
  int32_t timer, saveResults, saveLocalVisibility, saveCameFrom, saveLightSources, saveGlobalVisibility,
      saveVisibilityField, silent, ballRadius;
};

// 1 when parse() succeeded.  The banner goes to stdout exactly as the CLI prints it.
int vhp_host_parse_config(const char* path, vhp_host_config* out) {
  vbs::ConfigParser p;
  const bool ok = p.parse(path);
  const vbs::Config& c = p.getConfig();
  std::memset(out, 0, sizeof(*out));
  out->mode = c.mode;
  out->ncols = c.ncols; out->nrows = c.nrows; out->nb_of_obstacles = c.nb_of_obstacles;
  out->minWidth = c.minWidth; out->maxWidth = c.maxWidth; out->minHeight = c.minHeight; out->maxHeight = c.maxHeight;
  out->randomSeed = c.randomSeed; out->seedValue = c.seedValue;
  std::strncpy(out->imagePath, c.imagePath.c_str(), sizeof(out->imagePath) - 1);
  out->start_x = c.start.first; out->start_y = c.start.second; out->end_x = c.end.first; out->end_y = c.end.second;
  out->max_iter = c.max_iter;
  out->visibilityThreshold = c.visibilityThreshold;
  out->lightStrength = c.lightStrength;
  out->timer = c.timer; out->saveResults = c.saveResults; out->saveLocalVisibility = c.saveLocalVisibility;
  out->saveCameFrom = c.saveCameFrom; out->saveLightSources = c.saveLightSources;
  out->saveGlobalVisibility = c.saveGlobalVisibility; out->saveVisibilityField = c.saveVisibilityField;
  out->silent = c.silent; out->ballRadius = c.ballRadius;
  return ok ? 1 : 0;
}

// mode-1 environment for the given sizes / seed; occ must hold ncols*nrows bytes
int vhp_host_generate_env(uint64_t ncols, uint64_t nrows, uint64_t nb, uint64_t min_w, uint64_t max_w, uint64_t min_h,
                          uint64_t max_h, int seed, uint8_t* occ) {
  vbs::Config c;
  c.mode = 1;
  c.ncols = ncols; c.nrows = nrows; c.nb_of_obstacles = nb;
  c.minWidth = min_w; c.maxWidth = max_w; c.minHeight = min_h; c.maxHeight = max_h;
  c.randomSeed = false; c.seedValue = seed;
  c.saveResults = false; c.silent = true;
  vbs::environment env(c);
  const auto& g = env.getVisibilityField();
  std::memcpy(occ, g->cells.data(), g->cells.size());
  return 0;
}

// mode-2 loader; returns 0 and the size, or 1.  occ may be null to query the size.
int vhp_host_load_image(const char* path, uint8_t* occ, uint64_t cap, int32_t* nx, int32_t* ny) {
  vbs::Image img;
  std::string err;
  if (!vbs::loadImage(path, img, &err)) return 1;
  *nx = (int32_t)img.width;
  *ny = (int32_t)img.height;
  if (occ) {
    if (cap < (uint64_t)img.width * img.height) return 2;
    for (size_t k = 0; k < img.px.size(); ++k) occ[k] = img.px[k].r == 255 ? 1 : 0;
  }
  return 0;
}

int vhp_host_write_matrix_f64(const char* path, const double* data, uint64_t nx, uint64_t ny, int flip) {
  return vbs::ensureOutputDir(path) && vbs::writeMatrix(path, data, nx, ny, flip != 0) ? 0 : 1;
}
int vhp_host_write_matrix_u64(const char* path, const uint64_t* data, uint64_t nx, uint64_t ny, int flip) {
  return vbs::ensureOutputDir(path) && vbs::writeMatrix(path, data, nx, ny, flip != 0) ? 0 : 1;
}

// PNG round trip helper for tests: writes an RGBA image
int vhp_host_save_png(const char* path, const uint8_t* rgba, uint32_t w, uint32_t h) {
  vbs::Image img;
  img.create(w, h, vbs::Rgba{});
  std::memcpy(img.px.data(), rgba, (size_t)w * h * 4);
  std::string err;
  return vbs::savePng(path, img, &err) ? 0 : 1;
}

}  // extern "C"
