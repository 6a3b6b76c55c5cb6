#include "environment.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>

#include "image_io.hpp"

namespace vbs {

bool ensureOutputDir(const std::string& file_path) {
  namespace fs = std::filesystem;
  const fs::path dir = fs::path(file_path).parent_path();
  if (!fs::exists(dir) && !fs::create_directories(dir)) {
    std::cerr << "Failed to create directory " << dir.string() << std::endl;
    return false;
  }
  return true;
}

template <class T>
bool writeMatrix(const std::string& path, const T* data, std::size_t nx, std::size_t ny, bool flip) {
  std::ofstream os(path, std::ios::out | std::ios::trunc);
  if (!os.is_open()) {
    std::cerr << "Failed to open output file " << path << std::endl;
    return false;
  }
  for (std::size_t row = 0; row < ny; ++row) {
    const T* line = data + (flip ? ny - 1 - row : row) * nx;
    for (std::size_t x = 0; x < nx; ++x) os << line[x] << " ";
    os << "\n";
  }
  return true;
}
template bool writeMatrix<double>(const std::string&, const double*, std::size_t, std::size_t, bool);
template bool writeMatrix<unsigned long long>(const std::string&, const unsigned long long*, std::size_t, std::size_t, bool);
template bool writeMatrix<uint64_t>(const std::string&, const uint64_t*, std::size_t, std::size_t, bool);

environment::environment(Config& config) : config_(std::make_shared<Config>(config)) {
  if (config_->mode == 1) {
    if (!config_->randomSeed) seedValue_ = config_->seedValue;
    generateNewEnvironmentFromSettings();
    if (config_->saveResults) saveEnvironment();
  } else if (config_->mode == 2) {
    loadImage(config_->imagePath);
    if (config_->saveResults) saveEnvironment();
  }
}

// Four rand() draws per obstacle in the order column, width, row, height; both corners
// clamped to the last index; half-open fill (reference environment.cpp:55-78).  N is the
// integer type the size arithmetic is carried out in: std::size_t when the sizes come from
// the parsed Config (:57-66), int for the explicit-argument overload (:104-112).
template <class N>
static void scatter(Grid& g, std::size_t count, N min_w, N max_w, N min_h, N max_h, int seed) {
  std::srand(seed);
  const int last_x = (int)g.nx - 1, last_y = (int)g.ny - 1;
  for (std::size_t n = 0; n < count; ++n) {
    int x0 = (int)(1 + (std::rand() % (g.nx + 1)));
    int x1 = (int)(x0 + min_w + (std::rand() % (max_w - min_w + 1)));
    x0 = std::min(x0, last_x);
    x1 = std::min(x1, last_x);
    int y0 = (int)(1 + (std::rand() % (g.ny + 1)));
    int y1 = (int)(y0 + min_h + (std::rand() % (max_h - min_h + 1)));
    y0 = std::min(y0, last_y);
    y1 = std::min(y1, last_y);
    for (int x = x0; x < x1; ++x)
      for (int y = y0; y < y1; ++y) g.set(x, y, 0);
  }
}

void environment::generateNewEnvironmentFromSettings() {
  field_ = std::make_shared<Grid>();
  field_->nx = config_->ncols;
  field_->ny = config_->nrows;
  field_->cells.assign(field_->nx * field_->ny, 1);
  int seed = seedValue_;
  if (config_->randomSeed) {
    const auto ns = std::chrono::time_point_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now())
                        .time_since_epoch().count();
    seed = (int)ns;
  }
  scatter<std::size_t>(*field_, config_->nb_of_obstacles, config_->minWidth, config_->maxWidth, config_->minHeight,
                       config_->maxHeight, seed);
  if (!config_->silent)
    std::cout << "########################### Environment output ############################ \n"
              << "Generated new environment based on parsed settings at a seed value of: " << seed << std::endl;
}

void environment::generateNewEnvironment(std::size_t ncols, std::size_t nrows, int nb_of_obstacles, int min_width,
                                         int max_width, int min_height, int max_height, int seedValue) {
  field_ = std::make_shared<Grid>();
  field_->nx = ncols;
  field_->ny = nrows;
  field_->cells.assign(ncols * nrows, 1);
  scatter<int>(*field_, (std::size_t)std::max(nb_of_obstacles, 0), min_width, max_width, min_height, max_height, seedValue);
  if (!config_->silent)
    std::cout << "########################### Environment output ############################ \n"
              << "Generated new environment on request based on custom settings" << std::endl;
  if (config_->saveResults) saveEnvironment();
}

void environment::loadImage(const std::string& filename) {
  Image img;
  std::string err;
  if (!vbs::loadImage(filename, img, &err)) {
    std::cout << "Error: Failed to load image" << std::endl;
    field_ = std::make_shared<Grid>();
    return;
  }
  field_ = std::make_shared<Grid>();
  field_->nx = img.width;
  field_->ny = img.height;
  field_->cells.resize((std::size_t)img.width * img.height);
  // free <=> red channel == 255; field(x, y) = pixel(x, y), y = 0 at the top (environment.cpp:195-207)
  for (unsigned y = 0; y < img.height; ++y)
    for (unsigned x = 0; x < img.width; ++x) field_->set(x, y, img.at(x, y).r == 255 ? 1 : 0);
  std::cout << "Loaded image of dimensions " << field_->nx << "x" << field_->ny << " successfully" << std::endl;
}

void environment::saveEnvironment() {
  const std::string path = "./output/visibilityField.txt";
  if (!ensureOutputDir(path)) return;
  if (!config_->saveVisibilityField) return;
  // the reference streams doubles (1 / 0); the tokens are identical for 0/1 values
  std::vector<double> as_double(field_->cells.begin(), field_->cells.end());
  if (!writeMatrix(path, as_double.data(), field_->nx, field_->ny, /*flip=*/true)) return;
  if (!config_->silent) std::cout << "Saved visibility field" << std::endl;
}

}  // namespace vbs
