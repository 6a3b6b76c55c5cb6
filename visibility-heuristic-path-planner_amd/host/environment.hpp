// environment.hpp -- occupancy-grid construction (host side).
//
// Mirrors the reference's vbs::environment (include/environment/environment.h:16-88,
// src/environment.cpp): mode 1 draws random rectangles with std::srand/std::rand in the
// reference's call order, mode 2 reduces an image to a grid (free <=> red channel == 255);
// both optionally write output/visibilityField.txt.  The grid is the occupancy complement
// (1 free, 0 blocked) as one uint8 per cell, row-major x fastest -- what vhp_set_map takes.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "config.hpp"

namespace vbs {

struct Grid {
  std::size_t nx = 0, ny = 0;
  std::vector<uint8_t> cells;
  uint8_t get(std::size_t x, std::size_t y) const { return cells[x + y * nx]; }
  void set(std::size_t x, std::size_t y, uint8_t v) { cells[x + y * nx] = v; }
};

class environment {
 public:
  explicit environment(Config& config);
  void generateNewEnvironmentFromSettings();
  void generateNewEnvironment(std::size_t ncols, std::size_t nrows, int nb_of_obstacles, int min_width, int max_width,
                              int min_height, int max_height, int seedValue = 0);
  void loadImage(const std::string& filename);
  const std::shared_ptr<Grid>& getVisibilityField() const { return field_; }
  const std::shared_ptr<Config>& getConfig() const { return config_; }

 private:
  void saveEnvironment();
  std::shared_ptr<Grid> field_;
  std::shared_ptr<Config> config_;
  int seedValue_ = 1;
};

// `ny` lines of `nx` tokens, each followed by one space, default ostream formatting;
// flip = rows written from y = ny-1 down to 0 (reference: mode 2 and saveEnvironment)
template <class T>
bool writeMatrix(const std::string& path, const T* data, std::size_t nx, std::size_t ny, bool flip);
bool ensureOutputDir(const std::string& file_path);

}  // namespace vbs
