// config.hpp -- settings.config surface (host side, C++20).
//
// Mirrors the reference's vbs::Config / vbs::ConfigParser
// (include/parser/parser.h:11-50, src/parser.cpp:12-353): same 25 keys, same defaults,
// same tolerances (which bad values abort the parse and which only warn), same
// messages.  Written table-driven instead of as an if-chain.
#pragma once
#include <cstddef>
#include <string>
#include <utility>

namespace vbs {

using point = std::pair<int, int>;

struct Config {
  int mode = 1;
  std::size_t ncols = 100, nrows = 100;
  std::size_t nb_of_obstacles = 10;
  std::size_t minWidth = 10, maxWidth = 20, minHeight = 10, maxHeight = 20;
  bool randomSeed = true;
  int seedValue = 0;
  std::string imagePath = "C:\\...";
  point start{}, end{};
  std::size_t max_iter = 100;
  double visibilityThreshold = 0.5;
  float lightStrength = 1;  // parsed and echoed, never used by the solver (reference: .h:144 is a constant)
  bool timer = true;
  bool saveResults = true, saveLocalVisibility = true, saveCameFrom = true, saveLightSources = true,
       saveGlobalVisibility = true, saveVisibilityField = true;
  bool silent = false;
  int ballRadius = 5;
};

class ConfigParser {
 public:
  // false: the file could not be opened or a value was rejected (main then exits 1)
  bool parse(const std::string& filename);
  const Config& getConfig() const { return config_; }

 private:
  Config config_;
};

}  // namespace vbs
