// solver.hpp -- host-side mirror of vbs::visibilityBasedSolver
// (reference include/solver/visibilityBasedSolver.h:23-71): same public methods, same
// stdout messages, same output files -- with every numeric step delegated to the HIP
// library through the C ABI of include/vhp.h.  Nothing here computes a visibility value.
#pragma once
#include <string>
#include <vector>

#include "environment.hpp"
#include "vhp.h"

namespace vbs {

class visibilityBasedSolver {
 public:
  explicit visibilityBasedSolver(environment& env, int device_ordinal = 0);
  ~visibilityBasedSolver();
  visibilityBasedSolver(const visibilityBasedSolver&) = delete;
  visibilityBasedSolver& operator=(const visibilityBasedSolver&) = delete;

  void solve();                 // solver.cpp:76-160
  void standAloneVisibility();  // :165-189
  void benchmark();             // :194-262
  void benchmarkSeries(int num_points = 60);       // :295-374
  int getGlobalIter() const { return 0; }  // the reference never increments its counter (.h:35,150)

  bool ok() const { return ctx_ != nullptr; }
  const std::string& lastError() const { return error_; }

 private:
  void saveResults() const;  // :1022-1178
  void reconstructPath();    // :1183-1213
  void saveStandAloneVisibility(const std::vector<double>& field, point source, const std::string& name) const;  // :898-955
  void saveImageWithPath(const std::vector<point>& path) const;                                                    // :1218-1292
  bool checkStart(point start) const;

  std::shared_ptr<Grid> grid_;
  std::shared_ptr<Config> config_;
  vhp_ctx* ctx_ = nullptr;
  std::string error_;
  std::size_t nx_ = 0, ny_ = 0;

  // results of the last solve(), host copies for the writers
  std::vector<uint64_t> cameFrom_;
  std::vector<double> visibility_global_, visibility_;
  std::vector<int32_t> lightSources_;  // (x, y) pairs, nb_of_sources_ + 1 entries
  uint32_t nb_of_sources_ = 0;
  point end_{};
};

}  // namespace vbs
