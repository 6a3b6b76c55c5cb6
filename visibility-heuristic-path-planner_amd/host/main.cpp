// vhp -- command-line driver with the reference's behaviour (src/main.cpp:6-29): parse
// ./config/settings.config (relative to the working directory), build the environment,
// solve, benchmark.  Optional: `vhp <config file>` and `--device N`.
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "environment.hpp"
#include "solver.hpp"

int main(int argc, char** argv) {
  std::string config_path = "config/settings.config";
  int device = 0;
  bool series = false, standalone = false;
  int series_points = 60;  // the reference's 60 log-spaced sizes 50..5000 (solver.cpp:298-315); fewer for a quick run
  for (int k = 1; k < argc; ++k) {
    if (!std::strcmp(argv[k], "--device") && k + 1 < argc) device = std::atoi(argv[++k]);
    else if (!std::strcmp(argv[k], "--benchmark-series")) series = true;
    else if (!std::strcmp(argv[k], "--series-points") && k + 1 < argc) series_points = std::atoi(argv[++k]);
    else if (!std::strcmp(argv[k], "--standalone")) standalone = true;
    else config_path = argv[k];
  }
  vbs::ConfigParser parser;
  const bool parsed = parser.parse(config_path);
  std::cout << "################## Parsing results: ##################### \n";
  if (!parsed) {
    std::cout << "Error parsing config file" << std::endl;
    return 1;
  }
  std::cout << "Config file parsed successfully \n" << std::endl;
  auto config = parser.getConfig();

  vbs::environment env(config);
  vbs::visibilityBasedSolver solver(env, device);
  if (!solver.ok()) {
    std::cerr << "vhp: " << solver.lastError() << std::endl;
    return 2;
  }
  solver.solve();
  if (standalone) solver.standAloneVisibility();
  solver.benchmark();
  if (series) solver.benchmarkSeries(series_points);
  return 0;
}
