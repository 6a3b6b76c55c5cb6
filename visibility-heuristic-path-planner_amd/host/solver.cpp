#include "solver.hpp"

#include <chrono>
#include <cmath>
#include <fstream>
#include <iostream>

#include "image_io.hpp"

namespace vbs {
namespace {

const char* const kBanner = "############################## Solver output ##############################";

using Clock = std::chrono::high_resolution_clock;
long long micros_since(Clock::time_point t0) {
  return std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count();
}

const Rgba kBlack{0, 0, 0, 255}, kWhite{255, 255, 255, 255}, kRed{255, 0, 0, 255}, kGreen{0, 255, 0, 255},
    kYellow{255, 255, 0, 255}, kMagenta{255, 0, 255, 255}, kCyan{0, 255, 255, 255};

void disc(Image& img, int cx, int cy, int radius, Rgba colour) {
  for (int dx = -radius; dx <= radius; ++dx)
    for (int dy = -radius; dy <= radius; ++dy) {
      const int x = cx + dx, y = cy + dy;
      if (x < 0 || y < 0 || x >= (int)img.width || y >= (int)img.height) continue;
      if (dx * dx + dy * dy <= radius * radius) img.at(x, y) = colour;
    }
}

void ring(Image& img, int cx, int cy, int radius, Rgba colour) {
  for (int dx = -radius - 1; dx <= radius + 1; ++dx)
    for (int dy = -radius - 1; dy <= radius + 1; ++dy) {
      const int x = cx + dx, y = cy + dy, d2 = dx * dx + dy * dy;
      if (x < 0 || y < 0 || x >= (int)img.width || y >= (int)img.height) continue;
      if (d2 > radius * radius && d2 <= (radius + 1) * (radius + 1)) img.at(x, y) = colour;
    }
}

}  // namespace

visibilityBasedSolver::visibilityBasedSolver(environment& env, int device_ordinal)
    : grid_(env.getVisibilityField()), config_(env.getConfig()) {
  nx_ = grid_->nx;
  ny_ = grid_->ny;
  if (vhp_create(device_ordinal, &ctx_) != VHP_OK) {
    error_ = "no usable HIP device (this build has no CPU path)";
    ctx_ = nullptr;
    return;
  }
  if (nx_ == 0 || ny_ == 0 || vhp_set_map(ctx_, grid_->cells.data(), (int)nx_, (int)ny_) != VHP_OK) {
    error_ = std::string("vhp_set_map: ") + vhp_last_error(ctx_);
    vhp_destroy(ctx_);
    ctx_ = nullptr;
  }
}

visibilityBasedSolver::~visibilityBasedSolver() {
  if (ctx_) vhp_destroy(ctx_);
}

void visibilityBasedSolver::solve() {
  if (!ctx_) { std::cerr << error_ << std::endl; return; }
  const auto t0 = Clock::now();
  point start = config_->start, end = config_->end;
  if (config_->mode == 2) {  // image coordinates: y = 0 at the top
    start.second = (int)ny_ - 1 - start.second;
    end.second = (int)ny_ - 1 - end.second;
  }
  const std::size_t cells = nx_ * ny_;
  cameFrom_.assign(cells, VHP_UNLABELLED);
  visibility_global_.assign(cells, 0.0);
  visibility_.assign(cells, 0.0);
  lightSources_.assign(2 * (config_->max_iter + 2), 0);
  const int rc = vhp_planner_solve(ctx_, start.first, start.second, end.first, end.second, config_->visibilityThreshold,
                                   config_->max_iter, cameFrom_.data(), visibility_global_.data(), visibility_.data(),
                                   lightSources_.data(), &nb_of_sources_);
  switch (rc) {
    case VHP_OK: break;
    case VHP_ERR_START_OOB: std::cout << kBanner << std::endl << "Start point is out of bounds." << std::endl; return;
    case VHP_ERR_END_OOB: std::cout << kBanner << std::endl << "End point is out of bounds." << std::endl; return;
    case VHP_ERR_START_OCCUPIED: std::cout << kBanner << std::endl << "Start point is not valid (occupied)" << std::endl; return;
    case VHP_ERR_END_OCCUPIED: std::cout << kBanner << std::endl << "End point is not valid (occupied)" << std::endl; return;
    case VHP_ERR_MAX_ITER:  // like the reference: report and leave ./output untouched
      std::cout << "Max iters hit. Solution could not be found. Try lowering visibility threshold." << std::endl;
      return;
    default: std::cerr << "vhp_planner_solve failed (" << rc << "): " << vhp_last_error(ctx_) << std::endl; return;
  }
  end_ = end;
  if (!config_->silent && config_->timer)
    std::cout << kBanner << "\n" << "Execution time in us: " << micros_since(t0) << "us" << std::endl;
  saveResults();
  reconstructPath();
}

bool visibilityBasedSolver::checkStart(point start) const {
  if (!((std::size_t)start.first < nx_ && (std::size_t)start.second < ny_)) {
    std::cout << kBanner << std::endl << "Start point is out of bounds." << std::endl;
    return false;
  }
  if (grid_->get(start.first, start.second) == 0) {
    std::cout << kBanner << std::endl << "Start point is not valid (occupied)" << std::endl;
    return false;
  }
  return true;
}

void visibilityBasedSolver::standAloneVisibility() {
  if (!ctx_) { std::cerr << error_ << std::endl; return; }
  const point start = config_->start;
  if (!checkStart(start)) return;
  std::vector<double> field(nx_ * ny_);
  const int32_t src[2] = {start.first, start.second};
  if (vhp_sweep_batch(ctx_, src, 1, VHP_SWEEP_FULL, VHP_F64, field.data()) != VHP_OK) {
    std::cerr << "vhp_sweep_batch: " << vhp_last_error(ctx_) << std::endl;
    return;
  }
  saveStandAloneVisibility(field, start, "output/standAloneVisibility.png");
}

void visibilityBasedSolver::benchmark() {
  if (!ctx_) { std::cerr << error_ << std::endl; return; }
  const point start = config_->start;
  if (!checkStart(start)) return;
  std::vector<double> field(nx_ * ny_);
  const int32_t src[2] = {start.first, start.second};
  const auto t0 = Clock::now();
  const int rc = vhp_sweep_batch(ctx_, src, 1, VHP_SWEEP_FULL, VHP_F64, field.data());  // computeVisibility()
  const long long us_sweep = micros_since(t0);
  if (rc != VHP_OK) { std::cerr << "vhp_sweep_batch: " << vhp_last_error(ctx_) << std::endl; return; }
  float kernel_ms = 0;
  vhp_last_elapsed_ms(ctx_, &kernel_ms);
  saveStandAloneVisibility(field, start, "output/standAloneVisibility.png");

  std::vector<double> rays(nx_ * ny_);
  const auto t1 = Clock::now();
  const int rc2 = vhp_raycast_all(ctx_, start.first, start.second, rays.data());  // nx*ny Bresenham rays
  const long long us_rays = micros_since(t1);
  if (rc2 == VHP_OK) saveStandAloneVisibility(rays, start, "output/rayCastingVisibility.png");

  if (!config_->silent) {
    std::cout << kBanner << "\n" << "Visibility computation time in us: " << us_sweep << "us" << std::endl;
    std::cout << "Raycasting computation time in us: " << us_rays << "us" << std::endl;
    std::cout << "Ratio. Proposed method is: " << (double)us_rays / (double)us_sweep << " faster than typical raycasting."
              << std::endl;
    std::cout << "(device time of the sweep kernel alone: " << kernel_ms * 1000.0f << "us; the figures above include the "
              << "transfer of the field to the host)" << std::endl;
  }
  std::size_t blocked = 0;
  for (uint8_t c : grid_->cells) blocked += c == 0;
  std::cout << "Density of the occupancy grid: " << (double)blocked / (double)(nx_ * ny_) * 100 << "%" << std::endl;
}

void visibilityBasedSolver::benchmarkSeries(int num_points) {
  if (!ctx_) { std::cerr << error_ << std::endl; return; }
  // log-spaced sizes 50..5000 (60 of them in the reference) on an empty grid, source at the centre (solver.cpp:298-324)
  if (num_points < 2) num_points = 2;
  const double first = 50, last = 5000;
  std::vector<int> sizes;
  for (int k = 0; k < num_points; ++k) sizes.push_back((int)std::round(first * std::exp(std::log(last / first) * k / (num_points - 1))));
  std::vector<double> t_sweep, t_rays, ratios;
  for (int n : sizes) {
    std::vector<uint8_t> empty((std::size_t)n * n, 1);
    std::vector<double> field((std::size_t)n * n);
    if (vhp_set_map(ctx_, empty.data(), n, n) != VHP_OK) { std::cerr << vhp_last_error(ctx_) << std::endl; break; }
    const int32_t src[2] = {n / 2, n / 2};
    const auto t0 = Clock::now();
    if (vhp_sweep_batch(ctx_, src, 1, VHP_SWEEP_FULL, VHP_F64, field.data()) != VHP_OK) { std::cerr << vhp_last_error(ctx_) << std::endl; break; }
    const double us0 = (double)micros_since(t0);
    const auto t1 = Clock::now();
    vhp_raycast_all(ctx_, src[0], src[1], field.data());
    const double us1 = (double)micros_since(t1);
    std::cout << "***************************" << std::endl;
    std::cout << "For grid size: " << n << "x" << n << std::endl;
    std::cout << "Visibility computation time in us: " << us0 << "us" << std::endl;
    std::cout << "Raycasting computation time in us: " << us1 << "us" << std::endl;
    std::cout << "Ratio. Proposed method is: " << us1 / us0 << " faster than typical raycasting." << std::endl;
    t_sweep.push_back(us0);
    t_rays.push_back(us1);
    ratios.push_back(us1 / us0);
  }
  std::cout << kBanner << "\n" << "Ratios: " << std::endl;
  for (double r : ratios) std::cout << r << std::endl;
  if (!ensureOutputDir("./output/benchmark_results.txt")) return;
  std::ofstream file("output/benchmark_results.txt", std::ios::app);
  for (std::size_t k = 0; k < ratios.size(); ++k)
    file << t_sweep[k] << " " << t_rays[k] << " " << ratios[k] << " " << sizes[k] << "x" << sizes[k] << std::endl;
  // restore the configured map
  vhp_set_map(ctx_, grid_->cells.data(), (int)nx_, (int)ny_);
}

// The five text matrices interface.m reads (reference solver.cpp:1022-1178).  mode 2
// writes rows top-down in image orientation (y = ny-1 .. 0) and flips the pivots' y.
void visibilityBasedSolver::saveResults() const {
  if (!ensureOutputDir("./output/cameFrom.txt")) return;
  const bool flip = config_->mode == 2;
  const bool quiet = config_->silent;
  if (config_->saveCameFrom) {
    if (!writeMatrix("./output/cameFrom.txt", cameFrom_.data(), nx_, ny_, flip)) return;
    if (!quiet) std::cout << "Saved cameFrom_" << std::endl;
  }
  if (config_->saveLightSources) {
    std::ofstream os("./output/lightSources.txt", std::ios::out | std::ios::trunc);
    if (!os.is_open()) { std::cerr << "Failed to open output file ./output/lightSources.txt" << std::endl; return; }
    for (uint32_t k = 0; k < nb_of_sources_; ++k) {  // the trailing `end` entry is not written
      const int x = lightSources_[2 * k], y = lightSources_[2 * k + 1];
      if (flip)
        os << x << " " << ny_ - 1 - y;  // size_t arithmetic, as the reference
      else
        os << x << " " << y;
      os << "\n";
    }
    os.close();
    if (!quiet) std::cout << "Saved lightSources" << std::endl;
  }
  if (config_->saveGlobalVisibility) {
    if (!writeMatrix("./output/VisibilityMap.txt", visibility_global_.data(), nx_, ny_, flip)) return;
    if (!quiet) std::cout << "Saved GlobalVisibility" << std::endl;
  }
  if (config_->saveLocalVisibility) {
    if (!writeMatrix("./output/LocalVisibilityMap.txt", visibility_.data(), nx_, ny_, flip)) return;
    if (!quiet) std::cout << "Saved LocalVisibility" << std::endl;
  }
  if (config_->saveVisibilityField) {
    std::vector<double> occ(grid_->cells.begin(), grid_->cells.end());
    if (!writeMatrix("./output/visibilityField.txt", occ.data(), nx_, ny_, flip)) return;
    if (!quiet) std::cout << "Saved OccupancyComplement" << std::endl;
  }
}

void visibilityBasedSolver::reconstructPath() {
  std::vector<int32_t> pts(2 * (std::size_t)(nb_of_sources_ + 3));
  uint32_t n = 0;
  double length = 0;
  const int rc = vhp_reconstruct_path(cameFrom_.data(), lightSources_.data(), nb_of_sources_, (int)nx_, (int)ny_, end_.first, end_.second,
                                      pts.data(), (uint32_t)(pts.size() / 2), &n, &length);
  if (rc != VHP_OK) { std::cerr << "vhp_reconstruct_path failed (" << rc << ")" << std::endl; return; }
  if (!config_->silent) std::cout << "Path length: " << length << std::endl;
  if (config_->saveResults) {
    std::vector<point> path;
    for (uint32_t k = 0; k < n && 2 * k + 1 < pts.size(); ++k) path.push_back({pts[2 * k], pts[2 * k + 1]});
    saveImageWithPath(path);
  }
}

// grey field + yellow source ball with a black ring + red obstacles, image y flipped
// (reference solver.cpp:898-955; its loops skip field row 0, reproduced here)
void visibilityBasedSolver::saveStandAloneVisibility(const std::vector<double>& field, point source,
                                                     const std::string& name) const {
  Image img;
  img.create((unsigned)nx_, (unsigned)ny_, kBlack);
  for (std::size_t y = ny_ - 1; y > 0; --y)
    for (std::size_t x = 0; x < nx_; ++x) {
      const uint8_t g = (uint8_t)(255 * field[x + y * nx_]);
      img.at((unsigned)x, (unsigned)(ny_ - 1 - y)) = Rgba{g, g, g, 255};
    }
  const int r = config_->ballRadius;
  const int cx = source.first, cy = (int)ny_ - 1 - source.second;
  disc(img, cx, cy, r, kYellow);
  ring(img, cx, cy, r, kBlack);
  for (std::size_t y = ny_ - 1; y > 0; --y)
    for (std::size_t x = 0; x < nx_; ++x)
      if (grid_->get(x, y) == 0) img.at((unsigned)x, (unsigned)(ny_ - 1 - y)) = kRed;
  std::string err;
  if (ensureOutputDir("./" + name) && !savePng(name, img, &err)) std::cerr << err << std::endl;
}

// map + magenta Bresenham segments between pivots + cyan pivot balls, green start, red end
// (reference solver.cpp:1218-1292)
void visibilityBasedSolver::saveImageWithPath(const std::vector<point>& path) const {
  if (path.empty()) return;
  Image img;
  img.create((unsigned)nx_, (unsigned)ny_, kBlack);
  for (std::size_t y = ny_ - 1; y > 0; --y)
    for (std::size_t x = 0; x < nx_; ++x)
      img.at((unsigned)x, (unsigned)(ny_ - 1 - y)) = grid_->get(x, y) < 1 ? kBlack : kWhite;
  for (std::size_t k = 0; k + 1 < path.size(); ++k) {
    int x0 = path[k].first, y0 = (int)ny_ - 1 - path[k].second;
    const int x1 = path[k + 1].first, y1 = (int)ny_ - 1 - path[k + 1].second;
    const int dx = std::abs(x1 - x0), dy = std::abs(y1 - y0);
    const int stepx = x0 < x1 ? 1 : -1, stepy = y0 < y1 ? 1 : -1;
    int err = dx - dy;
    while (x0 != x1 || y0 != y1) {
      if (x0 >= 0 && y0 >= 0 && x0 < (int)nx_ && y0 < (int)ny_) img.at(x0, y0) = kMagenta;
      const int e2 = 2 * err;
      if (e2 > -dy) { err -= dy; x0 += stepx; }
      if (e2 < dx) { err += dx; y0 += stepy; }
    }
  }
  const int r = config_->ballRadius;
  for (const point& p : path) disc(img, p.first, (int)ny_ - 1 - p.second, r, kCyan);
  disc(img, path.front().first, (int)ny_ - 1 - path.front().second, r, kGreen);
  disc(img, path.back().first, (int)ny_ - 1 - path.back().second, r, kRed);
  std::string err;
  if (ensureOutputDir("./output/ResultingPath.png") && !savePng("output/ResultingPath.png", img, &err)) std::cerr << err << std::endl;
}

}  // namespace vbs
