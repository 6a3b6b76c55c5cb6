// vhp_oracle.cpp -- CPU restatement of the reference's visibility sweep and
// visibility-heuristic planner.  TEST INFRASTRUCTURE ONLY.
//
//   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
//     load this library.  The product (libvhp_hip.so, the vhp CLI, the python
//     binding) never links, loads or calls it.
//   * Strict IEEE-754 binary64: build with -O2 -ffp-contract=off, no fast-math.
//   * PARITY PIN.  The reference solver cannot be compiled in this image (it includes
//     <SFML/Graphics.hpp>, SFML is absent, and building it against a stand-in header is
//     not allowed), so there is no oracle/_ref for it.  This restatement is pinned against
//       (1) golden outputs the reference itself ships: Samples/SFMLrayCastingVisibility.png
//           and Samples/SFMLstandAloneVisibility.png (README.md:27-33), two 1000 x 1000
//           renderings of raycasting() and computeVisibility() from one map and source
//           (committed as data: tests/golden/samples_1000.npz).  Ray casting: all 951 360
//           comparable pixels equal.  The sweep: all 951 360 equal with the reference's
//           local `offset` (:573) = 1.0, the value the published build had; HEAD has 0.0,
//           which provably does not match (tests/test_oracle_kat.py::test_reference_sample_*).
//           This pins the loop nest, the stencil form, the quadrant extents (Q2), the
//           occupancy handling and the rounding -- not the stale diagonal (Q1), which this
//           map does not discriminate;
//       (2) known answers of runs of the real reference recorded in SURVEY.md section 8c /
//           Q1-Q9: maze_6 -> 64 pivots, first five pivots, path length 1529.55; 1000^2 seed 1
//           -> pivots (50,50),(273,350),(525,675), path 1346.71, density 20.3243 %; the 9x9
//           stale-diagonal probe (Q1); row/column 0 never swept.
//     Samples/SFMLResultingPath.png (a third map) was tried with offset 0 and 1 at thresholds
//     0.1-0.5 and is NOT reproduced (one of its five pivots, (468,592), appears at offset 1,
//     thr 0.1): an older planner build; it pins nothing.  The planner's heap/tie-break
//     (Q6) and heuristic are therefore pinned by (2) only.
//
// Every function cites the reference lines it follows
// (paths relative to /root/reference).
//
// Layout: flat row-major arrays, index = x + y*nx (include/environment/field.h:24-29).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <algorithm>
#include <chrono>
#include <queue>
#include <thread>
#include <utility>
#include <vector>

namespace {

constexpr uint64_t kUnlabelled = 1000000000000000ULL;  // (size_t)1e15, solver.cpp:46

struct Grid {
  int nx = 0, ny = 0;
  std::vector<double> occ;  // occupancy complement as double: 1 free, 0 blocked
  inline size_t at(size_t x, size_t y) const { return x + y * (size_t)nx; }
};

Grid make_grid(const uint8_t* occ_u8, int nx, int ny) {
  Grid g;
  g.nx = nx;
  g.ny = ny;
  g.occ.resize((size_t)nx * ny);
  for (size_t k = 0; k < g.occ.size(); ++k) g.occ[k] = occ_u8[k] ? 1.0 : 0.0;
  return g;
}

// One of the four quadrant loop nests of computeVisibility()/updateVisibility()
// (src/visibilityBasedSolver.cpp:579-605, 609-635, 639-665, 669-695 and
// :392-432, 436-476, 480-520, 524-564).  dirx/diry = +1/-1 select the nest;
// ni/nj are its max_x_/max_y_.  The loop order (i outer over x offset, j inner
// over y offset) and the single running `v` -- which is NOT reassigned when
// i == j >= 1, so the diagonal inherits the previous inner iteration's stored
// value (quirk Q1) -- are what make this a restatement rather than a
// re-derivation.  `visit(x, y, v)` runs after the cell is stored.
// `offset` is the reference's local of that name (:384, :573; added to both operands
// of every c_, :403-404 ... :686-687).  It is 0.0 at HEAD, and 0.0 is what every
// product-facing entry point of this oracle passes.  It is a parameter only so that
// tests/test_oracle_kat.py can pin this loop nest against the reference's own
// Samples/SFMLstandAloneVisibility.png, which was rendered by a build with offset = 1.
template <class Visit>
void quadrant_nest(const Grid& g, double* vis, int sx, int sy, int dirx, int diry,
                   size_t ni, size_t nj, double& v, Visit&& visit, double offset = 0.0) {
  for (size_t i = 0; i < ni; ++i) {
    const size_t x = dirx > 0 ? (size_t)sx + i : (size_t)sx - i;
    const size_t xm = dirx > 0 ? x - 1 : x + 1;  // the column one step back toward the source
    for (size_t j = 0; j < nj; ++j) {
      const size_t y = diry > 0 ? (size_t)sy + j : (size_t)sy - j;
      const size_t ym = diry > 0 ? y - 1 : y + 1;
      if (i == 0 && j == 0) {
        v = 1.0;  // lightStrength_, visibilityBasedSolver.h:144
      } else if (i == 0) {
        v = vis[g.at(x, ym)];
      } else if (j == 0) {
        v = vis[g.at(xm, y)];
      } else if (i > j) {
        const double c = ((double)j + offset) / ((double)i + offset);  // (offset = 0.0 at HEAD: j + 0.0 is exact, so this IS j / i)
        const double a = vis[g.at(xm, y)];
        v = a - c * (a - vis[g.at(xm, ym)]);
      } else if (j > i) {
        const double c = ((double)i + offset) / ((double)j + offset);
        const double a = vis[g.at(x, ym)];
        v = a - c * (a - vis[g.at(xm, ym)]);
      }
      v = v * g.occ[g.at(x, y)];
      vis[g.at(x, y)] = v;
      visit(x, y, v);
    }
  }
}

// Quadrant order and extents: Q1 (+,+) nx-sx by ny-sy; Q2 (-,+) sx by ny-sy;
// Q3 (-,-) sx by sy; Q4 (+,-) nx-sx by sy  (solver.cpp:576-577,607-608,637-638,667-668).
template <class Visit>
void four_quadrants(const Grid& g, double* vis, int sx, int sy, Visit&& visit, double offset = 0.0) {
  double v = 0.0;
  quadrant_nest(g, vis, sx, sy, +1, +1, (size_t)(g.nx - sx), (size_t)(g.ny - sy), v, visit, offset);
  quadrant_nest(g, vis, sx, sy, -1, +1, (size_t)sx, (size_t)(g.ny - sy), v, visit, offset);
  quadrant_nest(g, vis, sx, sy, -1, -1, (size_t)sx, (size_t)sy, v, visit, offset);
  quadrant_nest(g, vis, sx, sy, +1, -1, (size_t)(g.nx - sx), (size_t)sy, v, visit, offset);
}

// eval_d, visibilityBasedSolver.h:112-115: first product in double, second in int.
inline double eval_d(int ax, int ay, int bx, int by) {
  return std::sqrt((double)(ax - bx) * (ax - bx) + (ay - by) * (ay - by));
}

// Node + ordering, visibilityBasedSolver.h:16-21 (operator< is "h greater" => min-heap).
struct HeapNode {
  size_t x, y;
  double h;
  bool operator<(const HeapNode& o) const { return h > o.h; }
};

}  // namespace

extern "C" {

// computeVisibility(), src/visibilityBasedSolver.cpp:570-696.  `vis` is NOT
// cleared (quirk Q4): cells the nests never touch keep their contents.
int vhp_oracle_sweep_full(const uint8_t* occ, int nx, int ny, int sx, int sy, double* vis) {
  if (!occ || !vis || nx <= 0 || ny <= 0) return 1;
  if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return 2;
  Grid g = make_grid(occ, nx, ny);
  four_quadrants(g, vis, sx, sy, [](size_t, size_t, double) {});
  return 0;
}

// computeVisibility() with the reference's `offset` local (:573) set to something other
// than its HEAD value 0.0.  TEST-ONLY: exists to pin the loop nest against the reference's
// published Samples/SFMLstandAloneVisibility.png (offset = 1); nothing else may call it.
int vhp_oracle_sweep_full_offset(const uint8_t* occ, int nx, int ny, int sx, int sy, double offset, double* vis) {
  if (!occ || !vis || nx <= 0 || ny <= 0) return 1;
  if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return 2;
  Grid g = make_grid(occ, nx, ny);
  four_quadrants(g, vis, sx, sy, [](size_t, size_t, double) {}, offset);
  return 0;
}

// computeVisibilityUsingQueue(), src/visibilityBasedSolver.cpp:701-893.
// FIFO flood from the 8 neighbours; explicit diagonal rule; successors pushed
// only when the pre-occupancy value exceeds 0.001; pop-order dependent (Q8).
// Coordinates are size_t in the reference, so "ls - 1" wraps on the border and
// is rejected by isValid (.h:100-102); int64 with a range check is equivalent.
int vhp_oracle_sweep_queue(const uint8_t* occ, int nx, int ny, int sx, int sy, double* vis) {
  if (!occ || !vis || nx <= 0 || ny <= 0) return 1;
  if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return 2;
  Grid g = make_grid(occ, nx, ny);
  using P = std::pair<int64_t, int64_t>;
  std::queue<P> q;
  vis[g.at(sx, sy)] = 1.0;  // :708
  const int64_t seeds[8][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}, {1, 1}, {-1, 1}, {-1, -1}, {1, -1}};  // :710-717
  for (auto& s : seeds) q.push({sx + s[0], sy + s[1]});
  std::vector<uint8_t> visited((size_t)nx * ny, 0);

  while (!q.empty()) {
    const int64_t x = q.front().first, y = q.front().second;
    q.pop();
    if (x < 0 || y < 0 || x >= nx || y >= ny) continue;  // isValid, :726
    if (visited[g.at(x, y)]) continue;                  // :729
    if (g.occ[g.at(x, y)] == 0) continue;               // :732
    const int dx = (int)(x - sx), dy = (int)(y - sy);
    // Quadrant selection exactly as the four else-if arms (:739, :777, :815, :853):
    // (dx>=0,dy>=0) (dx<0,dy>=0) (dx<0,dy<0) (dx>=0,dy<0).
    const int qx = dx >= 0 ? +1 : -1;
    const int qy = dy >= 0 ? +1 : -1;
    const int64_t xb = x - qx, yb = y - qy;  // one step back toward the source
    const int adx = dx >= 0 ? dx : -dx, ady = dy >= 0 ? dy : -dy;
    double v = 0.0;
    auto V = [&](int64_t px, int64_t py) { return vis[g.at((size_t)px, (size_t)py)]; };
    // In the (dx<0, .) arms "dx == 0" can never hold and in the (., dy<0) arms
    // "dy == 0" can never hold, so the generic tests below select the same arm
    // the reference does.
    if (dx == 0) {  // on the vertical axis: only reachable when qx == +1
      v = V(x, yb);
      if (v > 0.001) q.push({x, y + qy});
    } else if (dy == 0) {  // on the horizontal axis: only reachable when qy == +1
      v = V(xb, y);
      if (v > 0.001) q.push({x + qx, y});
    } else if (adx == ady) {
      v = V(xb, yb);
      if (v > 0.001) {
        q.push({x, y + qy});
        q.push({x + qx, y});
        q.push({x + qx, y + qy});
      }
    } else if (adx > ady) {
      // c is written as dy/dx, dy/(-dx), dy/dx (both negative), (-dy)/dx in the
      // four arms (:758,:796,:834,:872): always |dy|/|dx| with exact integer
      // operands, and IEEE division of (-a)/(-b) equals a/b bit for bit.
      const double c = (double)ady / (double)adx;
      v = V(xb, y) - c * (V(xb, y) - V(xb, yb));
      if (v > 0.001) {
        q.push({x, y + qy});
        q.push({x + qx, y});
      }
    } else {
      const double c = (double)adx / (double)ady;
      v = V(x, yb) - c * (V(x, yb) - V(xb, yb));
      if (v > 0.001) {
        q.push({x + qx, y});
        q.push({x, y + qy});
      }
    }
    v = v * g.occ[g.at(x, y)];
    vis[g.at(x, y)] = v;
    visited[g.at(x, y)] = 1;
  }
  return 0;
}

// State of one planner run: what visibilityBasedSolver::reset() allocates
// (src/visibilityBasedSolver.cpp:42-60).
struct vhp_oracle_planner {
  Grid g;
  std::vector<double> vis_global, vis_local;
  std::vector<uint64_t> came_from;
  std::vector<int32_t> pivots;  // x,y pairs; lightSources_
  uint64_t nb = 0;              // nb_of_sources_
  double scale = 0, thr = 0;
  double offset = 0.0;  // the reference's local of updateVisibility() (:384); 0.0 except in the test-only *_offset entry
  int ex = 0, ey = 0;
  // result of the last planner step
  int top_x = 0, top_y = 0;
  double top_h = 0;
  uint64_t pushes = 0;
};

// updateVisibility(), src/visibilityBasedSolver.cpp:379-565, for the pivot
// (px,py) with label index `nb`; returns heap_->top() in (top_x, top_y, top_h).
static int planner_step(vhp_oracle_planner& P, int px, int py) {
  const Grid& g = P.g;
  std::fill(P.vis_local.begin(), P.vis_local.end(), 0.0);  // visibility_.reset(), :386
  std::vector<HeapNode> store;
  store.reserve((size_t)g.nx * g.ny);  // resetQueue(), :65-71
  std::priority_queue<HeapNode> heap(std::less<HeapNode>(), std::move(store));
  four_quadrants(g, P.vis_local.data(), px, py, [&](size_t x, size_t y, double v) {
    const size_t k = g.at(x, y);
    P.vis_global[k] = std::max(v, P.vis_global[k]);  // :417-418
    if (v >= P.thr) {                                // :419-423
      if ((double)P.came_from[k] == 1e15) P.came_from[k] = P.nb;
    }
    if (P.vis_global[k] >= P.thr) {  // :424-430
      const uint64_t lab = P.came_from[k];
      const int qx = P.pivots[2 * lab], qy = P.pivots[2 * lab + 1];
      const double h = (P.scale * P.vis_global[k]) +
                       (eval_d((int)x, (int)y, P.ex, P.ey) + eval_d((int)x, (int)y, qx, qy));
      heap.push(HeapNode{x, y, h});
    }
  }, P.offset);
  P.pushes = heap.size();
  if (heap.empty()) return 3;  // reference would call top() on an empty heap (UB)
  P.top_x = (int)heap.top().x;
  P.top_y = (int)heap.top().y;
  P.top_h = heap.top().h;
  return 0;
}

// solve(), src/visibilityBasedSolver.cpp:76-160 (the mode-2 y flip of :83-86 is
// the caller's job: coordinates here are field coordinates).
// status: 0 solved; 10 start out of bounds; 11 end out of bounds; 12 start
// occupied; 13 end occupied (:89-116); 20 max_iter hit (:134-139; outputs are
// still filled so tests can look at the live-lock); 3 nothing lit.
// pivots_xy must hold 2*(max_iter+2) int32.  On return *n_pivots = nb_of_sources_
// and pivots_xy[2*nb..] = end (:141).
static int planner_solve_impl(const uint8_t* occ, int nx, int ny, int start_x, int start_y,
                              int end_x, int end_y, double threshold, uint64_t max_iter,
                              uint64_t* came_from, double* vis_global, double* vis_local,
                              int32_t* pivots_xy, uint32_t* n_pivots, double* top_h_trace, double offset) {
  if (!occ || nx <= 0 || ny <= 0) return 1;
  auto valid = [&](int x, int y) { return (size_t)x < (size_t)nx && (size_t)y < (size_t)ny; };  // .h:100-102
  if (!valid(start_x, start_y)) return 10;
  if (!valid(end_x, end_y)) return 11;
  vhp_oracle_planner P;
  P.g = make_grid(occ, nx, ny);
  if (P.g.occ[P.g.at(start_x, start_y)] == 0) return 12;
  if (P.g.occ[P.g.at(end_x, end_y)] == 0) return 13;
  const size_t n = (size_t)nx * ny;
  P.vis_global.assign(n, 0.0);
  P.vis_local.assign(n, 0.0);
  P.came_from.assign(n, kUnlabelled);
  P.pivots.assign(2 * (max_iter + 2), 0);
  P.scale = std::sqrt((double)((size_t)ny * ny + (size_t)nx * nx));  // :49
  P.thr = threshold;
  P.offset = offset;
  P.ex = end_x;
  P.ey = end_y;
  int px = start_x, py = start_y;
  P.pivots[0] = px;
  P.pivots[1] = py;                               // :121
  P.came_from[P.g.at(start_x, start_y)] = P.nb;  // :122
  int status = 0;
  while (P.vis_global[P.g.at(end_x, end_y)] <= threshold) {  // :127
    int rc = planner_step(P, px, py);
    if (rc) { status = rc; break; }
    if (top_h_trace) top_h_trace[P.nb] = P.top_h;
    px = P.top_x;
    py = P.top_y;
    ++P.nb;
    P.pivots[2 * P.nb] = px;
    P.pivots[2 * P.nb + 1] = py;
    if (P.nb > max_iter) { status = 20; break; }  // :134-139
  }
  if (status == 0) {
    P.pivots[2 * P.nb] = end_x;  // :141
    P.pivots[2 * P.nb + 1] = end_y;
  }
  if (came_from) std::memcpy(came_from, P.came_from.data(), n * sizeof(uint64_t));
  if (vis_global) std::memcpy(vis_global, P.vis_global.data(), n * sizeof(double));
  if (vis_local) std::memcpy(vis_local, P.vis_local.data(), n * sizeof(double));
  if (pivots_xy) std::memcpy(pivots_xy, P.pivots.data(), 2 * (size_t)(P.nb + 1) * sizeof(int32_t));
  if (n_pivots) *n_pivots = (uint32_t)P.nb;
  return status;
}

int vhp_oracle_planner_solve(const uint8_t* occ, int nx, int ny, int start_x, int start_y,
                             int end_x, int end_y, double threshold, uint64_t max_iter,
                             uint64_t* came_from, double* vis_global, double* vis_local,
                             int32_t* pivots_xy, uint32_t* n_pivots, double* top_h_trace) {
  return planner_solve_impl(occ, nx, ny, start_x, start_y, end_x, end_y, threshold, max_iter, came_from,
                            vis_global, vis_local, pivots_xy, n_pivots, top_h_trace, 0.0);
}

// TEST-ONLY twin of the above with the reference's `offset` local (:384) exposed; see
// vhp_oracle_sweep_full_offset.
int vhp_oracle_planner_solve_offset(const uint8_t* occ, int nx, int ny, int start_x, int start_y,
                                    int end_x, int end_y, double threshold, uint64_t max_iter, double offset,
                                    uint64_t* came_from, double* vis_global, double* vis_local,
                                    int32_t* pivots_xy, uint32_t* n_pivots) {
  return planner_solve_impl(occ, nx, ny, start_x, start_y, end_x, end_y, threshold, max_iter, came_from,
                            vis_global, vis_local, pivots_xy, n_pivots, nullptr, offset);
}

// A single updateVisibility() on caller-owned state (for step-level parity tests).
// Returns 0 and the heap top; 3 when nothing was pushed.
int vhp_oracle_planner_step(const uint8_t* occ, int nx, int ny, int pivot_x, int pivot_y,
                            int end_x, int end_y, double threshold, uint64_t label,
                            const int32_t* pivots_xy, uint64_t* came_from, double* vis_global,
                            double* vis_local, int32_t* top_xy, double* top_h, uint64_t* n_pushed) {
  vhp_oracle_planner P;
  P.g = make_grid(occ, nx, ny);
  const size_t n = (size_t)nx * ny;
  P.vis_global.assign(vis_global, vis_global + n);
  P.vis_local.assign(n, 0.0);
  P.came_from.assign(came_from, came_from + n);
  P.pivots.assign(pivots_xy, pivots_xy + 2 * (label + 1));
  P.nb = label;
  P.scale = std::sqrt((double)((size_t)ny * ny + (size_t)nx * nx));
  P.thr = threshold;
  P.ex = end_x;
  P.ey = end_y;
  int rc = planner_step(P, pivot_x, pivot_y);
  std::memcpy(came_from, P.came_from.data(), n * sizeof(uint64_t));
  std::memcpy(vis_global, P.vis_global.data(), n * sizeof(double));
  if (vis_local) std::memcpy(vis_local, P.vis_local.data(), n * sizeof(double));
  if (top_xy) { top_xy[0] = P.top_x; top_xy[1] = P.top_y; }
  if (top_h) *top_h = P.top_h;
  if (n_pushed) *n_pushed = P.pushes;
  return rc;
}

// reconstructPath(), src/visibilityBasedSolver.cpp:1183-1213.  Follows
// t = cameFrom(x,y) -> (x,y) = lightSources[t] until t repeats, reverses, sums
// eval_d.  path_xy receives the reversed path (start first); returns its length
// in points through *n_path (capacity cap points) and the distance.
double vhp_oracle_reconstruct_path(const uint64_t* came_from, const int32_t* pivots_xy, int nx, int ny,
                                   int end_x, int end_y, int32_t* path_xy, uint32_t cap, uint32_t* n_path) {
  (void)ny;
  std::vector<std::pair<int, int>> path;
  int x = end_x, y = end_y;
  double t = (double)came_from[(size_t)x + (size_t)y * nx];
  double t_old = std::numeric_limits<double>::max();
  while (t != t_old) {
    path.push_back({x, y});
    t_old = t;
    if (t >= 1e15) break;  // unlabelled: the reference would index out of range here
    x = pivots_xy[2 * (size_t)t];
    y = pivots_xy[2 * (size_t)t + 1];
    t = (double)came_from[(size_t)x + (size_t)y * nx];
  }
  path.push_back({x, y});
  std::reverse(path.begin(), path.end());
  double total = 0;
  for (size_t i = 0; i + 1 < path.size(); ++i)
    total += eval_d(path[i].first, path[i].second, path[i + 1].first, path[i + 1].second);
  if (n_path) *n_path = (uint32_t)path.size();
  if (path_xy)
    for (size_t i = 0; i < path.size() && i < cap; ++i) {
      path_xy[2 * i] = path[i].first;
      path_xy[2 * i + 1] = path[i].second;
    }
  return total;
}

// generateNewEnvironmentFromSettings(), src/environment.cpp:40-88: srand(seed),
// then per obstacle four rand() calls in the order col_1, width, row_1, height.
// Writes the occupancy complement (1 free / 0 blocked) as uint8.
int vhp_oracle_generate_env(int nx, int ny, uint64_t nb_obstacles, uint64_t min_w, uint64_t max_w,
                            uint64_t min_h, uint64_t max_h, int seed, uint8_t* occ) {
  if (!occ || nx <= 0 || ny <= 0) return 1;
  std::memset(occ, 1, (size_t)nx * ny);
  std::srand(seed);
  const size_t unx = nx, uny = ny;
  for (uint64_t i = 0; i < nb_obstacles; ++i) {
    int col_1 = 1 + (int)(std::rand() % (unx - 0 + 1));
    int col_2 = (int)(col_1 + min_w + (std::rand() % (max_w - min_w + 1)));
    col_1 = std::min(col_1, nx - 1);
    col_2 = std::min(col_2, nx - 1);
    int row_1 = 1 + (int)(std::rand() % (uny - 0 + 1));
    int row_2 = (int)(row_1 + min_h + (std::rand() % (max_h - min_h + 1)));
    row_1 = std::min(row_1, ny - 1);
    row_2 = std::min(row_2, ny - 1);
    for (int j = col_1; j < col_2; ++j)
      for (int k = row_1; k < row_2; ++k) occ[(size_t)j + (size_t)k * nx] = 0;
  }
  return 0;
}

// raycasting(), src/visibilityBasedSolver.cpp:267-290, over all targets as in
// benchmark() (:228-232): Bresenham from the source; a blocked cell on the way
// zeroes both that cell and the target.  ray must be pre-filled with 1.0 (:45).
int vhp_oracle_raycast_all(const uint8_t* occ, int nx, int ny, int sx, int sy, double* ray) {
  if (!occ || !ray || nx <= 0 || ny <= 0) return 1;
  for (int tx = 0; tx < nx; ++tx)
    for (int ty = 0; ty < ny; ++ty) {
      int x0 = sx, y0 = sy;
      const int x1 = tx, y1 = ty;
      const int dx = std::abs(x1 - x0), dy = std::abs(y1 - y0);
      const int stepx = (x0 < x1) ? 1 : -1, stepy = (y0 < y1) ? 1 : -1;
      int err = dx - dy;
      while (x0 != x1 || y0 != y1) {
        if (occ[(size_t)x0 + (size_t)y0 * nx] == 0) {
          ray[(size_t)x0 + (size_t)y0 * nx] = 0;
          ray[(size_t)x1 + (size_t)y1 * nx] = 0;
          break;
        }
        const int e2 = 2 * err;
        if (e2 > -dy) { err -= dy; x0 += stepx; }
        if (e2 < dx) { err += dx; y0 += stepy; }
      }
    }
  return 0;
}

// Timing harness for bench.py's cpu_baseline leg.  Protocol of benchmark()
// (src/visibilityBasedSolver.cpp:217-222): a steady clock around computeVisibility()
// only, buffers allocated beforehand.  n_threads > 1 runs one independent source per
// thread over a shared read-only map (the reference itself is single-threaded).
// Returns wall seconds for all n_src sweeps; per_sweep_best receives the fastest
// single sweep.
double vhp_oracle_time_sweeps(const uint8_t* occ, int nx, int ny, const int32_t* src_xy, int n_src,
                              int n_threads, double* per_sweep_best) {
  if (!occ || !src_xy || n_src <= 0 || n_threads <= 0) return -1.0;
  const Grid g = make_grid(occ, nx, ny);
  std::vector<std::vector<double>> bufs(n_threads, std::vector<double>((size_t)nx * ny, 0.0));
  std::vector<double> best(n_threads, 1e30);
  auto worker = [&](int t) {
    for (int s = t; s < n_src; s += n_threads) {
      const auto t0 = std::chrono::steady_clock::now();
      four_quadrants(g, bufs[t].data(), src_xy[2 * s], src_xy[2 * s + 1], [](size_t, size_t, double) {});
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      best[t] = std::min(best[t], dt);
    }
  };
  const auto w0 = std::chrono::steady_clock::now();
  if (n_threads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
  }
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  if (per_sweep_best) *per_sweep_best = *std::min_element(best.begin(), best.end());
  return wall;
}

}  // extern "C"
