// vhp_oracle_matlab.cpp -- CPU restatement of the MATLAB-flavoured variants of the sweep and of the planner.
// TEST INFRASTRUCTURE ONLY (same rules as vhp_oracle.cpp: only tests/ may load it).
//
// PARITY: UNPINNED AGAINST MATLAB.  There is no MATLAB or Octave in this image, the reference ships no numeric output
// of these scripts, and its C++ program does not implement them (its sweep has no i == j rule, no alpha, no fac:
// SURVEY Q1).  This file restates, line for line, what the .m sources say; the GPU variant is checked against it.
//
//   sweep   : MATLAB_code/visibility/getAccessibilityMap.m:1-129 (decay `alpha`, curve factor `fac`, the explicit
//             diagonal rule `i == j*fac`), identical in structure to getAccessibilityMapPlanner.m with fac = 1;
//   planner : MATLAB_code/c_sample_planner_solving_random_environments.m:100-171 over
//             getAccessibilityMapPlanner.m:1-150 (first-lit labelling at v >= threshold, min-max-scaled heuristic,
//             stop when the NEW waypoint's own field sees the target).
//
// MATLAB arrays are 1-based and indexed (x, y) with x the first (fastest) index; here cells are 0-based and stored
// x + y*nx, which is the same linear order.  1-based loops `for i = 0:max_col` with currentX = lightPos(1) + i cover
// x up to nx (1-based) = nx-1 (0-based); `for i = 0:max_col-1` with currentX = lightPos(1) - i reach x = 1 (1-based) =
// 0 (0-based): unlike the C++ program every row and column is swept.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct G {
  int nx, ny;
  const uint8_t* occ;
  inline size_t at(int x, int y) const { return (size_t)x + (size_t)y * nx; }
};

// one quadrant nest of getAccessibilityMap.m (:10-37, :39-66, :67-95, :97-127).  (dirx, diry) = direction of the
// nest; ni / nj = number of i / j values the MATLAB loop runs over.  `visit(x, y, v)` after the cell is stored.
template <class Visit>
void nest(const G& g, double* a, int sx, int sy, int dirx, int diry, int ni, int nj, double alpha, double light, double fac, Visit&& visit) {
  for (int i = 0; i < ni; ++i) {
    const int x = sx + dirx * i, xm = x - dirx;
    for (int j = 0; j < nj; ++j) {
      const int y = sy + diry * j, ym = y - diry;
      double v;
      if (i == 0 && j == 0) {
        v = light;                                                        // :16-17
      } else if (i == 0) {
        v = alpha * a[g.at(x, ym)];                                       // :18-19
      } else if (j == 0) {
        v = alpha * a[g.at(xm, y)];                                       // :20-21
      } else if ((double)i == (double)j * fac) {
        v = alpha * a[g.at(xm, ym)];                                      // :22-23  the proper diagonal
      } else if ((double)i > (double)j * fac) {
        // :24-27.  In MATLAB c carries the sign of the two coordinate differences and the update is written with
        // "+ c" in the nests where that sign is negative: a + (-|c|)*(a-b) and a - |c|*(a-b) are the same IEEE value.
        const double c = ((double)j * fac) / (double)i;
        const double p = a[g.at(xm, y)];
        const double f = p - c * (p - a[g.at(xm, ym)]);
        v = alpha * f;
      } else {
        const double c = (double)i / ((double)j * fac);                   // :28-31
        const double p = a[g.at(x, ym)];
        const double f = p - c * (p - a[g.at(xm, ym)]);
        v = alpha * f;
      }
      v = v * (g.occ[g.at(x, y)] ? 1.0 : 0.0);                            // :33
      a[g.at(x, y)] = v;
      visit(x, y, v);
    }
  }
}

template <class Visit>
void four(const G& g, double* a, int sx, int sy, double alpha, double light, double fac, Visit&& visit) {
  nest(g, a, sx, sy, +1, +1, g.nx - sx, g.ny - sy, alpha, light, fac, visit);  // 1: i = 0:nx-lightPos(1), j = 0:ny-lightPos(2)
  nest(g, a, sx, sy, -1, +1, sx + 1, g.ny - sy, alpha, light, fac, visit);     // 2: i = 0:lightPos(1)-1 (1-based) = sx+1 values
  nest(g, a, sx, sy, -1, -1, sx + 1, sy + 1, alpha, light, fac, visit);        // 3
  nest(g, a, sx, sy, +1, -1, g.nx - sx, sy + 1, alpha, light, fac, visit);     // 4
}

}  // namespace

extern "C" {

// getAccessibilityMap(alpha, lightStrength, lightPos, obstacle, accessibilityMap, fac): `vis` is fully overwritten.
int vhp_oracle_sweep_matlab(const uint8_t* occ, int nx, int ny, int sx, int sy, double alpha, double fac, double* vis) {
  if (!occ || !vis || nx <= 0 || ny <= 0 || !(fac > 0)) return 1;
  if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return 2;
  G g{nx, ny, occ};
  four(g, vis, sx, sy, alpha, 1.0, fac, [](int, int, double) {});
  return 0;
}

// c_sample_planner_solving_random_environments.m:100-171 with getAccessibilityMapPlanner (fac = 1).
// label: 0-based index of the waypoint that first lit the cell (lightSource_enum - 1), UINT64 max where NaN.
// waypoints: (x, y) pairs, waypoints[0] = start; *n_way = number of waypoints (the target is not appended).
// Returns 0 solved, 20 max_iter reached (the script loops forever), 3 nothing above the threshold / degenerate scale.
int vhp_oracle_planner_matlab(const uint8_t* occ, int nx, int ny, int start_x, int start_y, int end_x, int end_y,
                              double threshold, double alpha, uint64_t max_iter, uint64_t* label, double* map_builder,
                              double* local, int32_t* waypoints, uint32_t* n_way) {
  if (!occ || nx <= 0 || ny <= 0) return 1;
  G g{nx, ny, occ};
  const size_t n = (size_t)nx * ny;
  const uint64_t none = std::numeric_limits<uint64_t>::max();
  std::vector<uint64_t> lab(n, none);
  std::vector<double> loc(n, 1.0), uni(n);
  std::vector<int32_t> way;
  uint64_t iter = 0;  // 0-based (MATLAB iter - 1)
  lab[g.at(start_x, start_y)] = 0;                                                         // :48-51 of f_comparison / c_sample
  auto sweep = [&](int wx, int wy) {
    four(g, loc.data(), wx, wy, alpha, 1.0, 1.0, [&](int x, int y, double v) {
      if (v >= threshold && lab[g.at(x, y)] == none) lab[g.at(x, y)] = iter;               // Planner.m:30-34
    });
  };
  way.push_back(start_x); way.push_back(start_y);
  sweep(start_x, start_y);
  uni = loc;                                                                                // map_builder = accessibilityMap_s
  int status = 0;
  int px = start_x, py = start_y;                                                          // wp_old
  if (!(loc[g.at(end_x, end_y)] > threshold)) {                                            // :113-119 trivial solution otherwise
    for (;;) {
      // :122-144  candidates = find(map_builder > threshold), heuristic with min-max scaled visibility
      double vmin = 0, vmax = 0, dmin = 0, dmax = 0;
      bool any = false;
      for (size_t k = 0; k < n; ++k) {
        if (!(uni[k] > threshold)) continue;
        const int x = (int)(k % nx), y = (int)(k / nx);
        const double dt = std::sqrt((double)(x - end_x) * (x - end_x) + (double)(y - end_y) * (y - end_y)) +
                          std::sqrt((double)(x - px) * (x - px) + (double)(y - py) * (y - py));
        if (!any) { vmin = vmax = uni[k]; dmin = dmax = dt; any = true; }
        else { vmin = std::fmin(vmin, uni[k]); vmax = std::fmax(vmax, uni[k]); dmin = std::fmin(dmin, dt); dmax = std::fmax(dmax, dt); }
      }
      if (!any || !(vmax > vmin)) { status = 3; break; }
      double best = 0;
      long long best_k = -1;
      for (size_t k = 0; k < n; ++k) {  // linear-index order = MATLAB's find() order; min() keeps the first minimum
        if (!(uni[k] > threshold)) continue;
        const int x = (int)(k % nx), y = (int)(k / nx);
        const double dt = std::sqrt((double)(x - end_x) * (x - end_x) + (double)(y - end_y) * (y - end_y)) +
                          std::sqrt((double)(x - px) * (x - px) + (double)(y - py) * (y - py));
        const double vs = (dmax - dmin) * (uni[k] - vmin) / (vmax - vmin) + dmin;           // :141
        const double fun = vs + dt;                                                         // :144
        if (best_k < 0 || fun < best) { best = fun; best_k = (long long)k; }
      }
      px = (int)(best_k % nx); py = (int)(best_k / nx);                                    // :147
      ++iter;
      way.push_back(px); way.push_back(py);
      if (iter > max_iter) { status = 20; break; }
      sweep(px, py);                                                                        // :157
      for (size_t k = 0; k < n; ++k) uni[k] = std::fmax(uni[k], loc[k]);                    // :158
      if (loc[g.at(end_x, end_y)] > threshold) break;                                       // :166
    }
  }
  if (label) std::memcpy(label, lab.data(), n * sizeof(uint64_t));
  if (map_builder) std::memcpy(map_builder, uni.data(), n * sizeof(double));
  if (local) std::memcpy(local, loc.data(), n * sizeof(double));
  if (waypoints) std::memcpy(waypoints, way.data(), way.size() * sizeof(int32_t));
  if (n_way) *n_way = (uint32_t)(way.size() / 2);
  return status;
}

}  // extern "C"
