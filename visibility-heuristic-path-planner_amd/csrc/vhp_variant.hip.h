// vhp_variant.hip.h -- the MATLAB-flavoured variants of the sweep and of the planner, gfx950.
//
// The reference's MATLAB demos differ from its C++ program: getAccessibilityMap.m has an explicit diagonal rule
// (`i == j*fac`: the cell takes alpha * its diagonal predecessor, instead of C++'s stale value), a decay factor
// `alpha` on every update, a curve factor `fac` that tilts the octant boundary, and it sweeps every row and column
// (1-based loops, no SURVEY Q2); c_sample_planner_solving_random_environments.m:122-147 picks the next waypoint with a
// min-max-scaled heuristic.  These are optional modes: correctness first, one workgroup per quadrant.
//
// Sweep: a cell (i, j) reads (i-1, j), (i, j-1) and (i-1, j-1), whatever `fac` is, so anti-diagonals d = i + j are
// fronts: the workgroup sweeps them one after the other with the last two fronts in LDS (indexed by i).
// Parity: against oracle/vhp_oracle_matlab.cpp (a line-by-line restatement of the .m files; UNPINNED against MATLAB
// itself, which is not available here).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vhp {
namespace variant {

constexpr uint32_t kNone = 0xffffffffu;

// getAccessibilityMap.m:10-127.  grid = 4 * n_src workgroups; dynamic LDS = 3 * (max(nx, ny) + 1) doubles.
__global__ void __launch_bounds__(1024) vhp_variant_sweep(int nx, int ny, const uint8_t* __restrict__ occ, const int32_t* __restrict__ src_xy,
                                                          double* __restrict__ out, long long field_stride, double alpha, double fac,
                                                          int* __restrict__ err_flag) {
  extern __shared__ double fronts[];
  const int s = blockIdx.x >> 2, q = blockIdx.x & 3;
  const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
  if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) {
    if (threadIdx.x == 0 && q == 0) atomicOr(err_flag, 1);
    return;
  }
  const int dirx = (q == 0 || q == 3) ? 1 : -1, diry = q < 2 ? 1 : -1;
  const int ni = dirx > 0 ? nx - sx : sx + 1, nj = diry > 0 ? ny - sy : sy + 1;  // every row and column is swept
  const int L = (nx > ny ? nx : ny) + 1;
  double* field = out + (size_t)s * field_stride;
  for (int d = 0; d <= ni + nj - 2; ++d) {
    double* cur = fronts + (d % 3) * L;
    const double* p1 = fronts + ((d + 2) % 3) * L;  // front d-1
    const double* p2 = fronts + ((d + 1) % 3) * L;  // front d-2
    const int i_lo = d - (nj - 1) > 0 ? d - (nj - 1) : 0, i_hi = d < ni - 1 ? d : ni - 1;
    for (int i = i_lo + (int)threadIdx.x; i <= i_hi; i += blockDim.x) {
      const int j = d - i;
      const int x = sx + dirx * i, y = sy + diry * j;
      double v;
      if (i == 0 && j == 0) v = 1.0;                                  // lightStrength
      else if (i == 0) v = alpha * p1[i];                             // (x, y -+ 1)
      else if (j == 0) v = alpha * p1[i - 1];                         // (x -+ 1, y)
      else if ((double)i == (double)j * fac) v = alpha * p2[i - 1];   // the proper diagonal
      else if ((double)i > (double)j * fac) {
        const double c = ((double)j * fac) / (double)i;
        const double a = p1[i - 1];
        const double f = a - c * (a - p2[i - 1]);
        v = alpha * f;
      } else {
        const double c = (double)i / ((double)j * fac);
        const double a = p1[i];
        const double f = a - c * (a - p2[i - 1]);
        v = alpha * f;
      }
      v = v * (occ[(size_t)y * nx + x] ? 1.0 : 0.0);
      cur[i] = v;
      field[(size_t)y * nx + x] = v;
    }
    __syncthreads();
  }
}

// computeVisibility() (reference src/visibilityBasedSolver.cpp:570-696) with its local `offset` (:573, added to both
// operands of every c_: :590-591 ... :686-687) as a parameter.  It is 0.0 at the reference's HEAD -- that case is what
// the tuned kernels compute -- but the reference's own published Samples/SFMLstandAloneVisibility.png was rendered by a
// build with offset = 1, and this kernel is how the library reproduces that image pixel for pixel
// (tests/test_gpu_golden_samples.py).  C++ semantics, not MATLAB's: the diagonal inherits the stored (x, y -+ 1)
// (on front d = 2i that is p1[i]), quadrants 2-4 stop short of column/row 0 (Q2), cells nobody writes
// keep the caller's zero fill.  Same anti-diagonal organisation as vhp_variant_sweep; correctness first.
__global__ void __launch_bounds__(1024) vhp_offset_sweep(int nx, int ny, const uint8_t* __restrict__ occ, const int32_t* __restrict__ src_xy,
                                                         double* __restrict__ out, long long field_stride, double offset,
                                                         int* __restrict__ err_flag) {
  extern __shared__ double fronts[];
  const int s = blockIdx.x >> 2, q = blockIdx.x & 3;
  const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
  if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) {
    if (threadIdx.x == 0 && q == 0) atomicOr(err_flag, 1);
    return;
  }
  const int dirx = (q == 0 || q == 3) ? 1 : -1, diry = q < 2 ? 1 : -1;
  const int ni = dirx > 0 ? nx - sx : sx, nj = diry > 0 ? ny - sy : sy;  // max_x_, max_y_ of the four nests (:576-577, 607-608, 637-638, 667-668)
  if (ni == 0 || nj == 0) return;
  const int L = (nx > ny ? nx : ny) + 1;
  double* field = out + (size_t)s * field_stride;
  for (int d = 0; d <= ni + nj - 2; ++d) {
    double* cur = fronts + (d % 3) * L;
    const double* p1 = fronts + ((d + 2) % 3) * L;  // front d-1
    const double* p2 = fronts + ((d + 1) % 3) * L;  // front d-2
    const int i_lo = d - (nj - 1) > 0 ? d - (nj - 1) : 0, i_hi = d < ni - 1 ? d : ni - 1;
    for (int i = i_lo + (int)threadIdx.x; i <= i_hi; i += blockDim.x) {
      const int j = d - i;
      const int x = sx + dirx * i, y = sy + diry * j;
      double v;
      if (i == 0 && j == 0) v = 1.0;
      else if (i == 0) v = p1[i];
      else if (j == 0) v = p1[i - 1];
      else if (i > j) {
        const double c = ((double)j + offset) / ((double)i + offset);
        const double a = p1[i - 1];
        const double t = a - p2[i - 1];
        const double u = c * t;
        v = a - u;
      } else if (j > i) {
        const double c = ((double)i + offset) / ((double)j + offset);
        const double a = p1[i];
        const double t = a - p2[i - 1];
        const double u = c * t;
        v = a - u;
      } else v = p1[i];                                               // i == j: the running v of the inner loop, i.e. (x, y -+ 1) as stored
      v = v * (occ[(size_t)y * nx + x] ? 1.0 : 0.0);
      cur[i] = v;
      // quadrants overlap on the axes with identical values; a later nest of the reference overwrites an
      // earlier one with the same number, so concurrent stores of equal bits are the same result
      field[(size_t)y * nx + x] = v;
    }
    __syncthreads();
  }
}

struct PlannerCtl {
  int n_way;    // waypoints so far (waypoints[0] = start)
  int done;
  int status;
  int iter;     // 0-based index of the current waypoint
};

// after a sweep from waypoint `iter`: map_builder = max(map_builder, local); first-lit labelling at v >= threshold
// (getAccessibilityMapPlanner.m:30-34, c_sample...m:158)
__global__ void vhp_variant_update(const double* __restrict__ local, double* __restrict__ uni, uint32_t* __restrict__ label, size_t cells,
                                   double threshold, const PlannerCtl* __restrict__ ctl) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cells || ctl->done) return;
  const double v = local[k];
  uni[k] = fmax(uni[k], v);
  if (v >= threshold && label[k] == kNone) label[k] = (uint32_t)ctl->iter;
}

// stop test (c_sample...m:113,166): the NEW waypoint's own field sees the target
__global__ void vhp_variant_check(const double* __restrict__ local, int nx, int end_x, int end_y, double threshold, PlannerCtl* ctl) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !ctl->done && local[(size_t)end_y * nx + end_x] > threshold) { ctl->done = 1; ctl->status = 0; }
}

struct Cand { double fun; unsigned long long k; };

// c_sample...m:122-147: over the cells with map_builder > threshold, d_tot = distance to the target + distance to the
// previous waypoint; visibility is min-max scaled onto the range of d_tot; next waypoint = first minimum (in linear
// index order, MATLAB's find/min) of scaled visibility + d_tot.  One workgroup.
__global__ void __launch_bounds__(1024) vhp_variant_pick(const double* __restrict__ uni, int nx, int ny, int end_x, int end_y, double threshold,
                                                         unsigned long long max_iter, int32_t* __restrict__ way, PlannerCtl* ctl) {
  __shared__ double s_vmin[16], s_vmax[16], s_dmin[16], s_dmax[16];
  __shared__ Cand s_best[16];
  __shared__ double b_vmin, b_vmax, b_dmin, b_dmax;
  __shared__ int b_any;
  if (ctl->done) return;
  const int px = way[2 * ctl->iter], py = way[2 * ctl->iter + 1];
  const size_t cells = (size_t)nx * ny;
  auto d_tot = [&](int x, int y) {
    return __builtin_sqrt((double)(x - end_x) * (x - end_x) + (double)(y - end_y) * (y - end_y)) +
           __builtin_sqrt((double)(x - px) * (x - px) + (double)(y - py) * (y - py));
  };
  double vmin = 1e300, vmax = -1e300, dmin = 1e300, dmax = -1e300;
  for (size_t k = threadIdx.x; k < cells; k += blockDim.x) {
    const double v = uni[k];
    if (!(v > threshold)) continue;
    const double dt = d_tot((int)(k % nx), (int)(k / nx));
    vmin = fmin(vmin, v); vmax = fmax(vmax, v); dmin = fmin(dmin, dt); dmax = fmax(dmax, dt);
  }
  for (int o = 32; o >= 1; o >>= 1) {
    vmin = fmin(vmin, __shfl_xor(vmin, o)); vmax = fmax(vmax, __shfl_xor(vmax, o));
    dmin = fmin(dmin, __shfl_xor(dmin, o)); dmax = fmax(dmax, __shfl_xor(dmax, o));
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_vmin[wv] = vmin; s_vmax[wv] = vmax; s_dmin[wv] = dmin; s_dmax[wv] = dmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = s_vmin[0], b = s_vmax[0], c = s_dmin[0], d = s_dmax[0];
    for (int w = 1; w < 16; ++w) { a = fmin(a, s_vmin[w]); b = fmax(b, s_vmax[w]); c = fmin(c, s_dmin[w]); d = fmax(d, s_dmax[w]); }
    b_vmin = a; b_vmax = b; b_dmin = c; b_dmax = d;
    b_any = (b >= a) && (b > a);
  }
  __syncthreads();
  if (!b_any) {
    if (threadIdx.x == 0) { ctl->status = 3; ctl->done = 1; }  // VHP_ERR_NOTHING_LIT: no candidate, or a degenerate scale
    return;
  }
  Cand best{1e300, ~0ull};
  for (size_t k = threadIdx.x; k < cells; k += blockDim.x) {
    const double v = uni[k];
    if (!(v > threshold)) continue;
    const double dt = d_tot((int)(k % nx), (int)(k / nx));
    const double vs = (b_dmax - b_dmin) * (v - b_vmin) / (b_vmax - b_vmin) + b_dmin;
    const double fun = vs + dt;
    if (fun < best.fun || (fun == best.fun && k < best.k)) { best.fun = fun; best.k = k; }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const double f = __shfl_xor(best.fun, o);
    const unsigned long long kk = ((unsigned long long)(unsigned)__shfl_xor((int)(best.k >> 32), o) << 32) | (unsigned)__shfl_xor((int)best.k, o);
    if (f < best.fun || (f == best.fun && kk < best.k)) { best.fun = f; best.k = kk; }
  }
  if ((threadIdx.x & 63) == 0) s_best[wv] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    Cand b = s_best[0];
    for (int w = 1; w < 16; ++w)
      if (s_best[w].fun < b.fun || (s_best[w].fun == b.fun && s_best[w].k < b.k)) b = s_best[w];
    const int it = ctl->iter + 1;
    ctl->iter = it;
    way[2 * it] = (int)(b.k % nx);
    way[2 * it + 1] = (int)(b.k / nx);
    ctl->n_way = it + 1;
    if ((unsigned long long)it > max_iter) { ctl->status = 20; ctl->done = 1; }  // VHP_ERR_MAX_ITER (the script has no bound)
  }
}

__global__ void vhp_variant_labels_to_u64(const uint32_t* __restrict__ lab, unsigned long long* __restrict__ out, size_t n, unsigned long long none) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = lab[k] == kNone ? none : (unsigned long long)lab[k];
}

}  // namespace variant
}  // namespace vhp
