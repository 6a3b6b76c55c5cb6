// Micro-benchmark of the sweep's per-step body on ONE wavefront: cycles per step for
// variants of the dependent chain.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../visibility-heuristic-path-planner_amd/csrc/vhp_sweep.hip.h"
using namespace vhp;

template <int MODE, int R>
__global__ void body(double* out, unsigned long long* cyc, int steps, uint64_t occbits) {
  const int lane = threadIdx.x & 63;
  double prev[R], jd[R];
  for (int r = 0; r < R; ++r) { prev[r] = 1.0 - 1e-3 * lane; jd[r] = (double)(64 * r + lane); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 1000; i < 1000 + steps; ++i) {
    const double di = (double)i;
    const double ri = __builtin_amdgcn_rcp(di);
    double fill = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double a = prev[r];
      double b;
      if (MODE & 1) b = shift_up(a, fill); else b = a * 0.5;
      double c;
      if (MODE & 2) c = ratio(jd[r], di, ri); else c = jd[r] * ri;
      double v = stencil(a, b, c);
      if (MODE & 4) v = and_mask(v, bit_mask(occbits + lane, i & 63));
      if (MODE & 8) fill = read_lane(a, 63);
      prev[r] = v;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int r = 0; r < R; ++r) s += prev[r];
  out[threadIdx.x & 63] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int R> void run(const char* name, double* d, unsigned long long* c, int waves = 1) {
  const int steps = 4000;
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((body<MODE, R>), dim3(1), dim3(64 * waves), 0, 0, d, c, steps, ~0ull);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-44s R=%d waves/CU=%2d: %7.1f cycles/step  %6.1f cycles/row-step (per wave); SIMD cycles per row-step %.1f\n", name, R, waves,
         (double)h / steps, (double)h / steps / R, (double)h / steps / R / (waves > 4 ? waves / 4.0 : 1.0));
}

// every CU loaded: `blocks_per_cu` blocks of `waves` wavefronts on each of the 256 CUs
template <int MODE, int R> void run_chip(const char* name, double* d, unsigned long long* c, int waves, int blocks_per_cu) {
  const int steps = 4000, nb = 256 * blocks_per_cu;
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((body<MODE, R>), dim3(nb), dim3(64 * waves), 0, 0, d, c, steps, ~0ull);
  hipDeviceSynchronize();
  static unsigned long long h[4096];
  hipMemcpy(h, c, 8 * nb, hipMemcpyDeviceToHost);
  double avg = 0; for (int b = 0; b < nb; ++b) avg += (double)h[b] / nb;
  const double wps = waves * blocks_per_cu / 4.0;
  printf("%-28s R=%d waves/SIMD=%4.1f: %7.1f cycles/row-step per wave; SIMD cycles per row-step %.1f\n", name, R, wps, avg / steps / R,
         avg / steps / R / wps);
}

int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 4096); hipMalloc(&c, 8 * 4096);
  run<0, 1>("stencil only (3 dep fp64)", d, c);
  run<2, 1>("+ratio (3 more fp64, independent of chain)", d, c);
  run<1, 1>("stencil + dpp shift", d, c);
  run<3, 1>("stencil + dpp + ratio", d, c);
  run<7, 1>("stencil + dpp + ratio + gate", d, c);
  run<15, 1>("all (+lane63 readlane)", d, c);
  run<15, 2>("all", d, c);
  run<15, 4>("all", d, c);
  run<7, 2>("no readlane", d, c);
  run<7, 4>("no readlane", d, c);
  run<0, 4>("stencil only", d, c);
  for (int w : {4, 8, 16}) for (int b : {1, 2}) { run_chip<7, 2>("chip: no readlane", d, c, w, b); run_chip<15, 2>("chip: all", d, c, w, b); }
  for (int w : {4, 8, 16}) { run<0, 4>("stencil only", d, c, w); run<7, 2>("no readlane", d, c, w); run<15, 2>("all", d, c, w); }
  return 0;
}
