// latbench.hip -- dependent-issue latency of the instructions on the sweep's critical chain (gfx950).
// One workgroup per CU-ish; `waves` wavefronts per workgroup, all on... (waves go to SIMDs round robin: 4 = one per SIMD,
// 8 = two per SIMD, 16 = four per SIMD).  Prints cycles per instruction of each chain for wavefront 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double dpp_shr(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, int n) {
  double a = out[threadIdx.x], b = out[threadIdx.x + 64], c = 0.999 + 1e-9 * threadIdx.x;
  int m = (threadIdx.x & 1) ? -1 : 0x7fffffff;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) { a = a + b; }                                   // dependent v_add_f64
      if (MODE == 1) { a = a * c; }                                   // dependent v_mul_f64
      if (MODE == 2) { a = __builtin_fma(a, c, b); }                  // dependent v_fma_f64
      if (MODE == 3) { a = dpp_shr(a, b); }                           // dependent pair of v_mov_dpp
      if (MODE == 4) { a = __hiloint2double(__double2hiint(a) & m, __double2loint(a) & m); asm volatile("" : "+v"(a)); }  // dependent v_and pair
      if (MODE == 5) {  // the sweep's step: dpp -> sub -> mul -> sub -> and
        const double bb = dpp_shr(a, b);
        const double t = a - bb;
        const double uu = c * t;
        double v = a - uu;
        v = __hiloint2double(__double2hiint(v) & m, __double2loint(v) & m);
        a = v;
      }
      if (MODE == 6) { a = a + b; c = c * 1.0000001; b = b + 1.0; }    // three independent chains
      if (MODE == 7) {  // the step with the ratio's three f64 ops beside it
        const double q = b * c; const double r = __builtin_fma(-a, q, b); const double cc = __builtin_fma(r, c, q);
        const double bb = dpp_shr(a, b);
        const double t = a - bb;
        const double uu = cc * t;
        double v = a - uu;
        v = __hiloint2double(__double2hiint(v) & m, __double2loint(v) & m);
        a = v;
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = a + b + c;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
double run(int waves, int blocks, double* d, unsigned long long* dc, int n, int ops) {
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves), 0, 0, d, dc, n);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
  return (double)h[0] / ((double)n * 8 * ops);
}
int main() {
  double* d; unsigned long long* dc;
  hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22); hipMalloc(&dc, 1 << 16);
  const int n = 2000;
  const char* names[] = {"v_add_f64 chain", "v_mul_f64 chain", "v_fma_f64 chain", "dpp pair chain (per pair)", "v_and pair chain (per pair)",
                         "sweep step (per step, 7 instr)", "3 independent f64 (per instr)", "step + ratio (per step, 10 instr)"};
  for (int waves : {1, 4, 8, 16}) {
    printf("waves per workgroup %2d (=%d per SIMD), 256 workgroups:\n", waves, (waves + 3) / 4);
    double r[8];
    r[0] = run<0>(waves, 256, d, dc, n, 1); r[1] = run<1>(waves, 256, d, dc, n, 1); r[2] = run<2>(waves, 256, d, dc, n, 1);
    r[3] = run<3>(waves, 256, d, dc, n, 1); r[4] = run<4>(waves, 256, d, dc, n, 1); r[5] = run<5>(waves, 256, d, dc, n, 1);
    r[6] = run<6>(waves, 256, d, dc, n, 3); r[7] = run<7>(waves, 256, d, dc, n, 1);
    for (int i = 0; i < 8; ++i) printf("   %-36s %.1f cycles\n", names[i], r[i]);
  }
  return 0;
}
