// placebench.hip -- does the speed of a store PATTERN depend on which allocation it writes?  (gfx950; diagnostic)
// Six 2 GB buffers from hipMalloc; on each: the sweep's two store patterns, two variants of the x-major pattern that
// touch fewer / more rows per instruction, and a plain fill.  512 workgroups of 8 wavefronts, 40 KB of dynamic LDS (4 per CU).
//   y    : a wavefront stores 1 KB of one row per instruction, row after row            (1 row  per instruction)
//   x8   : 8 rows x 128 B per instruction, rows 2 pitches (16 KB) apart, marching in x  (8 rows per instruction)
//   x4   : 4 rows x 256 B per instruction, adjacent rows                                (4 rows per instruction)
//   x16  : 16 rows x 64 B per instruction, adjacent rows                                (16 rows per instruction)
//   fill : consecutive 1 KB per instruction, grid-stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NX = 1000, NY = 1000;
template <int MODE>
__global__ void __launch_bounds__(512) pat(double* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* field = out + (size_t)(blockIdx.x >> 1) * NX * NY + (size_t)(blockIdx.x & 1) * 500 * NX;
  const double v0 = 1.0 + lane, v1 = 2.0 + lane;
  if (MODE == 0) {
    const int x0 = 128 * wave + 2 * lane;
    if (x0 + 1 < NX)
      for (int y = 0; y < 500; ++y) *reinterpret_cast<double2*>(field + (size_t)y * NX + x0) = make_double2(v0, v1);
  } else if (MODE == 1) {
    const int rslot = lane >> 3, pc = lane & 7;
    for (int xw = 0; xw + 16 <= NX + 8; xw += 8) {
      const int cls = (xw >> 3) & 1;
      const int xa = xw - 8 * cls >= 0 ? xw - (cls ? 8 : 0) : 0;
      for (int u = 0; u < 4; ++u) {
        const int r = 64 * wave + cls + 2 * (rslot + 8 * u);
        const int x = (xa & ~15) + 8 * ((r & 1) ? 1 : 0) + 2 * pc;
        if (r < 500 && x + 1 < NX && x >= 0) *reinterpret_cast<double2*>(field + (size_t)r * NX + x) = make_double2(v0, v1);
      }
    }
  } else if (MODE == 3) {  // 4 adjacent rows x 32 columns per instruction
    const int rslot = lane >> 4, pc = lane & 15;
    for (int xw = 0; xw + 32 <= NX + 24; xw += 32)
      for (int u = 0; u < 16; ++u) {
        const int r = 64 * wave + rslot + 4 * u;
        const int x = xw + 2 * pc;
        if (r < 500 && x + 1 < NX) *reinterpret_cast<double2*>(field + (size_t)r * NX + x) = make_double2(v0, v1);
      }
  } else if (MODE == 4) {  // 16 adjacent rows x 8 columns per instruction
    const int rslot = lane >> 2, pc = lane & 3;
    for (int xw = 0; xw + 8 <= NX; xw += 8)
      for (int u = 0; u < 4; ++u) {
        const int r = 64 * wave + rslot + 16 * u;
        const int x = xw + 2 * pc;
        if (r < 500 && x + 1 < NX) *reinterpret_cast<double2*>(field + (size_t)r * NX + x) = make_double2(v0, v1);
      }
  } else {
    double2* p = reinterpret_cast<double2*>(out);
    const size_t n2 = (size_t)256 * NX * NY / 2;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n2; k += (size_t)gridDim.x * blockDim.x) p[k] = make_double2(v0, v1);
  }
}
template <int MODE>
float run(double* d, int grid) {
  const int lds = 40000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(pat<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pat<MODE>, dim3(grid), dim3(512), lds, 0, d);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(pat<MODE>, dim3(grid), dim3(512), lds, 0, d);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / 10;
}
int main(int argc, char** argv) {
  const int nbuf = argc > 1 ? atoi(argv[1]) : 6;
  const bool one_allocation = argc > 2;  // any second argument: the buffers are 2 GB windows of ONE allocation
  double* d[16];
  const size_t each = (size_t)256 * NX * NY * 8 + 4096;
  if (one_allocation) {
    char* big = nullptr;
    if (hipMalloc(&big, each * nbuf) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    for (int k = 0; k < nbuf; ++k) d[k] = reinterpret_cast<double*>(big + each * k);
    printf("one allocation of %zu bytes at %p, %d windows\n", each * nbuf, (void*)big, nbuf);
  } else {
    for (int k = 0; k < nbuf; ++k) {
      if (hipMalloc(&d[k], each) != hipSuccess) { printf("hipMalloc %d failed\n", k); return 1; }
      printf("buffer %d at %p\n", k, (void*)d[k]);
    }
  }
  for (int rep = 0; rep < 2; ++rep)
    for (int k = 0; k < nbuf; ++k)
      printf("rep %d buffer %d:  y %.3f   x8 %.3f   x4 %.3f   x16 %.3f   fill %.3f  ms per 2 GB\n", rep, k, run<0>(d[k], 512), run<1>(d[k], 512),
             run<3>(d[k], 512), run<4>(d[k], 512), run<2>(d[k], 2048));
  return 0;
}
