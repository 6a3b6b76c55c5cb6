// patbench.hip -- store-only ceilings of the sweep's two store patterns at the sweep's occupancy (gfx950).
//   mode 0: "y-major": a wavefront stores 1 KB of one row per instruction (16 B per lane), row after row
//   mode 1: "x-major": a wavefront stores 8 rows x 128 B per instruction (rows 2 pitches apart), marching along x
//   mode 2: a plain fill (consecutive 1 KB per instruction, grid-stride)
// Each workgroup (8 wavefronts) writes a 512-row x 1000-column slab of one 1000x1000 field; 512 workgroups = 256 fields x 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NX = 1000, NY = 1000;
template <int MODE>
__global__ void __launch_bounds__(512) pat(double* out, int gap) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* field = out + (size_t)(blockIdx.x >> 1) * NX * NY + (size_t)(blockIdx.x & 1) * 500 * NX;
  const double v0 = 1.0 + lane, v1 = 2.0 + lane;
  if (MODE == 0) {  // wave owns columns [128*wave', ...): 7 full strips of 128 + ragged -> use 7 waves x 128 cols, wave 7 does cols 896..999 (104)
    const int x0 = 128 * wave + 2 * lane;
    if (x0 + 1 < NX)
      for (int y = 0; y < 500; ++y) {
        double2* p = reinterpret_cast<double2*>(field + (size_t)y * NX + x0);
        *p = make_double2(v0, v1);
        for (int g = 0; g < gap; ++g) asm volatile("v_mov_b32 %0, %0" : "+v"(x0 ? *(int*)&lds[0] : *(int*)&lds[1]) ::);
      }
  } else if (MODE == 1) {  // wave owns rows [64*wave .. +63] (of the 500; waves 0..7 -> 512 rows, clip), K=2 classes
    const int rslot = lane >> 3, pc = lane & 7;
    for (int xw = 0; xw + 16 <= NX + 8; xw += 8) {
      const int cls = (xw >> 3) & 1;  // which row parity completes a line at this window
      const int xa = xw - 8 * cls >= 0 ? xw - (cls ? 8 : 0) : 0;
      for (int u = 0; u < 4; ++u) {
        const int r = 64 * wave + cls + 2 * (rslot + 8 * u);
        const int x = (xa & ~15) + 8 * ((r & 1) ? 1 : 0) + 2 * pc;  // line start for this row parity
        if (r < 500 && x + 1 < NX && x >= 0) {
          double2* p = reinterpret_cast<double2*>(field + (size_t)r * NX + x);
          *p = make_double2(v0, v1);
        }
      }
      for (int g = 0; g < 8 * gap; ++g) asm volatile("v_mov_b32 %0, %0" : "+v"(*(int*)&lds[threadIdx.x & 1]) ::);
    }
  } else {
    double2* p = reinterpret_cast<double2*>(out);
    const size_t n2 = (size_t)256 * NX * NY / 2;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n2; k += (size_t)gridDim.x * blockDim.x) p[k] = make_double2(v0, v1);
  }
}
template <int MODE>
float run(double* d, int lds_bytes, int gap, int grid) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(pat<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pat<MODE>, dim3(grid), dim3(512), lds_bytes, 0, d, gap);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pat<MODE>, dim3(grid), dim3(512), lds_bytes, 0, d, gap);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20;
}
int main() {
  double* d; hipMalloc(&d, (size_t)256 * NX * NY * 8 + 4096);
  const double gb0 = 256.0 * 1000 * 1000 * 8 / 1e9;
  for (int lds : {1024, 40000, 80000}) {
    printf("dynamic LDS %d B per workgroup (%s workgroups per CU):\n", lds, lds > 54000 ? "2" : lds > 30000 ? "4" : "8 (wave-limited 4)");
    for (int gap : {0, 16, 64}) {
      float a = run<0>(d, lds, gap, 512), b = run<1>(d, lds, gap, 512);
      printf("   gap %2d: y-pattern %.3f ms (%.0f GB/s)   x-pattern %.3f ms (%.0f GB/s of ~2.0 GB)\n", gap, a, gb0 * 0.976 / a * 1e3, b, gb0 * 0.98 / b * 1e3);
    }
  }
  float f = run<2>(d, 1024, 0, 2048);
  printf("fill: %.3f ms (%.0f GB/s)\n", f, gb0 / f * 1e3);
  return 0;
}
