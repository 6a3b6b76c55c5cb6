// Measures the shader clock a kernel actually runs at: s_memtime (shader cycles) vs
// s_memrealtime (100 MHz), for a light (1 workgroup) and a heavy (all CUs) launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double a = threadIdx.x * 1e-3, b = 1.000001;
  for (int i = 0; i < iters; ++i) { a = a * b + 1e-9; a = a * b + 1e-9; a = a * b + 1e-9; a = a * b + 1e-9; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x*3+2] = (unsigned long long)a; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 4096 * 3 * 8);
  unsigned long long h[3];
  for (int rep = 0; rep < 3; ++rep)
  for (int blocks : {1, 9, 2048}) {
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), 0, 0, d, 200000);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks %4d: %llu shader cycles in %.1f us -> %.0f MHz ; %.2f cycles per dependent fp64 fma\n", blocks, h[0], h[1] / 100.0,
           h[0] / (h[1] / 100.0), (double)h[0] / (4.0 * 200000));
  }
  return 0;
}
