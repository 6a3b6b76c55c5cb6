// Is the allocation-dependent store rate of the sweep (tools/allocprobe4.py) visible with a plain synthetic pattern?
// Six 2 GB allocations; on each: (a) the mixed sweep-like pattern of tools/storemix.hip (x: 8 rows x 128 B per
// instruction over 64 rows, y: 8 rows x 1 KB), with compute in between; (b) a pure streaming fill.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(1024) mix(char* out, size_t band_bytes, int pitch, int spin, int iters) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  char* band = out + wave * band_bytes;
  const bool xpat = (wave & 1) == 0;
  const int row_in = lane >> 3, col = (lane & 7) * 16;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  double a = lane;
  for (int it = 0; it < iters; ++it) {
    for (int f = 0; f < spin; ++f) a = __builtin_fma(a, 1.0000001, 1e-9);
    dbl2 v = {a, a};
    if (xpat) {
      const int x = (it * 128) % 7936;
#pragma unroll 8
      for (int r0 = 0; r0 < 64; r0 += 8) *reinterpret_cast<dbl2*>(band + (size_t)(r0 + row_in) * pitch + x + col) = v;
    } else {
      const int x = ((it >> 4) * 1024) % 7168, r = (it & 15) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<dbl2*>(band + (size_t)(r + k) * pitch + x + lane * 16) = v;
    }
  }
  if (a == 123.456) out[0] = 1;
}

__global__ void fill(double2* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  char* d[6];
  for (int i = 0; i < 6; ++i) if (hipMalloc(&d[i], bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int pitch = 8192, nb = 512, w = 4, iters = 120;  // 2048 bands of 1 MB = 2 GB
  for (int rep = 0; rep < 2; ++rep) {
    for (int spin : {0, 100}) {
      printf("mix spin %3d:", spin);
      for (int i = 0; i < 6; ++i) {
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(mix, dim3(nb), dim3(64 * w), 0, 0, d[i], (size_t)128 * pitch, pitch, spin, iters);
          hipEventRecord(e1); hipDeviceSynchronize();
          float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("  %d:%.2f TB/s", i, (double)nb * w * iters * 8192 / best / 1e9);
      }
      printf("\n");
    }
    printf("fill        :");
    for (int i = 0; i < 6; ++i) {
      float best = 1e9;
      for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (double2*)d[i], bytes / 16);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("  %d:%.2f TB/s", i, (double)bytes / best / 1e9);
    }
    printf("\n");
  }
  return 0;
}
