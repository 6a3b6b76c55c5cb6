// What does a 1 KB store instruction cost the wavefront that issues it, and how many bytes per cycle
// does one CU push out?  Each wavefront loops: F dependent fp64 FMAs, then NS x global_store_dwordx4
// (1 KB per instruction, streaming).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// interleaved: NS x (filler/NS FMAs, one store)
template <int NS>
__global__ void __launch_bounds__(1024) ki(double* out, size_t per_wave_doubles, unsigned long long* cyc, int iters, int filler, double seed) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  double2* p = reinterpret_cast<double2*>(out + (size_t)wave * per_wave_doubles) + lane;
  double a = seed + lane, b = 1.0000001;
  const int fpart = filler / NS;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      for (int f = 0; f < fpart; ++f) a = __builtin_fma(a, b, 1e-9);
      p[(size_t)(it * NS + s) * 64] = make_double2(a, a + s);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[wave] = t1 - t0;
}

template <int NS>
__global__ void __launch_bounds__(1024) k(double* out, size_t per_wave_doubles, unsigned long long* cyc, int iters, int filler, double seed) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  double2* p = reinterpret_cast<double2*>(out + (size_t)wave * per_wave_doubles) + lane;
  double a = seed + lane, b = 1.0000001;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int f = 0; f < filler; ++f) a = __builtin_fma(a, b, 1e-9);
#pragma unroll
    for (int s = 0; s < NS; ++s) p[(size_t)(it * NS + s) * 64] = make_double2(a, a + s);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
  const int iters = 200, NS = 8;
  const size_t per_wave = (size_t)iters * NS * 128;  // doubles
  const int max_waves = 256 * 32;
  double* d; unsigned long long* c;
  hipMalloc(&d, per_wave * 8 * max_waves); hipMalloc(&c, 8 * max_waves);
  std::vector<unsigned long long> h(max_waves);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("cfg: blocks/CU x waves/block, filler FMAs per iteration -> cycles per iteration per wave (8 stores = 8 KB), GB/s chip\n");
  for (int filler : {0, 100, 300}) {
    for (int bpc : {1, 2}) {
      for (int w : {1, 4, 8, 16}) {
        const int nb = 256 * bpc;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k<NS>, dim3(nb), dim3(64 * w), 0, 0, d, per_wave, c, iters, filler, 1.0);
          hipEventRecord(e1);
          hipDeviceSynchronize();
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), c, 8 * nb * w, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < nb * w; ++i) avg += (double)h[i] / (nb * w);
        const double bytes = (double)nb * w * iters * NS * 1024;
        printf("filler %3d  %d x %2d waves: %8.0f ticks/iter (%.0f per store beyond filler~%d)  kernel %.3f ms  %.2f TB/s\n", filler, bpc, w,
               avg / iters, (avg / iters) / NS, filler * 8, ms, bytes / ms / 1e9);
      }
    }
  }
  printf("\nburst (B) vs interleaved (I), nb blocks total\n");
  for (int nb : {1, 64, 256, 512}) for (int w : {1, 4, 16}) for (int filler : {0, 96, 304}) for (int mode : {0, 1}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<NS>, dim3(nb), dim3(64 * w), 0, 0, d, per_wave, c, iters, filler, 1.0);
      else hipLaunchKernelGGL(ki<NS>, dim3(nb), dim3(64 * w), 0, 0, d, per_wave, c, iters, filler, 1.0);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), c, 8 * nb * w, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < nb * w; ++i) avg += (double)h[i] / (nb * w);
    printf("%s nb %3d x %2d waves filler %3d: %8.0f ticks/iter\n", mode ? "I" : "B", nb, w, filler, avg / iters);
  }
  return 0;
}
