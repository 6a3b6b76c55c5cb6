// Achievable HBM write rate as a function of the store pattern: every wavefront owns a band of 128
// rows of an 8000-byte-pitch image and fills it left to right in row segments of SB bytes per row
// (one 1 KB store instruction = 1024/SB rows x SB bytes).  SB = 64 is the x-strip flush pattern
// of the sweep, SB = 1024 a y-strip's.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int SB, int NT = 0>
__global__ void __launch_bounds__(1024) fill(char* out, size_t band_bytes, int pitch, int spin, int xoff) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  char* band = out + wave * band_bytes;
  constexpr int LPR = SB / 16;          // lanes per row segment
  constexpr int RPI = 64 / LPR;         // rows per instruction
  const int row_in = lane / LPR, col = (lane % LPR) * 16;
  double a = lane;
  for (int x = xoff; x + SB <= 8000; x += SB) {
    for (int f = 0; f < spin; ++f) a = __builtin_fma(a, 1.0000001, 1e-9);
#pragma unroll 4
    for (int r0 = 0; r0 < 128; r0 += RPI)
    {
      typedef double dbl2 __attribute__((ext_vector_type(2)));
      dbl2 v = {a, a};
      dbl2* q = reinterpret_cast<dbl2*>(band + (size_t)(r0 + row_in) * pitch + x + col);
      if (NT == 1) __builtin_nontemporal_store(v, q);
      else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
      else if (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
      else if (NT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(q), "v"(v) : "memory");
      else *q = v;
    }
  }
}

template <int SB, int NT = 0> void run(char* d, int nb, int w, int spin, int pitch = 8192, int xoff = 0) {
  const size_t band = (size_t)128 * pitch;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<SB, NT>), dim3(nb), dim3(64 * w), 0, 0, d, band, pitch, spin, xoff);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double bytes = (double)nb * w * 128 * ((8000 - xoff) / SB) * SB;
  printf("NT %d SB %4d pitch %d xoff %3d  %4d blocks x %2d waves spin %3d: %.3f ms  %.2f TB/s\n", NT, SB, pitch, xoff, nb, w, spin, best, bytes / best / 1e9);
}

int main() {
  char* d; const size_t total = (size_t)1024 * 16 * 128 * 8192;  // 16 GiB
  if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  for (int w : {4, 16}) {
    const int nb = 512;
    run<64, 0>(d, nb, w, 0); run<64, 1>(d, nb, w, 0); run<64, 2>(d, nb, w, 0); run<64, 3>(d, nb, w, 0); run<64, 4>(d, nb, w, 0);
    run<1024, 0>(d, nb, w, 0, 8192, 8); run<1024, 1>(d, nb, w, 0, 8192, 8); run<1024, 2>(d, nb, w, 0, 8192, 8); run<1024, 3>(d, nb, w, 0, 8192, 8); run<1024, 4>(d, nb, w, 0, 8192, 8);
    run<1024, 0>(d, nb, w, 0); run<1024, 1>(d, nb, w, 0); run<1024, 2>(d, nb, w, 0);
  }
  return 0;
}
