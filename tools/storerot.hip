// Does the memory pipeline form its 128-byte write requests from fixed groups of 16 lanes, or from the addresses
// of the whole wavefront?  Every wavefront stores aligned 512-byte row segments, 8 bytes per lane; with ROT the
// lanes are rotated by 8 inside the segment (lane l writes element (l + 8) mod 64), so each group of 16 lanes
// covers the halves of two different lines although the wavefront as a whole still covers whole lines.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ROT, int BYTES>
__global__ void __launch_bounds__(1024) fill(char* out, size_t per_wave, int iters) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  char* p = out + wave * per_wave + (size_t)((lane + ROT) & 63) * BYTES;
  for (int it = 0; it < iters; ++it) {
    if (BYTES == 8) *reinterpret_cast<double*>(p + (size_t)it * 64 * BYTES) = (double)it;
    else *reinterpret_cast<double2*>(p + (size_t)it * 64 * BYTES) = make_double2(it, lane);
  }
}

template <int ROT, int BYTES> void run(char* d, int nb, int w) {
  const int iters = 2048;
  const size_t per_wave = (size_t)iters * 64 * BYTES;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<ROT, BYTES>), dim3(nb), dim3(64 * w), 0, 0, d, per_wave, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("bytes/lane %2d rot %2d  %4d x %2d waves: %.3f ms  %.2f TB/s\n", BYTES, ROT, nb, w, best, (double)nb * w * per_wave / best / 1e9);
}

int main() {
  char* d; if (hipMalloc(&d, (size_t)512 * 16 * 2048 * 64 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  for (int w : {4, 16}) {
    run<0, 8>(d, 512, w); run<8, 8>(d, 512, w); run<4, 8>(d, 512, w); run<16, 8>(d, 512, w);
    run<0, 16>(d, 512, w); run<4, 16>(d, 512, w); run<8, 16>(d, 512, w);
  }
  return 0;
}
