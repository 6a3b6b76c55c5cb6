// Does L2 merge two 64-byte halves of a 128-byte line that are stored by different instructions?
// Each wavefront fills a band of 128 rows (pitch 8192 B) with 64 B row segments (16 rows x 64 B per
// instruction).  GAP = how many column-chunks lie between writing the left and the right half of a line:
// order of chunks c0, c0+GAP ... so that the right half (odd chunk) follows the left half after GAP
// chunk-passes (8 store instructions each).  GAP 0 = halves in consecutive passes.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(1024) fill(char* out, size_t band_bytes, int pitch, int gap) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  char* band = out + wave * band_bytes;
  const int row_in = lane >> 2, col = (lane & 3) * 16;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  dbl2 v = {(double)lane, 1.0};
  const int nlines = 8000 / 128;  // 62
  // sequence: left halves run `gap` lines ahead of right halves
  for (int s = 0; s < nlines + gap; ++s) {
    if (s < nlines) {
      const int x = s * 128;
#pragma unroll
      for (int r0 = 0; r0 < 128; r0 += 16) *reinterpret_cast<dbl2*>(band + (size_t)(r0 + row_in) * pitch + x + col) = v;
    }
    if (s >= gap) {
      const int x = (s - gap) * 128 + 64;
#pragma unroll
      for (int r0 = 0; r0 < 128; r0 += 16) *reinterpret_cast<dbl2*>(band + (size_t)(r0 + row_in) * pitch + x + col) = v;
    }
  }
}

int main() {
  char* d; const size_t band = (size_t)128 * 8192;
  if (hipMalloc(&d, band * 8192) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nb : {256, 512}) for (int w : {4, 8, 16}) for (int gap : {0, 1, 2, 4, 8, 16}) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fill, dim3(nb), dim3(64 * w), 0, 0, d, band, 8192, gap);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("blocks %4d x %2d waves gap %2d lines: %.3f ms  %.2f TB/s\n", nb, w, gap, best, (double)nb * w * 128 * 62 * 128 / best / 1e9);
  }
  return 0;
}
