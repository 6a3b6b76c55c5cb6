// Mixed store patterns with compute in between, as in the sweep: half of the wavefronts write an image
// band in row segments of XB bytes (the x-strip flush: XB = 64 today), the other half in 1 KB row
// segments (y strips).  Each wavefront does `spin` dependent FMAs per 8 KB it stores.  Reports the
// slowdown against the same loop with the stores removed.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int XB, bool STORE>
__global__ void __launch_bounds__(1024) mix(char* out, size_t band_bytes, int pitch, int spin, int iters, unsigned long long* cyc, int xalign, int yalign) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  char* band = out + wave * band_bytes;
  const bool xpat = (wave & 1) == 0;
  constexpr int LPR = XB / 16, RPI = 64 / LPR;
  const int row_in = lane / LPR, col = (lane % LPR) * 16;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  double a = lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int f = 0; f < spin; ++f) a = __builtin_fma(a, 1.0000001, 1e-9);
    dbl2 v = {a, a};
    if (STORE) {
      if (xpat) {
        // 8 KB: 128 rows x 64 B (XB = 64) or 64 rows x 128 B (XB = 128), next chunk of columns each iteration
        const int x = (it * XB) % 7936;
        const int rows = 8192 / XB;
#pragma unroll 8
        for (int r0 = 0; r0 < rows; r0 += RPI) {
          const int r = (r0 + row_in) & 127;
          // xalign: rows whose lines start 64 B off (odd rows at pitch 8000) flush a window slid by 64 B
          *reinterpret_cast<dbl2*>(band + (size_t)r * pitch + x + col + ((xalign && (r & 1)) ? 64 : 0)) = v;
        }
      } else {
        // 8 rows x 1 KB
        const int x = ((it >> 4) * 1024) % 7168, r = (it & 15) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<dbl2*>(band + (size_t)(r + k) * pitch + x + lane * 16 + ((yalign && ((r + k) & 1)) ? 64 : 0)) = v;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[wave] = t1 - t0;
  if (a == 123.456) out[0] = 1;
}

template <int XB> void run(char* d, unsigned long long* c, int nb, int w, int spin, int pitch, int xalign = 0, int yalign = 0) {
  const int iters = 120;
  const size_t band = (size_t)128 * pitch;
  static unsigned long long h[8192];
  double t[2];
  float ms[2];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int st = 0; st < 2; ++st) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (st) hipLaunchKernelGGL((mix<XB, true>), dim3(nb), dim3(64 * w), 0, 0, d, band, pitch, spin, iters, c, xalign, yalign);
      else hipLaunchKernelGGL((mix<XB, false>), dim3(nb), dim3(64 * w), 0, 0, d, band, pitch, spin, iters, c, xalign, yalign);
      hipEventRecord(e1); hipDeviceSynchronize();
    }
    hipEventElapsedTime(&ms[st], e0, e1);
    hipMemcpy(h, c, 8 * nb * w, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < nb * w; ++i) avg += (double)h[i] / (nb * w);
    t[st] = avg / iters;
  }
  const double bytes = (double)nb * w * iters * 8192;
  printf("xal %d yal %d XB %3d pitch %d  %3d x %2d waves spin %3d: compute %6.0f cyc/iter, with stores %6.0f (x%.2f)  kernel %.3f ms = %.2f TB/s\n", xalign, yalign, XB, pitch, nb, w,
         spin, t[0], t[1], t[1] / t[0], ms[1], bytes / ms[1] / 1e9);
}

int main() {
  char* d; unsigned long long* c;
  if (hipMalloc(&d, (size_t)8192 * 128 * 8192) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&c, 8 * 8192);
  for (int w : {8, 16}) {
    const int spin = 100;
    run<64>(d, c, 512, w, spin, 8000);
    run<128>(d, c, 512, w, spin, 8000, 0, 0);
    run<128>(d, c, 512, w, spin, 8000, 1, 0);
    run<128>(d, c, 512, w, spin, 8000, 0, 1);
    run<128>(d, c, 512, w, spin, 8000, 1, 1);
    run<64>(d, c, 512, w, spin, 8000, 0, 1);
    run<128>(d, c, 512, w, spin, 8192);
  }
  return 0;
}
