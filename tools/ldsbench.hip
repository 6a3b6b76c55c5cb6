// ldsbench.hip -- what LDS operations cost a LONE wavefront (the latency sweep's situation).  Diagnostic only.
// hipcc --offload-arch=gfx950 -O3 -o exp/ldsbench tools/ldsbench.hip && exp/ldsbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) int lds_int;
__global__ void k(int mode, int iters, unsigned long long* out, double* sink, int waves_active) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  if (w >= waves_active) return;
  double acc = 0;
  double* mine = lds + w * 1024;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {  // 8 broadcast reads, one wait
      double r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = mine[(it * 8 + k) & 63];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += r[k];
    } else if (mode == 1) {  // one broadcast read, wait
      acc += mine[it & 63];
    } else if (mode == 2) {  // poll-like: volatile int read + readfirstlane
      int v = __builtin_amdgcn_readfirstlane(*(volatile lds_int*)(lds + (it & 63)));
      acc += v;
    } else if (mode == 3) {  // 16 lane-strided writes (stride 17) then 8 broadcast reads
#pragma unroll
      for (int k = 0; k < 16; ++k) mine[lane * 17 + (k & 15)] = acc + k;
      __builtin_amdgcn_wave_barrier();
      double r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = mine[(it * 8 + k) & 63];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += r[k];
    } else if (mode == 4) {  // 4 x (two adjacent doubles per lane, rows of 34 doubles apart: the flush read)
      double a[4], b[4];
      const int t = (lane >> 3) * 34 + ((lane & 7) * 2 & 7) + ((lane & 7) >> 2) * 8;
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = mine[t + u * 8]; b[u] = mine[t + u * 8 + 1]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += a[u] + b[u];
    } else if (mode == 5) {  // dependent fp64 chain: 8 x (sub, mul, sub)
#pragma unroll
      for (int k = 0; k < 8; ++k) { double t = acc - 1.5; t = t * 0.999; acc = acc - t; }
    } else if (mode == 6) {  // DPP + chain as in a step
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int lo = __double2loint(acc), hi = __double2hiint(acc);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
        double b = __hiloint2double(hi, lo);
        double t = acc - b; t = t * 0.999; acc = acc - t;
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[w] = t1 - t0;
  sink[threadIdx.x] = acc;
}
int main() {
  unsigned long long* d; double* s;
  hipMalloc(&d, 16 * 8); hipMalloc(&s, 1024 * 8);
  const char* names[] = {"8 broadcast ds_read_b64 + wait", "1 broadcast read + wait", "poll (volatile b32 + readfirstlane)", "16 strided ds_write_b64 + 8 bcast reads + wait",
                         "flush read: 4 x 2 adjacent doubles (ds_read2_b64)", "8 x dependent (sub, mul, sub) fp64", "8 x (dpp shift + sub, mul, sub)"};
  for (int waves : {1, 4, 12}) {
    for (int mode = 0; mode < 7; ++mode) {
      const int iters = 2000;
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(768), 160 * 1024 - 1024, 0, mode, iters, d, s, waves);
      hipDeviceSynchronize();
      unsigned long long h[16];
      hipMemcpy(h, d, 16 * 8, hipMemcpyDeviceToHost);
      printf("%2d wave(s) active  %-52s %7.1f cycles per iteration\n", waves, names[mode], (double)h[0] / iters);
    }
  }
  return 0;
}
