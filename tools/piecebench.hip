// piecebench.hip -- which store shapes does the memory of a SLOW output buffer take at full speed?  (gfx950; diagnostic)
// 256 fields of 1000 x 1000 doubles (row pitch 8000 B), as the C3 launch writes them.  A task is what one wavefront streams:
//   mode x: R rows of one field, marched in x; a visit stores R rows x P bytes (whole 128-byte lines; P = 128 .. 4096)
//   mode y: a band of P bytes of one field, marched in y; a visit stores one row x P bytes
// Wavefronts are persistent and pull tasks from a counter; `wpc` wavefronts per CU (4 per workgroup, residency limited by LDS);
// `nap` s_sleep units after every store instruction (a sweeping wavefront computes ~300 cycles per KB it stores).
// Buffers: N hipMalloc allocations, classified by a fixed scattered pattern; the sweep runs on the slowest and the fastest.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int NX = 1000, NY = 1000, NF = 256;
constexpr size_t PITCH = 8000, FIELD = (size_t)NX * NY * 8;
struct Params { int mode, P, R, n_tasks, nap, tasks_per_field; };
__device__ __forceinline__ void nap(int n) { for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(1); }
__global__ void __launch_bounds__(256) pieces(char* out, Params p, unsigned* counter) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const double2 val = make_double2(1.0 + lane, 2.0 + lane);
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)p.n_tasks) break;
    // spread consecutive tasks over the fields (a launch's units are all over the output)
    const int f = t % NF, k = t / NF;
    char* field = out + (size_t)f * FIELD;
    if (p.mode == 0) {
      // rows [k R, k R + R), pieces of P bytes along x; an instruction = (1024 / P) rows x P bytes, or a 1 KB part of a row's piece
      const int r0 = k * p.R;
      const int n_x = 7936 / p.P;
      for (int ix = 0; ix < n_x; ++ix) {
        if (p.P <= 1024) {
          const int rpi = 1024 / p.P;               // rows per instruction
          const int lpr = 64 / rpi;                 // lanes per row
          for (int r = lane / lpr; r < p.R; r += rpi) {
            const int row = r0 + r;
            const size_t off = (size_t)row * PITCH + ((row & 1) ? 64 : 0) + (size_t)ix * p.P + (size_t)(lane % lpr) * 16;
            if (row < NY) *reinterpret_cast<double2*>(field + off) = val;
            nap(p.nap);
          }
        } else {
          for (int r = 0; r < p.R; ++r) {
            const int row = r0 + r;
            for (int part = 0; part < p.P / 1024; ++part) {
              const size_t off = (size_t)row * PITCH + ((row & 1) ? 64 : 0) + (size_t)ix * p.P + (size_t)part * 1024 + (size_t)lane * 16;
              if (row < NY) *reinterpret_cast<double2*>(field + off) = val;
              nap(p.nap);
            }
          }
        }
      }
    } else {
      // band k of P bytes, marched over all rows
      for (int row = 0; row < NY; ++row) {
        if (p.P <= 1024) {
          // P < 1024: the instruction covers 1024 / P consecutive rows' pieces (as a strip with fewer columns would not exist in the
          // sweep, this is only for symmetry)
          const int rpi = 1024 / p.P, lpr = 64 / rpi;
          const int rr = row * rpi + lane / lpr;
          if (row * rpi >= NY) break;
          const size_t off = (size_t)rr * PITCH + ((rr & 1) ? 64 : 0) + (size_t)k * p.P + (size_t)(lane % lpr) * 16;
          if (rr < NY) *reinterpret_cast<double2*>(field + off) = val;
          nap(p.nap);
        } else {
          for (int part = 0; part < p.P / 1024; ++part) {
            const size_t off = (size_t)row * PITCH + ((row & 1) ? 64 : 0) + (size_t)k * p.P + (size_t)part * 1024 + (size_t)lane * 16;
            *reinterpret_cast<double2*>(field + off) = val;
            nap(p.nap);
          }
        }
      }
    }
  }
}
static unsigned* d_counter;
static double run(char* d, Params p, int wpc, double* tbps) {
  const int wgs_per_cu = wpc / 4;
  size_t lds = 160 * 1024 / wgs_per_cu - 512;
  if (lds > 64 * 1024) lds = 64 * 1024 - 512;   // (cannot ask for more than 64 KB without the attribute; wpc >= 12 only)
  hipFuncSetAttribute(reinterpret_cast<const void*>(pieces), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  lds = 160 * 1024 / wgs_per_cu - 512;
  // bytes a launch writes
  double bytes;
  if (p.mode == 0) { p.tasks_per_field = (NY + p.R - 1) / p.R; bytes = (double)NF * NY * (7936 / p.P) * p.P; }
  else { p.tasks_per_field = 7936 / p.P; bytes = (double)NF * NY * (7936 / p.P) * p.P; }
  p.n_tasks = NF * p.tasks_per_field;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemsetAsync(d_counter, 0, 4, 0);
    hipEventRecord(a);
    hipLaunchKernelGGL(pieces, dim3(256 * wgs_per_cu), dim3(256), lds, 0, d, p, d_counter);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  hipEventDestroy(a); hipEventDestroy(b);
  *tbps = bytes / best / 1e9;
  return best;
}
int main(int argc, char** argv) {
  const int nbuf = argc > 1 ? atoi(argv[1]) : 40;
  hipMalloc(&d_counter, 4);
  std::vector<char*> bufs;
  for (int k = 0; k < nbuf; ++k) { char* p; if (hipMalloc(&p, NF * FIELD + 4096) != hipSuccess) break; bufs.push_back(p); }
  std::vector<std::pair<double, int>> cls;
  Params probe{1, 1024, 1, 0, 0, 0};
  for (size_t k = 0; k < bufs.size(); ++k) { double t; run(bufs[k], probe, 12, &t); cls.push_back({t, (int)k}); }
  printf("probe (mode y, 1 KB pieces, 12 wavefronts per CU) TB/s per buffer:");
  for (auto& c : cls) printf(" %.2f", c.first);
  printf("\n");
  std::sort(cls.begin(), cls.end());
  char* slow = bufs[cls.front().second];
  char* fast = bufs[cls.back().second];
  printf("slowest buffer %d (%.2f TB/s), fastest %d (%.2f TB/s)\n", cls.front().second, cls.front().first, cls.back().second, cls.back().first);
  printf("%-6s %5s %4s %4s %4s | %8s %8s\n", "mode", "P", "R", "wpc", "nap", "slow TB/s", "fast TB/s");
  const int Ps[] = {128, 256, 512, 1024, 2048, 4096};
  for (int mode = 0; mode < 2; ++mode)
    for (int P : Ps)
      for (int R : {8, 64})
        for (int wpc : {8, 12, 16, 32})
          for (int napv : {0, 4}) {
            if (mode == 1 && R != 8) continue;
            if (P == 3968 && false) continue;
            Params p{mode, P, mode == 0 ? R : 1, 0, napv, 0};
            if (P == 3968) { p.P = 3968; }
            if (p.P > 1024 && p.P % 1024) {  // 3968 = 31 lines: not a multiple of 1 KB -- round the instruction count up, the last is partial: skip, use 4096 on 7936? no: use 2 x 3968 = 7936
              continue;
            }
            double ts, tf;
            run(slow, p, wpc, &ts);
            run(fast, p, wpc, &tf);
            printf("%-6s %5d %4d %4d %4d | %8.2f %8.2f\n", mode == 0 ? "x" : "y", p.P, p.R, wpc, napv, ts, tf);
            fflush(stdout);
          }
  return 0;
}
