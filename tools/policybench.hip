// policybench.hip -- do the cache-policy bits of a store change what a partially written line costs?  (gfx950; diagnostic)
// The y1k_mixed pattern of shapebench.hip (a 1 KB store per row, every other row 64 B off the lines: two half lines per odd row)
// and its all-aligned twin, with global_store_dwordx4 carrying: nothing, nt, sc0, sc1, sc0 sc1, sc0 nt, sc1 nt, sc0 sc1 nt.
#include <hip/hip_runtime.h>
#include <cstdint>
constexpr int NX = 1000, NY = 1000, NF = 256;
constexpr size_t PITCH = 8000, FIELD = (size_t)NX * NY * 8;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int POL>
__device__ __forceinline__ void st16(char* p, u4 v) {
  if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
  else if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int POL>
__global__ void __launch_bounds__(256) pat(char* out, int aligned, int halves_pol, int n_tasks, unsigned* counter) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const u4 val = {(unsigned)lane, 0x3ff00000u, (unsigned)lane + 1u, 0x40000000u};
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)n_tasks) break;
    const int f = t % NF, k = t / NF;
    char* field = out + (size_t)f * FIELD;
    for (int row = 0; row < NY; ++row) {
      char* p = field + (size_t)row * PITCH + (aligned && (row & 1) ? 64 : 0) + (size_t)k * 1024 + (size_t)lane * 16;
      // halves_pol: only the lanes that write the two half lines of an off-line row carry the policy, the others store plainly
      const bool half = !aligned && (row & 1) && (lane < 4 || lane >= 60);
      if (halves_pol && !half) st16<0>(p, val); else st16<POL>(p, val);
    }
  }
}
static unsigned* d_counter = nullptr;
template <int POL>
static float run1(char* buf, int aligned, int halves_pol, int wpc) {
  const int n_tasks = NF * 7;
  const int wgs_per_cu = wpc / 4;
  const size_t lds = 160 * 1024 / wgs_per_cu - 512;
  hipFuncSetAttribute(reinterpret_cast<const void*>(pat<POL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemsetAsync(d_counter, 0, 4, 0);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(pat<POL>, dim3(256 * wgs_per_cu), dim3(256), lds, 0, buf, aligned, halves_pol, n_tasks, d_counter);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  hipEventDestroy(a); hipEventDestroy(b);
  return best;
}
extern "C" int policy_run(void* buf, int pol, int aligned, int halves_pol, int wpc, float* ms_out, double* bytes_out) {
  if (!d_counter && hipMalloc(&d_counter, 4) != hipSuccess) return 1;
  char* b = (char*)buf;
  float ms;
  switch (pol) {
    case 0: ms = run1<0>(b, aligned, halves_pol, wpc); break;
    case 1: ms = run1<1>(b, aligned, halves_pol, wpc); break;
    case 2: ms = run1<2>(b, aligned, halves_pol, wpc); break;
    case 3: ms = run1<3>(b, aligned, halves_pol, wpc); break;
    case 4: ms = run1<4>(b, aligned, halves_pol, wpc); break;
    case 5: ms = run1<5>(b, aligned, halves_pol, wpc); break;
    case 6: ms = run1<6>(b, aligned, halves_pol, wpc); break;
    default: ms = run1<7>(b, aligned, halves_pol, wpc); break;
  }
  *ms_out = ms; *bytes_out = (double)NF * NY * 7168;
  return (int)hipGetLastError();
}
