// shapebench.hip -- store shapes on a SLOW and a FAST output buffer (gfx950; diagnostic; driven by tools/shapebench.py, which
// classifies the buffers with the real C3 launch).  256 fields x 1000 rows x 8000 B.  Persistent wavefronts pull tasks.
//   pattern 0  y1k      : task = 128-column band of a field, marched over its 1000 rows; 1 KB per instruction, every piece on 128-byte lines
//   pattern 1  y1k_half : the same, every piece shifted by 64 B (two half lines per piece, as the odd rows of a 1000-wide field are)
//   pattern 2  y1k_mixed: even rows aligned, odd rows shifted (what the y-major strips of the C3 launch store)
//   pattern 3  x128     : task = 64 rows of a field marched in x; 8 rows x 128-byte line per instruction
//   pattern 4  x64      : 16 rows x 64-byte half line per instruction
//   pattern 5  x256     : 4 rows x 256 B
//   pattern 6  x512     : 2 rows x 512 B
//   pattern 7  fill     : compact, front to back
//   pattern 8  rand128  : every instruction stores 8 random whole lines of the buffer
//   pattern 9  rand1k   : every instruction stores one random aligned 1 KB piece
//   pattern 10 y1k_8B   : y1k, but every lane stores only the first 8 of its 16 bytes (partial sectors everywhere)
//   pattern 11 x128_rows2: x128 with the 8 rows of an instruction 2 rows apart (the C3 launch's r_stride = 2)
//   pattern 13 y1k_rot  : y1k_mixed with the odd rows' lanes rotated by four (lane m stores the 16 bytes at 16 ((m + 4) mod 64)): whole lines again
//   pattern 12 y2k      : 256-column band, two adjacent 1 KB instructions per row
#include <hip/hip_runtime.h>
#include <cstdint>
constexpr int NX = 1000, NY = 1000, NF = 256;
constexpr size_t PITCH = 8000, FIELD = (size_t)NX * NY * 8;
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NT> __device__ __forceinline__ void put2(void* p, double2 v) { if (NT) __builtin_nontemporal_store(d2{v.x, v.y}, (d2*)p); else *(double2*)p = v; }
template <int NT> __device__ __forceinline__ void put1(void* p, double v) { if (NT) __builtin_nontemporal_store(v, (double*)p); else *(double*)p = v; }
template <int NT>
__global__ void __launch_bounds__(256) shapes(char* out, int pattern, int n_tasks, unsigned* counter) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const double2 val = make_double2(1.0 + lane, 2.0 + lane);
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)n_tasks) break;
    const int f = t % NF, k = t / NF;
    char* field = out + (size_t)f * FIELD;
    if (pattern == 20 || pattern == 21) {
      // 20: y1k_mixed without its half lines, the whole lines of the odd rows issued by lanes 4..59 (8-lane groups that straddle lines)
      // 21: all rows aligned, but issued by lanes 4..59 of a rotated wavefront: lane m stores 16 bytes at 16 (m - 4)
      for (int row = 0; row < NY; ++row) {
        char* p = field + (size_t)row * PITCH + (pattern == 21 ? ((row & 1) ? 64 : 0) : 0) + (size_t)k * 1024;
        if (pattern == 20) { if (!(row & 1) || (lane >= 4 && lane < 60)) put2<NT>(p + (size_t)lane * 16, val); }
        else { if (lane >= 4 && lane < 60) put2<NT>(p + (size_t)(lane - 4) * 16, val); }
      }
    } else if (pattern >= 14 && pattern <= 19) {
      // 14..17: y1k with 1 / 2 / 4 / 8 rows of every 16 off the lines by 64 B; 18: y1k_mixed without the half lines (odd rows store their
      // 7 whole lines only); 19: y1k_mixed with the half lines as a separate, later instruction of the same wavefront
      for (int row = 0; row < NY; ++row) {
        const int m = pattern == 14 ? 1 : pattern == 15 ? 2 : pattern == 16 ? 4 : 8;
        const bool off_line = pattern >= 18 ? (row & 1) : (row & 15) < m;
        char* p = field + (size_t)row * PITCH + ((row & 1) ? 64 : 0) + (off_line ? 64 : 0) + (size_t)k * 1024 + (size_t)lane * 16;
        if (pattern >= 18 && off_line) { if (lane < 56) put2<NT>(p + 64, val); }
        else put2<NT>(p, val);
        if (pattern == 19 && off_line && lane >= 56) put2<NT>(p + (lane >= 60 ? 64 : 64 - 1024), val);
      }
    } else if (pattern == 13) {
      for (int row = 0; row < NY; ++row) {
        const int sl = (row & 1) ? ((lane + 4) & 63) : lane;
        // the same bytes as y1k_mixed (the odd rows of a 1000-wide field start 64 B off a line), other lanes issuing them
        char* p = field + (size_t)row * PITCH + (size_t)k * 1024 + (size_t)sl * 16;
        put2<NT>(p, val);
      }
    } else if (pattern <= 2 || pattern == 10) {  // 7 bands of 1 KB per row (7168 of 8000 B)
      for (int row = 0; row < NY; ++row) {
        const int shift = pattern == 0 || pattern == 10 ? ((row & 1) ? 64 : 0) : pattern == 1 ? ((row & 1) ? 0 : 64) : 0;
        char* p = field + (size_t)row * PITCH + shift + (size_t)k * 1024 + (size_t)lane * 16;
        if (pattern == 10) put1<NT>(p, val.x); else put2<NT>(p, val);
      }
    } else if (pattern == 12) {  // 3 bands of 2 KB
      for (int row = 0; row < NY; ++row) {
        char* p = field + (size_t)row * PITCH + ((row & 1) ? 64 : 0) + (size_t)k * 2048 + (size_t)lane * 16;
        put2<NT>(p, val);
        put2<NT>(p + 1024, val);
      }
    } else if (pattern >= 3 && pattern <= 6 || pattern == 11) {
      const int P = pattern == 3 || pattern == 11 ? 128 : pattern == 4 ? 64 : pattern == 5 ? 256 : 512;
      const int rpi = 1024 / P, lpr = 64 / rpi;
      const int r0 = k * 64;
      for (int ix = 0; ix < 7936 / P; ++ix)
        for (int u = 0; u < 64 / rpi; ++u) {
          int r;
          if (pattern == 11) { const int cls = u & 1; r = cls + 2 * ((lane / lpr) + rpi * (u >> 1)); }  // 8 rows two apart, then the other parity
          else r = (lane / lpr) + rpi * u;
          const int row = r0 + r;
          char* p = field + (size_t)row * PITCH + ((row & 1) ? 64 : 0) + (size_t)ix * P + (size_t)(lane % lpr) * 16;
          if (row < NY) put2<NT>(p, val);
        }
    } else if (pattern == 7) {
      char* p = out + (size_t)t * 65536;
      for (int i = 0; i < 64; ++i) put2<NT>(p + (size_t)i * 1024 + (size_t)lane * 16, val);
    } else if (pattern == 8) {
      for (int i = 0; i < 64; ++i) {
        const uint32_t h = hash32(t * 64u + i) * 8u + (lane >> 3);
        const size_t line = (size_t)(hash32(h) % (uint32_t)((size_t)NF * FIELD / 128));
        put2<NT>(out + line * 128 + (size_t)(lane & 7) * 16, val);
      }
    } else if (pattern == 9) {
      for (int i = 0; i < 64; ++i) {
        const size_t piece = (size_t)(hash32(t * 64u + i) % (uint32_t)((size_t)NF * FIELD / 1024));
        put2<NT>(out + piece * 1024 + (size_t)lane * 16, val);
      }
    }
  }
}
static unsigned* d_counter = nullptr;
extern "C" int shape_run(void* buf, int pattern_in, int wpc, float* ms_out, double* bytes_out) {
  const int nt = pattern_in >= 100; const int pattern = pattern_in % 100;
  if (!d_counter && hipMalloc(&d_counter, 4) != hipSuccess) return 1;
  int n_tasks; double bytes;
  if (pattern <= 2 || pattern >= 13) { n_tasks = NF * 7; bytes = (double)NF * NY * 7168; }
  else if (pattern == 10) { n_tasks = NF * 7; bytes = (double)NF * NY * 7168 / 2; }
  else if (pattern == 12) { n_tasks = NF * 3; bytes = (double)NF * NY * 6144; }
  else if (pattern <= 6 || pattern == 11) { n_tasks = NF * 16; bytes = (double)NF * NY * 7936; }
  else { n_tasks = (int)((size_t)NF * FIELD / 65536); bytes = (double)n_tasks * 65536; }
  const int wgs_per_cu = wpc / 4;
  const size_t lds = 160 * 1024 / wgs_per_cu - 512;
  hipFuncSetAttribute(reinterpret_cast<const void*>(shapes<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(shapes<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemsetAsync(d_counter, 0, 4, 0);
    hipEventRecord(a, 0);
    if (nt) hipLaunchKernelGGL(shapes<1>, dim3(256 * wgs_per_cu), dim3(256), lds, 0, (char*)buf, pattern, n_tasks, d_counter);
    else hipLaunchKernelGGL(shapes<0>, dim3(256 * wgs_per_cu), dim3(256), lds, 0, (char*)buf, pattern, n_tasks, d_counter);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  hipEventDestroy(a); hipEventDestroy(b);
  *ms_out = best; *bytes_out = bytes;
  return (int)hipGetLastError();
}
