// vhp_stream.hip -- gfx950 build of the streaming sweep (vhp_stream.hpp) and its launcher.
#include "vhp_stream_launch.h"

#include <hip/hip_runtime.h>

#include "vhp.h"
#include "vhp_stream.hpp"

namespace vhp {
namespace stream {

constexpr int kUnits = 4;  // quadrants per source

#ifdef VHP_EXP_WGTIME  // diagnostic builds only (tools/stream_timeline.py): when each workgroup ran and how busy its wavefronts were
__device__ unsigned long long g_wgtime[8 * 16384];
#define VHP_WG_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define VHP_WG_STAMP(var)
#endif

template <int DX, int DY, typename OutT>
__device__ __forceinline__ void run_quadrant(const Map& m, OutT* field, int sx, int sy, int W, double* lds) {
  Quad<DX, DY> g;
  g.init(m.nx, m.ny, sx, sy);
  if (g.empty()) return;  // uniform for the workgroup
  const Layout L = make_layout(W, m.nx, m.ny);
  {
    Progress<DX, DY> prog;
    prog.bind(lds, L, W);
    if (threadIdx.x == 0) prog.setup(g);
  }
  __syncthreads();  // the only workgroup barrier: from here on the wavefronts synchronise through their progress words
  const int wave = uniform((int)(threadIdx.x >> 6));
#ifndef VHP_EXP_NOPRIO
  // The launch is as long as its largest quadrants (a full-size one is 1/256 of a 256-source batch: a CU's fair
  // share all by itself), and two workgroups share a CU: the larger the quadrant, the higher the issue priority of its
  // wavefronts, so that it runs as if alone and the smaller neighbour fills the gaps.
  {
    const long area = (long)g.ni * g.nj, full = (long)m.nx * m.ny;
    if (4 * area >= 3 * full) __builtin_amdgcn_s_setprio(3);
    else if (2 * area >= full) __builtin_amdgcn_s_setprio(2);
    else if (4 * area >= full) __builtin_amdgcn_s_setprio(1);
  }
#endif
#ifdef VHP_EXP_WGTIME
  unsigned long long busy = 0;
  int units = 0;
  const unsigned long long wg_t0 = wall_clock64(), c_begin = __builtin_readcyclecounter();
#endif
  if (wave < W) {
    XWave<DX, DY, OutT> xw;
    xw.init(m, g, field, wave, W, lds, L);
    while (xw.active) {
      while (!xw.ready()) __builtin_amdgcn_s_sleep(4);
      lds_acquire();
      VHP_WG_STAMP(c0);
      xw.run_unit();
#ifdef VHP_EXP_WGTIME
      busy += __builtin_readcyclecounter() - c0;
      ++units;
#endif
    }
  } else {
    YWave<DX, DY, OutT> yw;
    yw.init(m, g, field, wave - W, W, lds, L);
    while (yw.active) {
      while (!yw.ready()) __builtin_amdgcn_s_sleep(4);
      lds_acquire();
      VHP_WG_STAMP(c0);
      yw.run_unit();
#ifdef VHP_EXP_WGTIME
      busy += __builtin_readcyclecounter() - c0;
      ++units;
#endif
    }
  }
#ifdef VHP_EXP_WGTIME
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 16384 / 2) {
    unsigned long long* wv = g_wgtime + (size_t)blockIdx.x * 16;
    if (wave == 0) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      wv[0] = wg_t0;
      wv[1] = wall_clock64();
      wv[2] = (unsigned long long)units | ((unsigned long long)g.ni << 16) | ((unsigned long long)g.nj << 32);
      wv[3] = ((unsigned long long)xcc << 32) | hwid;
      wv[4] = __builtin_readcyclecounter() - c_begin;
    }
    if (wave < 8) wv[8 + wave] = busy;   // waves 0..W-1: x-major, W..2W-1: y-major
  }
#endif
}

// grid = 4 * n_src workgroups of 128*W threads; dynamic LDS = make_layout(W, nx, ny).total doubles.
// Workgroup b sweeps unit order[b] (largest quadrants first: with in-order dispatch that is LPT scheduling).
template <typename OutT, int W>
__global__ void __launch_bounds__(128 * W, W == 4 ? 4 : 4)
vhp_stream_sweep(Map m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride, int* __restrict__ err_flag,
                 const int* __restrict__ order) {
  extern __shared__ double lds[];
  const int unit = order ? order[blockIdx.x] : (int)blockIdx.x;
  const int s = unit / kUnits, q = unit - s * kUnits;
  const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
  if (sx < 0 || sy < 0 || sx >= m.nx || sy >= m.ny) {
    if (threadIdx.x == 0 && q == 0) atomicOr(err_flag, 1);
    return;
  }
  OutT* field = out + (size_t)s * field_stride;
  if (q == 0) {
    // rows/columns no quadrant covers read as zero; quadrant 1 always exists
    if (sx > 0)
      for (int y = threadIdx.x; y < m.ny; y += blockDim.x) field[(size_t)y * m.nx] = OutT(0);
    if (sy > 0)
      for (int x = threadIdx.x; x < m.nx; x += blockDim.x) field[x] = OutT(0);
    run_quadrant<+1, +1>(m, field, sx, sy, W, lds);
  } else if (q == 1) {
    run_quadrant<-1, +1>(m, field, sx, sy, W, lds);
  } else if (q == 2) {
    run_quadrant<-1, -1>(m, field, sx, sy, W, lds);
  } else {
    run_quadrant<+1, -1>(m, field, sx, sy, W, lds);
  }
}

// Launch order of the units: a quadrant's work grows with its area.  One workgroup counting-sorts the units by
// area, largest first.  Units of out-of-range sources sort last.
constexpr int kBuckets = 1024;
__global__ void __launch_bounds__(1024) vhp_stream_order(const int32_t* __restrict__ src_xy, int n_src, int nx, int ny, int* __restrict__ order) {
  __shared__ int hist[kBuckets];
  __shared__ int start[kBuckets];
  __shared__ int wave_tot[16];
  const int n_units = n_src * kUnits;
  const double inv_area = 1.0 / ((double)nx * (double)ny);
  auto bucket_of = [&](int u) {
    const int s = u / kUnits, q = u - s * kUnits;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return kBuckets - 1;
    const int ni = (q == 0 || q == 3) ? nx - sx : sx;
    const int nj = (q < 2) ? ny - sy : sy;
    const double a = (ni <= 0 || nj <= 0) ? 0.0 : (double)ni * (double)nj * inv_area;
    return (kBuckets - 1) - (int)(a * (kBuckets - 1));  // bucket 0 = largest
  };
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (int u = threadIdx.x; u < n_units; u += blockDim.x) atomicAdd(&hist[bucket_of(u)], 1);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int v = hist[threadIdx.x];
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) before += k < wv ? wave_tot[k] : 0;
  start[threadIdx.x] = before + inc - v;
  __syncthreads();
  for (int u = threadIdx.x; u < n_units; u += blockDim.x) order[atomicAdd(&start[bucket_of(u)], 1)] = u;
}

}  // namespace stream

namespace {
constexpr size_t kLdsLimit = 160 * 1024;
size_t lds_bytes(int nx, int ny, int W) { return (size_t)stream::make_layout(W, nx, ny).total * sizeof(double); }

template <typename OutT, int W>
hipError_t launch_t(const StreamArgs& a) {
  using namespace stream;
  auto k = vhp_stream_sweep<OutT, W>;
  const size_t lds = lds_bytes(a.nx, a.ny, W);
  if (a.raise_lds) {
    hipError_t e = a.raise_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return e;
  }
  Map m;
  m.rows = a.rows; m.cols = a.cols; m.recip = a.recip;
  m.wpr = a.wpr; m.wpc = a.wpc; m.nx = a.nx; m.ny = a.ny;
  const int* order = nullptr;
  if (a.d_order && a.n_src >= 8) {
    hipLaunchKernelGGL(vhp_stream_order, dim3(1), dim3(1024), 0, a.stream, a.d_src, a.n_src, a.nx, a.ny, a.d_order);
    order = a.d_order;
  }
  if (a.ev_begin) (void)hipEventRecord(a.ev_begin, a.stream);
  hipLaunchKernelGGL(k, dim3((unsigned)(a.n_src * kUnits)), dim3(128 * W), lds, a.stream, m, a.d_src, static_cast<OutT*>(a.d_out),
                     a.field_stride, a.d_err, order);
  const hipError_t e = hipGetLastError();
  if (a.ev_end) (void)hipEventRecord(a.ev_end, a.stream);
  return e;
}
}  // namespace

#ifdef VHP_EXP_WGTIME
extern "C" int vhp_debug_read_wgtime(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(stream::g_wgtime), (size_t)n_words * 8);
}
#endif

int stream_strips(int nx, int ny) {
  if (nx <= 0 || ny <= 0 || (nx & 7) != 0 || nx > VHP_MAX_SIDE || ny > VHP_MAX_SIDE) return 0;
  if (2 * lds_bytes(nx, ny, 4) <= kLdsLimit) return 4;   // two 8-wavefront workgroups per CU
  if (lds_bytes(nx, ny, 8) <= kLdsLimit) return 8;       // one 16-wavefront workgroup per CU
  if (lds_bytes(nx, ny, 4) <= kLdsLimit) return 4;
  return 0;
}
bool stream_supported(int nx, int ny) { return stream_strips(nx, ny) != 0; }

hipError_t launch_stream(const StreamArgs& a) {
  const int W = stream_strips(a.nx, a.ny);
  if (W == 0) return hipErrorInvalidValue;
  if (a.dtype == VHP_F64) return W == 4 ? launch_t<double, 4>(a) : launch_t<double, 8>(a);
  return W == 4 ? launch_t<float, 4>(a) : launch_t<float, 8>(a);
}

}  // namespace vhp
