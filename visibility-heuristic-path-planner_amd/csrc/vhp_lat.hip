// vhp_lat.hip -- gfx950 build of the latency sweep (vhp_lat.hpp) and its launcher.
// Plain field stores here (not the batch kernels' non-temporal ones, vhp_lanes.hpp): a launch of this kernel writes a few fields
// that its caller reads right away -- the planner's epilogue, the host copy of a small batch -- out of the caches.
#define VHP_FIELD_STORE_PLAIN 1
#include "vhp_batch_launch.h"

#include <hip/hip_runtime.h>

#include "vhp.h"
#include "vhp_lat.hpp"

namespace vhp {
namespace pool {

// One workgroup per unit (octant of a quadrant of a source); strip p is wavefront p mod kLatWaves's.  Eight wavefronts: two per
// SIMD, 256 vector registers each (a window keeps its 17 + 16 operands and the 16 pairs of its tile read-out in registers).
#ifndef VHP_LAT_WAVES
#define VHP_LAT_WAVES 8
#endif
constexpr int kLatWaves = VHP_LAT_WAVES;

#ifdef VHP_DIAG_POOLPROF  // diagnostic builds only (tools/lat_timeline.py)
__device__ unsigned long long g_latprof[256 * 16 * 20];
__device__ unsigned long long g_lat_strip_times[64 * 48 * 4];
#endif

template <typename OutT, bool ODD>
__global__ void __launch_bounds__(64 * kLatWaves, 1) vhp_lat_sweep(LatArgs<OutT> a) {
  extern __shared__ double lds[];
  const Layout L = make_layout(kLatWaves, 1, a.m.nx, a.m.ny);
#ifdef VHP_DIAG_POOLPROF
  const unsigned long long t_begin = wall_clock64();
#endif
  LatWorker<OutT, ODD>::clear(lds, L, (int)threadIdx.x, 64 * kLatWaves);
  __syncthreads();
  LatWorker<OutT, ODD> wk;
  wk.init(a, lds, L, uniform((int)(threadIdx.x >> 6)));
  wk.run((int)blockIdx.x);
#ifdef VHP_DIAG_POOLPROF
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) {
    unsigned long long* o = g_latprof + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 20;
    for (int k = 0; k < 16; ++k) o[k] = wk.prof[k];
    o[16] = t_begin;
    o[17] = wall_clock64();
  }
#endif
}

}  // namespace pool

namespace {
constexpr size_t kLdsLimit = 160 * 1024;
size_t lat_lds_bytes(int nx, int ny) { return (size_t)pool::make_layout(pool::kLatWaves, 1, nx, ny).total * 8; }

template <typename OutT>
hipError_t launch_lat_t(const BatchArgs& a) {
  using namespace pool;
  const bool odd = lat_needs_odd<OutT>(a.nx, a.field_stride, static_cast<const OutT*>(a.d_out));
  auto k = odd ? vhp_lat_sweep<OutT, true> : vhp_lat_sweep<OutT, false>;
  const size_t lds = lat_lds_bytes(a.nx, a.ny);
  if (lds > kLdsLimit || a.pool_epoch == 0) return hipErrorInvalidValue;
  if (a.raise_lds) {
    hipError_t e = a.raise_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return e;
  }
  LatArgs<OutT> g;
  g.m.rows = a.rows; g.m.cols = a.cols; g.m.recip = a.recip;
  g.m.wpr = a.wpr; g.m.wpc = a.wpc; g.m.nx = a.nx; g.m.ny = a.ny;
  g.src_xy = a.d_src;
  g.out = static_cast<OutT*>(a.d_out);
  g.field_stride = a.field_stride;
  g.err_flag = a.d_err;
  g.lines = reinterpret_cast<vhp::lanes::Tagged*>(a.d_queue);
  g.unit_blocks = lat_unit_blocks(a.nx, a.ny);
  g.epoch = a.pool_epoch;
  g.src_index = a.d_src_index;
  g.skip = a.d_skip;
  g.slot_base = a.d_slot_base;
  g.run_if = a.d_run_if;
  g.dead_cells_are_zero = a.lat_dead_cells_are_zero;
  g.strip_times = nullptr;
#ifdef VHP_DIAG_POOLPROF
  { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(pool::g_lat_strip_times)) == hipSuccess) g.strip_times = static_cast<unsigned long long*>(p); }
#endif
  if (a.ev_begin) (void)hipEventRecord(a.ev_begin, a.stream);
  hipLaunchKernelGGL(k, dim3((unsigned)(a.n_src * kUnits)), dim3(64 * kLatWaves), lds, a.stream, g);
  const hipError_t e = hipGetLastError();
  if (a.ev_end) (void)hipEventRecord(a.ev_end, a.stream);
  return e;
}
}  // namespace

size_t lat_scratch_bytes(int n_src, int nx, int ny) {
  return (size_t)pool::lat_unit_blocks(nx, ny) * pool::kUnits * (size_t)n_src * 64 * sizeof(vhp::lanes::Tagged);
}

bool lat_supported(int nx, int ny) {
  if (nx <= 0 || ny <= 0 || nx > VHP_MAX_SIDE || ny > VHP_MAX_SIDE) return false;
  // (a y-major workgroup keeps its quadrant's diagonal where an x-major one has its tiles)
  return lat_lds_bytes(nx, ny) <= kLdsLimit && (size_t)pool::kLatWaves * pool::kXRows * pool::kTStride >= (size_t)(nx < ny ? nx : ny);
}

#ifdef VHP_DIAG_POOLPROF
extern "C" int vhp_debug_read_latprof(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pool::g_latprof), (size_t)n_words * 8);
}
extern "C" int vhp_debug_read_lat_strip_times(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pool::g_lat_strip_times), (size_t)n_words * 8);
}
#endif

hipError_t launch_lat(const BatchArgs& a) {
  if (!lat_supported(a.nx, a.ny)) return hipErrorInvalidValue;
  return a.dtype == VHP_F64 ? launch_lat_t<double>(a) : launch_lat_t<float>(a);
}

}  // namespace vhp
