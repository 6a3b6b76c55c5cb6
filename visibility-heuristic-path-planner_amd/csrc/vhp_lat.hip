// vhp_lat.hip -- gfx950 build of the latency sweep (vhp_lat.hpp) and its launcher.
// Plain field stores here (not the batch kernels' non-temporal ones, vhp_lanes.hpp): a launch of this kernel writes a few fields
// that its caller reads right away -- the planner's epilogue, the host copy of a small batch -- out of the caches.
#define VHP_FIELD_STORE_PLAIN 1
#include "vhp_batch_launch.h"

#include <hip/hip_runtime.h>

#include <algorithm>

#include "vhp.h"
#include "vhp_band.hpp"
#include "vhp_planner_dev.hip.h"

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

#ifdef VHP_LAT_STRIPS  // A/B builds only: the sweep in strips of rows (vhp_lat.hpp), what the kernel was until round 6
template <typename OutT, bool ODD, bool MULTI = false> using LatWorkerT = LatWorker<OutT, ODD>;
#else
template <typename OutT, bool ODD, bool MULTI = false> using LatWorkerT = BandWorker<OutT, ODD, MULTI>;
#endif
// wavefronts of a workgroup: kLatWaves sweepers and, in the band sweep, a storer beside each (16: four per SIMD, 128 vector registers)
constexpr int kLatThreads = 64 * kLatWaves * LatWorkerT<double, false>::kRoles;
constexpr int kLatTilePitch = LatWorkerT<double, false>::kTilePitch;

// MULTI: the build for launches with more than one workgroup per unit (lat_halves below): its bands can read across workgroups
template <typename OutT, bool ODD, bool MULTI>
__global__ void __launch_bounds__(kLatThreads, 1) vhp_lat_sweep(LatArgs<OutT> a) {
  extern __shared__ double lds[];
  const Layout L = make_layout(kLatWaves, 1, a.m.nx, a.m.ny, kLatTilePitch);
#ifdef VHP_DIAG_POOLPROF
  const unsigned long long t_begin = wall_clock64();
#endif
  using WorkerT = LatWorkerT<OutT, ODD, MULTI>;
  WorkerT::clear(lds, L, (int)threadIdx.x, kLatThreads);
  __syncthreads();
  WorkerT wk;
  wk.init(a, lds, L, uniform((int)(threadIdx.x >> 6)));
  wk.run((int)blockIdx.x);
#ifdef VHP_DIAG_POOLPROF
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && threadIdx.x < 64 * 16) {
    unsigned long long* o = g_latprof + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 20;
    for (int k = 0; k < 16; ++k) o[k] = wk.prof[k];
    o[16] = t_begin;
    o[17] = wall_clock64();
  }
#endif
}

// A whole planner iteration as ONE launch (the reference's loop body, src/visibilityBasedSolver.cpp:127-141 with updateVisibility()
// :379-565: sweep, union, labels, heuristic, next pivot): the first eight workgroups are the eight octants of the pivot's sweep, the
// others run the epilogue (vhp_planner_dev.hip.h) behind them -- the loads that do not depend on the sweep on their way while it runs,
// the local field read once the eight have counted themselves in.  (Until round 6: two launches per iteration, 1.7 us between them.)
constexpr int kPlanEpiBlocks = kEpilogueBlocks;   // epilogue workgroups: the epilogue kernel's shape, 128 x 512 (the upper half of such a workgroup returns at once: 64 x 1024 measured 8 us slower per iteration, as in round 5)
template <bool ODD>
__global__ void __launch_bounds__(kLatThreads, 1) vhp_planner_iteration(LatArgs<double> a, PlannerDev d) {
  extern __shared__ double lds[];
  if (blockIdx.x >= (unsigned)kUnits) {
    if (threadIdx.x < (unsigned)kEpilogueThreads)
      planner_epilogue_body<kEpilogueThreads>(a.m.nx, a.m.ny, d, (int)blockIdx.x - kUnits, (int)gridDim.x - kUnits, d.ticket + 1, (unsigned)kUnits);
    return;
  }
  const Layout L = make_layout(kLatWaves, 1, a.m.nx, a.m.ny, kLatTilePitch);
  using WorkerT = LatWorkerT<double, ODD>;
  WorkerT::clear(lds, L, (int)threadIdx.x, kLatThreads);
  __syncthreads();
  WorkerT wk;
  wk.init(a, lds, L, uniform((int)(threadIdx.x >> 6)));
  wk.run((int)blockIdx.x);
  // this octant's stores are out: counted for the epilogue's workgroups (every wavefront drains its stores, the barrier, then one
  // lane releases at agent scope -- the readers sit behind other L2s -- and counts)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!d.local_uncached) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    (void)__hip_atomic_fetch_add(d.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One thread per word of the four diagonal-packed maps (DiagMaps): its 64 cells of the byte map.  Once per vhp_set_map.
__global__ void vhp_pack_diag(const uint8_t* __restrict__ occ, uint64_t* __restrict__ dmap, int nx, int ny) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= DiagMaps::words(nx, ny)) return;
  const size_t by_y0 = DiagMaps::offset(nx, ny, 2);
  const bool by_y = k >= by_y0;
  const int wpd = by_y ? DiagMaps::wpdy(ny) : DiagMaps::wpdx(nx);
  const size_t r = by_y ? k - by_y0 : k;
  const size_t run_all = r / (size_t)wpd;
  const int word = (int)(r - run_all * (size_t)wpd), n_runs = DiagMaps::runs(nx, ny);
  const bool anti = run_all >= (size_t)n_runs;
  const int id = (int)(anti ? run_all - (size_t)n_runs : run_all);
  if (word == 0 || word == wpd - 1) return;  // (the pad words at either end of a run stay zero)
  uint64_t bits = 0;
  for (int t = 0; t < 64; ++t) {
    const int c = (word - 1) * 64 + t;  // x (by x) or y (by y)
    // main: y - x = id - (nx - 1); anti: y + x = id
    const int x = by_y ? (anti ? id - c : c - (id - (nx - 1))) : c;
    const int y = by_y ? c : (anti ? id - c : c + (id - (nx - 1)));
    if (x >= 0 && x < nx && y >= 0 && y < ny && occ[(size_t)y * nx + x]) bits |= 1ull << t;
  }
  dmap[k] = bits;
}

// The units of a launch (8 per source: quadrant x {x-major, y-major}) by falling length of their march, for launches of more workgroups
// than the chip holds at once: a counting sort by steps / 8 in one workgroup (up to 1024 x kLatOrderPerThread units).  The length of a march is what an
// octant's time goes by (T / 16 + T / 64 windows); units of sources outside the grid go last.
constexpr int kLatOrderPerThread = 2;   // (256 sources)
__global__ void __launch_bounds__(1024) vhp_lat_order(const int32_t* __restrict__ src_xy, int n_src, int nx, int ny, int* __restrict__ order) {
  __shared__ int hist[1024], start[1024];
  const int tid = (int)threadIdx.x, n_units = n_src * kUnits;
  hist[tid] = 0;
  __syncthreads();
  int bucket[kLatOrderPerThread], pos[kLatOrderPerThread];
#pragma unroll
  for (int r = 0; r < kLatOrderPerThread; ++r) {
    const int u = tid + 1024 * r;
    bucket[r] = 1023;
    pos[r] = 0;
    if (u < n_units) {
      const int s = u / kUnits, qo = u - s * kUnits, q = qo >> 1;
      const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
      int T = 0;
      if (sx >= 0 && sy >= 0 && sx < nx && sy < ny) {
        const int dx = (q == 0 || q == 3) ? 1 : -1, dy = (q == 0 || q == 1) ? 1 : -1;   // (BandWorker::run's table of directions)
        const int ni = dx > 0 ? nx - sx : sx, nj = dy > 0 ? ny - sy : sy;
        T = (ni > 0 && nj > 0) ? ((qo & 1) == 0 ? ni : nj) : 0;
      }
      bucket[r] = 1023 - imin(T >> 3, 1023);
      pos[r] = atomicAdd(&hist[bucket[r]], 1);
    }
  }
  __syncthreads();
  // inclusive prefix sums of the 1024 buckets (Hillis-Steele on the workgroup's 1024 threads)
  start[tid] = hist[tid];
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid >= d ? start[tid - d] : 0;
    __syncthreads();
    start[tid] += v;
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < kLatOrderPerThread; ++r) {
    const int u = tid + 1024 * r;
    if (u < n_units) order[start[bucket[r]] - hist[bucket[r]] + pos[r]] = u;
  }
}

}  // namespace pool

namespace {
constexpr size_t kLdsLimit = 160 * 1024;
#ifndef VHP_LAT_HALVES_MIN_SIDE
#define VHP_LAT_HALVES_MIN_SIDE 1024
#endif
// Workgroups per unit of a latency-sweep launch (LatArgs::halves; vhp_band.hpp BandWorker).  One up to 1024 cells a side: an octant
// has at most 16 bands there, and what bands 8-15 gain by not waiting for the sweepers of bands 0-7 the hand-over through global
// memory takes back (measured at 1000^2: 117.8 / 118.4 us over eight source positions).  Above: two up to 2048, four up to 4096, eight
// beyond -- an octant of P bands is swept in rounds of 8 x that number, and every round waits for the one before (8192^2, one source:
// 3.47 ms with one workgroup per unit, 1.92 with two, 1.18 with four), halved until the launch is at most twice the chip.  (asked: vhp_set_option "lat_workgroups", 1 / 2 / 4 / 8, for
// measurements and tests; 0: by the size.)
int lat_halves(int n_src, int nx, int ny, int n_cus, int asked) {
  const int cus = n_cus > 0 ? n_cus : 256, side = std::max(nx, ny);
  const int want = asked > 0 ? asked : side > 4 * VHP_LAT_HALVES_MIN_SIDE ? 8 : side > 2 * VHP_LAT_HALVES_MIN_SIDE ? 4 : side > VHP_LAT_HALVES_MIN_SIDE ? 2 : 1;
  int h = 1;
  // (up to twice as many workgroups as CUs -- a launch of more workgroups than CUs is safe, BandWorker::run, and a unit's later
  // workgroups start while its first ones are at their first bands: 16 sources at 4096^2 1265 us with two workgroups per unit, 1193 with
  // four; 8 at 8192^2 2718 with four, 2447 with eight; a number that was asked for is taken as it is)
  // (... beyond two that fit: 32 sources at 1536^2 take 280 us with one workgroup per unit and 316 with two on twice the chip)
  while (2 * h <= want && 2 * h <= 8 && (asked > 0 || 2 * h * pool::kUnits * n_src <= (h >= 2 ? 2 : 1) * cus)) h *= 2;
  return h;
}
size_t lat_lds_bytes(int nx, int ny) { return (size_t)pool::make_layout(pool::kLatWaves, 1, nx, ny, pool::kLatTilePitch).total * 8; }

template <typename OutT>
hipError_t launch_lat_t(const BatchArgs& a, const PlannerDev* pd = nullptr) {
  using namespace pool;
  const bool odd = lat_needs_odd<OutT>(a.nx, a.field_stride, static_cast<const OutT*>(a.d_out));
  int halves = lat_halves(a.n_src, a.nx, a.ny, a.n_cus, a.lat_workgroups);
#if defined(VHP_LAT_STRIPS)
  halves = 1;
#endif
#ifdef VHP_EXP_ONE_KERNEL  // compile-time experiments only: one instantiation instead of eight
  auto k = vhp_lat_sweep<double, false, false>;
  halves = 1;
  if (odd || sizeof(OutT) != 8) return hipErrorInvalidValue;
#else
  auto k = halves > 1 ? (odd ? vhp_lat_sweep<OutT, true, true> : vhp_lat_sweep<OutT, false, true>)
                      : (odd ? vhp_lat_sweep<OutT, true, false> : vhp_lat_sweep<OutT, false, false>);
#endif
  const size_t lds = lat_lds_bytes(a.nx, a.ny);
  if (lds > kLdsLimit || a.pool_epoch == 0) return hipErrorInvalidValue;
  if (a.raise_lds) {
    hipError_t e = a.raise_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return e;
  }
#ifdef VHP_EXP_ONE_KERNEL
  LatArgs<double> g;
#else
  LatArgs<OutT> g;
#endif
  g.m.rows = a.rows; g.m.cols = a.cols; g.m.recip = a.recip;
  g.m.wpr = a.wpr; g.m.wpc = a.wpc; g.m.nx = a.nx; g.m.ny = a.ny;
  g.src_xy = a.d_src;
#ifdef VHP_EXP_ONE_KERNEL
  g.out = static_cast<double*>(a.d_out);
#else
  g.out = static_cast<OutT*>(a.d_out);
#endif
  g.field_stride = a.field_stride;
  g.err_flag = a.d_err;
  g.lines = reinterpret_cast<vhp::lanes::Tagged*>(a.d_queue);
  g.unit_blocks = lat_unit_blocks(a.nx, a.ny);
  g.epoch = a.pool_epoch;
  g.src_index = a.d_src_index;
  g.skip = a.d_skip;
  g.pivot_rec = a.d_pivot_rec;
  g.slot_base = a.d_slot_base;
  g.run_if = a.d_run_if;
  g.dead_cells_are_zero = a.lat_dead_cells_are_zero;
  g.strip_times = nullptr;
  g.dmap = a.dmap;
  if (!g.dmap) return hipErrorInvalidValue;
  // Two workgroups per unit where an octant can have more bands than a workgroup has sweepers and the launch leaves the CUs for it
  // (vhp_band.hpp BandWorker: bands 8-15, 24-31, ... of an octant on the second one).
  g.n_units = a.n_src * kUnits;
  g.halves = halves;
#ifdef VHP_DIAG_POOLPROF
  { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(pool::g_lat_strip_times)) == hipSuccess) g.strip_times = static_cast<unsigned long long*>(p); }
#endif
  if (a.ev_begin) (void)hipEventRecord(a.ev_begin, a.stream);
#ifndef VHP_EXP_ONE_KERNEL
  if constexpr (sizeof(OutT) == 8) {
    if (pd) {  // the planner's iteration: the sweep's eight workgroups and the epilogue's in one launch
      auto kp = odd ? vhp_planner_iteration<true> : vhp_planner_iteration<false>;
      if (a.raise_lds) {
        hipError_t e2 = a.raise_lds(reinterpret_cast<const void*>(kp), lds);
        if (e2 != hipSuccess) return e2;
      }
      g.halves = 1;
      hipLaunchKernelGGL(kp, dim3((unsigned)(kUnits + kPlanEpiBlocks)), dim3(kLatThreads), lds, a.stream, g, *pd);
      const hipError_t e3 = hipGetLastError();
      if (a.ev_end) (void)hipEventRecord(a.ev_end, a.stream);
      return e3;
    }
  }
#endif
  if (g.halves == 1 && a.n_src * kUnits > (a.n_cus > 0 ? a.n_cus : 256) && a.n_src * kUnits <= 1024 * pool::kLatOrderPerThread && a.d_lat_order && !a.d_pivot_rec && !a.d_src_index && !a.d_slot_base) {
    // more workgroups than the chip holds at once: the long units first
    // (a list of its own, not a corner of the scratch: nothing but tagged entries may ever be written where a later launch looks for tags)
    hipLaunchKernelGGL(pool::vhp_lat_order, dim3(1), dim3(1024), 0, a.stream, a.d_src, a.n_src, a.nx, a.ny, a.d_lat_order);
    g.order = a.d_lat_order;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)(a.n_src * kUnits * g.halves)), dim3(kLatThreads), lds, a.stream, g);
  const hipError_t e = hipGetLastError();
  if (a.ev_end) (void)hipEventRecord(a.ev_end, a.stream);
  return e;
}
}  // namespace

size_t lat_scratch_bytes(int n_src, int nx, int ny) {
  return (size_t)pool::lat_unit_blocks(nx, ny) * pool::kUnits * (size_t)n_src * 64 * sizeof(vhp::lanes::Tagged);
}
size_t lat_order_bytes() { return (size_t)1024 * pool::kLatOrderPerThread * sizeof(int); }

size_t lat_diag_map_bytes(int nx, int ny) { return pool::DiagMaps::words(nx, ny) * sizeof(uint64_t); }

hipError_t lat_pack_diag_maps(const uint8_t* d_occ, int nx, int ny, uint64_t* d_dmap, hipStream_t stream) {
  const size_t words = pool::DiagMaps::words(nx, ny);
  hipLaunchKernelGGL(pool::vhp_pack_diag, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, d_occ, d_dmap, nx, ny);
  return hipGetLastError();
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

hipError_t launch_lat_planner(const BatchArgs& a, const PlannerDev& d) {
  if (!lat_supported(a.nx, a.ny) || a.dtype != VHP_F64 || a.n_src != 1 || !a.d_pivot_rec) return hipErrorInvalidValue;
  return launch_lat_t<double>(a, &d);
}

}  // namespace vhp
