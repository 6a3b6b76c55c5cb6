// vhp_pool.hip -- gfx950 build of the pool sweep (vhp_pool.hpp), its unit-order pre-kernel and its launcher.
#include "vhp_batch_launch.h"

#include <hip/hip_runtime.h>

#include "vhp.h"
#include "vhp_pool.hpp"

namespace vhp {
namespace pool {

// One persistent workgroup per CU; every wavefront is a Worker.  kWaves wavefronts: three per SIMD, 168 vector registers each
// (the builds use 121 and 114).
#ifndef VHP_POOL_WAVES
#define VHP_POOL_WAVES 12
#endif
constexpr int kWaves = VHP_POOL_WAVES;
// the build for widths that are not a multiple of 8: its tiles are three windows (12.8 KB a wavefront), nine wavefronts fit the LDS
#ifndef VHP_POOL_WAVES_ANYW
#define VHP_POOL_WAVES_ANYW 9
#endif
constexpr int kWavesAny = VHP_POOL_WAVES_ANYW;


template <typename OutT, bool ANYW>
__global__ void __launch_bounds__(64 * kWaves, 1) vhp_pool_sweep(Args<OutT> a, int n_ctx) {
  extern __shared__ double lds[];
  constexpr int W = ANYW ? kWavesAny : kWaves;
  const Layout L = make_layout(W, n_ctx, a.m.nx, a.m.ny, ANYW ? kTStrideAny : kTStride);
  Worker<OutT, ANYW>::clear(lds, L, (int)threadIdx.x, 64 * W);
  __syncthreads();
  Worker<OutT, ANYW> wk;
  wk.init(a, lds, L, uniform((int)(threadIdx.x >> 6)), (int)blockIdx.x);
  wk.run();
}

// Launch order: one workgroup counting-sorts the 8 n_src units by the length of their march (then by cell count), longest first,
// sets the pull queue to where the launch starts pulling (queue0: 0, or behind the units that the contexts take by workgroup
// index, Args::static_round), lays the units' boundary lines out in the scratch (first 64-entry block of unit u: exclusive prefix
// sum of UnitGeo::line_blocks in unit order) and writes the launch's records: recs[k] = {unit, sx | sy << 16, line base, 0} of the
// k-th unit in launch order.  Units of out-of-range sources weigh nothing, sort last and carry -1 in place of the source (they
// are rejected when they are installed).  If the lines do not fit `capacity_blocks` (the launcher sizes the scratch by an upper
// bound, so they do) nothing is swept and the error flag says so.
constexpr int kBuckets = 1024;
constexpr int kOrderLdsUnits = 8192;  // batches of up to 1024 sources keep the per-unit scratch of the ordering in LDS

// exclusive prefix sums over arr[0..n), in place, by the 1024 threads of the workgroup (a contiguous chunk each); returns the total
template <typename Arr>
__device__ __forceinline__ int block_exclusive_scan_1024(Arr arr, int n, int* wave_tot) {
  const int per = (n + 1023) / 1024;
  const int lo = (int)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  int mine = 0;
  for (int k = lo; k < hi; ++k) mine += arr[k];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { before += k < wv ? wave_tot[k] : 0; all += wave_tot[k]; }
  int at = before + inc - mine;
  for (int k = lo; k < hi; ++k) { const int v = arr[k]; arr[k] = at; at += v; }
  __syncthreads();
  return all;
}

// (PerUnit: int* in LDS, or line_base itself in global memory for batches whose units do not fit there)
template <typename PerUnit>
__device__ __forceinline__ void order_units(const int32_t* __restrict__ src_xy, int n_src, int nx, int ny, int* __restrict__ order,
                                            int* __restrict__ line_base, int* __restrict__ recs, long long capacity_blocks,
                                            unsigned long long* __restrict__ queue, unsigned long long queue0,
                                            int* __restrict__ err_flag, int* hist, int* wave_tot, PerUnit blocks) {
  const int n_units = n_src * kUnits;
  const double inv_area = 1.0 / ((double)nx * (double)ny), inv_side = 1.0 / (double)(nx > ny ? nx : ny);
  hist[threadIdx.x] = 0;
  __syncthreads();
  // one pass over the units: the blocks of a unit's boundary lines and its bucket (0 = first), kept for the two scatters below
  for (int u = threadIdx.x; u < n_units; u += 1024) {
    const int s = u / kUnits, qo = u - s * kUnits;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    int nb = 0, bucket = kBuckets - 1;
    if (!(sx < 0 || sy < 0 || sx >= nx || sy >= ny)) {
      UnitGeo g;
      g.init(nx, ny, qo, sx, sy);
      nb = g.line_blocks();
      double cells = 0.0;
      if (g.n_strips > 0) {
        if (g.x_major) { const double r = g.rows_total; cells = r * g.ni - r * (r - 1) * 0.5; }
        else { const double c = g.cols_total; cells = c * (g.nj - 1) - c * (c - 1) * 0.5; }
      }
      // Launch order = how early a unit has to start, not how large it is: a unit is as long as its march (a thin octant along
      // an axis is one strip, i.e. one wavefront, for 15 blocks: sorted by cells it started last and the launch ended on it),
      // so the key is mostly the march length; among equally long ones the larger first.
      const double march = (double)(g.x_major ? g.ni : g.nj) * inv_side;
      double f = 0.8 * march + 0.2 * (cells * inv_area * 1.6 > 1.0 ? 1.0 : cells * inv_area * 1.6);
      if (f > 1.0) f = 1.0;
      bucket = (kBuckets - 1) - (int)(f * (kBuckets - 1));
    }
    blocks[u] = nb | (bucket << 20);  // (nb < 2^20: a unit has at most 129 strips of 130 blocks)
    atomicAdd(&hist[bucket], 1);
  }
  __syncthreads();
  (void)block_exclusive_scan_1024(hist, kBuckets, wave_tot);
  for (int u = threadIdx.x; u < n_units; u += 1024) order[atomicAdd(&hist[blocks[u] >> 20], 1)] = u;
  __syncthreads();
  for (int u = threadIdx.x; u < n_units; u += 1024) blocks[u] &= (1 << 20) - 1;
  __syncthreads();
  const int total = block_exclusive_scan_1024(blocks, n_units, wave_tot);
  const bool fits = (long long)total <= capacity_blocks;
  // the records, in launch order (order[] was written by this workgroup before the barriers above).  If the boundary lines do not fit
  // the scratch, NOTHING is swept: the queue starts exhausted, and every record says so as well (-2) -- the first unit of a context is
  // handed out by workgroup index without a look at the queue (Args::static_round), and its lines would lie beyond the scratch.
  for (int k = threadIdx.x; k < n_units; k += 1024) {
    const int u = order[k], s = u / kUnits;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    const bool inside = !(sx < 0 || sy < 0 || sx >= nx || sy >= ny);
    reinterpret_cast<int4*>(recs)[k] = make_int4(u, !fits ? -2 : inside ? (sx | (sy << 16)) : -1, blocks[u], 0);
  }
  if (threadIdx.x == 0) {
    *queue = fits ? queue0 : (unsigned long long)n_units;
    if (!fits) atomicOr(err_flag, 4);
  }
}

__global__ void __launch_bounds__(1024) vhp_pool_order(const int32_t* __restrict__ src_xy, int n_src, int nx, int ny, int* __restrict__ order,
                                                       int* __restrict__ line_base, int* __restrict__ recs, long long capacity_blocks,
                                                       unsigned long long* __restrict__ queue, unsigned long long queue0,
                                                       int* __restrict__ err_flag) {
  __shared__ int hist[kBuckets];
  __shared__ int wave_tot[16];
  __shared__ int blocks[kOrderLdsUnits];
  VHP_DIAG_TL_RESET
  if (n_src * kUnits <= kOrderLdsUnits) order_units(src_xy, n_src, nx, ny, order, line_base, recs, capacity_blocks, queue, queue0, err_flag, hist, wave_tot, blocks);
  else order_units(src_xy, n_src, nx, ny, order, line_base, recs, capacity_blocks, queue, queue0, err_flag, hist, wave_tot, line_base);
}

}  // namespace pool

namespace {
constexpr size_t kLdsLimit = 160 * 1024;
constexpr int kQueueInts = 16;  // the pull counter (and padding) ahead of the order array

// scratch of a launch: [pull counter, recs[4 n_units], order[n_units], line_base[n_units]] [diagonal lines] [boundary lines]
int diag_stride_of(int nx, int ny) { return ((nx < ny ? nx : ny) + 64 + 15) & ~15; }
size_t head_bytes(int n_src) { return (((size_t)(kQueueInts + 6 * pool::kUnits * (size_t)n_src) * sizeof(int)) + 255) & ~(size_t)255; }
size_t diag_bytes(int n_src, int nx, int ny) { return (((size_t)n_src * 4 * (size_t)diag_stride_of(nx, ny) * sizeof(double)) + 255) & ~(size_t)255; }
// 64-entry blocks of boundary lines a source can need, an upper bound: over its four quadrants ni * nj sums to nx * ny;
// an x-major unit takes at most (min(ni,nj)/64) * (ni/64 + 2) blocks, a y-major one (ni/128 + 1) * (nj/64 + 2)
long long line_blocks_per_source(int nx, int ny) { return (3LL * nx * ny) / 8192 + (nx + ny) / 4 + 64; }

struct PoolShape { int n_ctx; size_t lds; };
// as many contexts (units a workgroup holds at once) as asked for (default 4) that fit the LDS
// Measured (tools/ab_libs.py on one buffer, final launch order): 256 sources at 1000^2, 2 / 3 / 4 / 5 contexts 0.69 / 0.67-0.70 /
// 0.74 / 0.77 ms; 128 sources at 2048^2, 1 / 2 / 3 contexts 1.25 / 1.38 / 1.41 ms; at 4096^2 1 / 2: 3.88 / 4.80 ms -- units that
// large (a 4096^2 octant is 67 MB, 64 strips) keep every wavefront busy by themselves and only lose to a neighbour.
PoolShape pool_shape(int nx, int ny, int force_ctx, bool anyw) {
  PoolShape s;
  // Round 4 (non-temporal stores, strips claimed ahead; 128 sources, 1 / 2 / 3 contexts, ms): 1280^2 0.617 / 0.538 / 0.561; 1536^2 0.771 /
  // 0.741 / 0.772; 1792^2 0.933 / 0.956 / 1.005; 2048^2 1.106 / 1.189 / 1.233; 3072^2 (64 sources) 1.463 / 1.569 / 1.659; 4096^2 3.46 / 4.18 /
  // 4.35; 1024^2 (256 sources) - / 0.624 / 0.608: three up to 1024, two up to 1664, one above.
  const int maxdim = nx > ny ? nx : ny;
  s.n_ctx = force_ctx > 0 ? force_ctx : (maxdim > 1664 ? 1 : maxdim > 1024 ? 2 : 3);
  if (s.n_ctx > 16) s.n_ctx = 16;
  for (;; --s.n_ctx) {
    s.lds = (size_t)(anyw ? pool::make_layout(pool::kWavesAny, s.n_ctx, nx, ny, pool::kTStrideAny) : pool::make_layout(pool::kWaves, s.n_ctx, nx, ny)).total * 8;
    if (s.lds <= kLdsLimit || s.n_ctx == 1) break;
  }
  return s;
}

template <typename OutT>
hipError_t launch_pool_t(const BatchArgs& a) {
  using namespace pool;
  const bool anyw = pool_needs_anyw<OutT>(a.nx, a.field_stride > 0 ? a.field_stride : (long long)a.nx * a.ny, static_cast<const OutT*>(a.d_out));
  auto k = anyw ? vhp_pool_sweep<OutT, true> : vhp_pool_sweep<OutT, false>;
  const PoolShape sh = pool_shape(a.nx, a.ny, a.pool_contexts, anyw);
  const int waves = anyw ? kWavesAny : kWaves;
  if (sh.lds > kLdsLimit || a.pool_epoch == 0) return hipErrorInvalidValue;
  if (a.raise_lds) {
    hipError_t e = a.raise_lds(reinterpret_cast<const void*>(k), sh.lds);
    if (e != hipSuccess) return e;
  }
  char* scratch = reinterpret_cast<char*>(a.d_queue);
  Args<OutT> g;
  g.m.rows = a.rows; g.m.cols = a.cols; g.m.recip = a.recip;
  g.m.wpr = a.wpr; g.m.wpc = a.wpc; g.m.nx = a.nx; g.m.ny = a.ny;
  g.out = static_cast<OutT*>(a.d_out);
  g.field_stride = a.field_stride;
  g.err_flag = a.d_err;
  g.queue = reinterpret_cast<unsigned long long*>(a.d_queue);
  g.n_units = a.n_src * kUnits;
  int* recs = a.d_queue + kQueueInts;  // (16-byte aligned: the scratch is, and kQueueInts is a multiple of 4)
  int* order = recs + 4 * (size_t)g.n_units;
  int* line_base = order + g.n_units;
  g.recs = recs;
  g.diag = reinterpret_cast<double*>(scratch + head_bytes(a.n_src));
  g.diag_stride = diag_stride_of(a.nx, a.ny);
  g.lines = reinterpret_cast<vhp::lanes::Tagged*>(scratch + head_bytes(a.n_src) + diag_bytes(a.n_src, a.nx, a.ny));
  g.epoch = a.pool_epoch;
  g.busy_cap = a.pool_busy_cap > 0 ? a.pool_busy_cap : waves;
  // two contexts take the largest units left, the others the smallest (0.75 against 0.78 ms with one head at 1000^2)
  g.n_head = a.pool_heads > 0 ? a.pool_heads : (sh.n_ctx >= 3 ? 2 : 1);  // (all three from the head: 0.51 / 0.70 ms on two boxes, this: 0.53 / 0.67)
  if (g.n_head > sh.n_ctx) g.n_head = sh.n_ctx;
  g.tail_limit = (int)((long long)g.n_units * (a.pool_tail_pct > 0 ? a.pool_tail_pct : 15) / 100);  // (100 / 50 / 25 / 15 %: 0.56 / 0.55 / 0.53 / - and - / - / - / 0.67 ms on two boxes; round 4, with non-temporal stores: 5 / 15 / 30 / 60 %: 0.580 / 0.608 / 0.608 / 0.616 ms on a slow buffer, level on a fast one -- within the noise of 1-2 %)
  g.early_ctx = a.pool_early_ctx > 0 ? a.pool_early_ctx : sh.n_ctx;
  g.late_after = (int)((long long)g.n_units * (a.pool_late_pct > 0 ? a.pool_late_pct : 50) / 100);
  // measured (tools/ab_slowfast.py, ab_libs.py; 0 / 16 / 32 / 48 / 64 steps): C3 on a fast buffer 0.485 / 0.468 / 0.463 / 0.461 / 0.460 ms, on a
  // slow one 0.583 / 0.582 / 0.597 / 0.589 / 0.591 (bound by the memory there); C5 3.567 / 3.483 / 3.476 / 3.474 / 3.482; 128 sources at
  // 2048^2 1.316 / - / 1.277 / - / 1.238; 512 at 512^2 0.445 / - / 0.419 / - / 0.416
  g.claim_ahead = a.pool_claim_ahead >= 0 ? a.pool_claim_ahead : 48;
  // the first unit of every context by workgroup index, the queue behind them (Args::static_round)
  g.n_groups = a.n_cus;
  g.static_snake = a.pool_static_round >= 2;
  g.static_round = a.pool_static_round != 0 && g.early_ctx >= sh.n_ctx && (long long)g.n_units >= (long long)sh.n_ctx * a.n_cus;
  const unsigned long long queue0 = g.static_round ? ((unsigned long long)(g.n_head * a.n_cus) | ((unsigned long long)((sh.n_ctx - g.n_head) * a.n_cus) << 32)) : 0ull;
  if (a.ev_begin) (void)hipEventRecord(a.ev_begin, a.stream);  // the order pre-kernel is part of what a launch costs
  hipLaunchKernelGGL(vhp_pool_order, dim3(1), dim3(1024), 0, a.stream, a.d_src, a.n_src, a.nx, a.ny, order, line_base, recs,
                     line_blocks_per_source(a.nx, a.ny) * a.n_src, reinterpret_cast<unsigned long long*>(a.d_queue), queue0, a.d_err);
  hipLaunchKernelGGL(k, dim3((unsigned)a.n_cus), dim3(64 * waves), sh.lds, a.stream, g, sh.n_ctx);
  const hipError_t e = hipGetLastError();
  if (a.ev_end) (void)hipEventRecord(a.ev_end, a.stream);
  return e;
}
}  // namespace

size_t pool_scratch_bytes(int n_src, int nx, int ny) {
  return head_bytes(n_src) + diag_bytes(n_src, nx, ny) + (size_t)line_blocks_per_source(nx, ny) * (size_t)n_src * 64 * sizeof(vhp::lanes::Tagged);
}

bool pool_supported(int nx, int ny) {
  if (nx <= 0 || ny <= 0 || nx > VHP_MAX_SIDE || ny > VHP_MAX_SIDE) return false;
  return pool_shape(nx, ny, 0, false).lds <= kLdsLimit && pool_shape(nx, ny, 0, true).lds <= kLdsLimit;
}


#ifdef VHP_DIAG_TIMELINE
extern "C" int vhp_debug_read_hist(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pool::g_pp_hist), (size_t)n_words * 8);
}
extern "C" int vhp_debug_read_units(unsigned* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pool::g_pp_unit), (size_t)n_words * 4);
}
#endif

hipError_t launch_pool(const BatchArgs& a) {
  if (!pool_supported(a.nx, a.ny)) return hipErrorInvalidValue;
  return a.dtype == VHP_F64 ? launch_pool_t<double>(a) : launch_pool_t<float>(a);
}

}  // namespace vhp
