// vhp_stream.hip -- gfx950 build of the streaming sweep (vhp_stream.hpp) and its launcher.
#include "vhp_stream_launch.h"

#include <hip/hip_runtime.h>

#include "vhp.h"
#include "vhp_stream.hpp"

namespace vhp {
namespace stream {

constexpr int kUnits = 8;  // units per source: 4 quadrants x {x-major, y-major} octant
constexpr int kCuSlots = 2048;  // per-CU arrival counters (XCC, SE, SH, CU packed into 11 bits)

#ifdef VHP_DIAG_WGTIME  // diagnostic builds only (tools/stream_timeline.py): when each workgroup ran and how busy its wavefronts were
__device__ unsigned long long g_wgtime[8 * 16384];
__device__ unsigned long long g_prof[8 * 8];  // per wavefront of workgroup 0: the XWave / YWave prof[] words
#define VHP_WG_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define VHP_WG_STAMP(var)
#endif

// A unit is the x-major (OCT == 0) or the y-major (OCT == 1) octant of one quadrant.  A TEAM of wavefronts sweeps it:
//   x-major: the whole workgroup -- W sweeping wavefronts and their W flushers;
//   y-major: W sweeping wavefronts and the DiagWave -- the whole workgroup (W = 7), or, where the LDS holds two such
//            units side by side, half of it (W = 3): two y-major units per workgroup.  (A y-major unit keeps 2 of its
//            wavefronts busy on average -- strips get shorter and start a block apart -- and y-major units were half of
//            all workgroup-time: tools/stream_timeline.py.)
// PHASE 0 (before the workgroup barrier): the team clears and sets up its progress words.  PHASE 1: the wavefronts run.
template <int DX, int DY, int OCT, int PHASE, typename OutT>
__device__ __forceinline__ void octant_phase(const Map& m, OutT* field, int sx, int sy, int W, int tile_slots, double* lds, int team_tid,
                                             int team_threads, int uid) {
  (void)uid;
  Quad<DX, DY> g;
  g.init(m.nx, m.ny, sx, sy);
  if (g.empty() || (OCT == 1 && g.Py == 0)) return;  // uniform for the team
  const Layout L = make_layout(W, m.nx, m.ny, OCT == 0, tile_slots);
  if (PHASE == 0) {
    Progress<DX, DY> prog;
    prog.bind(lds, L, W);
    prog.clear(OCT == 0, L, team_tid, team_threads);
    if (team_tid == 0) prog.setup(g, OCT == 0);  // (writes other words than clear())
    return;
  }
  // from here on the wavefronts synchronise through their progress words
  const int wave = uniform(team_tid >> 6);
#ifdef VHP_DIAG_WGTIME
  unsigned long long busy = 0;
  int units = 0;
  const unsigned long long wg_t0 = wall_clock64(), c_begin = __builtin_readcyclecounter();
#endif
  if (OCT == 0) {
    if (wave < W) {
      XWave<DX, DY, OutT> xw;
      xw.init(m, g, field, wave, W, lds, L);
      while (xw.active) {
        while (!xw.ready()) ready_backoff();
        lds_acquire();
        VHP_WG_STAMP(c0);
        xw.run_unit();
#ifdef VHP_DIAG_WGTIME
        busy += __builtin_readcyclecounter() - c0;
        ++units;
#endif
      }
      xw.finish();
#ifdef VHP_DIAG_WGTIME
      if (uid < 64 && (team_tid & 63) == 0)
        for (int k = 0; k < 6; ++k) g_prof[wave * 8 + k] = xw.prof[k];
#endif
    } else {
      XWave<DX, DY, OutT> fw;
      fw.init_flusher(m, g, field, wave - W, W, lds, L);
      for (;;) {
        VHP_WG_STAMP(c0);
        if (!fw.drain_one()) break;
#ifdef VHP_DIAG_WGTIME
        busy += __builtin_readcyclecounter() - c0;  // (includes the wait for the descriptor)
        ++units;
#endif
      }
    }
  } else {
    if (wave < W) {
      YWave<DX, DY, OutT> yw;
      yw.init(m, g, field, wave, W, lds, L);
      while (yw.active) {
        while (!yw.ready()) ready_backoff();
        lds_acquire();
        VHP_WG_STAMP(c0);
        yw.run_unit();
#ifdef VHP_DIAG_WGTIME
        busy += __builtin_readcyclecounter() - c0;
        ++units;
#endif
      }
#ifdef VHP_DIAG_WGTIME
      if (uid < 64 && wave < 4 && (team_tid & 63) == 0)
        for (int k = 0; k < 6; ++k) g_prof[(4 + wave) * 8 + k] = yw.prof[k];
#endif
    } else {
      DiagWave<DX, DY> dw;
      dw.init(m, g, W, lds, L);
      while (dw.active) {
        while (!dw.ready()) ready_backoff();
        lds_acquire();
        dw.run_unit();
      }
    }
  }
#ifdef VHP_DIAG_WGTIME
  if ((team_tid & 63) == 0 && uid < 16384 / 2) {
    unsigned long long* wv = g_wgtime + (size_t)uid * 16;
    if (wave == 0) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      wv[0] = wg_t0;
      wv[1] = wall_clock64();
      wv[2] = (unsigned long long)units | ((unsigned long long)g.ni << 16) | ((unsigned long long)g.nj << 32) | ((unsigned long long)OCT << 48);
      wv[3] = ((unsigned long long)xcc << 32) | hwid;
      wv[4] = __builtin_readcyclecounter() - c_begin;
    }
    wv[8 + (wave & 7)] = busy;
  }
#endif
}

// one phase of unit `unit` (source * 8 + quadrant * 2 + octant kind) for a team; units of rejected sources do nothing
template <int PHASE, typename OutT>
__device__ __forceinline__ void unit_phase(const Map& m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride,
                                           int* __restrict__ err_flag, int unit, int W, int tile_slots, double* lds, int team_tid,
                                           int team_threads) {
  if (unit < 0) return;
  const int s = unit / kUnits, qo = unit - s * kUnits;
  const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
  if (sx < 0 || sy < 0 || sx >= m.nx || sy >= m.ny) {
    if (PHASE == 0 && team_tid == 0 && qo == 0) atomicOr(err_flag, 1);
    return;
  }
  OutT* field = out + (size_t)s * field_stride;
  if (PHASE == 0 && qo == 0) {
    // rows/columns no quadrant covers read as zero; the x-major unit of quadrant 1 always exists
    if (sx > 0)
      for (int y = team_tid; y < m.ny; y += team_threads) field[(size_t)y * m.nx] = OutT(0);
    if (sy > 0)
      for (int x = team_tid; x < m.nx; x += team_threads) field[x] = OutT(0);
  }
  switch (qo) {
    case 0: octant_phase<+1, +1, 0, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
    case 1: octant_phase<+1, +1, 1, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
    case 2: octant_phase<-1, +1, 0, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
    case 3: octant_phase<-1, +1, 1, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
    case 4: octant_phase<-1, -1, 0, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
    case 5: octant_phase<-1, -1, 1, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
    case 6: octant_phase<+1, -1, 0, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
    default: octant_phase<+1, -1, 1, PHASE>(m, field, sx, sy, W, tile_slots, lds, team_tid, team_threads, unit); break;
  }
}

// Persistent workgroups: the grid is as many workgroups as the chip holds at once (two 8-wavefront ones per CU: LDS and
// registers), and each takes the next SLOT -- one x-major unit, or one or two y-major units (vhp_stream_order) -- from a
// global queue until the queue is empty.  A full-size octant is 4 MB, half of a CU's fair share of a 256-source batch:
// with one workgroup per unit, dealt out in launch order, the CUs that happened to get several large octants set the
// length of the launch.  Pulling balances the bytes per CU by itself: a workgroup busy with a large octant simply pulls
// nothing else.
// dynamic LDS = stream LDS size of the launcher (lds_bytes); y_half = doubles of one y-major team when two share a
// workgroup, 0 when a y-major unit has the whole workgroup.
template <typename OutT, int WX>
__global__ void __launch_bounds__(128 * WX, WX <= 4 ? 2 : 1)
vhp_stream_sweep(Map m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride, int* __restrict__ err_flag,
                 const int2* __restrict__ order, unsigned long long* __restrict__ queue, int* __restrict__ cu_slots,
                 const int* __restrict__ n_slots_ptr, int tile_slots, int y_half) {
  extern __shared__ double lds[];
  const int n_slots = uniform(*n_slots_ptr);  // written by vhp_stream_order
  __shared__ int next_slot;
  // Two workgroups share a CU.  The first to arrive on a CU pulls from the head of the queue (largest slots first),
  // the second from its tail (smallest first), until the two ends meet: every CU then carries one stream of large
  // units and one of small ones, instead of some CUs starting with two of the largest (which share that CU's path to
  // memory: measured 0.78 ms with every workgroup pulling from the head, 0.73 ms this way, 256 sources on 1000^2).
  // `queue` packs both ends in one word (low half: slots taken from the head, high half: from the tail) so that a pull
  // sees both consistently.
  __shared__ int from_tail;
  if (threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned key = (((xcc & 7u) << 8) | ((hwid >> 8) & 0xffu)) & (kCuSlots - 1);  // XCC | SE, SH, CU of this workgroup
    from_tail = atomicAdd(&cu_slots[key], 1) & 1;
  }
  __syncthreads();
  const bool tail = from_tail != 0;
  const int wave = uniform((int)(threadIdx.x >> 6));
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long old = atomicAdd(queue, tail ? (1ull << 32) : 1ull);
      const unsigned h = (unsigned)old, t = (unsigned)(old >> 32);
      // (Keeping the smallest units for the end of the launch, so that the last units running are small ones instead of
      // the middle-sized ones at which the two ends meet, was measured and lost 6 %: what a large unit needs beside it
      // on its CU are the very small ones.)
      next_slot = (h + t >= (unsigned)n_slots) ? n_slots : (tail ? n_slots - 1 - (int)t : (int)h);
    }
    __syncthreads();
    const int slot = uniform(next_slot);
    if (slot >= n_slots) return;
    const int2 us = order[slot];  // .x: the slot's unit; .y: a second y-major unit beside it, -1 (none), or -2: .x has the whole workgroup
    // the team this wavefront belongs to: the whole workgroup, or (two y-major units) one half of it
    int unit = us.x, W = WX, team_tid = (int)threadIdx.x, team_threads = 2 * 64 * WX;
    double* team_lds = lds;
    if ((us.x & 1) != 0) {  // y-major
      if (y_half > 0 && us.y != -2) {
        const int half = wave >= WX ? 1 : 0;
        unit = half ? us.y : us.x;
        W = WX - 1;
        team_tid = (int)threadIdx.x - half * 64 * WX;
        team_threads = 64 * WX;
        team_lds = lds + half * y_half;
      } else {
        W = 2 * WX - 1;
      }
    }
    unit_phase<0>(m, src_xy, out, field_stride, err_flag, unit, W, tile_slots, team_lds, team_tid, team_threads);
    __syncthreads();
    unit_phase<1>(m, src_xy, out, field_stride, err_flag, unit, W, tile_slots, team_lds, team_tid, team_threads);
    __syncthreads();  // every wavefront is through with this slot's LDS before the next slot's setup
  }
}

// Launch order.  One workgroup (1024 threads) counting-sorts the y-major units by cell count, pairs neighbours of that
// order (units of similar size: a pair takes as long as its larger member) when two y-major units share a workgroup,
// then counting-sorts the slots -- x-major units and y-major pairs / units -- by weight, largest first, and zeroes the
// pull queue.  Units of out-of-range sources weigh nothing and sort last.
constexpr int kBuckets = 1024;
__device__ __forceinline__ int exclusive_scan_1024(int v, int* wave_tot) {  // blockDim.x == 1024; returns the exclusive prefix of v
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
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
  __syncthreads();
  return before + inc - v;
}
constexpr int kWholeCols = 640;  // a y-major unit with more columns than this keeps the whole workgroup
__global__ void __launch_bounds__(1024) vhp_stream_order(const int32_t* __restrict__ src_xy, int n_src, int nx, int ny, int pair_y,
                                                         int2* __restrict__ order, int* __restrict__ ysorted, int* __restrict__ n_slots_out,
                                                         unsigned long long* __restrict__ queue, int* __restrict__ cu_slots) {
  if (threadIdx.x == 0) *queue = 0ull;
  for (int k = threadIdx.x; k < kCuSlots; k += blockDim.x) cu_slots[k] = 0;
  __shared__ int hist[kBuckets];
  __shared__ int start[kBuckets];
  __shared__ int wave_tot[16];
  const int n_units = n_src * kUnits;
  const double inv_area = 1.0 / ((double)nx * (double)ny);
  // cells of a unit, as a fraction of the grid (an octant holds at most ~5/8 of the grid's cells)
  auto cells_of = [&](int u) -> double {
    const int s = u / kUnits, qo = u - s * kUnits, q = qo >> 1;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return -1.0;
    const int ni = (q == 0 || q == 3) ? nx - sx : sx;
    const int nj = (q < 2) ? ny - sy : sy;
    double cells = 0.0;
    if (ni > 0 && nj > 0) {
      if ((qo & 1) == 0) {  // x-major: rows j < min(ni, nj), cells (i, j), j <= i < ni
        const double r = ni < nj ? ni : nj;
        cells = r * ni - r * (r - 1) * 0.5;
      } else {              // y-major: columns i < min(ni, nj - 1), cells (i, j), i < j < nj
        const double c = ni < nj - 1 ? ni : nj - 1;
        cells = c > 0 ? c * (nj - 1) - c * (c - 1) * 0.5 : 0.0;
      }
    }
    return cells * inv_area;
  };
  auto bucket_of_weight = [&](double a) {  // bucket 0 = largest
    if (a < 0.0) return kBuckets - 1;
    a *= 1.45;
    return (kBuckets - 1) - (int)((a > 1.0 ? 1.0 : a) * (kBuckets - 1));
  };
  const int n_y = n_units / 2;
  // A y-major unit with many strips keeps the whole workgroup (7 sweeping wavefronts): with 3, wavefront 0 would sweep
  // its strips 0, 3, 6 one after the other, and the largest units set the length of the launch (measured: 0.69 ms for
  // one such unit inside a 0.75 ms launch).
  __shared__ int n_big_sh;
  auto is_big = [&](int u) {
    if (!pair_y) return false;
    const int s = u / kUnits, q = (u - s * kUnits) >> 1;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return false;
    const int ni = (q == 0 || q == 3) ? nx - sx : sx, nj = (q < 2) ? ny - sy : sy;
    return (ni < nj - 1 ? ni : nj - 1) > kWholeCols;
  };
  // 1. the y-major units (odd unit numbers): the big ones first, then the others, each group largest first
  auto ybucket = [&](int u) { return (bucket_of_weight(cells_of(u)) >> 1) + (is_big(u) ? 0 : kBuckets / 2); };
  hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) n_big_sh = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < n_y; k += blockDim.x) {
    atomicAdd(&hist[ybucket(2 * k + 1)], 1);
    if (is_big(2 * k + 1)) atomicAdd(&n_big_sh, 1);
  }
  __syncthreads();
  start[threadIdx.x] = exclusive_scan_1024(hist[threadIdx.x], wave_tot);
  __syncthreads();
  for (int k = threadIdx.x; k < n_y; k += blockDim.x) ysorted[atomicAdd(&start[ybucket(2 * k + 1)], 1)] = 2 * k + 1;
  __syncthreads();
  // 2. the slots: every x-major unit (weighted 1.1: it takes a little longer per cell), the big y-major units on their
  //    own, the others in pairs of neighbours of the sorted order (weight of the larger) -- or all on their own
  const int n_big = n_big_sh, n_small = n_y - n_big;
  const int n_yslots = pair_y ? n_big + (n_small + 1) / 2 : n_y;
  const int n_slots = n_y + n_yslots;
  if (threadIdx.x == 0) *n_slots_out = n_slots;
  auto slot_units = [&](int k) -> int2 {
    if (k < n_y) return make_int2(2 * k, -1);
    const int j = k - n_y;
    if (!pair_y) return make_int2(ysorted[j], -1);
    if (j < n_big) return make_int2(ysorted[j], -2);
    const int a = n_big + 2 * (j - n_big);
    return make_int2(ysorted[a], a + 1 < n_y ? ysorted[a + 1] : -1);
  };
  auto slot_bucket = [&](int2 us) { return bucket_of_weight(cells_of(us.x) * ((us.x & 1) ? 1.0 : 1.1)); };
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < n_slots; k += blockDim.x) atomicAdd(&hist[slot_bucket(slot_units(k))], 1);
  __syncthreads();
  start[threadIdx.x] = exclusive_scan_1024(hist[threadIdx.x], wave_tot);
  __syncthreads();
  for (int k = threadIdx.x; k < n_slots; k += blockDim.x) {
    const int2 us = slot_units(k);
    order[atomicAdd(&start[slot_bucket(us)], 1)] = us;
  }
}

}  // namespace stream

namespace {
constexpr size_t kLdsLimit = 160 * 1024 - 256;  // dynamic LDS: the kernel's 8 bytes of static LDS and the allocation granule stay free
// sweeping wavefronts of an x-major unit; a workgroup is 2 * kWX wavefronts.  (8, i.e. one 16-wavefront workgroup per
// CU, was measured: 0.92 against 0.75 ms at 1000^2 with 256 sources, 1.39 / 1.45 at 2048^2, level at 4096^2.)
constexpr int kWX = 4;
struct StreamShape {
  int tile_slots;  // 3 (a window of slack for the flushers) where two workgroups still fit a CU's LDS, else 2
  int y_half;      // doubles of one y-major team when two y-major units share a workgroup (kWX - 1 sweeping wavefronts each), else 0
  size_t lds;      // dynamic LDS per workgroup, bytes
};
// Tile slots and workgroups per CU.  On large grids one workgroup per CU is the better shape, and slack between a
// sweeping wavefront and its flusher helps on top: 128 sources at 4096^2 take 5.67 ms with two slots and two workgroups
// per CU, 4.92 ms with three or four slots and two workgroups (a diagnostic build), 4.31 ms with three slots and ONE
// workgroup per CU, 4.28 with four, 4.37 with six; at 3072^2 3.30 / 2.51 / 2.45 ms.  Where three slots fit twice into a
// CU's LDS (sides up to ~2000) two workgroups stay better or level (1000^2: 0.64 against 0.87 ms with one workgroup and
// 4-8 slots; 2048^2: 1.41 against 1.38-1.39).
StreamShape shape_for(int nx, int ny, int slots) {
  using namespace stream;
  const int x_total = make_layout(kWX, nx, ny, true, slots).total;
  const int y_pair = (make_layout(kWX - 1, nx, ny, false, slots).total + 1) & ~1;
  const int y_whole = make_layout(2 * kWX - 1, nx, ny, false, slots).total;
  const int pair_total = 2 * y_pair > y_whole ? 2 * y_pair : y_whole;  // (a large y-major unit of a paired launch has the whole workgroup)
  const size_t paired = (size_t)(x_total > pair_total ? x_total : pair_total) * 8;
  const size_t whole = (size_t)(x_total > y_whole ? x_total : y_whole) * 8;
  // two y-major units per workgroup unless that costs the CU its second workgroup
  StreamShape sh{};
  sh.tile_slots = slots;
  const bool pair = paired * 2 <= kLdsLimit || paired <= whole;  // (with one workgroup per CU pairing is level: 4.36 / 4.26 ms at 4096^2)
  sh.y_half = pair ? y_pair : 0;
  sh.lds = pair ? paired : whole;
  return sh;
}
StreamShape pick_stream_shape(int nx, int ny, int force_tile_slots) {
  if (force_tile_slots) {
    const StreamShape forced = shape_for(nx, ny, force_tile_slots);
    if (forced.lds <= kLdsLimit) return forced;  // (a forced depth that does not fit this grid: the automatic shape instead)
  }
  StreamShape sh = shape_for(nx, ny, 3);
  if (sh.lds * 2 <= kLdsLimit) return sh;  // three slots, two workgroups per CU
  sh = shape_for(nx, ny, 4);
  if (sh.lds <= kLdsLimit) return sh;      // four slots, one workgroup per CU
  sh = shape_for(nx, ny, 3);
  if (sh.lds <= kLdsLimit) return sh;
  return shape_for(nx, ny, 2);
}

template <typename OutT>
hipError_t launch_t(const StreamArgs& a) {
  using namespace stream;
  auto k = vhp_stream_sweep<OutT, kWX>;
  const StreamShape sh = pick_stream_shape(a.nx, a.ny, a.force_tile_slots);
  if (sh.lds == 0 || sh.lds > kLdsLimit) return hipErrorInvalidValue;
  if (a.raise_lds) {
    hipError_t e = a.raise_lds(reinterpret_cast<const void*>(k), sh.lds);
    if (e != hipSuccess) return e;
  }
  Map m;
  m.rows = a.rows; m.cols = a.cols; m.recip = a.recip;
  m.wpr = a.wpr; m.wpc = a.wpc; m.nx = a.nx; m.ny = a.ny;
  // scratch: the queue word, the per-CU arrival counters, the launch order of the slots, the sorted y-major units
  const int n_units = a.n_src * kUnits, n_y = n_units / 2;
  unsigned long long* queue = reinterpret_cast<unsigned long long*>(a.d_queue);
  int* cu_slots = a.d_queue + 2;
  int2* ord = reinterpret_cast<int2*>(a.d_queue + 2 + kCuSlots);
  int* ysorted = a.d_queue + 2 + kCuSlots + 2 * n_units;
  int* n_slots_dev = ysorted + n_y;
  const int pair_y = sh.y_half > 0 ? 1 : 0;
  if (a.ev_begin) (void)hipEventRecord(a.ev_begin, a.stream);  // the order pre-kernel is part of what a launch costs
  hipLaunchKernelGGL(vhp_stream_order, dim3(1), dim3(1024), 0, a.stream, a.d_src, a.n_src, a.nx, a.ny, pair_y, ord, ysorted, n_slots_dev, queue,
                     cu_slots);
  const int n_slots = n_units;  // an upper bound: the order kernel counts them (big y-major units stay unpaired)
  int per_cu = (int)(kLdsLimit / sh.lds);
  if (per_cu > 16 / (2 * kWX)) per_cu = 16 / (2 * kWX);  // 16 wavefronts per CU: 128 vector registers each
  const int resident = per_cu * a.n_cus;
  const int grid = n_slots < resident ? n_slots : resident;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(128 * kWX), sh.lds, a.stream, m, a.d_src, static_cast<OutT*>(a.d_out), a.field_stride,
                     a.d_err, (const int2*)ord, queue, cu_slots, (const int*)n_slots_dev, sh.tile_slots, sh.y_half);
  const hipError_t e = hipGetLastError();
  if (a.ev_end) (void)hipEventRecord(a.ev_end, a.stream);
  return e;
}
}  // namespace

#ifdef VHP_DIAG_WGTIME
extern "C" int vhp_debug_read_wgtime(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(stream::g_wgtime), (size_t)n_words * 8);
}
extern "C" int vhp_debug_clear_wgtime() {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(stream::g_wgtime));
  if (e != hipSuccess) return (int)e;
  e = hipMemset(p, 0, sizeof(stream::g_wgtime));
  if (e != hipSuccess) return (int)e;
  return (int)hipDeviceSynchronize();
}
extern "C" int vhp_debug_read_prof(unsigned long long* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(stream::g_prof), 64 * 8);
}
#endif

size_t stream_queue_bytes(int n_src) {
  const size_t n_units = stream::kUnits * (size_t)n_src;
  return (size_t)(2 + stream::kCuSlots + 2 * n_units + n_units / 2 + 8) * sizeof(int);  // queue, per-CU counters, slots (int2), sorted y-major units
}

// sweeping wavefronts per x-major unit, or 0 if the grid is not one the streaming kernel takes
int stream_strips(int nx, int ny) {
  if (nx <= 0 || ny <= 0 || (nx & 7) != 0 || nx > VHP_MAX_SIDE || ny > VHP_MAX_SIDE) return 0;
  const StreamShape sh = pick_stream_shape(nx, ny, 0);
  return (sh.lds != 0 && sh.lds <= kLdsLimit) ? kWX : 0;
}
bool stream_supported(int nx, int ny) { return stream_strips(nx, ny) != 0; }

hipError_t launch_stream(const StreamArgs& a) {
  if (stream_strips(a.nx, a.ny) == 0) return hipErrorInvalidValue;
  return a.dtype == VHP_F64 ? launch_t<double>(a) : launch_t<float>(a);
}

}  // namespace vhp
