// vhp_stream.hip -- gfx950 build of the streaming sweep (vhp_stream.hpp) and its launcher.
#include "vhp_stream_launch.h"

#include <hip/hip_runtime.h>

#include "vhp.h"
#include "vhp_stream.hpp"

namespace vhp {
namespace stream {

constexpr int kUnits = 8;  // units per source: 4 quadrants x {x-major, y-major} octant
constexpr int kCuSlots = 2048;  // per-CU arrival counters (XCC, SE, SH, CU packed into 11 bits)

#ifdef VHP_EXP_WGTIME  // diagnostic builds only (tools/stream_timeline.py): when each workgroup ran and how busy its wavefronts were
__device__ unsigned long long g_wgtime[8 * 16384];
__device__ unsigned long long g_prof[8 * 8];  // per wavefront of workgroup 0: the XWave / YWave prof[] words
#define VHP_WG_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define VHP_WG_STAMP(var)
#endif

// One unit: the x-major (OCT == 0) or the y-major (OCT == 1) octant of one quadrant, swept by the workgroup of 2*WX
// wavefronts: WX sweeping wavefronts and their WX flushers, or 2*WX - 1 sweeping wavefronts and the DiagWave.
template <int DX, int DY, int OCT, int WX, typename OutT>
__device__ __forceinline__ void run_octant(const Map& m, OutT* field, int sx, int sy, int tile_slots, double* lds, int slot) {
  (void)slot;
  constexpr int W = OCT == 0 ? WX : 2 * WX - 1;
  Quad<DX, DY> g;
  g.init(m.nx, m.ny, sx, sy);
  if (g.empty() || (OCT == 1 && g.Py == 0)) return;  // uniform for the workgroup
  const Layout L = make_layout(W, m.nx, m.ny, OCT == 0, tile_slots);
  {
    Progress<DX, DY> prog;
    prog.bind(lds, L, W);
    prog.clear(OCT == 0, L, (int)threadIdx.x, (int)blockDim.x);
    if (threadIdx.x == 0) prog.setup(g, OCT == 0);  // (writes other words than clear())
  }
  __syncthreads();  // from here on the wavefronts synchronise through their progress words
  const int wave = uniform((int)(threadIdx.x >> 6));
#ifdef VHP_EXP_WGTIME
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
#ifdef VHP_EXP_WGTIME
        busy += __builtin_readcyclecounter() - c0;
        ++units;
#endif
      }
      xw.finish();
#ifdef VHP_EXP_WGTIME
      if (slot < 8 && (threadIdx.x & 63) == 0)
        for (int k = 0; k < 6; ++k) g_prof[wave * 8 + k] = xw.prof[k];
#endif
    } else {
      XWave<DX, DY, OutT> fw;
      fw.init_flusher(m, g, field, wave - W, W, lds, L);
#ifdef VHP_EXP_PRIO
      __builtin_amdgcn_s_setprio(3);
#endif
      for (;;) {
        VHP_WG_STAMP(c0);
        if (!fw.drain_one()) break;
#ifdef VHP_EXP_WGTIME
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
#ifdef VHP_EXP_WGTIME
        busy += __builtin_readcyclecounter() - c0;
        ++units;
#endif
      }
#ifdef VHP_EXP_WGTIME
      if (slot < 8 && wave < 4 && (threadIdx.x & 63) == 0)
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
#ifdef VHP_EXP_WGTIME
  if ((threadIdx.x & 63) == 0 && slot < 16384 / 2) {
    unsigned long long* wv = g_wgtime + (size_t)slot * 16;
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

// Persistent workgroups: the grid is as many workgroups as the chip holds at once (two 8-wavefront ones per CU: LDS and
// registers), and each takes the next unit -- one octant of one quadrant of one source -- from a global queue until the
// queue is empty.  A CU's store path moves only ~9 bytes per clock and a full-size octant is 4 MB, half of a CU's fair
// share of a 256-source batch: with one workgroup per unit, dealt out in launch order, the CUs that happened to get
// several large octants set the length of the launch.  Pulling balances the bytes per CU by itself: a workgroup busy with
// a large octant simply pulls nothing else.
// dynamic LDS = lds_doubles(WX, 2*WX - 1, nx, ny, tile_slots) doubles.
template <typename OutT, int WX>
__global__ void __launch_bounds__(128 * WX, 2)
vhp_stream_sweep(Map m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride, int* __restrict__ err_flag,
                 const int* __restrict__ order, unsigned long long* __restrict__ queue, int* __restrict__ cu_slots, int n_units,
                 int tile_slots) {
  extern __shared__ double lds[];
  __shared__ int next_unit;
  // Two workgroups share a CU.  The first to arrive on a CU pulls from the head of the queue (largest units first),
  // the second from its tail (smallest first), until the two ends meet: every CU then carries one stream of large
  // units and one of small ones, instead of some CUs starting with two of the largest (which share that CU's store
  // path: measured 0.78 ms with every workgroup pulling from the head, 0.73 ms this way, 256 sources on 1000^2).  `queue` packs both ends in
  // one word (low half: units taken from the head, high half: from the tail) so that a pull sees both consistently.
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
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long old = atomicAdd(queue, tail ? (1ull << 32) : 1ull);
      const unsigned h = (unsigned)old, t = (unsigned)(old >> 32);
      // (Keeping the smallest units for the end of the launch, so that the last units running are small ones instead of
      // the middle-sized ones at which the two ends meet, was measured and lost 6 %: what a large unit needs beside it
      // on its CU are the very small ones.)
      next_unit = (h + t >= (unsigned)n_units) ? n_units : (tail ? n_units - 1 - (int)t : (int)h);
    }
    __syncthreads();
    const int slot = uniform(next_unit);
    if (slot >= n_units) return;
    const int unit = order ? order[slot] : slot;
    const int s = unit / kUnits, qo = unit - s * kUnits;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= m.nx || sy >= m.ny) {
      if (threadIdx.x == 0 && qo == 0) atomicOr(err_flag, 1);
    } else {
      OutT* field = out + (size_t)s * field_stride;
      if (qo == 0) {
        // rows/columns no quadrant covers read as zero; the x-major unit of quadrant 1 always exists
        if (sx > 0)
          for (int y = threadIdx.x; y < m.ny; y += blockDim.x) field[(size_t)y * m.nx] = OutT(0);
        if (sy > 0)
          for (int x = threadIdx.x; x < m.nx; x += blockDim.x) field[x] = OutT(0);
      }
      switch (qo) {
        case 0: run_octant<+1, +1, 0, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
        case 1: run_octant<+1, +1, 1, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
        case 2: run_octant<-1, +1, 0, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
        case 3: run_octant<-1, +1, 1, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
        case 4: run_octant<-1, -1, 0, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
        case 5: run_octant<-1, -1, 1, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
        case 6: run_octant<+1, -1, 0, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
        default: run_octant<+1, -1, 1, WX>(m, field, sx, sy, tile_slots, lds, slot); break;
      }
    }
    __syncthreads();  // every wavefront is through with this unit's LDS before the next unit's setup
  }
}

// Launch order of the units: a quadrant's work grows with its area.  One workgroup counting-sorts the units by
// area, largest first.  Units of out-of-range sources sort last.
constexpr int kBuckets = 1024;
__global__ void __launch_bounds__(1024) vhp_stream_order(const int32_t* __restrict__ src_xy, int n_src, int nx, int ny, int* __restrict__ order,
                                                         unsigned long long* __restrict__ queue, int* __restrict__ cu_slots) {
  if (threadIdx.x == 0) *queue = 0ull;
  for (int k = threadIdx.x; k < 2 * kCuSlots; k += blockDim.x) cu_slots[k] = 0;
  if (!order) return;
  __shared__ int hist[kBuckets];
  __shared__ int start[kBuckets];
  __shared__ int wave_tot[16];
  const int n_units = n_src * kUnits;
  const double inv_area = 1.0 / ((double)nx * (double)ny);
  auto bucket_of = [&](int u) {
    const int s = u / kUnits, qo = u - s * kUnits, q = qo >> 1;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return kBuckets - 1;
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
    const double a = cells * inv_area * 1.6;  // an octant holds at most ~5/8 of the grid's cells (the whole grid when 1 cell thin)
    return (kBuckets - 1) - (int)((a > 1.0 ? 1.0 : a) * (kBuckets - 1));  // bucket 0 = largest
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
constexpr int kWX = 4;  // sweeping wavefronts of an x-major unit; a workgroup is 2 * kWX wavefronts
size_t lds_bytes(int nx, int ny, int tile_slots) { return (size_t)stream::lds_doubles(kWX, 2 * kWX - 1, nx, ny, tile_slots) * sizeof(double); }
// three tile slots (a window of slack for the flushers) where two workgroups still fit a CU's LDS, else two
int pick_tile_slots(int nx, int ny) { return 2 * lds_bytes(nx, ny, 3) <= kLdsLimit ? 3 : 2; }

template <typename OutT>
hipError_t launch_t(const StreamArgs& a) {
  using namespace stream;
  auto k = vhp_stream_sweep<OutT, kWX>;
  const int tile_slots = (a.force_strips == 2 || a.force_strips == 3) ? a.force_strips : pick_tile_slots(a.nx, a.ny);
  const size_t lds = lds_bytes(a.nx, a.ny, tile_slots);
  if (lds > kLdsLimit) return hipErrorInvalidValue;
  if (a.raise_lds) {
    hipError_t e = a.raise_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return e;
  }
  Map m;
  m.rows = a.rows; m.cols = a.cols; m.recip = a.recip;
  m.wpr = a.wpr; m.wpc = a.wpc; m.nx = a.nx; m.ny = a.ny;
  // scratch: the queue word, the per-CU arrival counters, the launch order of the units
  unsigned long long* queue = reinterpret_cast<unsigned long long*>(a.d_queue);
  int* cu_slots = a.d_queue + 2;
  int* ord = a.n_src >= 8 ? a.d_queue + 2 + 2 * kCuSlots : nullptr;
  hipLaunchKernelGGL(vhp_stream_order, dim3(1), dim3(1024), 0, a.stream, a.d_src, a.n_src, a.nx, a.ny, ord, queue, cu_slots);
  const int n_units = a.n_src * kUnits;
  int per_cu = (int)(kLdsLimit / lds);
  if (per_cu > 2) per_cu = 2;  // 16 wavefronts per CU: 128 vector registers each
  const int resident = per_cu * a.n_cus;
  const int grid = n_units < resident ? n_units : resident;
  if (a.ev_begin) (void)hipEventRecord(a.ev_begin, a.stream);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(128 * kWX), lds, a.stream, m, a.d_src, static_cast<OutT*>(a.d_out), a.field_stride,
                     a.d_err, (const int*)ord, queue, cu_slots, n_units, tile_slots);
  const hipError_t e = hipGetLastError();
  if (a.ev_end) (void)hipEventRecord(a.ev_end, a.stream);
  return e;
}
}  // namespace

#ifdef VHP_EXP_WGTIME
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

size_t stream_queue_bytes(int n_src) { return (size_t)(2 + 2 * stream::kCuSlots + stream::kUnits * (size_t)n_src) * sizeof(int); }

// sweeping wavefronts per x-major unit, or 0 if the grid is not one the streaming kernel takes
int stream_strips(int nx, int ny) {
  if (nx <= 0 || ny <= 0 || (nx & 7) != 0 || nx > VHP_MAX_SIDE || ny > VHP_MAX_SIDE) return 0;
  if (lds_bytes(nx, ny, 2) <= kLdsLimit) return kWX;
  return 0;
}
bool stream_supported(int nx, int ny) { return stream_strips(nx, ny) != 0; }

hipError_t launch_stream(const StreamArgs& a) {
  if (stream_strips(a.nx, a.ny) == 0) return hipErrorInvalidValue;
  return a.dtype == VHP_F64 ? launch_t<double>(a) : launch_t<float>(a);
}

}  // namespace vhp
