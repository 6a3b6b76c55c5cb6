// vhp_planner_dev.hip.h -- the device side of a planner iteration that is not the sweep: the types of the planner's state in HBM and
// the epilogue (union, labels, heuristic, arg-min, the pick of the next pivot).  Shared by vhp_planner.hip.h (the two-kernel iteration
// of the front sweep and the speculative solve) and vhp_lat.hip (the ONE-kernel iteration: eight workgroups sweep, the others run this
// epilogue behind them).  Reference: src/visibilityBasedSolver.cpp:417-430 (the loop body after the store), :127-141 (the loop).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vhp.h"

namespace vhp {

constexpr uint32_t kUnlabelled32 = 0xffffffffu;

struct PlannerCtl {
  int nb;        // nb_of_sources_
  int done;      // loop finished (any reason)
  int status;    // vhp_status of the solve
  int iters;     // planner steps executed
};

struct PlannerKey {
  unsigned long long h;     // bits of the heuristic (h >= 0, so the bit pattern orders like the value)
  unsigned long long rank;  // push order, lower = earlier
  int x, y;
};

struct PlannerDev {
  double* vis_global;
  double* vis_local;
  double* vis_other;    // the local field that is NOT in use this iteration (two take turns when the sweep leaves dark cells
                        // unwritten): the epilogue clears it for the next sweep; nullptr: one local field, fully written by every sweep
  uint32_t* label;
  int32_t* pivots;      // (x, y) pairs, lightSources_
  PlannerCtl* ctl;
  int* rec;             // {done, nb, x, y of pivot nb}: what the iteration's sweep needs to know, as ONE 16-byte record behind the control
                        // block (the latency sweep reads it with one load instead of the control words and then the pivot: LatArgs::pivot_rec)
  PlannerKey* partial;  // one per epilogue workgroup
  unsigned int* ticket; // epilogue workgroups that have delivered their partial (the last one picks the pivot)
  double threshold, scale;
  int end_x, end_y;
  unsigned long long max_iter;
  int local_uncached;   // the local fields live in uncached device memory: a one-kernel iteration needs no L2 write-back / invalidate between
                        // its sweep and its epilogue (vhp_lat.hip vhp_planner_iteration)
};

__device__ __forceinline__ bool key_less(const PlannerKey& a, const PlannerKey& b) {
  return a.h < b.h || (a.h == b.h && a.rank < b.rank);
}

// eval_d, visibilityBasedSolver.h:112-115: first product in double, second in int
__device__ __forceinline__ double eval_d_dev(int ax, int ay, int bx, int by) {
  const int dx = ax - bx, dy = ay - by;
  return __builtin_sqrt((double)dx * dx + (double)(dy * dy));
}

// push-order rank of cell (x, y) for the pivot (sx, sy): quadrants in the order Q1..Q4,
// inside a quadrant the x offset is the outer loop and the y offset the inner one
// (solver.cpp:392-395 etc.); a cell several quadrants visit counts where it is first pushed.
__device__ __forceinline__ unsigned long long push_rank(int nx, int ny, int sx, int sy, int x, int y) {
  const long long dx = x - sx, dy = y - sy;
  long long q, r;
  if (dx >= 0 && dy >= 0) { q = 0; r = dx * (ny - sy) + dy; }
  else if (dx < 0 && dy >= 0) { q = 1; r = (-dx) * (ny - sy) + dy; }
  else if (dy < 0 && (dx < 0 || (dx == 0 && sx >= 1))) { q = 2; r = (-dx) * (long long)sy + (-dy); }
  else { q = 3; r = dx * (long long)sy + (-dy); }
  (void)nx;
  return ((unsigned long long)q << 40) | (unsigned long long)r;
}

// The minimum over a whole wavefront (every lane active), in every lane: four DPP steps inside the rows of 16 (pairs, quads, the
// half-row and the row mirrored), the four rows' results through scalar registers.  (A butterfly of ds_bpermute takes 12 trips through
// the LDS crossbar per 64-bit value: 0.9 us per round of the speculative epilogue's pick.)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));  // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), e = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, e));
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  const unsigned hi = wave_min_u32((unsigned)(v >> 32));
  const unsigned lo = wave_min_u32((unsigned)(v >> 32) == hi ? (unsigned)v : 0xffffffffu);
  return ((unsigned long long)hi << 32) | lo;
}

// Is this lane the one that holds the wavefront's minimum of (h, rank)?  (The lowest such lane: ranks of lit cells are distinct, keys
// of lanes that found nothing are all alike.)  Every lane of the wavefront calls it.
__device__ __forceinline__ bool wave_holds_min(const PlannerKey& k) {
  const unsigned long long hmin = wave_min_u64(k.h);
  const unsigned long long rmin = wave_min_u64(k.h == hmin ? k.rank : ~0ull);
  const unsigned long long holders = __ballot(k.h == hmin && k.rank == rmin);
  return (int)(threadIdx.x & 63) == __ffsll((long long)holders) - 1;
}

// Step 2: the per-cell body of updateVisibility() that follows the store (solver.cpp:417-430)
// over every cell the sweep visited: max-union into vis_global, first-lit labelling, heuristic
// of every lit cell, arg-min of (h, push rank).  Embarrassingly parallel and coalesced.
// The epilogue's launch shape: kEpilogueBlocks workgroups of kEpilogueThreads (blocks <= threads: the last workgroup merges one partial
// per thread).  Measured on maze_6 (bench.py --workload c4, us per pivot of the device loop, one box): 256 x 256 31.3, 128 x 512 30.3,
// 64 x 1024 38.6 (profiles/r05_planner_epilogue_shapes.txt); later in round 5, with the polls overlapped: 96 x 512 27.2, 128 x 512 26.5,
// 192 x 512 27.3, 256 x 512 28.8.
#ifndef VHP_EPI_BLOCKS
#define VHP_EPI_BLOCKS 128
#endif
#ifndef VHP_EPI_THREADS
#define VHP_EPI_THREADS 512
#endif
constexpr int kEpilogueBlocks = VHP_EPI_BLOCKS;
constexpr int kEpilogueThreads = VHP_EPI_THREADS;
constexpr int kEpilogueWaves = kEpilogueThreads / 64;
// cells of a thread per batch of loads in vhp_planner_epilogue (measured on maze_6 -- 277 380 cells, 65 536 threads: 4.2 cells each --,
// us per pivot: 4: 26.4-26.6, 5 -- one batch instead of a full one and a quarter of a second --: 26.9, 6: 27.0)
#ifndef VHP_EPI_CELLS
#define VHP_EPI_CELLS 4
#endif
constexpr int kEpiCells = VHP_EPI_CELLS;
constexpr int kSpecPartials = kEpilogueBlocks * kEpilogueWaves;  // the speculative solve's epilogue (same launch shape) leaves one partial per wavefront
// THREADS: the workgroup's size; block of n_blocks: which share of the cells is this workgroup's.  sweeps_done (or nullptr): the
// iteration's sweep runs in the SAME launch (vhp_lat.hip vhp_planner_iteration): its workgroups count themselves in there when their
// stores are out, and this workgroup may read the local field only after all n_sweeps of them have -- the loads that do not depend on
// the sweep (the union, the labels, the other local field) are on their way by then.
template <int THREADS>
__device__ __forceinline__ void planner_epilogue_body(int nx, int ny, const PlannerDev& d, int block, int n_blocks, const unsigned* sweeps_done, unsigned n_sweeps) {
  constexpr int kWaves = THREADS / 64;
  __shared__ PlannerKey slots[kWaves];
  const size_t cells = (size_t)nx * ny;
  // kEpiCells cells of a thread at a time, their loads issued together -- the label's too, whether or not the cell turns out lit: the
  // kernel is a chain of memory latencies, not of bytes.  The first batch is asked for with the control block, before anybody knows
  // what that says (the loads are harmless if the loop has ended; the stores wait): one trip to memory less per iteration.
  const size_t stride = (size_t)n_blocks * THREADS;
  double vv[kEpiCells], oo[kEpiCells], ot[kEpiCells];
  uint32_t ll[kEpiCells];
  auto load_rest = [&](size_t k0) {   // (what the iteration's sweep does not write)
#pragma unroll
    for (int u = 0; u < kEpiCells; ++u) {
      const size_t k = k0 + u * stride;
      const bool in = k < cells;
      oo[u] = in ? d.vis_global[k] : 0.0;
      ll[u] = in ? d.label[k] : 0u;
      ot[u] = (d.vis_other && in) ? d.vis_other[k] : 0.0;   // (cleared without looking -- a store per cell instead of this load -- measured: 23.0 -> 23.4 us per pivot)
    }
  };
  auto load_local = [&](size_t k0) {
#pragma unroll
    for (int u = 0; u < kEpiCells; ++u) {
      const size_t k = k0 + u * stride;
      vv[u] = k < cells ? d.vis_local[k] : 0.0;
    }
  };
  size_t k0 = (size_t)block * THREADS + threadIdx.x;
  const int done = d.ctl->done, nb = d.ctl->nb;
  load_rest(k0);
  if (!sweeps_done) load_local(k0);
  if (done) return;
  if (sweeps_done) {
    // The sweep's workgroups are other workgroups of this launch: one lane waits for their count (an agent-scope load: they sit on
    // other CUs, behind other L2s), the workgroup's barrier, then every wavefront's acquire -- what it loads from here on is what
    // the sweep stored (MI355X_MICROARCH "Valid forms": the writers drained their stores and released at agent scope before counting).
    if (threadIdx.x == 0)
      while (__hip_atomic_load(sweeps_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_sweeps) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (!d.local_uncached) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    load_local(k0);
  }
  const int sx = d.pivots[2 * nb], sy = d.pivots[2 * nb + 1];
  PlannerKey best;
  best.h = ~0ull;
  best.rank = ~0ull;
  best.x = best.y = -1;
  // The pivots a label can name (lightSources_[0 .. nb]) into LDS, beside the first batch of loads: a lit cell's parent then
  // costs no round trip to memory of its own (beyond kPivLds entries: from global memory as before).
  constexpr int kPivLds = 1024;
  __shared__ int piv_lds[2 * kPivLds];
  const int n_piv = nb + 1 < kPivLds ? nb + 1 : kPivLds;
  for (int t = threadIdx.x; t < 2 * n_piv; t += THREADS) piv_lds[t] = d.pivots[t];
  __syncthreads();
  for (;;) {
#pragma unroll
    for (int u = 0; u < kEpiCells; ++u) {
      const size_t k = k0 + u * stride;
      if (k >= cells) continue;
      if (ot[u] != 0.0) d.vis_other[k] = 0.0;   // (what iteration n - 1 left in the field the next sweep writes)
      const int y = (int)((unsigned)k / (unsigned)nx), x = (int)((unsigned)k - (unsigned)y * (unsigned)nx);  // (cells < 2^31: VHP_MAX_SIDE^2)
      // column 0 / row 0 are swept only when the pivot lies on them: unvisited cells are
      // neither united, labelled nor pushed
      if ((x == 0 && sx > 0) || (y == 0 && sy > 0)) continue;
      const double v = vv[u];
      const double old = oo[u];
      const double g = fmax(v, old);  // :417-418
      if (g != old) d.vis_global[k] = g;  // (most cells of most iterations are dark: nothing to write)
      if (g >= d.threshold) {  // :424-430 (v >= threshold implies g >= threshold)
        uint32_t lab = ll[u];
        if (v >= d.threshold && lab == kUnlabelled32) {  // :419-423
          lab = (uint32_t)nb;
          d.label[k] = lab;
        }
        const int px = lab < (uint32_t)kPivLds ? piv_lds[2 * lab] : d.pivots[2 * lab];
        const int py = lab < (uint32_t)kPivLds ? piv_lds[2 * lab + 1] : d.pivots[2 * lab + 1];
        const double h = (d.scale * g) + (eval_d_dev(x, y, d.end_x, d.end_y) + eval_d_dev(x, y, px, py));
        PlannerKey c;
        c.h = (unsigned long long)__double_as_longlong(h);
        c.rank = push_rank(nx, ny, sx, sy, x, y);
        c.x = x;
        c.y = y;
        if (key_less(c, best)) best = c;
      }
    }
    k0 += kEpiCells * stride;
    if (k0 >= cells) break;
    load_rest(k0);
    load_local(k0);
  }
  const int wave = threadIdx.x >> 6;
  if (wave_holds_min(best)) slots[wave] = best;   // (the one lane that holds the wavefront's minimum of (h, rank))
  // The workgroup whose partial arrives last merges them all and picks the next pivot: no third kernel, no single-thread
  // walk over the partials.  Cross-CU hand-off (MI355X_MICROARCH "Valid forms"): every storing wavefront drains its
  // stores, the workgroup's barrier, then one lane: partial, agent-scope release, drained again, ticket.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    PlannerKey b = slots[0];
    for (int w = 1; w < kWaves; ++w)
      if (key_less(slots[w], b)) b = slots[w];
    d.partial[block] = b;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = __hip_atomic_fetch_add(d.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)n_blocks - 1 ? 1 : 0;
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!is_last) return;
  // Wavefront 0 alone, on registers: the partials (agent-scope loads: other CUs wrote them), the loop condition's cell (written by
  // some workgroup of this kernel) and the iteration count are asked for together; the minimum of (h, rank) by DPP; lane 0 stores
  // the pick -- ls_ = top(); ++nb_of_sources_; lightSources_[nb] = ls_; the loop condition (solver.cpp:127-141).
  // (Until round 5: a partial per thread, a butterfly of ds_bpermute per wavefront, a barrier, one thread over the wavefronts'
  // minima and then the pick's trips to memory one after the other.)
  if (wave != 0) return;
  const int lane = (int)threadIdx.x;
  const double ge = __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(d.vis_global + (size_t)d.end_y * nx + d.end_x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const int iters = d.ctl->iters;
  constexpr int kPerLane = (kEpilogueBlocks + 63) / 64;   // (n_blocks <= kEpilogueBlocks)
  PlannerKey k;
  k.h = ~0ull; k.rank = ~0ull; k.x = k.y = -1;
#pragma unroll
  for (int t = 0; t < kPerLane; ++t) {
    const int i = lane + 64 * t;
    const bool in = i < n_blocks;
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(d.partial + (in ? i : 0));
    PlannerKey o;
    o.h = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o.rank = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long xy = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o.x = (int)(unsigned)xy;
    o.y = (int)(unsigned)(xy >> 32);
    if (in && key_less(o, k)) k = o;
  }
  const unsigned long long hmin = wave_min_u64(k.h);
  const unsigned long long rmin = wave_min_u64(k.h == hmin ? k.rank : ~0ull);
  const unsigned long long holders = __ballot(k.h == hmin && k.rank == rmin);
  const int wl = holders ? __ffsll((long long)holders) - 1 : 0;
  const int wx = __builtin_amdgcn_readlane(k.x, wl), wy = __builtin_amdgcn_readlane(k.y, wl);  // (-1: nothing reached the threshold)
  if (lane != 0) return;
  d.ticket[0] = 0;  // for the next iteration (kernels of one stream run in order)
  d.ticket[1] = 0;  // (... and the count of the sweep's workgroups, where the sweep runs in the epilogue's launch)
  d.ctl->iters = iters + 1;
  if (wx < 0) {  // the reference would call top() on an empty heap
    d.ctl->status = VHP_ERR_NOTHING_LIT;
    d.ctl->done = 1;
    d.rec[0] = 1;
    return;
  }
  const int nbn = nb + 1;
  int px = wx, py = wy, status = -1;
  if ((unsigned long long)nbn > d.max_iter) status = VHP_ERR_MAX_ITER;  // :134-139
  else if (ge > d.threshold) { px = d.end_x; py = d.end_y; status = VHP_OK; }  // :127, :141
  d.ctl->nb = nbn;
  d.pivots[2 * nbn] = px;
  d.pivots[2 * nbn + 1] = py;
  if (status >= 0) { d.ctl->status = status; d.ctl->done = 1; }
  *reinterpret_cast<int4*>(d.rec) = make_int4(status >= 0 ? 1 : 0, nbn, px, py);
}

}  // namespace vhp
