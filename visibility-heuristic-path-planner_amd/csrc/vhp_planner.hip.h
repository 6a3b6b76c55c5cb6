// vhp_planner.hip.h -- the visibility-heuristic planner loop on the GPU.
//
// Replaces solve() (reference src/visibilityBasedSolver.cpp:76-160), updateVisibility()
// (:379-565) and the std::priority_queue / top() arg-min
// (include/solver/visibilityBasedSolver.h:16-21,138; resetQueue() .cpp:65-71).
//
// All planner state lives in HBM for the whole solve: vis_global, vis_local, labels
// (cameFrom_ as uint32), the pivot list (lightSources_) and a small control block.
// One iteration = two kernels on one stream, no host round trip:
//   vhp_planner_sweep    : the front sweep of vhp_sweep.hip.h (its fast path) from the current
//                          pivot into vis_local;
//   vhp_planner_epilogue : per visited cell max-union into vis_global, first-lit labelling,
//                          heuristic h of every lit cell, block arg-min of (h, push rank);
//                          (the workgroup that finishes last merges the block partials, appends the
//                          next pivot, evaluates the loop condition and raises `done`)
// The host enqueues a few iterations at a time and polls the control block; kernels
// of iterations queued past the end see `done` and return at once.
//
// The reference heap is only ever asked for top() and never popped, and libstdc++'s
// push_heap sifts up on strict comparison only, so top() is the FIRST-pushed node of
// minimal h.  Here that is a lexicographic min over (h, push rank): the
// rank of a cell is its position in the reference's push order, a closed form of
// (dx, dy) (tests/schedule_model.py first_touch_rank, checked against the oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "vhp.h"
#include "vhp_sweep.hip.h"
#include "vhp_planner_dev.hip.h"

namespace vhp {

// Step 1 of a planner iteration: the plain front sweep (vhp_sweep.hip.h, fast path) from the
// current pivot into vis_local -- the reference's visibility_ (solver.cpp:386-416).
template <int R, bool MULTI>
__global__ void __launch_bounds__((R == 2 && MULTI) ? 512 : 1024, R >= 4 ? 4 : ((MULTI && R == 2) ? 6 : 8)) vhp_planner_sweep(DevMap m, PlannerDev d) {
  extern __shared__ double lds[];
  if (d.ctl->done) return;
  const int nb = d.ctl->nb;
  const int sx = d.pivots[2 * nb], sy = d.pivots[2 * nb + 1];
  StoreEmit<double, MULTI> emit(d.vis_local, m.nx, m.ny);
  sweep_quadrant<R>(m, emit, sx, sy, blockIdx.x, lds, whole_workgroup());
}

// Step 2 (vhp_planner_dev.hip.h): the per-cell body of updateVisibility() that follows the store, and the pick of the next pivot.
__global__ void __launch_bounds__(kEpilogueThreads) vhp_planner_epilogue(DevMap m, PlannerDev d) {
  planner_epilogue_body<kEpilogueThreads>(m.nx, m.ny, d, (int)blockIdx.x, (int)gridDim.x, nullptr, 0u);
}

__global__ void vhp_planner_init(PlannerDev d, int nx, int start_x, int start_y) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    d.ctl->nb = 0;
    d.ctl->done = 0;
    d.ctl->status = VHP_OK;
    d.ctl->iters = 0;
    d.pivots[0] = start_x;  // lightSources_[0] = start; cameFrom_(start) = 0   (solver.cpp:121-122)
    d.pivots[1] = start_y;
    d.label[(size_t)start_y * nx + start_x] = 0;
    // while (visibility_global_(end) <= threshold), solver.cpp:127: visibility_global_ is all zero here, so a
    // negative threshold skips the loop altogether; lightSources_[0] = end then (:141)
    if (0.0 > d.threshold) {
      d.pivots[0] = d.end_x;
      d.pivots[1] = d.end_y;
      d.ctl->done = 1;
    }
    *reinterpret_cast<int4*>(d.rec) = make_int4(d.ctl->done, 0, d.pivots[0], d.pivots[1]);
  }
}

// labels -> cameFrom_ as the reference stores it (size_t, (size_t)1e15 where unlabelled)
__global__ void vhp_labels_to_u64(const uint32_t* __restrict__ lab, unsigned long long* __restrict__ out, size_t n) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = lab[k] == kUnlabelled32 ? VHP_UNLABELLED : (unsigned long long)lab[k];
}

struct PlannerState {
  size_t cells = 0;
  size_t pivot_cap = 0;
  double* vis_global = nullptr;
  double* vis_local = nullptr;
  double* vis_local2 = nullptr;      // the second local field of the plain solve (see PlannerDev::vis_other)
  bool local_uncached = false;       // the local fields are uncached device memory (PlannerDev::local_uncached)
  double* vis_local_out = nullptr;   // where the last solve left its local field (one of the two)
  uint32_t* label = nullptr;
  unsigned long long* came64 = nullptr;
  int32_t* pivots = nullptr;
  PlannerCtl* ctl = nullptr;
  PlannerKey* partial = nullptr;
  unsigned int* ticket = nullptr;
  // the host's polls of the control block: two pinned copies and their events, so that the next batch of iterations is enqueued
  // before the host waits for the last one's copy (planner_poll)
  PlannerCtl* h_ctl = nullptr;
  hipEvent_t poll_ev[2] = {nullptr, nullptr};
  const uint8_t* h_occ = nullptr;   // the caller's host copy of the map if it has one (vhp_set_map), for the start / end validation
  // launch shape of the front sweep and the per-device dynamic-LDS bookkeeping, set by the caller (vhp_capi.hip)
  int R = 2, W = 8;
  bool multi = false;
  std::function<hipError_t(const void*, size_t)> raise_lds;
  // set by the caller when the latency sweep (vhp_lat.hpp) can sweep this grid: launches it for source number *nb of pivots into
  // out unless *done is set (the same contract as vhp_planner_sweep: everything read on the device when the launch runs)
  // (dark_unwritten: the field holds +0.0 wherever the sweep does not write, dead strips store nothing)
  std::function<hipError_t(const int32_t* pivots, const int* nb, const int* done, const int* rec, double* out, bool dark_unwritten)> lat_sweep;
  // ... or the whole iteration -- that sweep and the epilogue over d -- as ONE launch (vhp_lat.hip vhp_planner_iteration)
  std::function<hipError_t(const PlannerDev& d)> lat_iteration;
  // ... of the speculative loop: the n candidates of `cand` into fields *slot_base .. of `cache`, if *run_if and not *done
  std::function<hipError_t(const int32_t* cand, int n, const int* slot_base, const int* run_if, const int* done, double* cache, bool dark_unwritten)> lat_sweep_k;
};

inline void planner_free(PlannerState& s) {
  if (s.vis_global) (void)hipFree(s.vis_global);
  if (s.vis_local) (void)hipFree(s.vis_local);
  if (s.vis_local2) (void)hipFree(s.vis_local2);
  s.vis_local2 = s.vis_local_out = nullptr;
  if (s.label) (void)hipFree(s.label);
  if (s.came64) (void)hipFree(s.came64);
  if (s.pivots) (void)hipFree(s.pivots);
  if (s.ctl) (void)hipFree(s.ctl);
  if (s.partial) (void)hipFree(s.partial);
  if (s.ticket) (void)hipFree(s.ticket);
  s.ticket = nullptr;
  if (s.h_ctl) (void)hipHostFree(s.h_ctl);
  s.h_ctl = nullptr;
  for (auto& e : s.poll_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  s.cells = 0;
  s.pivot_cap = 0;
  s.vis_global = s.vis_local = nullptr;
  s.label = nullptr;
  s.came64 = nullptr;
  s.pivots = nullptr;
  s.ctl = nullptr;
  s.partial = nullptr;
}

#define VHP_PL_HIP(call)                                                      \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess) {                                                   \
      *msg = std::string(#call) + ": " + hipGetErrorString(e_);               \
      return (int)VHP_ERR_HIP;                                                \
    }                                                                         \
  } while (0)

// The loop both solves run on the host: `enqueue` puts one batch of iterations on the stream (kernels of iterations past the end see
// `done` and return at once); the control block is copied out behind every batch, and the host waits for the copy of batch n only
// after batch n + 1 is on the stream -- the GPU does not idle while the host looks (until round 5 it did: 10-20 us per poll); the price is
// one batch of kernels that return at once when the loop ends.  maze_6: 1.87 -> 1.71 ms per solve (batches of 4 or 16: no better).
template <typename Enqueue>
inline int planner_poll(PlannerState& s, hipStream_t stream, Enqueue enqueue, PlannerCtl* out, std::string* msg) {
  if (!s.h_ctl) {
    VHP_PL_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_ctl), 2 * sizeof(PlannerCtl)));
    for (auto& e : s.poll_ev) VHP_PL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  auto post = [&](int slot) -> int {
    // (on the solve's own stream: the copy on a second stream behind an event of this one measured 6 % slower)
    VHP_PL_HIP(hipMemcpyAsync(&s.h_ctl[slot], s.ctl, sizeof(PlannerCtl), hipMemcpyDeviceToHost, stream));
    VHP_PL_HIP(hipEventRecord(s.poll_ev[slot], stream));
    return VHP_OK;
  };
  // (an error return leaves nothing in flight: kernels of this solve and a copy into h_ctl may be on the stream when a later enqueue fails)
  auto fail = [&](int rc) -> int { (void)hipStreamSynchronize(stream); return rc; };
  int rc = enqueue();
  if (rc != VHP_OK) return fail(rc);
  if ((rc = post(0)) != VHP_OK) return fail(rc);
  for (int q = 0;; ++q) {
    if ((rc = enqueue()) != VHP_OK) return fail(rc);
    if ((rc = post((q + 1) & 1)) != VHP_OK) return fail(rc);
    if (hipEventSynchronize(s.poll_ev[q & 1]) != hipSuccess) { *msg = "hipEventSynchronize (planner poll) failed"; return fail(VHP_ERR_HIP); }
    *out = s.h_ctl[q & 1];
    if (out->done) return VHP_OK;
  }
}

template <int R, bool MULTI>
inline hipError_t launch_planner_fronts(PlannerState& s, const DevMap& m, const PlannerDev& d, int W, hipStream_t stream) {
  const size_t lds = sweep_lds_bytes(R, W, MULTI);
  auto k = vhp_planner_sweep<R, MULTI>;
  {
    hipError_t e = s.raise_lds ? s.raise_lds(reinterpret_cast<const void*>(k), lds)
                               : hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(4), dim3(128 * W), lds, stream, m, d);
  return hipGetLastError();
}

inline int planner_solve(PlannerState& s, const DevMap& m, const uint8_t* d_occ, hipStream_t stream, hipEvent_t ev0,
                         hipEvent_t ev1, int start_x, int start_y, int end_x, int end_y, double threshold,
                         uint64_t max_iter, uint64_t* came_from, double* vis_global, double* vis_local,
                         int32_t* pivots_xy, uint32_t* n_pivots, std::string* msg) {
  const int nx = m.nx, ny = m.ny;
  // the four validity checks of solve(), in the reference's order (solver.cpp:89-116)
  auto valid = [&](int x, int y) { return (size_t)x < (size_t)nx && (size_t)y < (size_t)ny; };
  if (!valid(start_x, start_y)) { *msg = "Start point is out of bounds."; return VHP_ERR_START_OOB; }
  if (!valid(end_x, end_y)) { *msg = "End point is out of bounds."; return VHP_ERR_END_OOB; }
  uint8_t occ_s = 0, occ_e = 0;
  if (s.h_occ) {  // (the map came from the host: its copy answers, no trip to the device)
    occ_s = s.h_occ[(size_t)start_y * nx + start_x];
    occ_e = s.h_occ[(size_t)end_y * nx + end_x];
  } else {
    VHP_PL_HIP(hipMemcpyAsync(&occ_s, d_occ + (size_t)start_y * nx + start_x, 1, hipMemcpyDeviceToHost, stream));
    VHP_PL_HIP(hipMemcpyAsync(&occ_e, d_occ + (size_t)end_y * nx + end_x, 1, hipMemcpyDeviceToHost, stream));
    VHP_PL_HIP(hipStreamSynchronize(stream));
  }
  if (!occ_s) { *msg = "Start point is not valid (occupied)"; return VHP_ERR_START_OCCUPIED; }
  if (!occ_e) { *msg = "End point is not valid (occupied)"; return VHP_ERR_END_OCCUPIED; }
  if (max_iter > (1u << 24)) { *msg = "max_iter too large"; return VHP_ERR_ARG; }

  const size_t cells = (size_t)nx * ny;
  const size_t pcap = 2 * (size_t)(max_iter + 2);
  if (s.cells != cells) {
    planner_free(s);
    VHP_PL_HIP(hipMalloc(&s.vis_global, cells * 8));
    s.local_uncached = std::getenv("VHP_PLANNER_UNCACHED") != nullptr;   // (an experiment of the one-kernel iteration: vhp_capi.hip)
    if (s.local_uncached) VHP_PL_HIP(hipExtMallocWithFlags(reinterpret_cast<void**>(&s.vis_local), cells * 8, hipDeviceMallocUncached));
    else VHP_PL_HIP(hipMalloc(&s.vis_local, cells * 8));
    VHP_PL_HIP(hipMalloc(&s.label, cells * 4));
    VHP_PL_HIP(hipMalloc(&s.came64, cells * 8));
    VHP_PL_HIP(hipMalloc(&s.ctl, sizeof(PlannerCtl) + 16));   // (+ the 16-byte pivot record: PlannerDev::rec)
    VHP_PL_HIP(hipMalloc(&s.partial, kSpecPartials * sizeof(PlannerKey)));
    VHP_PL_HIP(hipMalloc(&s.ticket, 2 * sizeof(unsigned int)));   // ([1]: the sweep workgroups of a one-kernel iteration that have finished)
    s.cells = cells;
  }
  if (s.pivot_cap < pcap) {
    if (s.pivots) (void)hipFree(s.pivots);
    s.pivots = nullptr;
    VHP_PL_HIP(hipMalloc(&s.pivots, pcap * sizeof(int32_t)));
    s.pivot_cap = pcap;
  }
  // (the second local field only where the latency sweep runs the loop; the speculative solve, which shares this state, has one)
  if (s.lat_sweep && !s.vis_local2) {
    if (s.local_uncached) VHP_PL_HIP(hipExtMallocWithFlags(reinterpret_cast<void**>(&s.vis_local2), cells * 8, hipDeviceMallocUncached));
    else VHP_PL_HIP(hipMalloc(&s.vis_local2, cells * 8));
  }
  // reset(): visibility_global_ = 0, visibility_ = 0, cameFrom_ = 1e15   (solver.cpp:42-47)
  VHP_PL_HIP(hipMemsetAsync(s.vis_global, 0, cells * 8, stream));
  VHP_PL_HIP(hipMemsetAsync(s.vis_local, 0, cells * 8, stream));
  if (s.vis_local2) VHP_PL_HIP(hipMemsetAsync(s.vis_local2, 0, cells * 8, stream));
  VHP_PL_HIP(hipMemsetAsync(s.label, 0xff, cells * 4, stream));
  VHP_PL_HIP(hipMemsetAsync(s.pivots, 0, pcap * sizeof(int32_t), stream));
  VHP_PL_HIP(hipMemsetAsync(s.ticket, 0, 2 * sizeof(unsigned int), stream));

  PlannerDev d;
  d.vis_global = s.vis_global;
  d.vis_local = s.vis_local;
  d.vis_other = nullptr;
  // With the latency sweep the dark part of a field -- most of it, in a maze -- is not written at all: two local fields take
  // turns, and the epilogue that reads one clears what the sweep before last left in the other.
  const bool two_fields = (bool)s.lat_sweep;
  size_t launches = 0;
  d.label = s.label;
  d.pivots = s.pivots;
  d.ctl = s.ctl;
  d.rec = reinterpret_cast<int*>(s.ctl + 1);
  d.local_uncached = s.local_uncached ? 1 : 0;
  d.partial = s.partial;
  d.ticket = s.ticket;
  d.threshold = threshold;
  {
    volatile double q = (double)((size_t)ny * ny + (size_t)nx * nx);
    d.scale = std::sqrt(q);  // scale_, solver.cpp:49
  }
  d.end_x = end_x;
  d.end_y = end_y;
  d.max_iter = max_iter;

  const int R = s.R, W = s.W;
  const bool multi = s.multi;
  VHP_PL_HIP(hipEventRecord(ev0, stream));
  hipLaunchKernelGGL(vhp_planner_init, dim3(1), dim3(64), 0, stream, d, nx, start_x, start_y);
  VHP_PL_HIP(hipGetLastError());
  PlannerCtl ctl{};
#ifndef VHP_PLANNER_BATCH
#define VHP_PLANNER_BATCH 8
#endif
  const int batch = VHP_PLANNER_BATCH;  // iterations enqueued per host poll (those past the end see `done` and return at once)
  {
    const int rc = planner_poll(s, stream, [&]() -> int {
      for (int b = 0; b < batch; ++b, ++launches) {
        if (two_fields) {
          d.vis_local = (launches & 1) ? s.vis_local2 : s.vis_local;
          d.vis_other = (launches & 1) ? s.vis_local : s.vis_local2;
        }
        if (s.lat_iteration) {
          const hipError_t ei = s.lat_iteration(d);
          if (ei != hipSuccess) { *msg = std::string("planner launch: ") + hipGetErrorString(ei); return VHP_ERR_HIP; }
          continue;
        }
        hipError_t e = s.lat_sweep ? s.lat_sweep(d.pivots, &s.ctl->nb, &s.ctl->done, d.rec, d.vis_local, true)
                     : R == 1 ? (multi ? launch_planner_fronts<1, true>(s, m, d, W, stream) : launch_planner_fronts<1, false>(s, m, d, W, stream))
                     : R == 2 ? (multi ? launch_planner_fronts<2, true>(s, m, d, W, stream) : launch_planner_fronts<2, false>(s, m, d, W, stream))
                              : (multi ? launch_planner_fronts<4, true>(s, m, d, W, stream) : launch_planner_fronts<4, false>(s, m, d, W, stream));
        if (e != hipSuccess) { *msg = std::string("planner launch: ") + hipGetErrorString(e); return VHP_ERR_HIP; }
        hipLaunchKernelGGL(vhp_planner_epilogue, dim3(kEpilogueBlocks), dim3(kEpilogueThreads), 0, stream, m, d);
        VHP_PL_HIP(hipGetLastError());
      }
      return VHP_OK;
    }, &ctl, msg);
    if (rc != VHP_OK) return rc;
  }
  VHP_PL_HIP(hipEventRecord(ev1, stream));

  // (launch number n is iteration number n while the loop runs; the last iteration that ran is number iters - 1)
  s.vis_local_out = (two_fields && ctl.iters > 0 && ((ctl.iters - 1) & 1)) ? s.vis_local2 : s.vis_local;
  const uint32_t nb = (uint32_t)ctl.nb;
  if (n_pivots) *n_pivots = nb;
  if (pivots_xy) VHP_PL_HIP(hipMemcpyAsync(pivots_xy, s.pivots, 2 * (size_t)(nb + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  if (came_from) {
    hipLaunchKernelGGL(vhp_labels_to_u64, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, s.label, s.came64, cells);
    VHP_PL_HIP(hipGetLastError());
    VHP_PL_HIP(hipMemcpyAsync(came_from, s.came64, cells * 8, hipMemcpyDeviceToHost, stream));
  }
  if (vis_global) VHP_PL_HIP(hipMemcpyAsync(vis_global, s.vis_global, cells * 8, hipMemcpyDeviceToHost, stream));
  if (vis_local) VHP_PL_HIP(hipMemcpyAsync(vis_local, s.vis_local_out, cells * 8, hipMemcpyDeviceToHost, stream));
  VHP_PL_HIP(hipStreamSynchronize(stream));
  if (ctl.status == VHP_ERR_MAX_ITER) *msg = "Max iters hit. Solution could not be found. Try lowering visibility threshold.";
  if (ctl.status == VHP_ERR_NOTHING_LIT) *msg = "no cell reached the visibility threshold";
  return ctl.status;
}

// ---------------------------------------------------------------------------------------------------------------------
// The speculative planner (second half).  The planner loop is sequential in its pivots -- pivot k+1 is the
// arg-min of a heuristic that needs the union after pivot k -- but a sweep depends on nothing but its pivot, and one
// launch sweeps K sources in the time it sweeps one (a single-source sweep is latency-bound: 8 workgroups on 256 CUs; the launches
// take the latency sweep where a batch of K would: PlannerState::lat_sweep_k).
// So every sweep launch takes the next pivot AND the K-1 best other candidates of the last heuristic evaluation along,
// into a cache of fields keyed by the source cell:
//   mode 0 (exact): an iteration whose pivot is already in the cache skips its sweep.  Pivots, labels, union and the
//     last local field are those of planner_solve bit for bit (the cached field is the same kernel's output from the
//     same cell); only the number of sweep launches changes.  The gain is the hit rate (reported), which is what the
//     map makes it: a pivot that repeats always hits.
//   mode 1 (fast, NOT the reference's result): every candidate that was swept is committed as a pivot in the same
//     iteration, in rank order -- what the reference would do if its heap handed out its K best entries before it
//     re-evaluated the heuristic.  Fewer, fatter iterations; the labels still form a valid parent table (every pivot
//     was lit by an earlier one), the path and the pivot list differ from the reference's.
// Candidates: the committed pivot is the exact arg-min of the epilogue's 1024 wavefront minima; the runner-ups are picked among the
// 64 lanes' best of 16 of those each (a guess is allowed to be a guess), at least kSpecSep cells apart.  Mode 1 keeps its slots
// clean (two groups of K take turns, vhp_spec_epilogue) so that its sweeps store nothing for dead strips.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSpecMaxK = 8;
constexpr int kSpecSlots = 32;  // cached fields (a multiple of every K)
#ifndef VHP_SPEC_SEP
#define VHP_SPEC_SEP 8
#endif
constexpr int kSpecSep = VHP_SPEC_SEP;     // Chebyshev distance between candidates of one launch (maze_6, fast k = 4, iterations at threshold
                                           // 0.1 / 0.25: 4: 43 / max_iter, 8: 38 / 40, 16: 39 / 42, 32: 36 / max_iter, 64: 51 / max_iter)

struct SpecCtl {
  int cand[2 * kSpecMaxK];       // sources of the next sweep launch: [0] the pivot, then runner-ups (x < 0: none)
  int slot_key[2 * kSpecSlots];  // source cell of the field in each cache slot (x < 0: empty)
  int head;                      // the slot group the next sweep launch overwrites
  int cur_slot;                  // where the current pivot's field is
  int sweep;                     // this iteration sweeps (a miss, or mode 1)
  int n_commit;                  // pivots this iteration commits (1 in mode 0)
  int hits, misses, fields_swept;
#ifdef VHP_DIAG_SPEC_TAIL  // (diagnostic builds: where the last epilogue workgroup's time goes, 10 ns units, summed over the iterations)
  unsigned long long tail_t[8];
#endif
  int prev_last;                 // mode 1: the slot of the previous iteration's last committed field (-1: none), see vhp_spec_epilogue
};

struct SpecDev {
  SpecCtl* sc;
  double* cache;  // kSpecSlots fields
  size_t cells;
  int K, mode;
};

// Before the sweep of an iteration: is the pivot's field cached?  Otherwise the launch's slots are named after its sources.  One
// wavefront (all 64 lanes call it; lane 0 writes): vhp_spec_init's for the first iteration, wavefront 0 of the epilogue's last
// workgroup -- the one that has just picked the pivot -- for every other (a kernel of its own until round 5: 4.9 us per iteration).
__device__ inline void spec_lookup(const PlannerDev& d, const SpecDev& sp) {
  const int lane = (int)(threadIdx.x & 63);
  SpecCtl* c = sp.sc;
  // (what lane 0 has just written -- the control block, the pivot -- the other lanes get from it)
  int done = 0, nb = 0, px = 0, py = 0;
  if (lane == 0) {
    done = d.ctl->done;
    nb = d.ctl->nb;
    px = d.pivots[2 * nb];
    py = d.pivots[2 * nb + 1];
  }
  done = __shfl(done, 0);
  nb = __shfl(nb, 0);
  px = __shfl(px, 0);
  py = __shfl(py, 0);
  if (done) return;
  int hit = -1;
  if (sp.mode == 0) {
    // (agent-scope loads: in vhp_spec_init the keys were written by lane 0 a moment ago)
    const bool mine = lane < kSpecSlots && __hip_atomic_load(&c->slot_key[2 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == px &&
                      __hip_atomic_load(&c->slot_key[2 * lane + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == py;
    const unsigned long long found = __ballot(mine);
    if (found) hit = __ffsll((long long)found) - 1;  // (the lowest slot, as the scan found it)
  }
  if (lane != 0) return;
  c->n_commit = 1;
  if (hit >= 0) {
    c->cur_slot = hit;
    c->sweep = 0;
    c->hits += 1;
    return;
  }
  const int base = c->head;
  c->head = (base + sp.K) % (sp.mode == 1 ? 2 * sp.K : kSpecSlots);  // (mode 1: two groups take turns, vhp_spec_epilogue)
  c->cand[0] = px;  // (it is: the pick wrote both)
  c->cand[1] = py;
  int n = 0, committed = 1;
  for (int j = 0; j < sp.K; ++j) {
    const int x = c->cand[2 * j], y = c->cand[2 * j + 1];
    c->slot_key[2 * (base + j)] = x;
    c->slot_key[2 * (base + j) + 1] = y;
    if (x >= 0) ++n;
    // (the candidates are packed -- the pick fills them front to back -- so the committed ones are cand[0 .. committed))
    if (sp.mode == 1 && j > 0 && x >= 0 && committed == j && (unsigned long long)(nb + j) <= d.max_iter + 1) {
      // committed in rank order: lightSources_[nb + j]
      d.pivots[2 * (nb + j)] = x;
      d.pivots[2 * (nb + j) + 1] = y;
      committed = j + 1;
    }
  }
  if (sp.mode == 1) c->n_commit = committed;  // only what has a pivot slot: a candidate past max_iter + 1 is swept but not committed
  c->cur_slot = base;
  c->sweep = 1;
  c->misses += 1;
  c->fields_swept += n;
}

// (behind vhp_planner_init on the stream: the control block is set)
__global__ void __launch_bounds__(64) vhp_spec_init(PlannerDev d, SpecDev sp, int start_x, int start_y) {
  if (blockIdx.x != 0) return;
  SpecCtl* c = sp.sc;
  if (threadIdx.x == 0) {
    for (int k = 0; k < 2 * kSpecMaxK; ++k) c->cand[k] = -1;
    for (int k = 0; k < 2 * kSpecSlots; ++k) c->slot_key[k] = -1;
    c->cand[0] = start_x;
    c->cand[1] = start_y;
    c->head = c->cur_slot = c->sweep = 0;
    c->n_commit = 1;
    c->hits = c->misses = c->fields_swept = 0;
    c->prev_last = -1;
#ifdef VHP_DIAG_SPEC_TAIL
    for (int k = 0; k < 8; ++k) c->tail_t[k] = 0;
#endif
    __threadfence();  // (the other lanes read the slot keys)
  }
  spec_lookup(d, sp);
}

// The sweep launch of an iteration where the latency sweep does not take it (PlannerState::lat_sweep_k): workgroup b sweeps quadrant
// b & 3 of candidate b >> 2 into its cache slot.
template <int R, bool MULTI>
__global__ void __launch_bounds__((R == 2 && MULTI) ? 512 : 1024, R >= 4 ? 4 : ((MULTI && R == 2) ? 6 : 8)) vhp_spec_sweep(DevMap m, PlannerDev d, SpecDev sp) {
  extern __shared__ double lds[];
  if (d.ctl->done || !sp.sc->sweep) return;
  const int j = blockIdx.x >> 2;
  const int sx = sp.sc->cand[2 * j], sy = sp.sc->cand[2 * j + 1];
  if (sx < 0) return;
  StoreEmit<double, MULTI> emit(sp.cache + (size_t)(sp.sc->cur_slot + j) * sp.cells, m.nx, m.ny);
  sweep_quadrant<R>(m, emit, sx, sy, blockIdx.x & 3, lds, whole_workgroup());
}

// updateVisibility()'s per-cell body over the committed field(s), the arg-min, and -- in the last workgroup -- the next pivot, the
// runner-ups and the look-up of the next iteration.  With one committed field (mode 0) the cells see vhp_planner_epilogue's operations
// one for one, in its launch shape; every wavefront leaves a partial minimum (kSpecPartials of them: the runner-ups are picked
// among those).
// NF = the committed fields a launch can have (mode 0: 1; mode 1: k).  CLEAN (mode 1): a field is read exactly once, by the epilogue
// of the iteration that swept it, so that epilogue puts +0.0 back wherever it read something else -- except into the iteration's
// last field, which is the solve's local field if the loop ends here: the next epilogue clears that one (one more load per cell).
// Two groups of k slots take turns, both zero when the solve starts: every sweep finds +0.0 wherever it will not write, and its dead
// strips store nothing (LatArgs::dead_cells_are_zero) -- k fields of zeros per launch were 10 us of the sweep and as much of the
// epilogue behind it, whose loads queued behind their write-back.
template <int NF, bool CLEAN>
__global__ void __launch_bounds__(kEpilogueThreads) vhp_spec_epilogue(DevMap m, PlannerDev d, SpecDev sp) {
#ifdef VHP_DIAG_SPEC_TAIL
  unsigned long long tt[8];
  tt[0] = wall_clock64();
#define VHP_SPEC_STAMP(i) tt[i] = wall_clock64()
#else
#define VHP_SPEC_STAMP(i)
#endif
  // (the control words together, and nobody waits for `done` before the others are on their way: a trip to memory less)
  const int done_now = d.ctl->done, nb = d.ctl->nb, nc = sp.sc->n_commit, cur = sp.sc->cur_slot;
  const int prev_last = CLEAN ? sp.sc->prev_last : -1;
  if (done_now) return;
  const int nx = m.nx, ny = m.ny;
  double* field0 = sp.cache + (size_t)cur * sp.cells;
  double* prev_field = sp.cache + (size_t)(prev_last < 0 ? 0 : prev_last) * sp.cells;
  PlannerKey best;
  best.h = ~0ull;
  best.rank = ~0ull;
  best.x = best.y = -1;
  const size_t cells = (size_t)nx * ny;
  constexpr int kPivLds = 1024;
  __shared__ int piv_lds[2 * kPivLds];
  const int n_piv = nb + nc < kPivLds ? nb + nc : kPivLds;   // (a label can name the pivots this iteration commits)
  for (int t = threadIdx.x; t < 2 * n_piv; t += blockDim.x) piv_lds[t] = d.pivots[t];
  auto pivot_x = [&](uint32_t i) { return i < (uint32_t)kPivLds ? piv_lds[2 * i] : d.pivots[2 * i]; };
  auto pivot_y = [&](uint32_t i) { return i < (uint32_t)kPivLds ? piv_lds[2 * i + 1] : d.pivots[2 * i + 1]; };
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  constexpr int CB = NF > 4 ? 2 : 4;   // cells of a thread per batch of loads (8 fields x 4 cells spilled: 348 bytes of scratch per lane)
  double vv[NF][CB], oo[CB];   // (every committed field's cell with the batch: the kernel is a chain of latencies)
  uint32_t ll[CB];
  auto load_batch = [&](size_t k0) {
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const size_t k = k0 + u * stride;
      const bool in = k < cells;
#pragma unroll
      for (int j = 0; j < NF; ++j) vv[j][u] = (in && (NF == 1 || j < nc)) ? field0[(size_t)j * sp.cells + k] : 0.0;
      if (CLEAN && in && prev_last >= 0 && prev_field[k] != 0.0) prev_field[k] = 0.0;
      oo[u] = in ? d.vis_global[k] : 0.0;
      ll[u] = in ? d.label[k] : 0u;
    }
  };
  size_t k0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  load_batch(k0);
  __syncthreads();
  const int sx0 = pivot_x((uint32_t)nb), sy0 = pivot_y((uint32_t)nb);
  for (;;) {
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const size_t k = k0 + u * stride;
      if (k >= cells) continue;
      const int y = (int)((unsigned)k / (unsigned)nx), x = (int)((unsigned)k - (unsigned)y * (unsigned)nx);
      double g = oo[u];
      uint32_t lab = ll[u];
      bool visited = false;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        if (NF > 1 && j >= nc) continue;
        if (CLEAN && j < nc - 1 && vv[j][u] != 0.0) field0[(size_t)j * sp.cells + k] = 0.0;
        const int sx = j == 0 ? sx0 : pivot_x((uint32_t)(nb + j)), sy = j == 0 ? sy0 : pivot_y((uint32_t)(nb + j));
        if ((x == 0 && sx > 0) || (y == 0 && sy > 0)) continue;  // not swept from this pivot
        visited = true;
        const double v = vv[j][u];
        g = fmax(v, g);  // :417-418
        if (v >= d.threshold && lab == kUnlabelled32) lab = (uint32_t)(nb + j);  // :419-423
      }
      if (!visited) continue;
      if (g != oo[u]) d.vis_global[k] = g;
      if (lab != ll[u]) d.label[k] = lab;
      if (g >= d.threshold) {  // :424-430
        const double h = (d.scale * g) + (eval_d_dev(x, y, d.end_x, d.end_y) + eval_d_dev(x, y, pivot_x(lab), pivot_y(lab)));
        PlannerKey c;
        c.h = (unsigned long long)__double_as_longlong(h);
        c.rank = push_rank(nx, ny, sx0, sy0, x, y);
        c.x = x;
        c.y = y;
        if (key_less(c, best)) best = c;
      }
    }
    k0 += CB * stride;
    if (k0 >= cells) break;
    load_batch(k0);
  }
  const int wave = threadIdx.x >> 6;
  if (wave_holds_min(best)) d.partial[blockIdx.x * kEpilogueWaves + wave] = best;
  // (the hand-off of vhp_planner_epilogue: every wavefront drains its stores -- its partial among them --, the barrier, one lane:
  // agent-scope release, ticket)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = __hip_atomic_fetch_add(d.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!is_last) return;
  VHP_SPEC_STAMP(1);
  // Wavefront 0 picks alone -- no barrier and no trip through one thread per round (round 5: 2.4 us per round that way).  What the
  // pick and the look-up need from memory is asked for now, together -- the partials (agent scope: other CUs wrote them), the loop
  // condition's cell (written by some workgroup of this kernel), the control words, the slot keys (lane s: slot s) --, so that the
  // serial part below runs on registers: a trip to memory per word, one after the other, was 10 us of this kernel at k = 4 and 20 at
  // k = 8.
  if (wave != 0) return;
  VHP_SPEC_STAMP(2);
  SpecCtl* c = sp.sc;
  const int lane = (int)threadIdx.x;
  const double ge = __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(d.vis_global + (size_t)d.end_y * m.nx + d.end_x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const int iters = d.ctl->iters;
  const int head = c->head;
  const int key_x = lane < kSpecSlots ? c->slot_key[2 * lane] : -1, key_y = lane < kSpecSlots ? c->slot_key[2 * lane + 1] : -1;
  constexpr int kPerLane = kSpecPartials / 64;   // (16 candidates per lane: every wavefront's partial minimum is one)
  static_assert(kSpecPartials % 64 == 0 && kPerLane <= 32, "one bit per candidate");
  PlannerKey cand[kPerLane];
#pragma unroll
  for (int t = 0; t < kPerLane; ++t) {
    const int i = lane + 64 * t;
    const bool in = i < (int)gridDim.x * kEpilogueWaves;
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(d.partial + (in ? i : 0));
    cand[t].h = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cand[t].rank = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long xy = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cand[t].x = in ? (int)(unsigned)xy : -1;
    cand[t].y = in ? (int)(unsigned)(xy >> 32) : -1;
    if (!in) { cand[t].h = ~0ull; cand[t].rank = ~0ull; }
  }
  // round 0: the arg-min (the next pivot, exactly: every partial is somebody's); rounds 1 .. K-1: the best of the rest at least
  // kSpecSep away from what has been chosen (guesses).  A candidate too close to
  // a choice is struck off when the choice is made; the choices go to LDS (nothing here is an array the compiler would index at
  // run time: such arrays lived in scratch memory, 2.6 us per round at k = 8).
  __shared__ int next_src[2 * kSpecMaxK];   // the sources of the next sweep launch, packed: [0] the pivot, then the runner-ups
  VHP_SPEC_STAMP(3);
  // (the rounds run on ONE candidate per lane, the best of its 16 partials: 800 instructions per round on all of them, by a lone
  // wavefront, were 1.4 us per round)
  PlannerKey mine = cand[0];
#pragma unroll
  for (int t = 1; t < kPerLane; ++t)
    if (key_less(cand[t], mine)) mine = cand[t];
  int first_x = -1, first_y = -1, n_c = 1;
  for (int r = 0; r < sp.K; ++r) {
    // the wavefront's minimum of h, of the rank among the lanes that hold it (round 0: the arg-min must be the reference's; a guess
    // takes the lowest lane), and the cell from the lane that has both
    const unsigned long long hmin = wave_min_u64(mine.h);
    unsigned long long rk = mine.h == hmin ? mine.rank : ~0ull;
    if (r == 0) { const unsigned long long rmin = wave_min_u64(rk); rk = rk == rmin ? 0ull : ~0ull; }
    const unsigned long long holders = __ballot(mine.h == hmin && rk != ~0ull);
    const int wl = holders ? __ffsll((long long)holders) - 1 : 0;
    const int wx = __builtin_amdgcn_readlane(mine.x, wl), wy = __builtin_amdgcn_readlane(mine.y, wl);  // (-1 from every lane when nothing is left)
    if (wx < 0) break;
    if (r == 0) { first_x = wx; first_y = wy; }
    else {
      if (lane == 0) { next_src[2 * n_c] = wx; next_src[2 * n_c + 1] = wy; }
      ++n_c;
    }
    if (abs(mine.x - wx) < kSpecSep && abs(mine.y - wy) < kSpecSep) { mine.h = ~0ull; mine.rank = ~0ull; mine.x = mine.y = -1; }
  }
  VHP_SPEC_STAMP(4);
  // the pick of vhp_planner_epilogue, on registers (every lane computes, lane 0 stores): ls_ = top(); ++nb_of_sources_; the loop condition
  // (solver.cpp:127-141).  The runner-ups this iteration committed (mode 1) are pivots nb+1 .. nb+nc-1.
  const bool st = lane == 0;
  if (st) { *d.ticket = 0; d.ctl->iters = iters + 1; if (CLEAN) c->prev_last = cur + nc - 1; }
  int nbn = nb + nc - 1;
  bool done = false;
  int px = first_x, py = first_y;
  if (px < 0) {  // nothing reached the threshold: the reference would call top() on an empty heap
    if (st) { d.ctl->nb = nbn; d.ctl->status = VHP_ERR_NOTHING_LIT; d.ctl->done = 1; }
    done = true;
    // (next_src[0] below: the pivot as it stands)
    px = pivot_x((uint32_t)nbn);
    py = pivot_y((uint32_t)nbn);
  } else {
    nbn += 1;
    int status = -1;
    if ((unsigned long long)nbn > d.max_iter) status = VHP_ERR_MAX_ITER;  // :134-139
    else if (ge > d.threshold) { px = d.end_x; py = d.end_y; status = VHP_OK; }  // :127, :141
    done = status >= 0;
    if (st) {
      d.ctl->nb = nbn;
      d.pivots[2 * nbn] = px;
      d.pivots[2 * nbn + 1] = py;
      if (done) { d.ctl->status = status; d.ctl->done = 1; }
    }
  }
  // the sources of the next sweep launch: the pivot (the end point if the loop is over: nothing sweeps it), the runner-ups, then none
  if (lane < 2 * kSpecMaxK && (lane < 2 || lane >= 2 * n_c)) next_src[lane] = lane == 0 ? px : lane == 1 ? py : -1;
  __builtin_amdgcn_wave_barrier();
  const int my_src = lane < 2 * kSpecMaxK ? next_src[lane] : -1;   // (lane l: word l)
  if (lane < 2 * kSpecMaxK) c->cand[lane] = my_src;
  VHP_SPEC_STAMP(5);
#ifdef VHP_DIAG_SPEC_TAIL
  if (st) for (int i = 1; i <= 5; ++i) c->tail_t[i] += tt[i] - tt[i - 1];
#endif
  if (done) return;
  // the look-up of the next iteration (spec_lookup, on registers)
  int hit = -1;
  if (sp.mode == 0) {
    const unsigned long long found = __ballot(key_x == px && key_y == py && lane < kSpecSlots);
    if (found) hit = __ffsll((long long)found) - 1;
  }
  if (hit >= 0) {
    if (st) {
      c->n_commit = 1;
      c->cur_slot = hit;
      c->sweep = 0;
      atomicAdd(&c->hits, 1);
    }
    return;
  }
  // a miss: the launch's slots are named after its sources (lane l: word l of the k pairs); mode 1 commits the runner-ups in rank
  // order as lightSources_[nb + j] -- they are packed, so those are sources 1 .. committed-1; a candidate past max_iter + 1 is
  // swept but not committed
  const int n_src = n_c < sp.K ? n_c : sp.K;
  long long room = (long long)d.max_iter + 2 - nbn;   // j <= max_iter + 1 - nb
  const int committed = sp.mode == 1 ? (int)(room < 1 ? 1 : (room < n_src ? room : n_src)) : 1;
  if (lane < 2 * sp.K) c->slot_key[2 * head + lane] = my_src;
  if (sp.mode == 1 && lane >= 2 && lane < 2 * committed) d.pivots[2 * nbn + lane] = my_src;
  if (st) {
    c->head = (head + sp.K) % (CLEAN ? 2 * sp.K : kSpecSlots);
    c->n_commit = committed;
    c->cur_slot = head;
    c->sweep = 1;
    atomicAdd(&c->misses, 1);
    atomicAdd(&c->fields_swept, n_src);
  }
}
#undef VHP_SPEC_STAMP

// the last committed pivot's field becomes vis_local; cells its sweep does not visit read as zero (visibility_.reset(), :386)
__global__ void vhp_spec_export_local(DevMap m, PlannerDev d, SpecDev sp) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= sp.cells) return;
  const int last = d.ctl->iters > 0 ? sp.sc->n_commit - 1 : 0;
  const double* f = sp.cache + (size_t)(sp.sc->cur_slot + last) * sp.cells;
  const int sx = sp.sc->slot_key[2 * (sp.sc->cur_slot + last)], sy = sp.sc->slot_key[2 * (sp.sc->cur_slot + last) + 1];
  const int y = (int)(k / m.nx), x = (int)(k - (size_t)y * m.nx);
  const bool unvisited = (x == 0 && sx > 0) || (y == 0 && sy > 0);
  d.vis_local[k] = (d.ctl->iters == 0 || unvisited) ? 0.0 : f[k];
}

struct SpecState {
  size_t cells = 0;
  double* cache = nullptr;
  SpecCtl* sc = nullptr;
};
inline void spec_free(SpecState& s) {
  if (s.cache) (void)hipFree(s.cache);
  if (s.sc) (void)hipFree(s.sc);
  s.cache = nullptr;
  s.sc = nullptr;
  s.cells = 0;
}

template <int R, bool MULTI>
inline hipError_t launch_spec_fronts(PlannerState& s, const DevMap& m, const PlannerDev& d, const SpecDev& sp, int W, hipStream_t stream) {
  const size_t lds = sweep_lds_bytes(R, W, MULTI);
  auto k = vhp_spec_sweep<R, MULTI>;
  {
    hipError_t e = s.raise_lds ? s.raise_lds(reinterpret_cast<const void*>(k), lds)
                               : hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(4 * sp.K), dim3(128 * W), lds, stream, m, d, sp);
  return hipGetLastError();
}

// The loop of planner_solve with the speculative sweep launches.  stats (3 ints, may be null): iterations whose pivot was
// cached, iterations that swept, fields swept.  The caller has validated start / end and sized the round scratch for 4 K units.
inline int planner_solve_speculative(PlannerState& s, SpecState& ss, const DevMap& m, const uint8_t* d_occ, hipStream_t stream, hipEvent_t ev0,
                                     hipEvent_t ev1, int start_x, int start_y, int end_x, int end_y, double threshold, uint64_t max_iter, int K,
                                     int mode, uint64_t* came_from, double* vis_global, double* vis_local, int32_t* pivots_xy,
                                     uint32_t* n_pivots, int* stats, std::string* msg) {
  const int nx = m.nx, ny = m.ny;
  // the four validity checks of solve(), in the reference's order (solver.cpp:89-116)
  auto valid = [&](int x, int y) { return (size_t)x < (size_t)nx && (size_t)y < (size_t)ny; };
  if (!valid(start_x, start_y)) { *msg = "Start point is out of bounds."; return VHP_ERR_START_OOB; }
  if (!valid(end_x, end_y)) { *msg = "End point is out of bounds."; return VHP_ERR_END_OOB; }
  uint8_t occ_s = 0, occ_e = 0;
  if (s.h_occ) {  // (the map came from the host: its copy answers, no trip to the device)
    occ_s = s.h_occ[(size_t)start_y * nx + start_x];
    occ_e = s.h_occ[(size_t)end_y * nx + end_x];
  } else {
    VHP_PL_HIP(hipMemcpyAsync(&occ_s, d_occ + (size_t)start_y * nx + start_x, 1, hipMemcpyDeviceToHost, stream));
    VHP_PL_HIP(hipMemcpyAsync(&occ_e, d_occ + (size_t)end_y * nx + end_x, 1, hipMemcpyDeviceToHost, stream));
    VHP_PL_HIP(hipStreamSynchronize(stream));
  }
  if (!occ_s) { *msg = "Start point is not valid (occupied)"; return VHP_ERR_START_OCCUPIED; }
  if (!occ_e) { *msg = "End point is not valid (occupied)"; return VHP_ERR_END_OCCUPIED; }
  if (max_iter > (1u << 24)) { *msg = "max_iter too large"; return VHP_ERR_ARG; }
  if ((K != 1 && K != 2 && K != 4 && K != 8) || (mode != 0 && mode != 1)) { *msg = "speculative planner: k must be 1, 2, 4 or 8 and mode 0 or 1"; return VHP_ERR_ARG; }
  const size_t cells = (size_t)nx * ny;
  const size_t pcap = 2 * (size_t)(max_iter + 2 + kSpecMaxK);
  if (s.cells != cells) {
    planner_free(s);
    VHP_PL_HIP(hipMalloc(&s.vis_global, cells * 8));
    VHP_PL_HIP(hipMalloc(&s.vis_local, cells * 8));
    VHP_PL_HIP(hipMalloc(&s.label, cells * 4));
    VHP_PL_HIP(hipMalloc(&s.came64, cells * 8));
    VHP_PL_HIP(hipMalloc(&s.ctl, sizeof(PlannerCtl) + 16));   // (+ the 16-byte pivot record: PlannerDev::rec)
    VHP_PL_HIP(hipMalloc(&s.partial, kSpecPartials * sizeof(PlannerKey)));
    VHP_PL_HIP(hipMalloc(&s.ticket, 2 * sizeof(unsigned int)));   // ([1]: the sweep workgroups of a one-kernel iteration that have finished)
    s.cells = cells;
  }
  if (s.pivot_cap < pcap) {
    if (s.pivots) (void)hipFree(s.pivots);
    s.pivots = nullptr;
    VHP_PL_HIP(hipMalloc(&s.pivots, pcap * sizeof(int32_t)));
    s.pivot_cap = pcap;
  }
  if (ss.cells != cells) {
    spec_free(ss);
    VHP_PL_HIP(hipMalloc(&ss.cache, (size_t)kSpecSlots * cells * 8));
    VHP_PL_HIP(hipMalloc(&ss.sc, sizeof(SpecCtl)));
    ss.cells = cells;
  }
  VHP_PL_HIP(hipMemsetAsync(s.vis_global, 0, cells * 8, stream));
  VHP_PL_HIP(hipMemsetAsync(s.label, 0xff, cells * 4, stream));
  VHP_PL_HIP(hipMemsetAsync(s.pivots, 0, s.pivot_cap * sizeof(int32_t), stream));
  VHP_PL_HIP(hipMemsetAsync(s.ticket, 0, 2 * sizeof(unsigned int), stream));
  if (mode == 1) VHP_PL_HIP(hipMemsetAsync(ss.cache, 0, (size_t)2 * K * cells * 8, stream));  // (the two groups of k slots that take turns: vhp_spec_epilogue)

  PlannerDev d;
  d.vis_global = s.vis_global;
  d.vis_local = s.vis_local;
  d.vis_other = nullptr;
  s.vis_local_out = s.vis_local;
  d.label = s.label;
  d.pivots = s.pivots;
  d.ctl = s.ctl;
  d.rec = reinterpret_cast<int*>(s.ctl + 1);
  d.local_uncached = s.local_uncached ? 1 : 0;
  d.partial = s.partial;
  d.ticket = s.ticket;
  d.threshold = threshold;
  {
    volatile double q = (double)((size_t)ny * ny + (size_t)nx * nx);
    d.scale = std::sqrt(q);  // scale_, solver.cpp:49
  }
  d.end_x = end_x;
  d.end_y = end_y;
  d.max_iter = max_iter;
  SpecDev sp;
  sp.sc = ss.sc;
  sp.cache = ss.cache;
  sp.cells = cells;
  sp.K = K;
  sp.mode = mode;

  const int R = s.R, W = s.W;
  const bool multi = s.multi;
  VHP_PL_HIP(hipEventRecord(ev0, stream));
  hipLaunchKernelGGL(vhp_planner_init, dim3(1), dim3(64), 0, stream, d, nx, start_x, start_y);
  hipLaunchKernelGGL(vhp_spec_init, dim3(1), dim3(64), 0, stream, d, sp, start_x, start_y);
  VHP_PL_HIP(hipGetLastError());
  PlannerCtl ctl{};
  const int batch = VHP_PLANNER_BATCH;
  {
    const int rc = planner_poll(s, stream, [&]() -> int {
      for (int b = 0; b < batch; ++b) {
        hipError_t e = s.lat_sweep_k ? s.lat_sweep_k(ss.sc->cand, K, &ss.sc->cur_slot, &ss.sc->sweep, &s.ctl->done, ss.cache, mode == 1)
                     : R == 1 ? (multi ? launch_spec_fronts<1, true>(s, m, d, sp, W, stream) : launch_spec_fronts<1, false>(s, m, d, sp, W, stream))
                     : R == 2 ? (multi ? launch_spec_fronts<2, true>(s, m, d, sp, W, stream) : launch_spec_fronts<2, false>(s, m, d, sp, W, stream))
                              : (multi ? launch_spec_fronts<4, true>(s, m, d, sp, W, stream) : launch_spec_fronts<4, false>(s, m, d, sp, W, stream));
        if (e != hipSuccess) { *msg = std::string("speculative planner launch: ") + hipGetErrorString(e); return VHP_ERR_HIP; }
        auto epi = mode == 0 ? vhp_spec_epilogue<1, false> : K == 1 ? vhp_spec_epilogue<1, true> : K == 2 ? vhp_spec_epilogue<2, true>
                           : K == 4 ? vhp_spec_epilogue<4, true> : vhp_spec_epilogue<8, true>;
        hipLaunchKernelGGL(epi, dim3(kEpilogueBlocks), dim3(kEpilogueThreads), 0, stream, m, d, sp);
        VHP_PL_HIP(hipGetLastError());
      }
      return VHP_OK;
    }, &ctl, msg);
    if (rc != VHP_OK) return rc;
  }
  hipLaunchKernelGGL(vhp_spec_export_local, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, m, d, sp);
  VHP_PL_HIP(hipGetLastError());
  VHP_PL_HIP(hipEventRecord(ev1, stream));
  SpecCtl hc{};
  VHP_PL_HIP(hipMemcpyAsync(&hc, ss.sc, sizeof(hc), hipMemcpyDeviceToHost, stream));

  const uint32_t nb = (uint32_t)ctl.nb;
  if (n_pivots) *n_pivots = nb;
  if (pivots_xy) VHP_PL_HIP(hipMemcpyAsync(pivots_xy, s.pivots, 2 * (size_t)(nb + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  if (came_from) {
    hipLaunchKernelGGL(vhp_labels_to_u64, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, s.label, s.came64, cells);
    VHP_PL_HIP(hipGetLastError());
    VHP_PL_HIP(hipMemcpyAsync(came_from, s.came64, cells * 8, hipMemcpyDeviceToHost, stream));
  }
  if (vis_global) VHP_PL_HIP(hipMemcpyAsync(vis_global, s.vis_global, cells * 8, hipMemcpyDeviceToHost, stream));
  if (vis_local) VHP_PL_HIP(hipMemcpyAsync(vis_local, s.vis_local, cells * 8, hipMemcpyDeviceToHost, stream));
  VHP_PL_HIP(hipStreamSynchronize(stream));
  if (stats) { stats[0] = hc.hits; stats[1] = hc.misses; stats[2] = hc.fields_swept; }
#ifdef VHP_DIAG_SPEC_TAIL
  fprintf(stderr, "spec tail (us, summed over the solve): start->last %.1f  gather %.1f  loads %.1f  rounds %.1f  pick %.1f\n", hc.tail_t[1] / 100.0,
          hc.tail_t[2] / 100.0, hc.tail_t[3] / 100.0, hc.tail_t[4] / 100.0, hc.tail_t[5] / 100.0);
#endif
  if (ctl.status == VHP_ERR_MAX_ITER) *msg = "Max iters hit. Solution could not be found. Try lowering visibility threshold.";
  if (ctl.status == VHP_ERR_NOTHING_LIT) *msg = "no cell reached the visibility threshold";
  return ctl.status;
}

}  // namespace vhp
