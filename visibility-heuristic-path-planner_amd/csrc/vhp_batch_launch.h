// vhp_batch_launch.h -- host-side interface of the persistent batch kernels: the pool sweep (vhp_pool.hip) and the latency
// sweep (vhp_lat.hip), used by vhp_capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

namespace vhp {

struct PlannerDev;  // (vhp_planner_dev.hip.h)

struct BatchArgs {
  const uint64_t* rows;   // packed maps and reciprocal table of the context (vhp_set_map)
  const uint64_t* cols;
  const double* recip;
  const uint64_t* dmap = nullptr;  // the occupancy packed along diagonals (lat_pack_diag_maps): what the latency sweep reads
  int wpr, wpc, nx, ny;
  const int32_t* d_src;   // n_src (x, y) pairs, device
  int n_src;
  void* d_out;            // n_src fields
  int dtype;              // VHP_F64 / VHP_F32
  long long field_stride; // elements
  int* d_err;             // device flag: a source outside the grid
  int* d_queue;           // the launch's scratch (pool_scratch_bytes / lat_scratch_bytes)
  int n_cus;              // compute units of the device (the persistent grid is sized to what the chip holds at once)
  int lat_workgroups = 0; // latency sweep: workgroups per octant asked for (0: by the grid's size, lat_halves)
  int* d_lat_order = nullptr;  // latency sweep: room for the launch order of its units (lat_order_bytes; vhp_lat_order), or null: none
  hipStream_t stream;
  // called with (kernel, bytes) before a launch that needs more than the default dynamic LDS: the per-device
  // bookkeeping lives with the context
  std::function<hipError_t(const void*, size_t)> raise_lds;
  // optional per-launch timing events, recorded around the sweep kernel only
  hipEvent_t ev_begin, ev_end;
  int pool_contexts = 0;  // pool sweep: units a workgroup holds at once (0: automatic; vhp_set_option "pool_contexts")
  int pool_claim_ahead = -1;  // pool sweep: steps by which a strip is claimed ahead of the strip below's progress (-1: automatic)
  int pool_heads = 0;     // pool sweep: contexts that pull from the head of the size-sorted queue (0: one)
  int pool_tail_pct = 0;  // pool sweep: share of the units (by count, smallest first) that the filler contexts may take from the small end (0: 50)
  int pool_early_ctx = 0, pool_late_pct = 0;  // pool sweep: contexts >= early_ctx open once late_pct % of the units are taken (0: all open)
  int pool_busy_cap = 0;  // pool sweep: a workgroup takes another unit only while fewer wavefronts than this are sweeping (0: no cap)
  int pool_static_round = 2;  // pool sweep: every context's first unit by workgroup index, no pull (vhp_pool.hpp Args::static_round; 2: odd head contexts count down, Args::static_snake); 0: every unit pulled
  const int* d_src_index = nullptr;  // latency sweep in the planner's loop: sweep source number *d_src_index of d_src (n_src = 1) ...
  const int* d_skip = nullptr;       // ... and nothing at all if *d_skip is set
  const int* d_pivot_rec = nullptr;  // ... or, instead of both: the 16-byte record {done, nb, x, y} of the planner's loop (LatArgs::pivot_rec)
  const int* d_slot_base = nullptr;  // the speculative planner's launches (LatArgs::slot_base, run_if)
  const int* d_run_if = nullptr;
  bool lat_dead_cells_are_zero = false;  // ... and dead strips store nothing: the field holds +0.0 wherever the launch does not write
  unsigned long long pool_epoch = 0;  // pool sweep: the tag of this launch's boundary-line entries: never 0, never reused on this scratch
};

// The pool sweep (vhp_pool.hip): d_queue is scratch of pool_scratch_bytes (pull counter, unit order, the
// diagonal lines of the y-major units, the boundary lines of the strips) that is ZERO when it is first used and is
// written by nothing else; pool_epoch differs from launch to launch.
bool pool_supported(int nx, int ny);
hipError_t launch_pool(const BatchArgs& a);
size_t pool_scratch_bytes(int n_src, int nx, int ny);

// The latency sweep (vhp_lat.hip): one workgroup per octant, for launches of a few sources.  d_queue is scratch of
// lat_scratch_bytes with the pool sweep's rules (zero when first used, written by nothing but these two kernels; the two
// share the epoch counter, so either may follow the other on one allocation).
bool lat_supported(int nx, int ny);
hipError_t launch_lat(const BatchArgs& a);
// A planner iteration as one launch: the latency sweep of the pivot named by a.d_pivot_rec (fp64, into a.d_out) and, in the same
// grid, the epilogue over d (union, labels, heuristic, next pivot: vhp_planner_dev.hip.h).  d.ticket[1] counts the sweep's workgroups.
hipError_t launch_lat_planner(const BatchArgs& a, const PlannerDev& d);
size_t lat_scratch_bytes(int n_src, int nx, int ny);
size_t lat_order_bytes();
// The latency sweep's lanes run along diagonals of the grid: it reads the occupancy packed along them (vhp_band.hpp DiagMaps),
// built once per map from the byte map: lat_diag_map_bytes of device memory, zero-filled and packed by lat_pack_diag_maps.
size_t lat_diag_map_bytes(int nx, int ny);
hipError_t lat_pack_diag_maps(const uint8_t* d_occ, int nx, int ny, uint64_t* d_dmap, hipStream_t stream);

}  // namespace vhp
