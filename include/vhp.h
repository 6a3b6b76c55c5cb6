/* vhp.h -- C ABI of the MI355X visibility-sweep / visibility-heuristic-planner library
 * (libvhp_hip.so).  This is the drop-in boundary: plain pointers and sizes, no C++
 * or torch types.
 *
 * The reference (IbrahimSquared/visibility-heuristic-path-planner) has no FFI or
 * plugin interface; its de-facto boundary is the public surface of
 * vbs::visibilityBasedSolver (include/solver/visibilityBasedSolver.h:23-71), whose
 * methods return void and leave results in members.  Each entry point below names
 * the reference code it replaces.
 *
 * Conventions
 *   - grids are row-major, x fastest: index = x + y*nx (include/environment/field.h:24-29)
 *   - occupancy is the reference's "occupancy complement": 1 = free, 0 = blocked
 *     (environment.cpp:200-207), one uint8 per cell
 *   - sources / pivots are int32 (x, y) pairs (parser.h:9 `point`)
 *   - every call returns a vhp_status; vhp_last_error() gives the message
 *   - a context is bound to one GPU and is not thread-safe (the reference object is
 *     stateful and single-threaded as well, visibilityBasedSolver.h:148-171)
 *   - there is NO CPU fallback: without a usable HIP device vhp_create fails.
 */
#ifndef VHP_H
#define VHP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vhp_ctx vhp_ctx;

typedef enum vhp_status {
  VHP_OK = 0,
  VHP_ERR_ARG = 1,            /* null pointer / bad size / bad enum */
  VHP_ERR_SOURCE_OOB = 2,     /* a sweep source lies outside the grid */
  VHP_ERR_NOTHING_LIT = 3,    /* planner: no cell reached the threshold (reference: top() of an empty heap) */
  VHP_ERR_START_OOB = 10,     /* "Start point is out of bounds."        solver.cpp:89-95  */
  VHP_ERR_END_OOB = 11,       /* "End point is out of bounds."          solver.cpp:96-102 */
  VHP_ERR_START_OCCUPIED = 12,/* "Start point is not valid (occupied)"  solver.cpp:103-109 */
  VHP_ERR_END_OCCUPIED = 13,  /* "End point is not valid (occupied)"    solver.cpp:110-116 */
  VHP_ERR_MAX_ITER = 20,      /* "Max iters hit. ..."                   solver.cpp:134-139 */
  VHP_ERR_HIP = 100,          /* a HIP runtime call failed */
  VHP_ERR_NO_MAP = 101,       /* vhp_set_map has not been called */
  VHP_ERR_TOO_LARGE = 102     /* grid side exceeds VHP_MAX_SIDE */
} vhp_status;

/* which sweep: computeVisibility (solver.cpp:570-696) or
 * computeVisibilityUsingQueue (solver.cpp:701-893) */
typedef enum vhp_variant { VHP_SWEEP_FULL = 0, VHP_SWEEP_QUEUE = 1 } vhp_variant;

/* element type of the STORED field; arithmetic is always IEEE binary64 */
typedef enum vhp_dtype { VHP_F64 = 0, VHP_F32 = 1 } vhp_dtype;

#define VHP_MAX_SIDE 8192
#define VHP_UNLABELLED 1000000000000000ULL /* (size_t)1e15, solver.cpp:46 */

/* Replaces the visibilityBasedSolver constructor + reset() (solver.cpp:13-60).
 * device_ordinal: HIP device index. */
int vhp_create(int device_ordinal, vhp_ctx** out);
int vhp_destroy(vhp_ctx* ctx);
const char* vhp_last_error(const vhp_ctx* ctx);

/* Launch everything on the caller's HIP stream (hipStream_t as void*; NULL = the
 * context's own stream).  Lets a host framework time and order the work itself. */
int vhp_set_stream(vhp_ctx* ctx, void* hip_stream);

/* Replaces environment::getVisibilityField() hand-over (environment.h:58-60,
 * solver.cpp:14-17): uploads the occupancy complement and builds the bit-packed
 * row-major and column-major copies the kernels read.  The map is immutable until
 * the next vhp_set_map. */
int vhp_set_map(vhp_ctx* ctx, const uint8_t* occ_rowmajor, int nx, int ny);
/* Same, the uint8 map already resident in device memory. */
int vhp_set_map_device(vhp_ctx* ctx, const uint8_t* d_occ_rowmajor, int nx, int ny);

/* Replaces computeVisibility() / computeVisibilityUsingQueue() for a batch of
 * independent sources (the reference handles one source per call through the
 * member ls_, solver.cpp:214-218).  out holds n_src fields of nx*ny elements of
 * `dtype`, field s at out + s*nx*ny.  Cells the reference never writes (row 0 /
 * column 0 unless the source lies on them; for the queue variant
 * blocked and unreached cells) are returned as 0, i.e. the result of running the
 * reference on a freshly reset() solver. */
int vhp_sweep_batch(vhp_ctx* ctx, const int32_t* src_xy, int n_src, int variant, int dtype, void* out_host);
/* Device-resident form: d_src_xy and d_out are device pointers; asynchronous on the
 * context stream.  Sources are validated on the device; query with vhp_sync().  d_out may start at any element of the caller's
 * buffer (aligned to the element type, else VHP_ERR_ARG); fields that start on a 128-byte line, on a width that is a multiple of 8,
 * are stored fastest. */
int vhp_sweep_batch_device(vhp_ctx* ctx, const int32_t* d_src_xy, int n_src, int variant, int dtype, void* d_out);
/* Waits for the stream and returns the status of device-side validation
 * (VHP_ERR_SOURCE_OOB if any source of an earlier *_device call was out of range). */
int vhp_sync(vhp_ctx* ctx);

/* Replaces solve() (solver.cpp:76-160) incl. updateVisibility() (:379-565), the
 * heap / top() arg-min (visibilityBasedSolver.h:16-21,138) and resetQueue().
 * Coordinates are FIELD coordinates (the mode-2 y flip of solver.cpp:83-86 is the
 * caller's).  Outputs (any may be NULL): came_from nx*ny (cameFrom_, VHP_UNLABELLED
 * where unlabelled), vis_global / vis_local nx*ny (visibility_global_, visibility_ of
 * the last pivot), pivots_xy 2*(max_iter+2) (lightSources_[0..*n_pivots], the last
 * entry is `end`, solver.cpp:141), *n_pivots = nb_of_sources_.
 * Returns VHP_OK, one of the four validation codes, VHP_ERR_MAX_ITER (outputs are
 * still filled) or VHP_ERR_NOTHING_LIT. */
int vhp_planner_solve(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold,
                      uint64_t max_iter, uint64_t* came_from, double* vis_global, double* vis_local,
                      int32_t* pivots_xy, uint32_t* n_pivots);

/* The same solve with every result left in device memory (a batched or multi-GPU planner consumes labels and union on
 * the GPU; the host copy of vhp_planner_solve costs 10x the device loop at 1000^2).  *n_pivots = nb_of_sources_.
 * vhp_planner_results_device then hands out the device arrays of the most recent solve on this context, valid until the
 * next solve or vhp_set_map: labels nx*ny uint32 (0xFFFFFFFF where the reference holds (size_t)1e15), vis_global and
 * vis_local nx*ny doubles, pivots_xy int32 pairs 0 .. n_pivots.  Any output pointer may be NULL. */
int vhp_planner_solve_device(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold, uint64_t max_iter,
                             uint32_t* n_pivots);
int vhp_planner_results_device(vhp_ctx* ctx, const uint32_t** labels, const double** vis_global, const double** vis_local,
                               const int32_t** pivots_xy);

/* The speculative planner (the loop of solver.cpp:127-140 with every sweep launch taking the k - 1 best other candidates of
 * the last heuristic evaluation along; a launch sweeps k sources in the time it sweeps one).  k = 1, 2, 4 or 8.
 *   mode 0, exact: the extra fields go into a cache keyed by their source cell, and an iteration whose pivot is cached
 *     skips its sweep.  Every output equals vhp_planner_solve's bit for bit; only the number of sweep launches differs.
 *   mode 1, fast, NOT the reference's result: all k candidates of a launch are committed as pivots in that iteration, in
 *     rank order.  The outputs are a valid planner result (labels index pivots, every pivot was lit by an earlier one, the
 *     path reconstructs) but not the reference's pivots or path.
 * stats (may be NULL): [0] iterations whose pivot was cached, [1] iterations that swept, [2] fields swept.
 * pivots_xy: at least 2*(max_iter+2+8) ints -- in mode 1 an iteration commits up to k pivots before the max_iter test, so
 * *n_pivots can reach max_iter + 8.  Their launches take the latency sweep (8 k workgroups) wherever a batch of k would.  Measured
 * on maze_6 (threshold 0.1, 64 pivots; vhp_planner_solve 1.70 ms): mode 1 with k = 4 1.44 ms -- 149 pivots in 38 launches --, k = 2
 * 1.62; mode 0 2.44 ms: it sweeps k fields where the plain loop sweeps one and skips 13 of 64 sweeps, which does not pay there.  It
 * pays where pivots repeat (the reference's live-lock, threshold 0.25 on the same map: 3.8 ms against 5.6 until max_iter = 250; mode
 * 1 SOLVES that instance, 157 pivots in 1.51 ms).  DESIGN.md section 7.
 * Results stay on the device as with vhp_planner_solve_device (vhp_planner_results_device); host outputs may be NULL. */
int vhp_planner_solve_speculative(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold, uint64_t max_iter,
                                  int k, int mode, uint64_t* came_from, double* vis_global, double* vis_local, int32_t* pivots_xy,
                                  uint32_t* n_pivots, int32_t* stats);

/* Replaces reconstructPath() (solver.cpp:1183-1213): walks came_from -> pivots from
 * `end` until the label repeats; writes the path start-first into path_xy (capacity
 * cap points), its point count into *n_path, the summed eval_d length into *length.
 * pivots_xy holds entries 0 .. n_pivots (vhp_planner_solve's *n_pivots).  A label above
 * n_pivots (an unlabelled cell included) or a walk longer than n_pivots + 2 hops is
 * VHP_ERR_ARG; a path longer than cap is VHP_ERR_TOO_LARGE with *n_path = the size
 * needed and nothing written. */
int vhp_reconstruct_path(const uint64_t* came_from, const int32_t* pivots_xy, uint32_t n_pivots, int nx, int ny,
                         int end_x, int end_y, int32_t* path_xy, uint32_t cap, uint32_t* n_path, double* length);

/* Replaces raycasting() driven over all targets as benchmark() does (solver.cpp:226-232,
 * 267-290): a Bresenham ray from the source to every cell; a blocked cell met on the way
 * zeroes that cell and the target.  out: nx*ny doubles, 1 = visible (visibilityRayCasting_,
 * initialised to 1, solver.cpp:45).  Every write is a zero, so the union does not depend
 * on the order the rays are cast in and one thread per ray reproduces the reference. */
int vhp_raycast_all(vhp_ctx* ctx, int src_x, int src_y, double* out_host);

/* MATLAB-flavoured variants.  The reference's MATLAB demos are a different algorithm from its C++ program:
 * MATLAB_code/visibility/getAccessibilityMap.m has an explicit diagonal rule (cell (i,i*1/fac) = alpha * its diagonal
 * predecessor), a decay `alpha` on every update, a curve factor `fac`, and sweeps every row and column.
 * vhp_sweep_batch_variant replaces getAccessibilityMap(alpha, 1, lightPos, 1-obstacle, ., fac) for a batch of sources
 * (fp64 fields, host buffers).  vhp_planner_solve_variant replaces the exploration loop of
 * MATLAB_code/c_sample_planner_solving_random_environments.m:100-171 over getAccessibilityMapPlanner.m (fac = 1):
 * first-lit labels at v >= threshold (0-based waypoint index, UINT64_MAX where none), min-max-scaled heuristic, stop when
 * the newest waypoint's own field sees the target; waypoints_xy holds 2*(max_iter+3) ints, waypoints_xy[0..1] = start.
 * Returns VHP_OK, VHP_ERR_MAX_ITER (the script itself has no bound) or VHP_ERR_NOTHING_LIT.
 * Parity: against a line-by-line CPU restatement of the .m files (oracle/vhp_oracle_matlab.cpp); MATLAB itself is not
 * available to the build, so these modes are unpinned against it.  Grid sides up to 4096. */
int vhp_sweep_batch_variant(vhp_ctx* ctx, const int32_t* src_xy, int n_src, double alpha, double fac, double* out_host);
int vhp_planner_solve_variant(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold, double alpha,
                              uint64_t max_iter, uint64_t* label, double* map_builder, double* local, int32_t* waypoints_xy,
                              uint32_t* n_waypoints);

/* computeVisibility() (solver.cpp:570-696) with the reference's local `offset` (:573; added to both operands of every c_,
 * :590-591 ... :686-687) exposed.  offset = 0 is the reference at HEAD and equals vhp_sweep_batch bit for bit (use that:
 * this entry point runs a plain anti-diagonal kernel, not the tuned ones).  It exists because the reference's published
 * Samples/SFMLstandAloneVisibility.png was rendered by a build with offset = 1; with it the library reproduces that
 * image.  fp64 fields, host buffers, cells the reference never writes are 0, grid sides up to 4096. */
int vhp_sweep_batch_offset(vhp_ctx* ctx, const int32_t* src_xy, int n_src, double offset, double* out_host);

/* Elapsed milliseconds between the first and last kernel of the most recent
 * vhp_sweep_batch_device / planner call, from hipEvents recorded on the context
 * stream.  Blocks until that work has finished.  vhp_probe_stores and vhp_alloc_output time their probes with the same pair of
 * events: after either of them there is nothing to report (VHP_ERR_ARG, "nothing timed yet") until the next sweep or solve. */
int vhp_last_elapsed_ms(vhp_ctx* ctx, float* ms);

/* Per-launch kernel timing for benchmarks.  vhp_timing(ctx, 1) makes every following
 * vhp_sweep_batch_device call bracket what it launches -- the unit-ordering pre-kernel and the
 * sweep kernel -- with a pair of hipEvents on the context stream; vhp_timing_collect waits for them,
 * writes up to `cap` durations in milliseconds (oldest first), returns their count in *n and
 * clears the list.  vhp_timing(ctx, 0) switches it off. */
int vhp_timing(vhp_ctx* ctx, int enable);  /* enable > 1: also pre-creates that many event pairs */
int vhp_timing_collect(vhp_ctx* ctx, float* ms_out, int cap, int* n);

/* Launch-shape overrides for tuning and for parity tests that must reach every compiled shape
 * (0 / -1 = automatic, the default).  Keys: "rows_per_lane" (1, 2, 4), "strips" (1..8 wavefront
 * strips per octant), "multi_round" (1 = force the multi-round build), "slide" (0 / 1: y-major
 * column grid slid onto 128-byte lines), "pack" (1 = pack short quadrants), "lat_workgroups" (1, 2, 4, 8:
 * workgroups per octant of a latency-sweep launch; 0: one up to 1024 cells a side, two up to 2048, four up to 4096, eight beyond,
 * halved until the launch is resident at once), "kernel" (1 = front
 * sweep, 3 = pool sweep, 4 = latency sweep; 2 was the streaming sweep, retired in round 4 and refused),
 * "pool_contexts" (1..16: units a workgroup of the pool sweep holds
 * at once), "pool_static_round" (0: every unit of a pool-sweep launch is pulled from its queue; 1: the first unit of
 * every context is handed out by workgroup index; 2, the default: ... and a workgroup's second unit counts down from the end of the
 * round, so that the longest units share their CU with the shortest of the round), "field_stride" (elements from one field of a DEVICE-pointer batch to the next;
 * 0 = nx * ny, packed; a value below nx * ny makes vhp_sweep_batch_device fail with VHP_ERR_ARG; the host-buffer entry points
 * ignore it -- their results are packed --, the queue variant refuses it, and vhp_set_map resets it to 0),
 * "alloc_budget_pct" (vhp_alloc_output below).  The
 * results never depend on these; only the schedule (and, with field_stride, the placement of the fields) does. */
int vhp_set_option(vhp_ctx* ctx, const char* key, long long value);
/* Which kernel the last batch sweep of this context launched: 1 = front sweep (vhp_sweep_fronts),
 * 3 = pool sweep (vhp_pool_sweep), 4 = latency sweep
 * (vhp_lat_sweep), 0 = none yet.  For benchmarks and profiles. */
int vhp_last_sweep_kernel(const vhp_ctx* ctx);

/* The max-union of a batch of fields and the source that attains it, on the device: best[c] = max over k of field k at cell c,
 * arg[c] = first_index + the lowest k that attains it (a sequential max-union that replaces on strict improvement only) -- the
 * reference's union, src/visibilityBasedSolver.cpp:417-418 (visibility_global_ = max(visibility_, visibility_global_)), over a
 * whole batch at once, with the label the planner derives from it (:419-423).  What a batch sharded over several devices exchanges
 * instead of its fields (option 2: one union field and one label field per device).  d_fields: n_fields fields of nx * ny
 * elements of `dtype` (the map's grid; "field_stride" elements apart as in vhp_sweep_batch_device); d_best: nx * ny elements of
 * dtype; d_arg: nx * ny int32.  n_fields = 0: best = -1, arg = INT32_MAX everywhere.  One pass over the fields, no temporaries;
 * asynchronous on the context's stream.
 * vhp_union_partials_device: the same reduction over n_parts PARTIAL results -- d_bests: n_parts packed union fields, d_args: their
 * n_parts packed label fields (e.g. the partials of all devices after an all-gather); a tie goes to the lowest label. */
int vhp_union_fields_device(vhp_ctx* ctx, const void* d_fields, int n_fields, int dtype, int first_index, void* d_best, int32_t* d_arg);
int vhp_union_partials_device(vhp_ctx* ctx, const void* d_bests, const int32_t* d_args, int n_parts, int dtype, void* d_best, int32_t* d_arg);

/* A measurement aid, not part of the reference's surface (it has no device memory): the rate at which the memory behind a
 * device buffer takes two store patterns of a sweep launch, in TB/s of bytes stored -- 1 KB row pieces in many concurrent
 * streams, (a) every piece on the 128-byte line grid, (b) every other piece half a line off it, so that two lines per piece
 * are written in halves by different wavefronts at different times, with PLAIN stores.  The same physical memory answers (a)
 * with 4.9-5.0 or 5.9-6.1 and (b) with 3.6-3.7 or 5.2-5.4 depending on where the allocation landed;
 * bench.py reports both for the buffer it timed so that a result can be read against the state of its memory; vhp_alloc_output
 * (below) places a result buffer by it.  d_buf: 128-byte aligned, at least 128 MB; its contents are overwritten with zeros
 * (up to 16 GB of it are used).  Blocks until done. */
int vhp_probe_stores(vhp_ctx* ctx, void* d_buf, unsigned long long bytes, float* whole_lines_TBps, float* split_lines_TBps);

/* Device memory for results, placed by the library (the reference has no device memory; a maintainer's binding allocates its
 * result fields with this instead of hipMalloc).  The memory behind an allocation is of a faster or a slower kind, and no allocation
 * API chooses (a launch of 256 fields at 1000^2 takes 15-20 % longer on the one than on the other): up to
 * max_candidates allocations of `bytes` are made, each is probed (vhp_probe_stores; from 128 MB up), the one whose two rates add up
 * highest is kept, the others are freed.  BEST EFFORT: where the fast kind is rare the keeper is merely the best of what was tried.
 * The search is a guest on the device: the candidates -- all held until the choice is made, a freed one would be handed out
 * again -- stay within 25 % of the device memory that is free when the search starts (vhp_set_option "alloc_budget_pct", 1..90)
 * and within 64; it ends on the first buffer of the fast kind and after 8 candidates in a row that are no better than the best so
 * far.  While it runs, other users of the device see that much less free memory.  d_buf: 256-byte aligned, contents undefined.
 * whole_lines_TBps / split_lines_TBps (may be NULL): the probe's rates of the buffer kept (0 below 128 MB); n_tried (may be NULL):
 * allocations made.  Blocks; a one-off cost of 5-10 ms per candidate (vhp_alloc_output_cost tells what the last search took).
 * Buffers still allocated when the context is destroyed are freed with it. */
int vhp_alloc_output(vhp_ctx* ctx, unsigned long long bytes, int max_candidates, void** d_buf, float* whole_lines_TBps,
                     float* split_lines_TBps, int* n_tried);
/* What the last vhp_alloc_output of this context cost: wall-clock milliseconds of the search, and the device memory its
 * candidates held at the peak (bytes).  Either pointer may be NULL. */
int vhp_alloc_output_cost(const vhp_ctx* ctx, double* search_ms, unsigned long long* peak_bytes);
int vhp_free_output(vhp_ctx* ctx, void* d_buf);

/* ---- several devices of one node (the reference is one thread on one CPU and has nothing to replace here) ----
 * The sources of a batch are independent, so a batch shards over devices with no exchange step: device d of n sweeps the
 * block vhp_multi_shard_bounds(n_src, n, d) of the sources into its own buffer.  vhp_multi_create makes one context per listed
 * device (an ordinal may be listed twice: two contexts on one device), each with a stream of its own; vhp_multi_set_map gives
 * every device the map; vhp_multi_sweep_batch uploads the shards' sources, launches every device's sweep before it waits for
 * any, and returns the first device-side status that is not VHP_OK (d_out_per_device[d]: device memory ON device d for its
 * shard's fields, packed; NULL allowed where the shard is empty; on an error of device d the sweeps already enqueued on the
 * devices before it are waited for before the call returns); vhp_multi_allgather_fields then lands all n_src fields in source
 * order in d_all_per_device[d] on every device.  By default by direct peer copies: xGMI is point to point, every pair of devices
 * has a link of its own, and every (destination, source) pair has a STREAM of its own ("lane" k of device `to` carries the copy
 * from device (to + k) mod N), so the N - 1 inbound copies of a device are in flight together, one per link, and in round k no
 * two destinations pull from one source.  vhp_multi_allgather_plan returns that enqueue plan as data -- (to, from, lane, first
 * and one-past-last source of the piece) per copy, in enqueue order, up to `cap` entries; its return value is the number of
 * copies; host arithmetic only.  vhp_multi_use_rccl(m, 1) switches to RCCL instead (ncclAllGather on a single-process
 * communicator over the listed devices, one group call; librccl is loaded at run time; distinct devices only -- VHP_ERR_ARG when
 * an ordinal is listed twice --; a batch that does not divide by the number of devices still takes the peer copies).  The planner
 * does not shard (pivot k + 1 needs the union after pivot k): use vhp_multi_context(m, d) for replicas.  (torch.distributed /
 * RCCL form of the same: dist.py.)  Not thread-safe; every call restores the caller's current device.  Never run across devices
 * so far (no multi-GPU node was available): correct by construction, tested with one ordinal listed several times. */
typedef struct vhp_multi vhp_multi;
int vhp_multi_create(const int* device_ordinals, int n_devices, vhp_multi** out);
int vhp_multi_destroy(vhp_multi* m);
const char* vhp_multi_last_error(const vhp_multi* m);
int vhp_multi_devices(const vhp_multi* m);
vhp_ctx* vhp_multi_context(vhp_multi* m, int d);
void vhp_multi_shard_bounds(int n_src, int n_devices, int d, int* lo, int* hi);
int vhp_multi_set_map(vhp_multi* m, const uint8_t* occ_rowmajor, int nx, int ny);
int vhp_multi_sweep_batch(vhp_multi* m, const int32_t* src_xy, int n_src, int variant, int dtype, void* const* d_out_per_device);
int vhp_multi_allgather_fields(vhp_multi* m, int n_src, int dtype, void* const* d_shard_per_device, void* const* d_all_per_device);
int vhp_multi_allgather_plan(int n_src, int n_devices, int* to, int* from, int* lane, int* lo, int* hi, int cap);
int vhp_multi_use_rccl(vhp_multi* m, int enable);
/* The max-union of ALL n_src fields of a sharded batch and the source that attains it (vhp_union_fields_device), on every device:
 * each device reduces its own shard (one pass over its fields), the N partial results -- one union field and one label field per
 * device, whatever the batch -- are exchanged (peer copies on the lanes of vhp_multi_allgather_plan, or ncclAllGather after
 * vhp_multi_use_rccl), and every device merges them (vhp_union_partials_device; a tie goes to the lowest source index).  SURVEY 8e,
 * option 2: what the planner needs of a batch is this, not the fields (at BASELINE config 5: 0.2 GB per device instead of 137).
 * d_shard_per_device[d]: device d's fields as vhp_multi_sweep_batch left them; d_best_per_device[d]: nx * ny elements of dtype on
 * device d; d_arg_per_device[d]: nx * ny int32 there.  Blocks.  Like the rest of vhp_multi_*: NEVER RUN ACROSS DEVICES in this
 * repository's test environment (one GPU; the tests list its ordinal twice). */
int vhp_multi_union_fields(vhp_multi* m, int n_src, int dtype, void* const* d_shard_per_device, void* const* d_best_per_device,
                           int32_t* const* d_arg_per_device);

/* Library / build identification: "vhp-hip <version> gfx950". */
const char* vhp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VHP_H */
