// vhp_capi.hip -- the C ABI of include/vhp.h over the HIP kernels (gfx950).
// No CPU fallback lives here: every compute entry point launches HIP kernels and
// fails with VHP_ERR_HIP when the device or the runtime is unusable.
#include "vhp.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vhp_batch_launch.h"
#include "vhp_sweep.hip.h"
#include "vhp_planner.hip.h"
#include "vhp_queue.hip.h"
#include "vhp_variant.hip.h"
#include "vhp_union.hip.h"

struct vhp_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  std::string err;

  int n_cus = 256;
  std::vector<void*> placed;   // buffers handed out by vhp_alloc_output
  unsigned* d_probe_counter = nullptr;  // task counter of vhp_probe_stores
  double last_alloc_ms = 0.0;  // what the last vhp_alloc_output cost: wall time of the search ...
  unsigned long long last_alloc_peak_bytes = 0;  // ... and the device memory it held at its peak (vhp_alloc_output_cost)
  int opt_alloc_budget_pct = 25;  // vhp_alloc_output: share of the free device memory its candidates may hold at once
  int nx = 0, ny = 0;
  uint8_t* d_occ = nullptr;    // uint8 map (kept for the planner's validation and packing)
  std::vector<uint8_t> h_occ;  // ... its host copy when vhp_set_map brought it (empty after vhp_set_map_device)
  uint64_t* d_rows = nullptr;  // packed along x
  uint64_t* d_cols = nullptr;  // packed along y
  double* d_recip = nullptr;
  uint64_t* d_dmap = nullptr;  // packed along both diagonals, by x and by y (the latency sweep's maps: vhp_batch_launch.h lat_pack_diag_maps)
  int wpr = 0, wpc = 0;
  int* d_err = nullptr;

  // scratch for the host-buffer sweep entry point
  int32_t* d_src = nullptr;
  size_t d_src_cap = 0;
  void* d_out = nullptr;
  size_t d_out_cap = 0;
  double* d_bnd = nullptr;  // boundary rows of multi-round sweeps (sides above W*64*R)
  size_t d_bnd_cap = 0;
  int* d_order = nullptr;   // launch order of the (source, quadrant) units (+ one int4 descriptor per workgroup)
  int* d_lat_order = nullptr;  // launch order of the latency sweep's units where it launches more of them than the device has CUs
  size_t d_order_cap = 0;
  int* d_pool = nullptr;    // pool sweep: pull counter, unit order, diagonal lines, tagged boundary lines (zeroed when allocated)
  size_t d_pool_cap = 0;
  unsigned long long pool_epoch = 0x5A17000000000000ull;  // tag of the last pool launch
  bool timing = false;      // per-launch event pairs around the sweep kernel (vhp_timing)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed_launches;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;  // recycled pairs: no hipEventCreate inside a timed loop

  // launch-shape overrides (vhp_set_option); 0 / -1 = automatic
  int opt_rows_per_lane = 0;  // R: 1, 2 or 4
  int opt_strips = 0;         // W: 1..8
  int opt_multi = 0;          // 1: force the multi-round build
  int opt_slide = -1;         // 0 / 1: y-major column grid slid onto 128-byte lines
  int opt_pack = 0;           // 1: pack short quadrants into one workgroup
  int opt_lat_workgroups = 0; // latency sweep: workgroups per octant (0 automatic, 1 / 2 / 4 / 8: vhp_lat.hip lat_halves)
  const int* lat_src_index = nullptr;  // set around a latency-sweep launch of the planner's loop (vhp_planner_solve)
  const int* lat_skip = nullptr;
  const int* lat_pivot_rec = nullptr;  // ... or both and the pivot in one 16-byte record (vhp_planner.hip.h PlannerDev::rec)
  const vhp::PlannerDev* planner_dev = nullptr;  // ... and the launch is a whole planner iteration (vhp_batch_launch.h launch_lat_planner)
  const int* lat_slot_base = nullptr;  // ... of the speculative planner's (vhp_planner_solve_speculative)
  const int* lat_run_if = nullptr;
  bool lat_dark_unwritten = false;
  int opt_kernel = 0;         // 0 auto, 1 front sweep (vhp_sweep_fronts), 3 pool sweep (vhp_pool), 4 latency sweep (vhp_lat); 2 was the streaming sweep (retired in round 4)
  int last_kernel = 0;         // what the last batch sweep launched: 1 front sweep, 3 pool sweep, 4 latency sweep
  long long opt_field_stride = 0;  // device-pointer batch sweeps: elements from one field to the next (0: nx * ny, packed)
  int opt_pool_claim_ahead = -1;  // pool sweep: claim a strip this many steps ahead (-1 auto)
  int opt_pool_contexts = 0;   // pool sweep: units a workgroup holds at once (0 auto)
  int opt_pool_heads = 0;      // pool sweep: contexts that pull the largest units (0 auto)
  int opt_pool_tail_pct = 0;   // pool sweep: share of the units the filler contexts take from the small end (0 auto)
  int opt_pool_early_ctx = 0, opt_pool_late_pct = 0;  // pool sweep: late contexts (0 auto)
  int opt_pool_busy_cap = 0;   // pool sweep: no new unit while this many wavefronts of the workgroup are sweeping (0 auto)
  int opt_pool_static_round = 2;  // pool sweep: every context's first unit by workgroup index (2: the second head context counts down; 0: every unit pulled from the queue)

  vhp::PlannerState pl;  // device-resident planner state
  vhp::SpecState spec;   // field cache of the speculative planner
  vhp::QueueScratch qs;  // scratch of the queue-variant sweep
};

namespace {

// Every entry point runs on the context's device and leaves the caller's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (device, function) pair, not to a context: two contexts on
// one device share it.  Kept process-wide and monotonic -- the attribute is never set to a smaller value than before, so a
// context that raised it for a large grid is not undercut by another context's small one.
hipError_t raise_lds_limit(vhp_ctx* c, const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> raised;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = raised[{c->device, fn}];
  if (have >= bytes) return hipSuccess;
  hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (r == hipSuccess) have = bytes;
  return r;
}

int fail(vhp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define VHP_HIP(call)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(ctx, VHP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
  } while (0)

#define VHP_ON_DEVICE(ctx)                   \
  DeviceGuard device_guard_((ctx)->device); \
  if (!device_guard_.ok) return fail((ctx), VHP_ERR_HIP, "hipSetDevice failed")

vhp::DevMap dev_map(const vhp_ctx* c) {
  vhp::DevMap m;
  m.rows = c->d_rows;
  m.cols = c->d_cols;
  m.recip = c->d_recip;
  m.wpr = c->wpr;
  m.wpc = c->wpc;
  m.nx = c->nx;
  m.ny = c->ny;
  m.bnd = nullptr;
  m.bnd_len = 0;
  m.slide = 0;  // set per launch (launch_sweep_t)
  return m;
}

// A workgroup sweeps one quadrant with 2*W wavefronts (W strips per octant) and R rows/columns
// per lane.  Fronts longer than W*64*R are swept in rounds (`multi`).  Overridable for tuning
// with VHP_R / VHP_W / VHP_MULTI.
void pick_shape(const vhp_ctx* c, int maxdim, int* R, int* W, bool* multi, int n_src = 1, bool f64 = true, bool pitch64 = false) {
  if (maxdim <= 64) { *R = 1; *W = 1; }
  else if (maxdim <= 128) { *R = 2; *W = 1; }   // (one strip of two rows per lane per octant, no hand-over: 101^2 x 4096 sources 0.203 -> 0.191 ms, tools/small_grid_shapes.py)
  else if (maxdim <= 256) { *R = 1; *W = 4; }
  else if (maxdim <= 512) { *R = 2; *W = 4; }
  else if (maxdim <= 1024) { *R = 2; *W = 8; }
  else { *R = 4; *W = 8; }
  // (Round 1 sent batches of 256+ sources to the one-row-per-lane shape with whole-line flushes: -3 % at 1000^2 then,
  // +2 % when re-measured in round 2 (0.780 against 0.767 ms on one buffer); above 1024 such batches take the streaming
  // sweep now.  The shape stays reachable through vhp_set_option and is parity-tested.)
  // fp32 fields still do: there the 16 staged columns of that shape are one whole 64-byte sector per row, and the
  // two-rows-per-lane shape spills 52 registers in its fp32 build (0.53 against 0.61 ms at 1000^2, round 1).
  if (!f64 && maxdim > 256 && maxdim <= 1024 && n_src >= 256 && pitch64) { *R = 1; *W = 8; }
  if (c && c->opt_rows_per_lane) *R = c->opt_rows_per_lane;
  if (c && c->opt_strips) *W = c->opt_strips;
  *multi = (*W) * 64 * (*R) < maxdim;
  if (c && c->opt_multi) *multi = true;
  if (*R == 2 && *multi && *W > 4) *W = 4;  // that build is compiled for 8-wavefront workgroups
}

}  // namespace

namespace {

void free_map(vhp_ctx* c) {
  if (c->d_occ) hipFree(c->d_occ);
  if (c->d_rows) hipFree(c->d_rows);
  if (c->d_cols) hipFree(c->d_cols);
  if (c->d_recip) hipFree(c->d_recip);
  if (c->d_dmap) hipFree(c->d_dmap);
  c->d_occ = nullptr; c->d_rows = nullptr; c->d_cols = nullptr; c->d_recip = nullptr; c->d_dmap = nullptr;
  c->h_occ.clear();
  c->pl.h_occ = nullptr;
  vhp::planner_free(c->pl);
  vhp::spec_free(c->spec);
  vhp::queue_scratch_free(c->qs);
  c->nx = c->ny = 0;
}

template <int R, bool MULTI, typename OutT>
hipError_t launch_sweep_t(vhp_ctx* c, const int32_t* d_src, int n_src, OutT* d_out, int W) {
  const bool pack = c->opt_pack != 0;
  const size_t lds = vhp::sweep_lds_bytes(R, W, MULTI, pack);
  auto k = vhp::vhp_sweep_fronts<R, MULTI, OutT>;
  {
    hipError_t e = raise_lds_limit(c, reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return e;
  }
  const long long stride = c->opt_field_stride > 0 ? c->opt_field_stride : (long long)c->nx * c->ny;
  vhp::DevMap m = dev_map(c);
  // Sliding the y-major column grid onto 128-byte lines pays in the store-bound regime (many quadrants in flight:
  // +1 % at 1000^2, +3.5 % at 4096^2); a lone quadrant is latency-bound, and there the predicated stores of the
  // slid strip 0 -- the busiest wavefront -- cost 9 %.
  m.slide = n_src >= 96 ? 1 : 0;
  if (c->opt_slide >= 0) m.slide = c->opt_slide;
  hipError_t eb = vhp::attach_round_scratch(m, W * 64 * R, (size_t)n_src * vhp::kUnitsPerSource, &c->d_bnd, &c->d_bnd_cap);
  if (eb != hipSuccess) return eb;
  const size_t n_units = (size_t)n_src * vhp::kUnitsPerSource;
  const int* order = nullptr;
  const int4* desc = nullptr;
  // per-launch timing: from before the unit-ordering pre-kernel (part of what a launch costs) to after the sweep
  hipEvent_t ta = nullptr, tb = nullptr;
  if (c->timing) {
    if (!c->event_pool.empty()) {
      ta = c->event_pool.back().first;
      tb = c->event_pool.back().second;
      c->event_pool.pop_back();
    } else {
      if (hipEventCreate(&ta) != hipSuccess) return hipErrorOutOfMemory;
      if (hipEventCreate(&tb) != hipSuccess) { (void)hipEventDestroy(ta); return hipErrorOutOfMemory; }
    }
  }
  auto give_back = [&]() { if (ta) c->event_pool.push_back({ta, tb}); };
  if (ta) (void)hipEventRecord(ta, c->stream);
  if (n_src >= 8) {  // worth a 1-workgroup pre-kernel once the batch spans many CUs
    if (c->d_order_cap < n_units) {
      if (c->d_order) (void)hipFree(c->d_order);
      c->d_order = nullptr;
      c->d_order_cap = 0;
      hipError_t eo = hipMalloc(&c->d_order, n_units * (sizeof(int) + sizeof(int4)) + 16);
      if (eo != hipSuccess) { give_back(); return eo; }
      c->d_order_cap = n_units;
    }
    int4* d_desc = reinterpret_cast<int4*>(c->d_order);                      // n_units descriptors first (16-byte aligned)
    int* d_ord = reinterpret_cast<int*>(d_desc + n_units);                   // then the order list
    // Packing short quadrants into one workgroup is implemented and parity-tested, but measured slower
    // on MI355X: off unless VHP_PACK is set.
    const int pack_w = (!MULTI && W == 8 && pack) ? W : 0;
    hipLaunchKernelGGL(vhp::vhp_order_units, dim3(1), dim3(1024), 0, c->stream, d_src, n_src, c->nx, c->ny, 64 * R, pack_w, d_ord,
                       d_desc);
    order = d_ord;
    desc = d_desc;
  }
  const unsigned grid = (unsigned)n_units;
  hipLaunchKernelGGL(k, dim3(grid), dim3(128 * W), lds, c->stream, m, d_src, d_out, stride, c->d_err, order, desc);
  const hipError_t el = hipGetLastError();
  if (ta) {
    (void)hipEventRecord(tb, c->stream);
    c->timed_launches.push_back({ta, tb});
  }
  return el;
}

// Which kernel sweeps a batch: the pool sweep (vhp_pool.hpp) is built for throughput -- a pool of wavefronts per CU pulling
// strips of many units at once, whole-line non-temporal stores -- the front sweep (vhp_sweep.hip.h) takes what the two others
// do not (33-191 sources, widths that are not a multiple of 8).
// The latency sweep (vhp_lat.hpp): one workgroup per octant.  Measured on MI355X against the front sweep (tools/lat_vs_front.py,
// kernel time in us, front / latency sweep): one source at 256^2 33 / 28, 512^2 70 / 55, 690^2 94 / 67, 1000^2 132 / 96, 1536^2
// 291 / 176, 2048^2 380 / 244; 8 sources 65 / 47, 123 / 88, 176 / 130, 244 / 177, 607 / 362, 844 / 559; 32 sources 67 / 48, 138 / 92,
// 178 / 131, 296 / 196, 644 / 377, 866 / 604; 64 sources (two workgroups per CU wanted, one fits) 68 / 85, 139 / 153, 187 / 213,
// 300 / 317, 664 / 559, 999 / 1015; 3072^2: 1 / 8 / 32 sources 649 / 413, 1661 / 1171, 1771 / 1351; 4096^2: 8 / 32 sources 2543 / 1800,
// 2886 / 2554.  So: every launch whose octants have a CU each.
bool use_lat_kernel(const vhp_ctx* c, int n_src) {
  if (c->opt_kernel != 0 && c->opt_kernel != 4) return false;
  if (!vhp::lat_supported(c->nx, c->ny)) return false;
  if (c->opt_kernel == 4) return n_src <= 256 && vhp::lat_scratch_bytes(n_src, c->nx, c->ny) <= ((size_t)2 << 30);
  // (a caller that sets a launch shape of the front sweep is asking for the front sweep)
  if (c->opt_rows_per_lane || c->opt_strips || c->opt_multi || c->opt_slide >= 0 || c->opt_pack) return false;
  // Round 6, late: a long octant's bands go to two, four or eight workgroups (vhp_lat.hip lat_halves) -- while 16 x sources <= CUs.  With
  // 17-32 sources on a large grid the octants are back to one workgroup each, rounds of eight bands, and the pool sweep is ahead
  // (latency / pool sweep, us, profiles/r06_exp_workgroups_per_unit.txt): 2048^2: 16 sources 349 / 478, 24: 440 / 497, 32: 535 / 489;
  // 3072^2: 16: 664 / 904, 24: 1094 / 927, 32: 1377 / 1010; 4096^2: 16: 1226 / 1575, 24: 2053 / 1646, 32: 2414 / 1660; 1536^2: 24: 227 / 332,
  // 32: 280 / 332.  (Widths that are a multiple of 8: the pool sweep's other build was not measured against it.)
  const int maxdim = std::max(c->nx, c->ny);
  if ((c->nx & 7) == 0 && c->opt_kernel == 0 && vhp::pool_supported(c->nx, c->ny)) {
    if (n_src > 16 && maxdim > 2560) return false;
    if (n_src >= 28 && maxdim > 1792) return false;
  }
  // Round 6, last: MORE sources than octants fit the chip at once, up to 1024 cells a side.  The launch is as long as its longest octant
  // while most octants are short, so the workgroups beyond the first 256 find CUs long before the long ones are through (latency /
  // pool / front sweep, us, profiles/r06_exp_lat_beyond_one_octant_per_cu.txt): 320^2: 48 sources 51 / 92 / 87, 64: 62 / 96 / 87;
  // 512^2: 48: 84 / 131 / 128, 64: 107 / 134 / 128; 640^2: 64: 132 / 159 / 165; 768^2: 48: 157 / 192 / 221, 64: 190 / 193 / 220; 896^2: 48: 166 /
  // 207 / 250, 64: 214 / 210 / 252; 1000^2: 36: 155 / 234, 44: 201 / 234, 52: 220 / 234, 56: 269 / 237, 64: 273 / 242; from 96 sources the others
  // are ahead everywhere (512^2: 96: 128 / 127 / 138, 128: 156 / 132 / 145).  Other widths (the pool sweep's slower build): 64 sources at
  // 1001 x 971 273 against the front sweep's 295 and the pool sweep's ~ 340; at 101^2 28 / 34, at 255^2 60 / 71 (front).
  // ... and with the units launched by falling length of their march (vhp_lat.hip vhp_lat_order: the long octants start first, the short
  // ones fill the CUs they leave; unordered / ordered / pool sweep): 1000^2: 40 sources 171 / 157 / 233, 48: 212 / 171 / 234, 64: 273 / 216 / 239,
  // 96: 361 / 311 / 237; 768^2: 64: 190 / 125 / 197, 96: 234 / 187 / 181; 512^2: 64: 107 / 72 / 131, 96: 139 / 99 / 126, 128: 145 / 126 / 132.
  // 896^2: 80: 177 / 213 (ordered / pool); 640^2: 96: 133 / 154, 128: 177 / 157; 256^2: 96: 54 / 71, 128: 67 / 71 (front); 128^2: 96: 36 / 46, 128: 44 / 46
  // (front); 1280^2: 40: 229 / 291, 64: 342 / 287; 1536^2: 40: 365 / 335.
  // So, in rounds of octants (32 sources on 256 CUs): four up to 256 cells a side, three up to 640, two and a half up to 896, two up to
  // 1024, one and a quarter up to 1280, one above.
  const int one_round = c->n_cus / 8;   // sources whose octants have a CU each
  int cap = maxdim <= 256 ? 4 * one_round : maxdim <= 640 ? 3 * one_round : maxdim <= 896 ? one_round * 5 / 2 : maxdim <= 1024 ? 2 * one_round
          : maxdim <= 1280 ? one_round * 5 / 4 : one_round;
  // Widths that are not a multiple of 8, where the other two kernels store 8-byte cells or run their slower build (tools/kernel_choice_other_widths.py,
  // latency / pool / front, us): 1001 x 971: 64 sources 234 / 360 / 287, 128: 335 / 375 / 433; 1002 x 1000: 96: 281 / 327 / 378; 689^2: 80: 137 / 254 /
  // 191, 128: 230 / 262 / 278; 690 x 402: 128: 160 / 182 / 171; 500^2: 128: 146 / 170 / 147; 1101 x 1100: 64: 274 / 380 / 392; 1201 x 1200: 40: 241 / 416 /
  // 435; but 255^2: 128: 78 / 112 / 72, 101^2: 128: 44 / 57 / 39.  Four rounds up to 1280 cells a side, three up to 256.
  if ((c->nx & 7) != 0) cap = maxdim <= 256 ? 3 * one_round : maxdim <= 1280 ? 4 * one_round : one_round;
  // (the boundary lines of a launch -- 16 bytes per strip and step -- stay below two gigabytes: 32 sources at 8192^2 would take four)
  return n_src <= std::min(cap, 128) && vhp::lat_scratch_bytes(n_src, c->nx, c->ny) <= ((size_t)2 << 30);
}

bool use_pool_kernel(const vhp_ctx* c, int n_src) {
  if (c->opt_kernel == 1 || c->opt_kernel == 4) return false;
  if (!vhp::pool_supported(c->nx, c->ny)) return false;
  if (c->opt_kernel == 3) return true;
  // Measured on MI355X in round 3 (tools/ab_libs.py: three kernels on one buffer in one process; pool / streaming (since retired) / front, ms):
  // 256 sources at 1000^2 0.51-0.55 / 0.61 / 0.56 on one box and 0.67-0.70 / 0.73 / 0.74 on another; 512: 0.96 / - / 1.03;
  // 128: 0.44-0.46 / - / 0.45; 128 sources at 2048^2 1.25 / 1.42 / -; at 4096^2 3.88 / 4.29 / -.  The pool sweep from 192
  // sources up to side 1024, and wherever the streaming sweep used to be picked above it.
  // Round 4, with non-temporal stores and strips claimed ahead (tools/ab_libs.py, one buffer; front / pool, ms): 1000^2: 48 sources
  // 0.293 / 0.289, 64: 0.298 / 0.295, 96: 0.353 / 0.321, 128: 0.424 / 0.399, 192: 0.593 / 0.511; 512^2: 96: 0.137 / 0.165, 192: 0.199 /
  // 0.191, 384: 0.341 / 0.328; 2048^2: 32: 0.82 / 0.64, 48: 0.92 / 0.66, 64: 1.15 / 0.77, 96: 1.57 / 1.03; 4096^2: 24: 2.67 / 1.95, 48:
  // 3.43 / 2.18, 64: 4.42 / 2.42.  (Up to 32 sources the latency sweep has taken the launch before this is asked.)
  const int maxdim = std::max(c->nx, c->ny);
  if ((c->nx & 7) != 0) {
    // Widths that are not a multiple of 8 (the pool sweep's ANYW build; the front sweep stores
    // 8-byte cells there).  Front / pool, us: 1002x1000: 48 sources 373 / 336, 96: 621 / 384, 192: 1107 / 670; 1001x971: 48: 367 / 361,
    // 96: 639 / 394; 690x402: 48: 152 / 172, 96: 236 / 187, 192: 395 / 204; 500^2: 96: 166 / 196, 192: 280 / 209, 512: 663 / 462;
    // 250^2: 192: 94 / 123, 512: 220 / 259; 101^2: 512: 50 / 191.
    // (with the whole-line build: 1002x1000: 32: 309 / 329, 48: 365 / 320, 96: 621 / 394, 192: 1116 / 604; 690x402: 48: 152 / 189, 96: 240 / 195;
    // 500^2: 96: 167 / 185, 192: 285 / 208; 398^2: 96: 132 / 150, 192: 213 / 174)
    // Round 5, the front sweep with plain stores (1.5-1.9 x faster on these widths: vhp_sweep.hip.h StoreEmit; front / pool, us,
    // profiles/r05_front_vs_pool_plain_front_stores.txt): 1002x1000: 48 sources 298 / 321, 96: 438 / 322, 256: 1013 / 626; 1001x971: 48: 281 /
    // 337, 96: 425 / 353; 690x402: 96: 170 / 176, 192: 235 / 208, 256: 325 / 244; 500^2: 192: 171 / 182, 256: 265 / 238, 512: 521 / 438;
    // 398^2: 256: 139 / 199, 512: 356 / 353; 250^2: 512: 108 / 248.
    if (maxdim < 450) return false;
    if (maxdim <= 600) return n_src >= 256;
    if (maxdim <= 768) return n_src >= 128;
    if (maxdim <= 1100) return n_src >= 64;
    return n_src >= 24;
  }
  // (later in round 4, tools/kernel_ab.py, front / pool, us: 256^2: 192 sources 72 / 113, 384: 91 / 191; 384^2: 192: 124 / 148, 384: 175 / 249;
  // 512^2: 192: 157 / 184, 384: 284 / 315; 640^2: 96: 186 / 211, 192: 296 / 247, 384: 477 / 403; 768^2: 96: 227 / 237, 192: 397 / 329)
  // Round 5, with the strips in windows of 16 steps (profiles/r05_front_vs_pool_multiples_of_8.txt; front / pool, us): 320^2: 96 sources
  // 99 / 92, 192: 101 / 108, 1024: 314 / 475; 384^2: 96: 116 / 107, 384: 226 / 209, 1024: 495 / 529; 448^2: 96: 132 / 122, 192: 168 / 142,
  // 384: 286 / 244, 1024: 521 / 583; 512^2: 48: 135 / 137, 96: 144 / 131, 192: 201 / 168, 1024: 859 / 694; 576^2: 48: 150 / 146, 96: 164 / 142,
  // 384: 446 / 332; 640^2: 48: 169 / 163, 384: 519 / 384; 768^2: 48: 220 / 193, 96: 248 / 207; 1000^2: 48: 296 / 233, 96: 328 / 257, 384: 1104 / 809.
  // 256^2 and below: the front sweep at every batch size (256^2 x 4096: 911 / 1734; 104^2 x 4096: 247 / 1483).
  // Again with the front sweep's plain stores (same file as above): 384^2: 96 sources 115 / 102, 192: 126 / 121, 384: 189 / 202; 448^2: 48:
  // 118 / 124, 96: 133 / 118, 384: 274 / 242, 1024: 655 / 587; 512^2: 48: 136 / 135, 96: 145 / 128, 1024: 823 / 691; 576^2: 48: 152 / 147;
  // 640^2: 48: 172 / 162; 768^2: 48: 216 / 198; 1000^2: 48: 300 / 233.
  if (maxdim < 448) return false;
  if (maxdim < 576) return n_src >= 96;
  if (maxdim < 768) return n_src >= 48;
  if (maxdim <= 1024) return n_src >= 33;
  if (maxdim > 2560) return n_src >= 17;   // (where the latency sweep has left the launch to this: use_lat_kernel)
  return n_src >= 24;
}

// the pool sweep's scratch: its own allocation (nothing else may write the tagged lines), zero when new
hipError_t ensure_pool_scratch(vhp_ctx* c, size_t bytes) {
  if (c->d_pool_cap >= bytes) return hipSuccess;
  if (c->d_pool) (void)hipFree(c->d_pool);
  c->d_pool = nullptr;
  c->d_pool_cap = 0;
  hipError_t e = hipMalloc(&c->d_pool, bytes);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(c->d_pool, 0, bytes, c->stream);
  if (e != hipSuccess) return e;
  c->d_pool_cap = bytes;
  return hipSuccess;
}

template <typename OutT>
hipError_t launch_batch_sweep(vhp_ctx* c, const int32_t* d_src, int n_src, OutT* d_out, bool lat) {
  {
    hipError_t eo = ensure_pool_scratch(c, lat ? vhp::lat_scratch_bytes(n_src, c->nx, c->ny) : vhp::pool_scratch_bytes(n_src, c->nx, c->ny));
    if (eo != hipSuccess) return eo;
  }
  vhp::BatchArgs a;
  a.rows = c->d_rows; a.cols = c->d_cols; a.recip = c->d_recip; a.dmap = c->d_dmap;
  a.wpr = c->wpr; a.wpc = c->wpc; a.nx = c->nx; a.ny = c->ny;
  a.d_src = d_src; a.n_src = n_src; a.d_out = d_out;
  a.dtype = sizeof(OutT) == 8 ? VHP_F64 : VHP_F32;
  a.field_stride = c->opt_field_stride > 0 ? c->opt_field_stride : (long long)c->nx * c->ny;
  a.d_err = c->d_err;
  a.d_queue = c->d_pool;
  a.pool_epoch = ++c->pool_epoch;
  a.d_src_index = lat ? c->lat_src_index : nullptr;
  a.d_skip = lat ? c->lat_skip : nullptr;
  a.d_pivot_rec = lat ? c->lat_pivot_rec : nullptr;
  a.d_slot_base = lat ? c->lat_slot_base : nullptr;
  a.d_run_if = lat ? c->lat_run_if : nullptr;
  a.lat_dead_cells_are_zero = lat && c->lat_dark_unwritten;
  a.lat_workgroups = c->opt_lat_workgroups;
  if (lat && !c->d_lat_order) {
    hipError_t eo = hipMalloc(&c->d_lat_order, vhp::lat_order_bytes());
    if (eo != hipSuccess) return eo;
  }
  a.d_lat_order = c->d_lat_order;
  a.n_cus = c->n_cus;
  a.stream = c->stream;
  a.raise_lds = [c](const void* fn, size_t bytes) { return raise_lds_limit(c, fn, bytes); };
  a.ev_begin = a.ev_end = nullptr;
  a.pool_contexts = c->opt_pool_contexts;
  a.pool_claim_ahead = c->opt_pool_claim_ahead;
  a.pool_busy_cap = c->opt_pool_busy_cap;
  a.pool_early_ctx = c->opt_pool_early_ctx;
  a.pool_late_pct = c->opt_pool_late_pct;
  a.pool_tail_pct = c->opt_pool_tail_pct;
  a.pool_heads = c->opt_pool_heads;
  a.pool_static_round = c->opt_pool_static_round;
  if (c->timing) {
    if (!c->event_pool.empty()) {
      a.ev_begin = c->event_pool.back().first;
      a.ev_end = c->event_pool.back().second;
      c->event_pool.pop_back();
    } else {
      if (hipEventCreate(&a.ev_begin) != hipSuccess) return hipErrorOutOfMemory;
      if (hipEventCreate(&a.ev_end) != hipSuccess) { (void)hipEventDestroy(a.ev_begin); return hipErrorOutOfMemory; }
    }
  }
  const hipError_t e = lat ? (c->planner_dev ? vhp::launch_lat_planner(a, *c->planner_dev) : vhp::launch_lat(a)) : vhp::launch_pool(a);
  if (c->timing) {
    // (a launch that failed before its events were recorded must not leave a pair that can never be waited for)
    if (e == hipSuccess) c->timed_launches.push_back({a.ev_begin, a.ev_end});
    else c->event_pool.push_back({a.ev_begin, a.ev_end});
  }
  return e;
}

template <typename OutT>
hipError_t launch_sweep(vhp_ctx* c, const int32_t* d_src, int n_src, OutT* d_out) {
  if (use_lat_kernel(c, n_src)) {
    c->last_kernel = 4;
    return launch_batch_sweep<OutT>(c, d_src, n_src, d_out, true);
  }
  if (use_pool_kernel(c, n_src)) {
    c->last_kernel = 3;
    return launch_batch_sweep<OutT>(c, d_src, n_src, d_out, false);
  }
  c->last_kernel = 1;
  int R, W;
  bool multi;
  pick_shape(c, std::max(c->nx, c->ny), &R, &W, &multi, n_src, sizeof(OutT) == 8, (c->nx & 7) == 0);
  if constexpr (sizeof(OutT) == 4) {
    // fp32 fields: only the one-row-per-lane shape of the front sweep is built.  The two- and four-rows-per-lane fp32
    // instantiations needed 72-276 bytes of scratch per lane at 128 registers (round-2 verdict), and a scratch reload in
    // the flush path serialises the stores with their own completion; they are gone.  (Sides above 1024 take the streaming
    // sweep from 64-96 sources up as before; smaller fp32 batches there run one row per lane in rounds of 512 rows.)
    // (pick_shape's strip count where it already chose one row per lane -- small and odd-width grids keep their small
    // workgroups --, else as many strips as the rows need, up to 8; "rows_per_lane" other than 1 is ignored for fp32 fields)
    const int Wf = (c->opt_strips || R == 1) ? W : std::min(W * R, 8);
    const bool multi_f = Wf * 64 < std::max(c->nx, c->ny) || c->opt_multi;
    return multi_f ? launch_sweep_t<1, true, OutT>(c, d_src, n_src, d_out, Wf) : launch_sweep_t<1, false, OutT>(c, d_src, n_src, d_out, Wf);
  } else {
    switch (R) {
      case 1: return multi ? launch_sweep_t<1, true, OutT>(c, d_src, n_src, d_out, W) : launch_sweep_t<1, false, OutT>(c, d_src, n_src, d_out, W);
      case 2: return multi ? launch_sweep_t<2, true, OutT>(c, d_src, n_src, d_out, W) : launch_sweep_t<2, false, OutT>(c, d_src, n_src, d_out, W);
      default: return multi ? launch_sweep_t<4, true, OutT>(c, d_src, n_src, d_out, W) : launch_sweep_t<4, false, OutT>(c, d_src, n_src, d_out, W);
    }
  }
}

int finish_set_map(vhp_ctx* ctx, int nx, int ny) {
  // packed copies + reciprocal table
  ctx->wpr = (nx + 63) / 64 + 2;
  ctx->wpc = (ny + 63) / 64 + 2;
  const size_t rows_words = (size_t)ny * ctx->wpr, cols_words = (size_t)nx * ctx->wpc;
  VHP_HIP(hipMalloc(&ctx->d_rows, rows_words * 8));
  VHP_HIP(hipMalloc(&ctx->d_cols, cols_words * 8));
  VHP_HIP(hipMemsetAsync(ctx->d_rows, 0, rows_words * 8, ctx->stream));
  VHP_HIP(hipMemsetAsync(ctx->d_cols, 0, cols_words * 8, ctx->stream));
  {
    const long long waves = (long long)(ctx->wpr - 2) * ny;
    const int blocks = (int)((waves * 64 + 255) / 256);
    hipLaunchKernelGGL(vhp::vhp_pack_rows, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_occ, ctx->d_rows, nx, ny, ctx->wpr);
    VHP_HIP(hipGetLastError());
  }
  {
    const long long waves = (long long)(ctx->wpc - 2) * nx;
    const int blocks = (int)((waves * 64 + 255) / 256);
    hipLaunchKernelGGL(vhp::vhp_pack_cols, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_occ, ctx->d_cols, nx, ny, ctx->wpc);
    VHP_HIP(hipGetLastError());
  }
  const int nrec = std::max(nx, ny) + 1 + vhp::kRecipPad;
  std::vector<double> recip(nrec);
  recip[0] = 0.0;
  for (int k = 1; k < nrec; ++k) {
    volatile double d = (double)k;
    recip[k] = 1.0 / d;  // correctly rounded IEEE division on the host
  }
  if (vhp::lat_supported(nx, ny)) {
    const size_t bytes = vhp::lat_diag_map_bytes(nx, ny);
    VHP_HIP(hipMalloc(&ctx->d_dmap, bytes));
    VHP_HIP(hipMemsetAsync(ctx->d_dmap, 0, bytes, ctx->stream));
    VHP_HIP(vhp::lat_pack_diag_maps(ctx->d_occ, nx, ny, ctx->d_dmap, ctx->stream));
  }
  VHP_HIP(hipMalloc(&ctx->d_recip, nrec * sizeof(double)));
  VHP_HIP(hipMemcpyAsync(ctx->d_recip, recip.data(), nrec * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  VHP_HIP(hipStreamSynchronize(ctx->stream));
  ctx->nx = nx;
  ctx->ny = ny;
  return VHP_OK;
}

}  // namespace

extern "C" {

const char* vhp_version(void) { return "vhp-hip 0.1 gfx950"; }

int vhp_create(int device_ordinal, vhp_ctx** out) {
  if (!out) return VHP_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VHP_ERR_HIP;
  if (device_ordinal < 0 || device_ordinal >= n) return VHP_ERR_ARG;
  vhp_ctx* ctx = new vhp_ctx();
  ctx->device = device_ordinal;
  DeviceGuard guard(device_ordinal);
  if (!guard.ok || hipStreamCreate(&ctx->own_stream) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipMalloc(&ctx->d_err, sizeof(int)) != hipSuccess || hipMemset(ctx->d_err, 0, sizeof(int)) != hipSuccess) {
    delete ctx;
    return VHP_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cus = prop.multiProcessorCount;
  }
  *out = ctx;
  return VHP_OK;
}

int vhp_destroy(vhp_ctx* ctx) {
  if (!ctx) return VHP_ERR_ARG;
  DeviceGuard guard(ctx->device);
  hipStreamSynchronize(ctx->stream);
  free_map(ctx);
  if (ctx->d_src) hipFree(ctx->d_src);
  if (ctx->d_out) hipFree(ctx->d_out);
  if (ctx->d_bnd) hipFree(ctx->d_bnd);
  if (ctx->d_order) hipFree(ctx->d_order);
  if (ctx->d_lat_order) hipFree(ctx->d_lat_order);
  if (ctx->d_pool) hipFree(ctx->d_pool);
  for (auto& pr : ctx->timed_launches) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& pr : ctx->event_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (ctx->d_err) hipFree(ctx->d_err);
  for (void* p : ctx->placed) (void)hipFree(p);
  if (ctx->d_probe_counter) (void)hipFree(ctx->d_probe_counter);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return VHP_OK;
}

const char* vhp_last_error(const vhp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int vhp_set_stream(vhp_ctx* ctx, void* hip_stream) {
  if (!ctx) return VHP_ERR_ARG;
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return VHP_OK;
}

static int set_map_common(vhp_ctx* ctx, const uint8_t* src, int nx, int ny, bool from_device) {
  if (!ctx || !src || nx <= 0 || ny <= 0) return fail(ctx, VHP_ERR_ARG, "vhp_set_map: bad argument");
  if (nx > VHP_MAX_SIDE || ny > VHP_MAX_SIDE) return fail(ctx, VHP_ERR_TOO_LARGE, "vhp_set_map: grid side exceeds VHP_MAX_SIDE");
  VHP_ON_DEVICE(ctx);
  VHP_HIP(hipStreamSynchronize(ctx->stream));
  free_map(ctx);
  ctx->opt_field_stride = 0;  // (a stride belongs to a grid: one left over from a smaller grid would make the fields overlap)
  const size_t n = (size_t)nx * ny;
  VHP_HIP(hipMalloc(&ctx->d_occ, n));
  VHP_HIP(hipMemcpyAsync(ctx->d_occ, src, n, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  ctx->h_occ.clear();
  ctx->pl.h_occ = nullptr;
  const int rc = finish_set_map(ctx, nx, ny);
  if (rc != VHP_OK) {  // (nothing half-set: neither the old grid's sides with the new grid's arrays nor a host copy of a map that is not there)
    free_map(ctx);
    ctx->nx = ctx->ny = 0;
    return rc;
  }
  if (!from_device) ctx->h_occ.assign(src, src + n);
  ctx->pl.h_occ = ctx->h_occ.empty() ? nullptr : ctx->h_occ.data();
  return VHP_OK;
}

int vhp_set_map(vhp_ctx* ctx, const uint8_t* occ, int nx, int ny) { return set_map_common(ctx, occ, nx, ny, false); }
int vhp_set_map_device(vhp_ctx* ctx, const uint8_t* d_occ, int nx, int ny) { return set_map_common(ctx, d_occ, nx, ny, true); }

int vhp_sweep_batch_device(vhp_ctx* ctx, const int32_t* d_src_xy, int n_src, int variant, int dtype, void* d_out) {
  if (!ctx || !d_src_xy || !d_out || n_src < 0) return fail(ctx, VHP_ERR_ARG, "vhp_sweep_batch_device: bad argument");
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_sweep_batch_device: no map set");
  if (dtype != VHP_F64 && dtype != VHP_F32) return fail(ctx, VHP_ERR_ARG, "bad dtype");
  if (variant != VHP_SWEEP_FULL && variant != VHP_SWEEP_QUEUE) return fail(ctx, VHP_ERR_ARG, "bad variant");
  // (any alignment of whole elements is swept -- the kernels' builds for fields off the 16-byte grid --, a pointer inside an element is not)
  if (reinterpret_cast<uintptr_t>(d_out) % (dtype == VHP_F64 ? 8 : 4) != 0) return fail(ctx, VHP_ERR_ARG, "vhp_sweep_batch_device: d_out is not aligned to its element type");
  if (ctx->opt_field_stride > 0 && ctx->opt_field_stride < (long long)ctx->nx * ctx->ny)
    return fail(ctx, VHP_ERR_ARG, "vhp_sweep_batch_device: field_stride is smaller than a field (nx * ny elements): the fields would overlap");
  if (variant == VHP_SWEEP_QUEUE && ctx->opt_field_stride > 0 && ctx->opt_field_stride != (long long)ctx->nx * ctx->ny)
    return fail(ctx, VHP_ERR_ARG, "vhp_sweep_batch_device: the queue variant writes packed fields (field_stride must be 0)");
  if (n_src == 0) return VHP_OK;
  VHP_ON_DEVICE(ctx);
  VHP_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  hipError_t e;
  if (variant == VHP_SWEEP_QUEUE) {
    e = vhp::launch_queue_sweep_impl(ctx->qs, dev_map(ctx), ctx->d_occ, d_src_xy, n_src, dtype, d_out, ctx->d_err, ctx->stream);
  } else if (dtype == VHP_F64) {
    e = launch_sweep<double>(ctx, d_src_xy, n_src, static_cast<double*>(d_out));
  } else {
    e = launch_sweep<float>(ctx, d_src_xy, n_src, static_cast<float*>(d_out));
  }
  if (e != hipSuccess) return fail(ctx, VHP_ERR_HIP, std::string("sweep launch: ") + hipGetErrorString(e));
  VHP_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  ctx->timed = true;
  return VHP_OK;
}

int vhp_sync(vhp_ctx* ctx) {
  if (!ctx) return VHP_ERR_ARG;
  VHP_ON_DEVICE(ctx);
  VHP_HIP(hipStreamSynchronize(ctx->stream));
  int flag = 0;
  VHP_HIP(hipMemcpy(&flag, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) {
    VHP_HIP(hipMemset(ctx->d_err, 0, sizeof(int)));
    return fail(ctx, VHP_ERR_SOURCE_OOB, "a sweep source lies outside the grid");
  }
  return VHP_OK;
}

int vhp_sweep_batch(vhp_ctx* ctx, const int32_t* src_xy, int n_src, int variant, int dtype, void* out_host) {
  if (!ctx || !src_xy || !out_host || n_src < 0) return fail(ctx, VHP_ERR_ARG, "vhp_sweep_batch: bad argument");
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_sweep_batch: no map set");
  if (dtype != VHP_F64 && dtype != VHP_F32) return fail(ctx, VHP_ERR_ARG, "bad dtype");
  for (int s = 0; s < n_src; ++s)
    if (src_xy[2 * s] < 0 || src_xy[2 * s + 1] < 0 || src_xy[2 * s] >= ctx->nx || src_xy[2 * s + 1] >= ctx->ny)
      return fail(ctx, VHP_ERR_SOURCE_OOB, "a sweep source lies outside the grid");
  if (n_src == 0) return VHP_OK;
  VHP_ON_DEVICE(ctx);
  // (the library's own scratch holds packed fields and is copied out packed: "field_stride" is a property of a caller's device buffer)
  struct PackedHere { long long& v; long long keep; ~PackedHere() { v = keep; } } packed{ctx->opt_field_stride, ctx->opt_field_stride};
  ctx->opt_field_stride = 0;
  const size_t esz = dtype == VHP_F64 ? 8 : 4;
  const size_t cells = (size_t)ctx->nx * ctx->ny;
  // bound device scratch: process the batch in slices of at most ~1 GiB of output
  const int slice = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_src, ((size_t)1 << 30) / (cells * esz)));
  if (ctx->d_src_cap < (size_t)slice) {
    if (ctx->d_src) hipFree(ctx->d_src);
    ctx->d_src = nullptr;
    VHP_HIP(hipMalloc(&ctx->d_src, (size_t)slice * 2 * sizeof(int32_t)));
    ctx->d_src_cap = slice;
  }
  if (ctx->d_out_cap < (size_t)slice * cells * esz) {
    if (ctx->d_out) hipFree(ctx->d_out);
    ctx->d_out = nullptr;
    VHP_HIP(hipMalloc(&ctx->d_out, (size_t)slice * cells * esz));
    ctx->d_out_cap = (size_t)slice * cells * esz;
  }
  for (int s0 = 0; s0 < n_src; s0 += slice) {
    const int n = std::min(slice, n_src - s0);
    VHP_HIP(hipMemcpyAsync(ctx->d_src, src_xy + 2 * (size_t)s0, (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    int rc = vhp_sweep_batch_device(ctx, ctx->d_src, n, variant, dtype, ctx->d_out);
    if (rc != VHP_OK) return rc;
    VHP_HIP(hipMemcpyAsync(static_cast<char*>(out_host) + (size_t)s0 * cells * esz, ctx->d_out, (size_t)n * cells * esz,
                           hipMemcpyDeviceToHost, ctx->stream));
    VHP_HIP(hipStreamSynchronize(ctx->stream));
  }
  return vhp_sync(ctx);
}

int vhp_raycast_all(vhp_ctx* ctx, int src_x, int src_y, double* out_host) {
  if (!ctx || !out_host) return fail(ctx, VHP_ERR_ARG, "vhp_raycast_all: bad argument");
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_raycast_all: no map set");
  if (src_x < 0 || src_y < 0 || src_x >= ctx->nx || src_y >= ctx->ny) return fail(ctx, VHP_ERR_SOURCE_OOB, "source outside the grid");
  VHP_ON_DEVICE(ctx);
  const size_t cells = (size_t)ctx->nx * ctx->ny;
  if (ctx->d_out_cap < cells * 8) {
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    ctx->d_out = nullptr;
    VHP_HIP(hipMalloc(&ctx->d_out, cells * 8));
    ctx->d_out_cap = cells * 8;
  }
  double* d = static_cast<double*>(ctx->d_out);
  const unsigned blocks = (unsigned)((cells + 255) / 256);
  VHP_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(vhp::vhp_fill_f64, dim3(blocks), dim3(256), 0, ctx->stream, d, 1.0, cells);
  hipLaunchKernelGGL(vhp::vhp_raycast, dim3(blocks), dim3(256), 0, ctx->stream, ctx->nx, ctx->ny, ctx->d_occ, src_x, src_y, d);
  VHP_HIP(hipGetLastError());
  VHP_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  ctx->timed = true;
  VHP_HIP(hipMemcpyAsync(out_host, d, cells * 8, hipMemcpyDeviceToHost, ctx->stream));
  VHP_HIP(hipStreamSynchronize(ctx->stream));
  return VHP_OK;
}

// ---- max-union + arg-source of a batch of fields (vhp_union.hip.h) ---------------------------------------------------
static int union_common(vhp_ctx* ctx, const char* who, const void* d_fields, int n, int dtype, long long stride, const int32_t* d_labels, int first_index,
                        void* d_best, int32_t* d_arg) {
  if (!ctx || !d_best || !d_arg || n < 0 || (n > 0 && !d_fields)) return fail(ctx, VHP_ERR_ARG, std::string(who) + ": bad argument");
  if (dtype != VHP_F64 && dtype != VHP_F32) return fail(ctx, VHP_ERR_ARG, std::string(who) + ": bad dtype");
  if (ctx->nx <= 0) return fail(ctx, VHP_ERR_NO_MAP, std::string(who) + ": no map set (the grid's size is the map's)");
  const long long cells = (long long)ctx->nx * ctx->ny;
  if (stride == 0) stride = cells;
  if (stride < cells) return fail(ctx, VHP_ERR_ARG, std::string(who) + ": field_stride is smaller than a field");
  const size_t el = dtype == VHP_F64 ? 8 : 4;
  if (reinterpret_cast<uintptr_t>(d_fields) % el || reinterpret_cast<uintptr_t>(d_best) % el || reinterpret_cast<uintptr_t>(d_arg) % 4)
    return fail(ctx, VHP_ERR_ARG, std::string(who) + ": a pointer is not aligned to its element type");
  VHP_ON_DEVICE(ctx);
  VHP_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  const hipError_t e = dtype == VHP_F64 ? vhp::launch_union<double>(d_fields, n, stride, d_labels, first_index, cells, d_best, d_arg, ctx->n_cus, ctx->stream)
                                        : vhp::launch_union<float>(d_fields, n, stride, d_labels, first_index, cells, d_best, d_arg, ctx->n_cus, ctx->stream);
  if (e != hipSuccess) return fail(ctx, VHP_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
  VHP_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  ctx->timed = true;
  return VHP_OK;
}

int vhp_union_fields_device(vhp_ctx* ctx, const void* d_fields, int n_fields, int dtype, int first_index, void* d_best, int32_t* d_arg) {
  return union_common(ctx, "vhp_union_fields_device", d_fields, n_fields, dtype, ctx ? ctx->opt_field_stride : 0, nullptr, first_index, d_best, d_arg);
}

int vhp_union_partials_device(vhp_ctx* ctx, const void* d_bests, const int32_t* d_args, int n_parts, int dtype, void* d_best, int32_t* d_arg) {
  if (n_parts > 0 && !d_args) return fail(ctx, VHP_ERR_ARG, "vhp_union_partials_device: bad argument");
  return union_common(ctx, "vhp_union_partials_device", d_bests, n_parts, dtype, 0, d_args, 0, d_best, d_arg);
}

// ---- MATLAB-flavoured variants (vhp_variant.hip.h) -------------------------------------------------------------------
static int variant_launch_sweep(vhp_ctx* ctx, const int32_t* d_src, int n_src, double alpha, double fac, double* d_out) {
  const size_t lds = (size_t)3 * (std::max(ctx->nx, ctx->ny) + 1) * sizeof(double);
  auto k = vhp::variant::vhp_variant_sweep;
  hipError_t e = raise_lds_limit(ctx, reinterpret_cast<const void*>(k), lds);
  if (e != hipSuccess) return fail(ctx, VHP_ERR_HIP, std::string("variant sweep: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(k, dim3((unsigned)(4 * n_src)), dim3(1024), lds, ctx->stream, ctx->nx, ctx->ny, ctx->d_occ, d_src, d_out,
                     (long long)ctx->nx * ctx->ny, alpha, fac, ctx->d_err);
  VHP_HIP(hipGetLastError());
  return VHP_OK;
}

int vhp_sweep_batch_variant(vhp_ctx* ctx, const int32_t* src_xy, int n_src, double alpha, double fac, double* out_host) {
  if (!ctx || !src_xy || !out_host || n_src < 0 || !(fac > 0)) return fail(ctx, VHP_ERR_ARG, "vhp_sweep_batch_variant: bad argument");
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_sweep_batch_variant: no map set");
  if (std::max(ctx->nx, ctx->ny) > 4096) return fail(ctx, VHP_ERR_TOO_LARGE, "variant sweeps: grid side above 4096");
  for (int s = 0; s < n_src; ++s)
    if (src_xy[2 * s] < 0 || src_xy[2 * s + 1] < 0 || src_xy[2 * s] >= ctx->nx || src_xy[2 * s + 1] >= ctx->ny)
      return fail(ctx, VHP_ERR_SOURCE_OOB, "a sweep source lies outside the grid");
  if (n_src == 0) return VHP_OK;
  VHP_ON_DEVICE(ctx);
  const size_t cells = (size_t)ctx->nx * ctx->ny;
  const int slice = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_src, ((size_t)1 << 30) / (cells * 8)));
  if (ctx->d_src_cap < (size_t)slice) {
    if (ctx->d_src) hipFree(ctx->d_src);
    ctx->d_src = nullptr;
    VHP_HIP(hipMalloc(&ctx->d_src, (size_t)slice * 2 * sizeof(int32_t)));
    ctx->d_src_cap = slice;
  }
  if (ctx->d_out_cap < (size_t)slice * cells * 8) {
    if (ctx->d_out) hipFree(ctx->d_out);
    ctx->d_out = nullptr;
    VHP_HIP(hipMalloc(&ctx->d_out, (size_t)slice * cells * 8));
    ctx->d_out_cap = (size_t)slice * cells * 8;
  }
  for (int s0 = 0; s0 < n_src; s0 += slice) {
    const int n = std::min(slice, n_src - s0);
    VHP_HIP(hipMemcpyAsync(ctx->d_src, src_xy + 2 * (size_t)s0, (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const int rc = variant_launch_sweep(ctx, ctx->d_src, n, alpha, fac, static_cast<double*>(ctx->d_out));
    if (rc != VHP_OK) return rc;
    VHP_HIP(hipMemcpyAsync(out_host + (size_t)s0 * cells, ctx->d_out, (size_t)n * cells * 8, hipMemcpyDeviceToHost, ctx->stream));
    VHP_HIP(hipStreamSynchronize(ctx->stream));
  }
  return VHP_OK;
}

int vhp_sweep_batch_offset(vhp_ctx* ctx, const int32_t* src_xy, int n_src, double offset, double* out_host) {
  if (!ctx || !src_xy || !out_host || n_src < 0 || !(offset >= 0)) return fail(ctx, VHP_ERR_ARG, "vhp_sweep_batch_offset: bad argument");
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_sweep_batch_offset: no map set");
  if (std::max(ctx->nx, ctx->ny) > 4096) return fail(ctx, VHP_ERR_TOO_LARGE, "offset sweeps: grid side above 4096");
  for (int s = 0; s < n_src; ++s)
    if (src_xy[2 * s] < 0 || src_xy[2 * s + 1] < 0 || src_xy[2 * s] >= ctx->nx || src_xy[2 * s + 1] >= ctx->ny)
      return fail(ctx, VHP_ERR_SOURCE_OOB, "a sweep source lies outside the grid");
  if (n_src == 0) return VHP_OK;
  VHP_ON_DEVICE(ctx);
  const size_t cells = (size_t)ctx->nx * ctx->ny;
  const int slice = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_src, ((size_t)1 << 30) / (cells * 8)));
  if (ctx->d_src_cap < (size_t)slice) {
    if (ctx->d_src) hipFree(ctx->d_src);
    ctx->d_src = nullptr;
    VHP_HIP(hipMalloc(&ctx->d_src, (size_t)slice * 2 * sizeof(int32_t)));
    ctx->d_src_cap = slice;
  }
  if (ctx->d_out_cap < (size_t)slice * cells * 8) {
    if (ctx->d_out) hipFree(ctx->d_out);
    ctx->d_out = nullptr;
    VHP_HIP(hipMalloc(&ctx->d_out, (size_t)slice * cells * 8));
    ctx->d_out_cap = (size_t)slice * cells * 8;
  }
  const size_t lds = (size_t)3 * (std::max(ctx->nx, ctx->ny) + 1) * sizeof(double);
  auto k = vhp::variant::vhp_offset_sweep;
  hipError_t e = raise_lds_limit(ctx, reinterpret_cast<const void*>(k), lds);
  if (e != hipSuccess) return fail(ctx, VHP_ERR_HIP, std::string("offset sweep: ") + hipGetErrorString(e));
  for (int s0 = 0; s0 < n_src; s0 += slice) {
    const int n = std::min(slice, n_src - s0);
    VHP_HIP(hipMemcpyAsync(ctx->d_src, src_xy + 2 * (size_t)s0, (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    VHP_HIP(hipMemsetAsync(ctx->d_out, 0, (size_t)n * cells * 8, ctx->stream));  // a freshly reset() solver
    hipLaunchKernelGGL(k, dim3((unsigned)(4 * n)), dim3(1024), lds, ctx->stream, ctx->nx, ctx->ny, ctx->d_occ, ctx->d_src,
                       static_cast<double*>(ctx->d_out), (long long)cells, offset, ctx->d_err);
    VHP_HIP(hipGetLastError());
    VHP_HIP(hipMemcpyAsync(out_host + (size_t)s0 * cells, ctx->d_out, (size_t)n * cells * 8, hipMemcpyDeviceToHost, ctx->stream));
    VHP_HIP(hipStreamSynchronize(ctx->stream));
  }
  return VHP_OK;
}

int vhp_planner_solve_variant(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold, double alpha,
                              uint64_t max_iter, uint64_t* label, double* map_builder, double* local, int32_t* waypoints_xy,
                              uint32_t* n_waypoints) {
  using namespace vhp::variant;
  if (!ctx) return VHP_ERR_ARG;
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_planner_solve_variant: no map set");
  const int nx = ctx->nx, ny = ctx->ny;
  if (std::max(nx, ny) > 4096) return fail(ctx, VHP_ERR_TOO_LARGE, "variant planner: grid side above 4096");
  if ((unsigned)start_x >= (unsigned)nx || (unsigned)start_y >= (unsigned)ny) return fail(ctx, VHP_ERR_START_OOB, "Start point is out of bounds.");
  if ((unsigned)end_x >= (unsigned)nx || (unsigned)end_y >= (unsigned)ny) return fail(ctx, VHP_ERR_END_OOB, "End point is out of bounds.");
  if (max_iter > (1u << 20)) return fail(ctx, VHP_ERR_ARG, "max_iter too large");
  VHP_ON_DEVICE(ctx);
  const size_t cells = (size_t)nx * ny;
  double *d_uni = nullptr, *d_loc = nullptr;
  uint32_t* d_lab = nullptr;
  unsigned long long* d_lab64 = nullptr;
  int32_t* d_way = nullptr;
  PlannerCtl* d_ctl = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_uni, (void*)d_loc, (void*)d_lab, (void*)d_lab64, (void*)d_way, (void*)d_ctl})
      if (p) (void)hipFree(p);
  };
#define VHP_V(call)                                                                      \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) { cleanup(); return fail(ctx, VHP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } \
  } while (0)
  VHP_V(hipMalloc(&d_uni, cells * 8));
  VHP_V(hipMalloc(&d_loc, cells * 8));
  VHP_V(hipMalloc(&d_lab, cells * 4));
  VHP_V(hipMalloc(&d_lab64, cells * 8));
  VHP_V(hipMalloc(&d_way, 2 * (size_t)(max_iter + 3) * sizeof(int32_t)));
  VHP_V(hipMalloc(&d_ctl, sizeof(PlannerCtl)));
  VHP_V(hipMemsetAsync(d_uni, 0, cells * 8, ctx->stream));
  VHP_V(hipMemsetAsync(d_lab, 0xff, cells * 4, ctx->stream));
  const int32_t w0[2] = {start_x, start_y};
  const PlannerCtl c0{1, 0, 0, 0};
  const uint32_t zero = 0;
  VHP_V(hipMemcpyAsync(d_way, w0, sizeof(w0), hipMemcpyHostToDevice, ctx->stream));
  VHP_V(hipMemcpyAsync(d_ctl, &c0, sizeof(c0), hipMemcpyHostToDevice, ctx->stream));
  VHP_V(hipMemcpyAsync(d_lab + ((size_t)start_y * nx + start_x), &zero, 4, hipMemcpyHostToDevice, ctx->stream));  // lightSource_enum(start) = 1
  const unsigned eb = (unsigned)((cells + 255) / 256);
  PlannerCtl h{};
  for (;;) {
    // sweep from the current waypoint (d_way[2*iter]), union + labels, stop test
    int rc = variant_launch_sweep(ctx, d_way + 2 * h.iter, 1, alpha, 1.0, d_loc);
    if (rc != VHP_OK) { cleanup(); return rc; }
    hipLaunchKernelGGL(vhp_variant_update, dim3(eb), dim3(256), 0, ctx->stream, d_loc, d_uni, d_lab, cells, threshold, d_ctl);
    hipLaunchKernelGGL(vhp_variant_check, dim3(1), dim3(64), 0, ctx->stream, d_loc, nx, end_x, end_y, threshold, d_ctl);
    hipLaunchKernelGGL(vhp_variant_pick, dim3(1), dim3(1024), 0, ctx->stream, d_uni, nx, ny, end_x, end_y, threshold,
                       (unsigned long long)max_iter, d_way, d_ctl);
    VHP_V(hipGetLastError());
    VHP_V(hipMemcpyAsync(&h, d_ctl, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    VHP_V(hipStreamSynchronize(ctx->stream));
    if (h.done) break;
  }
  if (n_waypoints) *n_waypoints = (uint32_t)h.n_way;
  if (waypoints_xy) VHP_V(hipMemcpyAsync(waypoints_xy, d_way, 2 * (size_t)h.n_way * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (label) {
    hipLaunchKernelGGL(vhp_variant_labels_to_u64, dim3(eb), dim3(256), 0, ctx->stream, d_lab, d_lab64, cells, ~0ull);
    VHP_V(hipMemcpyAsync(label, d_lab64, cells * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (map_builder) VHP_V(hipMemcpyAsync(map_builder, d_uni, cells * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (local) VHP_V(hipMemcpyAsync(local, d_loc, cells * 8, hipMemcpyDeviceToHost, ctx->stream));
  VHP_V(hipStreamSynchronize(ctx->stream));
#undef VHP_V
  cleanup();
  if (h.status == VHP_ERR_MAX_ITER) ctx->err = "variant planner: max_iter reached";
  if (h.status == VHP_ERR_NOTHING_LIT) ctx->err = "variant planner: no candidate above the threshold";
  return h.status;
}

int vhp_timing(vhp_ctx* ctx, int enable) {
  if (!ctx) return VHP_ERR_ARG;
  ctx->timing = enable != 0;
  if (enable > 1) {  // pre-create `enable` event pairs so that no launch inside a timed loop has to
    VHP_ON_DEVICE(ctx);
    while ((int)ctx->event_pool.size() < enable) {
      hipEvent_t a = nullptr, b = nullptr;
      VHP_HIP(hipEventCreate(&a));
      VHP_HIP(hipEventCreate(&b));
      ctx->event_pool.push_back({a, b});
    }
  }
  return VHP_OK;
}

int vhp_set_option(vhp_ctx* ctx, const char* key, long long value) {
  if (!ctx || !key) return VHP_ERR_ARG;
  const std::string k(key);
  const int v = (int)value;
  if (k == "rows_per_lane") { if (v != 0 && v != 1 && v != 2 && v != 4) return fail(ctx, VHP_ERR_ARG, "rows_per_lane: 0, 1, 2 or 4"); ctx->opt_rows_per_lane = v; }
  else if (k == "strips") { if (v < 0 || v > 8) return fail(ctx, VHP_ERR_ARG, "strips: 0..8"); ctx->opt_strips = v; }
  else if (k == "multi_round") { ctx->opt_multi = v != 0; }
  else if (k == "slide") { if (v < -1 || v > 1) return fail(ctx, VHP_ERR_ARG, "slide: -1, 0 or 1"); ctx->opt_slide = v; }
  else if (k == "pack") { ctx->opt_pack = v != 0; }
  else if (k == "lat_workgroups") { if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8) return fail(ctx, VHP_ERR_ARG, "lat_workgroups: 0 (automatic), 1, 2, 4 or 8"); ctx->opt_lat_workgroups = v; }
  else if (k == "kernel") { if (v < 0 || v > 4 || v == 2) return fail(ctx, VHP_ERR_ARG, "kernel: 0 auto, 1 fronts, 3 pool, 4 latency (2, the streaming sweep, was retired)"); ctx->opt_kernel = v; }
  else if (k == "field_stride") { if (value < 0) return fail(ctx, VHP_ERR_ARG, "field_stride: 0 (packed) or elements per field"); ctx->opt_field_stride = value; }
  else if (k == "pool_claim_ahead") { if (v < -1 || v > 64) return fail(ctx, VHP_ERR_ARG, "pool_claim_ahead: -1 (automatic) .. 64"); ctx->opt_pool_claim_ahead = v; }
  else if (k == "pool_heads") { if (v < 0 || v > 16) return fail(ctx, VHP_ERR_ARG, "pool_heads: 0 (automatic) .. 16"); ctx->opt_pool_heads = v; }
  else if (k == "pool_tail_pct") { if (v < 0 || v > 100) return fail(ctx, VHP_ERR_ARG, "pool_tail_pct: 0 (automatic) .. 100"); ctx->opt_pool_tail_pct = v; }
  else if (k == "pool_early_ctx") { if (v < 0 || v > 16) return fail(ctx, VHP_ERR_ARG, "pool_early_ctx: 0 (automatic) .. 16"); ctx->opt_pool_early_ctx = v; }
  else if (k == "pool_late_pct") { if (v < 0 || v > 100) return fail(ctx, VHP_ERR_ARG, "pool_late_pct: 0 (automatic) .. 100"); ctx->opt_pool_late_pct = v; }
  else if (k == "pool_busy_cap") { if (v < 0 || v > 16) return fail(ctx, VHP_ERR_ARG, "pool_busy_cap: 0 (automatic) .. 16"); ctx->opt_pool_busy_cap = v; }
  else if (k == "pool_contexts") { if (v < 0 || v > 16) return fail(ctx, VHP_ERR_ARG, "pool_contexts: 0 (automatic) .. 16"); ctx->opt_pool_contexts = v; }
  else if (k == "pool_static_round") { if (v < 0 || v > 2) return fail(ctx, VHP_ERR_ARG, "pool_static_round: 0, 1 or 2"); ctx->opt_pool_static_round = v; }
  else if (k == "alloc_budget_pct") { if (v < 1 || v > 90) return fail(ctx, VHP_ERR_ARG, "alloc_budget_pct: 1 .. 90 (per cent of the free device memory)"); ctx->opt_alloc_budget_pct = v; }
  else return fail(ctx, VHP_ERR_ARG, "vhp_set_option: unknown key '" + k + "'");
  return VHP_OK;
}

int vhp_last_sweep_kernel(const vhp_ctx* ctx) { return ctx ? ctx->last_kernel : 0; }

int vhp_timing_collect(vhp_ctx* ctx, float* ms_out, int cap, int* n) {
  if (!ctx || !n || cap < 0 || (cap > 0 && !ms_out)) return VHP_ERR_ARG;
  VHP_ON_DEVICE(ctx);
  int k = 0, bad = 0;
  for (auto& pr : ctx->timed_launches) {
    float ms = 0.f;
    // a pair that cannot be read (e.g. never recorded) is dropped, not left to fail every later call
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      if (k < cap) ms_out[k] = ms;
      ++k;
    } else {
      ++bad;
      (void)hipGetLastError();
    }
    ctx->event_pool.push_back(pr);  // recycled by the next timed launches
  }
  ctx->timed_launches.clear();
  *n = std::min(k, cap);
  if (bad) return fail(ctx, VHP_ERR_HIP, "vhp_timing_collect: " + std::to_string(bad) + " timed launch(es) could not be read");
  return VHP_OK;
}

// ---- vhp_probe_stores: what the memory behind a buffer does with whole and with split lines (a measurement aid) -----------
namespace {
// Persistent wavefronts pull tasks; a task = 1000 rows of 8000 B (the C3 field's pitch: every other row starts half a 128-byte
// line off the grid) x a band of 1 KB; a store instruction = one row of the band, plain stores.  split = 0: every piece moved
// onto the line grid (whole lines only); split = 1: the pieces where they fall (two half lines per odd row).
__global__ void __launch_bounds__(256) vhp_store_probe_kernel(char* out, int n_tasks, int split, unsigned* counter) {
  extern __shared__ double probe_lds[];
  const int lane = threadIdx.x & 63;
  const double2 val = make_double2(0.0, 0.0);
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)n_tasks) break;
    const unsigned blk = t / 7u, band = t - blk * 7u;
    char* base = out + (size_t)blk * 8000000u + (size_t)band * 1024u + (size_t)lane * 16u;
    for (int row = 0; row < 1000; ++row) *reinterpret_cast<double2*>(base + (size_t)row * 8000u + ((!split && (row & 1)) ? 64 : 0)) = val;
  }
}
}  // namespace

int vhp_probe_stores(vhp_ctx* ctx, void* d_buf, unsigned long long bytes, float* whole_lines_TBps, float* split_lines_TBps) {
  if (!ctx || !d_buf || !whole_lines_TBps || !split_lines_TBps) return VHP_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(d_buf) & 127u) != 0) return fail(ctx, VHP_ERR_ARG, "vhp_probe_stores: the buffer must start on a 128-byte line");
  const unsigned long long blocks = bytes / 8000000ull;
  if (blocks < 16) return fail(ctx, VHP_ERR_ARG, "vhp_probe_stores: needs at least 128 MB to say anything about the memory");
  VHP_ON_DEVICE(ctx);
  const int n_tasks = (int)std::min<unsigned long long>(blocks, 2048) * 7;
  // (the task counter lives with the context: vhp_alloc_output probes up to 64 buffers, and a hipFree per probe synchronises the device;
  // the timing pair is the context's own ev0 / ev1 -- nothing here can leak on an early return)
  if (!ctx->d_probe_counter) VHP_HIP(hipMalloc(&ctx->d_probe_counter, 8));
  float res[2] = {0.f, 0.f};
  for (int split = 0; split < 2; ++split) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      VHP_HIP(hipMemsetAsync(ctx->d_probe_counter, 0, 4, ctx->stream));
      VHP_HIP(hipEventRecord(ctx->ev0, ctx->stream));
      // three workgroups of four wavefronts per CU (52 KB of LDS each keeps a fourth out): what a launch of the pool sweep holds
      hipLaunchKernelGGL(vhp_store_probe_kernel, dim3((unsigned)ctx->n_cus * 3), dim3(256), 52 * 1024, ctx->stream, static_cast<char*>(d_buf), n_tasks, split, ctx->d_probe_counter);
      VHP_HIP(hipGetLastError());
      VHP_HIP(hipEventRecord(ctx->ev1, ctx->stream));
      VHP_HIP(hipEventSynchronize(ctx->ev1));
      float ms = 0.f;
      VHP_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      if (rep > 0 && ms < best) best = ms;
    }
    res[split] = (float)((double)n_tasks * 1000.0 * 1024.0 / ((double)best * 1e-3) / 1e12);
  }
  ctx->timed = false;  // (ev0 / ev1 no longer bracket a sweep)
  *whole_lines_TBps = res[0];
  *split_lines_TBps = res[1];
  return VHP_OK;
}

int vhp_alloc_output(vhp_ctx* ctx, unsigned long long bytes, int max_candidates, void** d_buf, float* whole_lines_TBps, float* split_lines_TBps,
                     int* n_tried) {
  if (!ctx || !d_buf || bytes == 0 || max_candidates < 1) return fail(ctx, VHP_ERR_ARG, "vhp_alloc_output: bad argument");
  VHP_ON_DEVICE(ctx);
  *d_buf = nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  size_t free_b = 0, total_b = 0;
  VHP_HIP(hipMemGetInfo(&free_b, &total_b));
  // Every candidate stays allocated until the choice is made -- a freed one would be handed out again.  The search is a guest on
  // the device: what it holds at once stays within "alloc_budget_pct" (default 25) per cent of the memory that is free when it
  // starts, it ends on the first buffer of the fast kind, and it gives up after 8 candidates in a row that are no better than the
  // best so far (where the fast kind is rare a longer search mostly finds more of the same).
  const unsigned long long budget = (unsigned long long)free_b / 100ull * (unsigned long long)ctx->opt_alloc_budget_pct;
  const int cap = (int)std::min<unsigned long long>((unsigned long long)std::min(max_candidates, 64), std::max<unsigned long long>(1, budget / bytes));
  const bool probed = bytes >= 16ull * 8000000ull;   // (vhp_probe_stores says nothing about less than 128 MB)
  struct Held {  // (whatever way this function is left, only the keeper survives)
    std::vector<void*> v;
    void* keep = nullptr;
    ~Held() { for (void* q : v) if (q != keep) (void)hipFree(q); }
  } cand;
  int best = -1, tried = 0, since_best = 0;
  float best_w = 0.f, best_s = 0.f;
  for (int k = 0; k < cap; ++k) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
    cand.v.push_back(p);
    ++tried;
    float w = 0.f, sp = 0.f;
    if (probed) {
      const int rc = vhp_probe_stores(ctx, p, bytes, &w, &sp);
      if (rc != VHP_OK) return rc;
    }
    // (the two rates move together -- 4.9 / 3.6 on the slow kind, 6.0 / 5.3 on the fast, anything between on a buffer that straddles
    // both --: their sum ranks the candidates; an improvement is more than the probe's own scatter of ~0.05 TB/s)
    if (best < 0 || w + sp > best_w + best_s + 0.05f) { best = k; best_w = w; best_s = sp; since_best = 0; } else ++since_best;
    if (!probed || (w >= 5.5f && sp >= 4.6f)) break;   // the fast kind (5.6-6.1 / 4.6-5.4 by box): nothing better to find
    if (since_best >= 8) break;
  }
  ctx->last_alloc_peak_bytes = (unsigned long long)tried * bytes;
  ctx->last_alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (best < 0) return fail(ctx, VHP_ERR_HIP, "vhp_alloc_output: out of device memory");
  cand.keep = cand.v[best];
  ctx->placed.push_back(cand.keep);
  *d_buf = cand.keep;
  if (whole_lines_TBps) *whole_lines_TBps = best_w;
  if (split_lines_TBps) *split_lines_TBps = best_s;
  if (n_tried) *n_tried = tried;
  return VHP_OK;
}

int vhp_alloc_output_cost(const vhp_ctx* ctx, double* search_ms, unsigned long long* peak_bytes) {
  if (!ctx) return VHP_ERR_ARG;
  if (search_ms) *search_ms = ctx->last_alloc_ms;
  if (peak_bytes) *peak_bytes = ctx->last_alloc_peak_bytes;
  return VHP_OK;
}

int vhp_free_output(vhp_ctx* ctx, void* d_buf) {
  if (!ctx || !d_buf) return VHP_ERR_ARG;
  auto it = std::find(ctx->placed.begin(), ctx->placed.end(), d_buf);
  if (it == ctx->placed.end()) return fail(ctx, VHP_ERR_ARG, "vhp_free_output: not a buffer of vhp_alloc_output of this context");
  VHP_ON_DEVICE(ctx);
  VHP_HIP(hipStreamSynchronize(ctx->stream));
  ctx->placed.erase(it);
  VHP_HIP(hipFree(d_buf));
  return VHP_OK;
}

int vhp_last_elapsed_ms(vhp_ctx* ctx, float* ms) {
  if (!ctx || !ms) return VHP_ERR_ARG;
  if (!ctx->timed) return fail(ctx, VHP_ERR_ARG, "nothing timed yet");
  VHP_ON_DEVICE(ctx);
  VHP_HIP(hipEventSynchronize(ctx->ev1));
  VHP_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return VHP_OK;
}

int vhp_planner_solve(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold,
                      uint64_t max_iter, uint64_t* came_from, double* vis_global, double* vis_local,
                      int32_t* pivots_xy, uint32_t* n_pivots) {
  if (!ctx) return VHP_ERR_ARG;
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_planner_solve: no map set");
  VHP_ON_DEVICE(ctx);
  std::string msg;
  vhp::DevMap pm = dev_map(ctx);
  {
    int R, W;
    bool multi;
    pick_shape(ctx, std::max(ctx->nx, ctx->ny), &R, &W, &multi);
    hipError_t eb = vhp::attach_round_scratch(pm, W * 64 * R, 4, &ctx->d_bnd, &ctx->d_bnd_cap);
    if (eb != hipSuccess) return fail(ctx, VHP_ERR_HIP, std::string("scratch: ") + hipGetErrorString(eb));
    ctx->pl.R = R;
    ctx->pl.W = W;
    ctx->pl.multi = multi;
    ctx->pl.raise_lds = [ctx](const void* fn, size_t bytes) { return raise_lds_limit(ctx, fn, bytes); };
    // one source per sweep: the latency sweep wherever a batch of one would take it (94 against 67 us per sweep at 690^2)
    ctx->pl.lat_sweep = nullptr;
    ctx->last_kernel = use_lat_kernel(ctx, 1) ? 4 : 1;  // (vhp_last_sweep_kernel after a solve: what swept its iterations)
    if (use_lat_kernel(ctx, 1))
      ctx->pl.lat_sweep = [ctx](const int32_t* pivots, const int* nb, const int* done, const int* rec, double* out, bool dark_unwritten) {
        ctx->lat_src_index = nb;
        ctx->lat_skip = done;
        ctx->lat_pivot_rec = rec;
        ctx->lat_dark_unwritten = dark_unwritten;
        // (no per-launch event pairs inside a planner loop: vhp_timing times sweep launches, and the loop enqueues a batch of
        // launches past its end that return at once -- they would fill the pool with pairs that time nothing)
        const bool timing = ctx->timing;
        ctx->timing = false;
        const hipError_t e = launch_batch_sweep<double>(ctx, pivots, 1, out, true);
        ctx->timing = timing;
        ctx->lat_src_index = ctx->lat_skip = ctx->lat_pivot_rec = nullptr;
        ctx->lat_dark_unwritten = false;
        return e;
      };
    // ... or the iteration as ONE launch (vhp_lat.hip vhp_planner_iteration) -- built in round 6, bit-exact, and SLOWER on this part: the
    // epilogue's workgroups sit behind other L2s than the sweep's, so the hand-off inside a launch costs an L2 write-back and an
    // invalidate (39.6 us per pivot on maze_6 against 22.8 with two launches; 32.6 with the local fields in uncached memory and no
    // fences, VHP_PLANNER_UNCACHED: profiles/r06_planner_one_kernel_ab.txt).  Only with VHP_PLANNER_ONE_KERNEL set in the environment.
    ctx->pl.lat_iteration = nullptr;
    static const bool one_kernel = std::getenv("VHP_PLANNER_ONE_KERNEL") != nullptr;
    if (ctx->pl.lat_sweep && one_kernel)
      ctx->pl.lat_iteration = [ctx](const vhp::PlannerDev& d) {
        ctx->lat_pivot_rec = d.rec;
        ctx->lat_dark_unwritten = true;
        ctx->planner_dev = &d;
        const bool timing = ctx->timing;
        ctx->timing = false;
        const hipError_t e = launch_batch_sweep<double>(ctx, d.pivots, 1, d.vis_local, true);
        ctx->timing = timing;
        ctx->planner_dev = nullptr;
        ctx->lat_pivot_rec = nullptr;
        ctx->lat_dark_unwritten = false;
        return e;
      };
  }
  int rc = vhp::planner_solve(ctx->pl, pm, ctx->d_occ, ctx->stream, ctx->ev0, ctx->ev1, start_x, start_y, end_x,
                              end_y, threshold, max_iter, came_from, vis_global, vis_local, pivots_xy, n_pivots, &msg);
  ctx->timed = true;
  if (rc != VHP_OK) ctx->err = msg;
  return rc;
}

int vhp_planner_solve_speculative(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold, uint64_t max_iter, int k,
                                  int mode, uint64_t* came_from, double* vis_global, double* vis_local, int32_t* pivots_xy,
                                  uint32_t* n_pivots, int32_t* stats) {
  if (!ctx) return VHP_ERR_ARG;
  if (!ctx->d_rows) return fail(ctx, VHP_ERR_NO_MAP, "vhp_planner_solve_speculative: no map set");
  VHP_ON_DEVICE(ctx);
  std::string msg;
  vhp::DevMap pm = dev_map(ctx);
  {
    int R, W;
    bool multi;
    pick_shape(ctx, std::max(ctx->nx, ctx->ny), &R, &W, &multi);
    hipError_t eb = vhp::attach_round_scratch(pm, W * 64 * R, (size_t)4 * vhp::kSpecMaxK, &ctx->d_bnd, &ctx->d_bnd_cap);
    if (eb != hipSuccess) return fail(ctx, VHP_ERR_HIP, std::string("scratch: ") + hipGetErrorString(eb));
    ctx->pl.R = R;
    ctx->pl.W = W;
    ctx->pl.multi = multi;
    ctx->pl.raise_lds = [ctx](const void* fn, size_t bytes) { return raise_lds_limit(ctx, fn, bytes); };
    // k sources per launch: the latency sweep (8 k workgroups) wherever a batch of k would take it
    ctx->pl.lat_sweep_k = nullptr;
    ctx->last_kernel = use_lat_kernel(ctx, k) ? 4 : 1;
    if (use_lat_kernel(ctx, k))
      ctx->pl.lat_sweep_k = [ctx](const int32_t* cand, int n, const int* slot_base, const int* run_if, const int* done, double* cache, bool dark_unwritten) {
        ctx->lat_skip = done;
        ctx->lat_slot_base = slot_base;
        ctx->lat_run_if = run_if;
        ctx->lat_dark_unwritten = dark_unwritten;
        const long long stride = ctx->opt_field_stride;
        ctx->opt_field_stride = 0;  // (the cache holds packed fields)
        const bool timing = ctx->timing;
        ctx->timing = false;  // (as in vhp_planner_solve's loop)
        const hipError_t e = launch_batch_sweep<double>(ctx, cand, n, cache, true);
        ctx->timing = timing;
        ctx->opt_field_stride = stride;
        ctx->lat_skip = ctx->lat_slot_base = ctx->lat_run_if = nullptr;
        ctx->lat_dark_unwritten = false;
        return e;
      };
  }
  int st[3] = {0, 0, 0};
  int rc = vhp::planner_solve_speculative(ctx->pl, ctx->spec, pm, ctx->d_occ, ctx->stream, ctx->ev0, ctx->ev1, start_x, start_y, end_x, end_y,
                                          threshold, max_iter, k, mode, came_from, vis_global, vis_local, pivots_xy, n_pivots, st, &msg);
  if (stats) { stats[0] = st[0]; stats[1] = st[1]; stats[2] = st[2]; }
  ctx->timed = true;
  if (rc != VHP_OK) ctx->err = msg;
  return rc;
}

int vhp_planner_solve_device(vhp_ctx* ctx, int start_x, int start_y, int end_x, int end_y, double threshold, uint64_t max_iter,
                             uint32_t* n_pivots) {
  return vhp_planner_solve(ctx, start_x, start_y, end_x, end_y, threshold, max_iter, nullptr, nullptr, nullptr, nullptr, n_pivots);
}

int vhp_planner_results_device(vhp_ctx* ctx, const uint32_t** labels, const double** vis_global, const double** vis_local,
                               const int32_t** pivots_xy) {
  if (!ctx) return VHP_ERR_ARG;
  if (!ctx->pl.vis_global) return fail(ctx, VHP_ERR_ARG, "vhp_planner_results_device: no planner solve has run on this map");
  if (labels) *labels = ctx->pl.label;
  if (vis_global) *vis_global = ctx->pl.vis_global;
  if (vis_local) *vis_local = ctx->pl.vis_local_out ? ctx->pl.vis_local_out : ctx->pl.vis_local;
  if (pivots_xy) *pivots_xy = ctx->pl.pivots;
  return VHP_OK;
}

// eval_d of visibilityBasedSolver.h:112-115 (host side, used only for the path length)
static inline double eval_d_host(int ax, int ay, int bx, int by) {
  return std::sqrt((double)(ax - bx) * (ax - bx) + (ay - by) * (ay - by));
}

int vhp_reconstruct_path(const uint64_t* came_from, const int32_t* pivots_xy, uint32_t n_pivots, int nx, int ny, int end_x,
                         int end_y, int32_t* path_xy, uint32_t cap, uint32_t* n_path, double* length) {
  if (!came_from || !pivots_xy || nx <= 0 || ny <= 0) return VHP_ERR_ARG;
  if (end_x < 0 || end_y < 0 || end_x >= nx || end_y >= ny) return VHP_ERR_END_OOB;
  // walk labels back to the start: the label of a pivot's own cell is the pivot that
  // lit it, the start labels itself, so the walk stops when the label repeats.  Labels index
  // pivots_xy[0 .. n_pivots]; a consistent table needs at most n_pivots + 1 hops.
  std::vector<std::pair<int, int>> rev;
  int x = end_x, y = end_y;
  uint64_t t = came_from[(size_t)x + (size_t)y * nx];
  uint64_t t_old = std::numeric_limits<uint64_t>::max();
  while (t != t_old) {
    rev.push_back({x, y});
    t_old = t;
    if (t > n_pivots) return VHP_ERR_ARG;  // unlabelled cell (VHP_UNLABELLED) or a label outside the pivot list
    if (rev.size() > (size_t)n_pivots + 2) return VHP_ERR_ARG;  // the labels form a cycle: not a planner result
    x = pivots_xy[2 * t];
    y = pivots_xy[2 * t + 1];
    if (x < 0 || y < 0 || x >= nx || y >= ny) return VHP_ERR_ARG;
    t = came_from[(size_t)x + (size_t)y * nx];
  }
  rev.push_back({x, y});
  std::reverse(rev.begin(), rev.end());
  double total = 0.0;
  for (size_t k = 0; k + 1 < rev.size(); ++k)
    total += eval_d_host(rev[k].first, rev[k].second, rev[k + 1].first, rev[k + 1].second);
  if (n_path) *n_path = (uint32_t)rev.size();  // the size needed, also when it exceeds cap
  if (length) *length = total;
  if (path_xy) {
    if (rev.size() > cap) return VHP_ERR_TOO_LARGE;  // nothing written; *n_path says how many points there are
    for (size_t k = 0; k < rev.size(); ++k) {
      path_xy[2 * k] = rev[k].first;
      path_xy[2 * k + 1] = rev[k].second;
    }
  }
  return VHP_OK;
}

}  // extern "C"
