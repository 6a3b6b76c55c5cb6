// vhp_sweep.hip.h -- CDNA4 (gfx950) device code of the visibility-transport sweep.
//
// Replaces the four loop nests of computeVisibility()
// (reference src/visibilityBasedSolver.cpp:570-696) for a batch of sources, and --
// through the Emit policy -- the same nests inside updateVisibility() (:379-565).
//
// Decomposition (tests/schedule_model.py is the executable statement of it and is
// checked bit-for-bit against the oracle):
//   source -> 4 quadrants, one workgroup each; a quadrant = an x-major octant
//   (|dx| > |dy|, plus the diagonal) and a y-major octant (|dy| > |dx|).  Inside an
//   octant a cell depends only on the previous "front" (previous column for x-major,
//   previous row for y-major): on itself and on its neighbour one row/column below,
//       v = (a - c*(a - b)) * occ,   a = own previous, b = lower neighbour's previous,
//   so a front is swept by lanes.  A lane owns R consecutive rows (x-major) or
//   columns (y-major): the neighbour of its register r > 0 is its own register r-1
//   and only register 0 needs the lane below, one DPP wave shift.  Nothing but
//   registers sits on the dependent chain.
//   A workgroup is 2*W wavefronts: wavefronts 0..W-1 sweep strips of the x-major
//   octant, W..2W-1 strips of the y-major one (strip = 64*R rows/columns).  Strip p
//   runs one pipeline slot (a 16-aligned chunk of the marching coordinate, one
//   workgroup barrier) behind strip p-1, which hands it its boundary lane through a
//   small LDS ring: the LDS-staged front.
//
//   x-major: a lane produces consecutive x of its rows, i.e. the wavefront produces a
//   column per step.  Values are staged in a wave-private LDS tile and emitted as
//   64-byte row segments with 16-byte stores (coalesced, sector aligned).
//   y-major: a wavefront produces 64*R consecutive x of one row per step and emits
//   them directly, 16 bytes per lane.
//
//   The reference's stale diagonal (cell (k,k) = cell (k,k-1) * occ) is
//   produced by the x-major strips (the row below hands its NEW value up); the
//   y-major strips need diag(k) as the seed of column k and recompute it from the
//   private two-term recurrence sub(k) = V(k,k-1), diag(k) = sub(k)*occ(k,k).
//
// Each strip has two code paths.  The generic step handles everything (triangular
// start-up where the diagonal lives inside the strip, partial windows, ragged edges).
// Once a strip is past its diagonal, full 8-aligned windows of the marching coordinate
// run the fast path: 8 fully unrolled steps with no branches, no loads and no scalar
// bookkeeping -- reciprocals arrive by one scalar load per window, occupancy bits come
// from a lane-private word that is pre-shifted once per window, the boundary lane of
// the strip below rides in a register that is rotated by DPP.  (A wavefront issues
// about one instruction per 4-5 cycles whatever its kind, so instruction count per
// step, scalar ones included, is what sets the latency of a front.)
//
// Memory: occupancy is read from two bit-packed copies of the map.  A lane keeps one
// 64-bit word per owned row/column, packed along the marching direction: 64 steps of
// occupancy per load, refilled once per 64 steps.
//
// Arithmetic is IEEE binary64 with contraction off.  The per-cell division
// c = j/i is replaced by Markstein's correction with a host-computed table of
// correctly rounded reciprocals: q = j*y; r = fma(-i,q,j); c = fma(r,y,q), which
// oracle/markstein_check.c proves bit-identical to j/i for all j < i <= 16384.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vhp_diag.h"

#include <algorithm>
#include <type_traits>

namespace vhp {

constexpr int kChunk = 8;        // marching-coordinate cells per pipeline slot (aligned): one 8-step window
constexpr int kRing = 32;        // entries of a boundary ring (>= 4*kChunk), indexed by marching coordinate & (kRing-1);
                                 // sized so that a 16-wavefront workgroup at 1000^2 needs < 80 KB of LDS (two per CU)
constexpr int kTileCols = 8;     // columns computed per window = 64 B of fp64
// Staging tile of an x-major strip: one row of `tile_stride(R)` doubles per strip row (odd: spreads the column
// writes over the LDS banks).  R >= 2: the 8 columns of the current window.  R == 1: a ring of 16 columns
// (x & 15), so that two consecutive windows can leave as one whole 128-byte line per row ("line mode", x_strip):
// HBM takes whole lines at 5.8 TB/s but lone 64-byte halves at 3.5.  Only the one-row-per-lane shape has the LDS
// for it: at R = 2 a 16-wavefront workgroup would need 139 KB.
__host__ __device__ constexpr int tile_stride(int R) { return R == 1 ? 17 : 9; }
__host__ __device__ constexpr int tile_cols(int R) { return R == 1 ? 16 : 8; }
constexpr int kUnitsPerSource = 4;
constexpr int kYLag = 2;         // y-major strips run this many slots behind the x-major ones (they consume diag(k))
constexpr int kStage = 128;      // LDS staging of a boundary row that arrives from the previous round (per octant)
constexpr int kRecipPad = 8;     // the reciprocal table is readable 8 entries past max(nx,ny)

struct DevMap {
  const uint64_t* rows;  // bit x&63 of rows[y*wpr + 1 + (x>>6)] = occ(x,y); word 0 and the last word of a line are zero pads
  const uint64_t* cols;  // bit y&63 of cols[x*wpc + 1 + (y>>6)] = occ(x,y)
  const double* recip;   // recip[k] = RN(1/k), k = 1..max(nx,ny); recip[0] = 0
  int wpr, wpc;
  int nx, ny;
  // multi-round sweeps (fronts longer than one workgroup's W*64*R rows): per workgroup
  // 4 arrays of `bnd_len` doubles (x-major / y-major boundary rows, double-buffered by round)
  double* bnd;
  int bnd_len;
  int slide;  // 1: slide the y-major column grid onto 128-byte lines (y_grid_slide)
};

// the reciprocal table never changes during a launch: wave-uniform reads through the
// constant address space become scalar loads
typedef const __attribute__((address_space(4))) double* crecip_p;

// lane l <- lane l-1, lane 0 keeps `fill`'s lane 0.  DPP wave_shr:1 (gfx9 encoding
// 0x138); with bound_ctrl off the lane without a source keeps `old`.
__device__ __forceinline__ double shift_up(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  const int flo = __double2loint(fill), fhi = __double2hiint(fill);
  lo = __builtin_amdgcn_update_dpp(flo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// lane l <- lane l+1 (lane 63 <- lane 0): DPP wave_rol:1 (0x134)
__device__ __forceinline__ double rotate_down(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// wave-uniform read of lane `l` (l uniform)
__device__ __forceinline__ double read_lane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// RN(num/den) for integers 0 <= num < den <= 16384, given rden = RN(1/den)
__device__ __forceinline__ double ratio(double num, double den, double rden) {
  const double q = num * rden;
  const double r = __builtin_fma(-den, q, num);
  return __builtin_fma(r, rden, q);
}

// the reference's update (solver.cpp:592-594 / 598-600): a - c*(a - b), no contraction
__device__ __forceinline__ double stencil(double a, double b, double c) {
  const double t = a - b;
  const double u = c * t;
  return a - u;
}

// v * occ for occ in {0,1} and finite v >= 0 (solver.cpp:602): AND with 0 / ~0
__device__ __forceinline__ double and_mask(double v, int msk) {
  return __hiloint2double(__double2hiint(v) & msk, __double2loint(v) & msk);
}
// bit `b` (uniform, 0..63) of a lane-private word as 0 / ~0
__device__ __forceinline__ int bit_mask(uint64_t w, int b) {
  const uint32_t half = (b & 32) ? (uint32_t)(w >> 32) : (uint32_t)w;
  return __builtin_amdgcn_sbfe(half, b & 31, 1);
}

// Makes a just-loaded value count as "used here": the compiler then waits for the load at
// this point instead of at the first real use.
__device__ __forceinline__ void pin_loaded(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_loaded(uint64_t& v) { asm volatile("" : "+v"(v)); }

// Load of a value that another wavefront (same workgroup, earlier round) stored to global
// memory: agent-scope atomic load = sc1, served from L2 rather than this CU's L1.
__device__ __forceinline__ double load_shared_f64(const double* p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}

struct UnitGeom {
  int sx, sy, ni, nj;
  int ya;  // y-major strips own columns [p*S - ya, (p+1)*S - ya): the grid is slid so that a strip's row
           // segment starts on a 128-byte line of the output (partial 64-byte sectors cost a read-modify-write)
};

// Slide of the y-major column grid for a quadrant (0 .. 15), 0 if the slid grid would need one strip more
// than `max_strips`.
template <int DX>
__device__ __forceinline__ int y_grid_slide(int sx, int cols_total, int S, int max_strips) {
  // DX > 0: strip p starts at x = sx + p*S - ya;  DX < 0: its lowest x is sx - (p+1)*S + 1 + ya
  const int a = DX > 0 ? (sx & 15) : ((-(sx + 1)) & 15);
  return (cols_total + a + S - 1) / S <= max_strips ? a : 0;
}

// A workgroup sweeps one quadrant with all its wavefronts, or -- packing -- several short
// quadrants side by side, each with an equal share ("subgroup") of the wavefronts and of the LDS.
// All subgroups execute the same number of barriers (`tmax_floor` = the longest partner's).
struct SubGroup {
  int W;           // strips per octant available to this subgroup
  int wave;        // wavefront index inside the subgroup, 0 .. 2W-1
  int tid;         // thread index inside the subgroup
  int nthreads;    // threads of the subgroup
  int tmax_floor;  // run at least this many + 1 pipeline slots
};

// Where a strip sits in the current round of a quadrant sweep.
struct StripSlot {
  int pg;      // global strip index: rows/columns [pg*S, (pg+1)*S)
  int w;       // wavefront slot inside the round: pipeline position and ring index
  int nbase;   // chunk sequence number the round starts at (slot T works on chunk T - w + nbase)
  int tmax;    // last slot of the round (uniform for the workgroup)
  bool tail;   // last wavefront of a round that is followed by another: publish the boundary row to global memory
  const double* bnd_in;  // boundary row from the previous round (w == 0, pg > 0), else unused
  double* bnd_out;       // boundary row for the next round (tail)
  double* stage;         // kStage doubles of LDS for bnd_in
  double* dummy;         // 8 doubles of LDS per wavefront: where the lanes that are not the boundary lane "write" theirs
};

// Pipeline chunks are kChunk-aligned in the marching coordinate mc = s + DIR*step.
constexpr int kChunkShift = 3;  // log2(kChunk)
template <int DIR>
__device__ __forceinline__ int chunk_seq(int s, int step) {
  const int mc = s + DIR * step;
  return DIR > 0 ? (mc >> kChunkShift) - (s >> kChunkShift) : (s >> kChunkShift) - (mc >> kChunkShift);
}
template <int DIR>
__device__ __forceinline__ void chunk_steps(int s, int n, int* lo, int* hi) {
  const int c = DIR > 0 ? (s >> kChunkShift) + n : (s >> kChunkShift) - n;
  if (DIR > 0) { *lo = kChunk * c - s; *hi = kChunk * c + kChunk - 1 - s; }
  else { *lo = s - (kChunk * c + kChunk - 1); *hi = s - kChunk * c; }
}

// Emit policy of the plain sweep: store cells.  A "pair" is cells (x, y) and (x+1, y).
// (Raw buffer stores with out-of-range offsets as predication were measured slower than
// exec-masked global stores here: every store instruction costs the issuing wavefront
// ~60-80 cycles whether or not its lanes are dropped.)
// NT: non-temporal stores (the nt bit).  The POOL sweep's field stores carry it (vhp_lanes.hpp VHP_FIELD_STORE: its x-major strips
// flush whole 128-byte lines, and on most of the device's memory a plain store that covers part of a line costs several lines'
// worth: DESIGN.md appendix A.7).  THIS kernel's do not, since round 5: it flushes 64-byte row segments, which a wavefront's L2
// merges into lines when they are plain stores, and it runs the small and the unaligned grids, where rows are short and segments
// straddle sectors.  Measured, nt / plain, ms (profiles/r05_front_sweep_nt_vs_plain.txt): 101^2 x 4096 sources 0.368 / 0.192,
// 250^2 x 256 0.121 / 0.081, 250^2 x 1024 0.426 / 0.245, 398^2 x 192 0.209 / 0.133, 128^2 x 4096 0.307 / 0.252, 384^2 x 256 0.156 /
// 0.135, 512^2 x 256 0.242 / 0.222, 1002^2 x 40 0.337 / 0.295; level (+-1 %) at 248^2 x 256, 640^2 x 48, 768^2 x 48, 1000^2 x 40.
// (Round 4 had set the bit for every batch launch on the strength of 1000^2 x 256 -- 0.766 -> 0.747 ms --, a launch the pool sweep
// takes.)  -DVHP_FRONT_NT=true builds the round-4 behaviour.
template <typename OutT, bool MULTI = false, bool NT = false>
struct StoreEmit {
  static constexpr int kCellBytes = sizeof(OutT);
  static constexpr bool kFastPath = true;  // use the unrolled windows
  static constexpr bool kMulti = MULTI;    // fronts may be longer than one round of W strips
  OutT* __restrict__ out;
  int nx;
  // A pair is stored with ONE instruction wherever it lies: on an odd pitch, or in a field that starts off the 16-byte grid, every
  // other pair is aligned to its cells only.  The type says so (aligned to one cell), which keeps the access defined; gfx950 serves
  // global accesses at any 4-byte alignment (the unaligned access mode ROCm runs the device in), so it stays one global_store.
  typedef OutT Two __attribute__((ext_vector_type(2), aligned(sizeof(OutT))));
  static __device__ __forceinline__ void put(OutT* p, OutT v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
  // (not a template over the pointer type: a template argument drops the typedef's alignment)
  static __device__ __forceinline__ void put2(char* at, OutT a, OutT b) {
    Two* p = reinterpret_cast<Two*>(at);
    const Two v = {a, b};
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
  }
  __device__ __forceinline__ StoreEmit(OutT* field, int nx_, int) : out(field), nx(nx_) {}
  // both cells valid; off = (y*nx + x) * kCellBytes, maintained incrementally by the caller
  __device__ __forceinline__ void pair_at(uint32_t off, int, int, double v0, double v1) {
    VHP_DIAG_FRONT_STORE_GUARD
    put2(reinterpret_cast<char*>(out) + off, static_cast<OutT>(v0), static_cast<OutT>(v1));
  }
  // `both`: store the pair; else `one`: store only the cell at off + sel*kCellBytes (value vs)
  __device__ __forceinline__ void pair_or_single_at(uint32_t off, int, int, double v0, double v1, bool both, bool one,
                                                    int sel, double vs) {
    asm volatile("" : "+v"(vs));  // keep the compiler from splitting the 16-byte store to share a half with the single
    VHP_DIAG_FRONT_STORE_GUARD
    if (both)
      put2(reinterpret_cast<char*>(out) + off, static_cast<OutT>(v0), static_cast<OutT>(v1));
    else if (one)
      put(reinterpret_cast<OutT*>(reinterpret_cast<char*>(out) + off + (uint32_t)(sel * kCellBytes)), static_cast<OutT>(vs));
  }
  // one cell at byte offset off
  __device__ __forceinline__ void single_at(uint32_t off, int, int, double v) {
    VHP_DIAG_FRONT_STORE_GUARD
    put(reinterpret_cast<OutT*>(reinterpret_cast<char*>(out) + off), static_cast<OutT>(v));
  }
  __device__ __forceinline__ void pair(int x, int y, double v0, double v1, bool ok0, bool ok1) {
    VHP_DIAG_FRONT_STORE_GUARD
    OutT* p = out + (size_t)y * nx + x;
    if (ok0) put(p, static_cast<OutT>(v0));
    if (ok1) put(p + 1, static_cast<OutT>(v1));
  }
  __device__ __forceinline__ void zero(int x, int y) { out[(size_t)y * nx + x] = OutT(0); }
};

// ---------------------------------------------------------------------------
// x-major strip: rows j = j0 + R*lane + r, steps i = j0 .. ni-1, cells (i, j), i >= j.
// x = sx + DX*i, y = sy + DY*j.
// ---------------------------------------------------------------------------
template <int R, int DX, int DY, typename Emit>
__device__ __forceinline__ void x_strip(const DevMap& m, Emit& emit_, const UnitGeom g, const StripSlot ss, double* ring_base,
                                        double* tile, double* diag_ring) {
  Emit& emit = emit_;
  const int p = ss.pg, tmax = ss.tmax;
  constexpr int S = 64 * R;
  constexpr int CB = Emit::kCellBytes;
  const int lane = threadIdx.x & 63;
  const int rows_total = min(g.nj, g.ni);
  const int P = (rows_total + S - 1) / S;
  const int j0 = p * S;
  const bool strip_on = p < P;
  const bool has_consumer = p + 1 < P;
  const int nlast = chunk_seq<DX>(g.sx, g.ni - 1);
  const int nfirst = chunk_seq<DX>(g.sx, min(j0, g.ni - 1));
  double* ring_out = ring_base + ss.w * kRing;
  // lane 0's lower neighbour: the LDS ring of the wavefront below, or (first wavefront of a later
  // round) the staged copy of the boundary row the previous round left in global memory
  constexpr bool kMulti = Emit::kMulti;  // sweeping in rounds is compiled in only where the launch shape needs it
  const bool from_prev_round = kMulti && ss.w == 0 && p > 0;
  const double* ring_in = from_prev_round ? ss.stage : ring_base + (ss.w > 0 ? ss.w - 1 : 0) * kRing;
  const int rin_mask = from_prev_round ? kStage - 1 : kRing - 1;
  const int rows_here = max(min(S, rows_total - j0), 0);

  double prev[R], jd[R];
  int dmask[R];
  uint64_t ow[R];   // occupancy word of the current 64-block of x, one per owned row
  double rv = 0.0;  // generic path: lane t holds 1/i of the step whose x is (block, t)
  int cur_blk = INT32_MIN;
  bool first_refill = true;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = j0 + R * lane + r;
    const bool on = strip_on && j < rows_total;
    const int y = on ? g.sy + DY * j : g.sy;
    const int xd = on ? g.sx + DX * j : g.sx;  // x of this row's diagonal cell
    prev[r] = 0.0;
    jd[r] = (double)j;
    ow[r] = 0;
    dmask[r] = 0;
    if (strip_on) dmask[r] = ((m.rows[(size_t)y * m.wpr + 1 + (xd >> 6)] >> (xd & 63)) & 1ull) ? -1 : 0;
  }
  constexpr int kTileStride = tile_stride(R);
  constexpr int kRingCols = tile_cols(R);
  double* tile_lane = tile + R * lane * kTileStride;
  // Line mode (R == 1, fp64 output, pitch a multiple of 64 B): in the steady windows the rows whose 128-byte line
  // is only half computed are held back in the tile and leave one window later as a whole line.
  // (y*nx + x)/8 even <=> chunk x>>3 is the lower half of its line.
  constexpr bool kLineCapable = kRingCols == 16;  // fp32 output: the 16 columns are one whole 64-byte sector
  const bool line_mode = kLineCapable && (m.nx & 7) == 0;
  const bool rows_alternate = ((m.nx >> 3) & 1) != 0;  // consecutive rows start half a line apart
  int held_chunk = INT32_MIN;                           // chunk whose hold-class rows are still in the tile
  // flush geometry: lane <-> (row-in-group = lane>>2, column pair = lane&3), 16 rows x 64 B per pass
  // (cp, rsub and the lane's byte offset are derived from the lane id inside each flush, behind an opaque move: held in
  // registers across the step loops they pushed three values into scratch, and a scratch reload waits on vmcnt -- for
  // every global store the wavefront has in flight)
  const uint32_t flush_pass_stride = (uint32_t)(16 * DY * m.nx * CB);

  auto refill = [&](int blk) {  // blocking: once per 64 steps
    cur_blk = blk;
    const int xt = blk * 64 + lane;
    const int it = DX > 0 ? xt - g.sx : g.sx - xt;
    rv = (it >= 0 && it < g.ni) ? m.recip[it] : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = j0 + R * lane + r;
      const int y = (strip_on && j < rows_total) ? g.sy + DY * j : g.sy;
      ow[r] = m.rows[(size_t)y * m.wpr + 1 + blk];
    }
    // pin the s_waitcnt vmcnt for these loads INSIDE this (rare) block: left to the compiler it
    // lands at the first use in the common path, where it would also drain every store in flight
    pin_loaded(rv);
#pragma unroll
    for (int r = 0; r < R; ++r) pin_loaded(ow[r]);
    if (from_prev_round) {
      // stage this 64-block of the previous round's boundary row (and, the first time, the block
      // before it in marching order, which holds step j0-1) into LDS, indexed like the rings
      for (int b = first_refill ? blk - DX : blk; DX > 0 ? b <= blk : b >= blk; b += DX) {
        const int xs = b * 64 + lane;
        const int is = DX > 0 ? xs - g.sx : g.sx - xs;
        double bv = 0.0;
        if (is >= 0 && is < g.ni) bv = load_shared_f64(ss.bnd_in + is);
        ss.stage[xs & (kStage - 1)] = bv;
      }
      first_refill = false;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  // ---- line mode helpers ---------------------------------------------------------------
  // The rows of this strip for which chunk c (= x >> 3) is the half of their line that is computed first
  // (`first_half`) or second: local rows rb, rb + rs, ...  Marching up in x the lower half comes first.
  auto class_first_row = [&](int c, bool first_half) -> int {  // -1: no row of the strip is in the class
    const int want = (first_half == (DX > 0)) ? 0 : 1;         // 0: c is the lower half of the row's line
    if (rows_alternate) return (want ^ c ^ g.sy ^ j0) & 1;     // the parity of the row's y decides
    return ((c & 1) == want) ? 0 : -1;
  };
  // bytes [64*h, 64*h + 64*nh) of the 128-byte line that starts at chunk clo, for the class rows rb, rb+rs, ...:
  // 8 rows per store instruction, lanes outside the byte range idle
  auto flush_rows = [&](int clo, int rb, int h, int nh) {
    if (rb < 0) return;
    int l = lane;
    asm volatile("" : "+v"(l));  // keep this geometry out of the step loops' registers
    const int q8 = l & 7, mloc = l >> 3;
    const int rs = rows_alternate ? 2 : 1;
    const int xl = 8 * clo + 2 * q8;
    const bool on = (q8 >> 2) >= h && (q8 >> 2) < h + nh;
    const int jr0 = rb + rs * mloc;
    const double* q = tile + (xl & (kRingCols - 1)) + jr0 * kTileStride;
    uint32_t off = (uint32_t)(((g.sy + DY * (j0 + jr0)) * m.nx + xl) * CB);
    const uint32_t qstep = (uint32_t)(rs * 8 * kTileStride), ostep = (uint32_t)(rs * 8 * DY * m.nx * CB);
    const int passes = rows_alternate ? 4 : 8;
    for (int pass = 0; pass < passes; pass += 2) {
      const double a0 = q[0], b0 = q[1], a1 = q[qstep], b1 = q[qstep + 1];
      if (on) {
        emit.pair_at(off, xl, 0, a0, b0);
        emit.pair_at(off + ostep, xl, 0, a1, b1);
      }
      q += 2 * qstep;
      off += 2 * ostep;
    }
  };
  // leaving the steady windows: the rows still waiting for the second half of their line go out as they are
  auto drain_held = [&]() {
    if (kLineCapable && held_chunk != INT32_MIN) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      flush_rows(DX > 0 ? held_chunk : held_chunk - 1, class_first_row(held_chunk, true), DX > 0 ? 0 : 1, 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      held_chunk = INT32_MIN;
    }
  };

  // ---- generic step: any i, diagonal handling, predicated flush ----------------
  auto slow_step = [&](int i) {
    drain_held();
    const int x = g.sx + DX * i;
    const int blk = x >> 6, t = x & 63;
    if (blk != cur_blk) refill(blk);
    const double di = (double)i;
    const double ri = read_lane(rv, t);
    double fill = 0.0;  // OLD value of the row just below lane 0's first row
    double dsrc = 1.0;  // NEW value of that row (feeds the diagonal cell); 1.0 = light strength at the origin
    if (p > 0) {
      fill = ring_in[(x - DX) & rin_mask];
      dsrc = ring_in[x & rin_mask];
    }
    double v[R];
    {
      const double b0 = shift_up(prev[R - 1], fill);
      v[0] = and_mask(stencil(prev[0], b0, ratio(jd[0], di, ri)), bit_mask(ow[0], t));
    }
#pragma unroll
    for (int r = 1; r < R; ++r) v[r] = and_mask(stencil(prev[r], prev[r - 1], ratio(jd[r], di, ri)), bit_mask(ow[r], t));
    if (i < j0 + S && i < rows_total) {
      // the diagonal cell (i,i) is one of this strip's rows: it inherits the NEW value of
      // the row below it times its own occupancy
      const int k = i - j0;
      const int ld = k / R, rd = k - ld * R;
      const double up = shift_up(v[R - 1], dsrc);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (rd == r) {
          const double below = (r == 0) ? up : v[r > 0 ? r - 1 : 0];
          if (lane == ld) {
            v[r] = and_mask(below, dmask[r]);
            diag_ring[i & (kRing - 1)] = v[r];  // the y-major strips seed column i with it two slots later
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      prev[r] = v[r];
      tile_lane[r * kTileStride + (x & (kRingCols - 1))] = v[r];
    }
    if (has_consumer && lane == 63) {
      ring_out[x & (kRing - 1)] = v[R - 1];
      if (kMulti && ss.tail) ss.bnd_out[i] = v[R - 1];
    }

    const bool endwin = DX > 0 ? ((x & (kTileCols - 1)) == kTileCols - 1) : ((x & (kTileCols - 1)) == 0);
    if (endwin || i == g.ni - 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int xbase = x & ~(kTileCols - 1);
      int lf = lane;
      asm volatile("" : "+v"(lf));
      const int cp = lf & 3, rsub = lf >> 2;
      const int xc = xbase + 2 * cp;
      const int ic0 = DX > 0 ? xc - g.sx : g.sx - xc;
      const int ic1 = DX > 0 ? ic0 + 1 : ic0 - 1;
      const bool c0 = ic0 >= 0 && ic0 <= i, c1 = ic1 >= 0 && ic1 <= i;
      const int rows_live = min(rows_here, i - j0 + 1);  // rows j <= i
      for (int rb = 0; rb < rows_live; rb += 32) {
        double ta[2], tb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const double* q = tile + (rb + 16 * u + rsub) * kTileStride + ((xbase + 2 * cp) & (kRingCols - 1));
          ta[u] = q[0];
          tb[u] = q[1];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int rl = rb + 16 * u + rsub;
          const int j = j0 + rl;
          const bool rowok = rl < rows_here;
          emit.pair(xc, g.sy + DY * j, ta[u], tb[u], rowok && c0 && j <= ic0, rowok && c1 && j <= ic1);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  // ---- fast window: 8 steps i..i+7 covering one 8-aligned window of x, strip past its
  // diagonal (i >= j0 + S): every row active, every staged cell valid -------------------
  const int jbase = j0 + R * lane;  // row index of register 0, for the "am I this step's diagonal cell" test
  auto fast_window = [&](int i, auto diag_tag, auto prod_tag, auto par_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;  // the strip's diagonal may fall inside this window
    constexpr bool PROD = decltype(prod_tag)::value;  // a strip below feeds lane 0 (p > 0)
    constexpr int PAR = decltype(par_tag)::value;     // R == 2: parity of (i - j0), i.e. which register holds step i's diagonal row
    const int x0 = g.sx + DX * i;             // x of the first step
    const int xb = x0 & ~(kTileCols - 1);     // lowest x of the window
    const int blk = x0 >> 6, t0 = x0 & 63;
    if (blk != cur_blk) refill(blk);
    // eight reciprocals: lane reads of the block's table (measured faster than one scalar load per
    // window, whose latency sits exposed at the top of the window)
    double rr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rr[k] = read_lane(rv, t0 + DX * k);
    // occupancy bits of the window, pre-shifted so that step k's bit sits at a fixed position
    int hs[R];
    const int sh = DX > 0 ? (t0 & 31) : (t0 & 31) - 7;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t half = (t0 & 32) ? (uint32_t)(ow[r] >> 32) : (uint32_t)ow[r];
      hs[r] = (int)(half >> sh);
    }
    // boundary row of the strip below: lane t holds its value at step i-1+t
    double ringv = 0.0;
    if (PROD) ringv = ring_in[(x0 - DX + DX * lane) & rin_mask];
    double* ring_w = ring_out + (xb & (kRing - 1));
    // every lane writes "its boundary value" each step -- lane 63 into the ring, the others into a dummy slot:
    // one ds_write instead of an exec-masked region per step
    double* ring_wl = lane == 63 ? ring_w : ss.dummy;
    double* tile_win = tile_lane + (xb & (kRingCols - 1));  // R == 1: the window's half of the 16-column ring
    double di = (double)i;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int col = DX > 0 ? k : 7 - k;  // x & 7 of this step, compile time
      double v[R];
      {
        const double b0 = shift_up(prev[R - 1], ringv);
        v[0] = and_mask(stencil(prev[0], b0, ratio(jd[0], di, rr[k])), __builtin_amdgcn_sbfe(hs[0], col, 1));
      }
#pragma unroll
      for (int r = 1; r < R; ++r)
        v[r] = and_mask(stencil(prev[r], prev[r - 1], ratio(jd[r], di, rr[k])), __builtin_amdgcn_sbfe(hs[r], col, 1));
      double ringn = 1.0;  // lane 0: NEW value of the row below this strip (1.0 = light strength at the origin)
      if (PROD) ringn = rotate_down(ringv);
      if (DIAG) {
        // the diagonal cell (i,i) inherits the NEW value of the row below it times its own
        // occupancy; it is row ji[r] of the lane for which ji[r] == i
        if constexpr (R == 2) {
          // which register holds the diagonal row of step i+k is known at compile time from the
          // parity tag; the lane that holds it is wave-uniform
          const int ld = (i + k - j0) >> 1;
          if (((PAR + k) & 1) == 0) {  // folds after unrolling: register 0 holds the diagonal row
            const double dcell = and_mask(shift_up(v[1], ringn), dmask[0]);
            if (lane == ld) {
              v[0] = dcell;
              diag_ring[(i + k) & (kRing - 1)] = dcell;  // the y-major strips seed column i+k with it
            }
          } else {
            const double dcell = and_mask(v[0], dmask[1]);
            if (lane == ld) {
              v[1] = dcell;
              diag_ring[(i + k) & (kRing - 1)] = dcell;
            }
          }
        } else {
          const double up = shift_up(v[R - 1], ringn);
          bool hit_any = false;
          double dval = 0.0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const double below = (r == 0) ? up : v[r > 0 ? r - 1 : 0];
            const bool hit = jbase + r == i + k;
            const double dcell = and_mask(below, dmask[r]);
            v[r] = hit ? dcell : v[r];
            dval = hit ? dcell : dval;
            hit_any |= hit;
          }
          if (hit_any) diag_ring[(i + k) & (kRing - 1)] = dval;  // the y-major strips seed column i+k with it
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        prev[r] = v[r];
        tile_win[r * kTileStride + col] = v[r];
      }
      ring_wl[col] = v[R - 1];
      if (PROD) ringv = ringn;
      di += 1.0;
    }
    // flush the whole window: S rows x 64 B
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (kMulti && ss.tail) {  // the next round's first strip reads this strip's boundary row from global memory
      if (lane < 8) ss.bnd_out[i + lane] = ring_w[DX > 0 ? lane : 7 - lane];
    }
    int lf = lane;
    asm volatile("" : "+v"(lf));
    const int cp = lf & 3, rsub = lf >> 2;
    const uint32_t off0 = (uint32_t)(((g.sy + DY * (j0 + rsub)) * m.nx + 2 * cp + xb) * CB);
    const int xc = xb + 2 * cp;
    const int y0 = g.sy + DY * (j0 + rsub);
    const double* q0 = tile + rsub * kTileStride + 2 * cp + (xb & (kRingCols - 1));
    if (DIAG) {
      // staged cell (column step ic, row j) is real iff j <= ic: below the diagonal both cells of
      // a pair are, on the diagonal only the later column's
      const int ic0 = DX > 0 ? xc - g.sx : g.sx - xc;  // step index of the pair's first cell
      const int icmin = DX > 0 ? ic0 : ic0 - 1;         // the later column is icmin + 1
      const int jrow = j0 + rsub;
      const int room = rows_here - 1 - rsub;            // 16*pass must not exceed this (ragged last strip)
      const int lim_both = min(icmin - jrow, room);
      const int at_single = (icmin + 1 - jrow <= room) ? icmin + 1 - jrow : -1;
      const int rows_live = min(rows_here, i + 8 - j0);  // rows j <= last step of the window
      const int rows_full = min(rows_here, i - j0 + 1) & ~15;  // whole passes of rows j <= first step: no predicate
#pragma unroll
      for (int pass = 0; pass < S / 16; pass += 2) {
        if (pass * 16 < rows_live) {
          double ta[2], tb[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            ta[u] = q0[(pass + u) * 16 * kTileStride];
            tb[u] = q0[(pass + u) * 16 * kTileStride + 1];
          }
          if ((pass + 2) * 16 <= rows_full) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
              emit.pair_at(off0 + (uint32_t)(pass + u) * flush_pass_stride, xc, y0 + DY * 16 * (pass + u), ta[u], tb[u]);
          } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int d = 16 * (pass + u);
              emit.pair_or_single_at(off0 + (uint32_t)(pass + u) * flush_pass_stride, xc, y0 + DY * d, ta[u], tb[u],
                                     d <= lim_both, d == at_single, DX > 0 ? 1 : 0, DX > 0 ? tb[u] : ta[u]);
            }
          }
        }
      }
    } else if (rows_here == S && line_mode) {
      // rows whose line this chunk completes leave whole; the others wait for the next window
      const int c = xb >> 3;
      const int rb = class_first_row(c, false);
      const int clo = DX > 0 ? c - 1 : c;  // lower chunk of the line that c completes
      if (held_chunk == c - DX) flush_rows(clo, rb, 0, 2);
      else flush_rows(clo, rb, DX > 0 ? 1 : 0, 1);  // first steady window: their first half left on its own
      held_chunk = c;
    } else if (rows_here == S) {
#pragma unroll
      for (int pass = 0; pass < S / 16; pass += 2) {
        double ta[2], tb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          ta[u] = q0[(pass + u) * 16 * kTileStride];
          tb[u] = q0[(pass + u) * 16 * kTileStride + 1];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          emit.pair_at(off0 + (uint32_t)(pass + u) * flush_pass_stride, xc, y0 + DY * 16 * (pass + u), ta[u], tb[u]);
      }
    } else {
      for (int pass = 0; pass * 16 < rows_here; ++pass) {
        const double a = q0[pass * 16 * kTileStride], b = q0[pass * 16 * kTileStride + 1];
        if (pass * 16 + rsub < rows_here) emit.pair_at(off0 + (uint32_t)pass * flush_pass_stride, xc, y0 + DY * 16 * pass, a, b);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  for (int T = 0; T <= tmax; ++T) {
    const int n = T - ss.w + ss.nbase;
    if (strip_on && n >= nfirst && n <= nlast) {
      int ilo, ihi;
      chunk_steps<DX>(g.sx, n, &ilo, &ihi);
      ilo = max(ilo, j0);
      ihi = min(ihi, g.ni - 1);
      int i = ilo;
      while (i <= ihi) {
        const int x = g.sx + DX * i;
        const bool aligned = DX > 0 ? (x & 7) == 0 : (x & 7) == 7;
        if (Emit::kFastPath && aligned && i + 7 <= ihi) {
          const bool steady = i >= j0 + S;
          using P0 = std::integral_constant<int, 0>;
          using P1 = std::integral_constant<int, 1>;
          const bool odd = ((i - j0) & 1) != 0;
          if (p > 0) {
            if (steady) fast_window(i, std::false_type(), std::true_type(), P0());
            else if (odd) fast_window(i, std::true_type(), std::true_type(), P1());
            else fast_window(i, std::true_type(), std::true_type(), P0());
          } else {
            if (steady) fast_window(i, std::false_type(), std::false_type(), P0());
            else if (odd) fast_window(i, std::true_type(), std::false_type(), P1());
            else fast_window(i, std::true_type(), std::false_type(), P0());
          }
          i += 8;
        } else {
          slow_step(i);
          i += 1;
        }
      }
    }
    __syncthreads();
  }
  drain_held();
}

// ---------------------------------------------------------------------------
// y-major strip: columns i = i0 + R*lane + r, steps j = i0 .. nj-1, cells (i, j), j > i.
// ---------------------------------------------------------------------------
template <int R, int DX, int DY, typename Emit>
__device__ __forceinline__ void y_strip(const DevMap& m, Emit& emit_, const UnitGeom g, const StripSlot ss, double* ring_base,
                                        const double* diag_ring) {
  Emit& emit = emit_;
  const int p = ss.pg, tmax = ss.tmax;
  constexpr int S = 64 * R;
  constexpr int CB = Emit::kCellBytes;
  const int lane = threadIdx.x & 63;
  const int cols_total = max(min(g.ni, g.nj - 1), 0);
  const int P = (cols_total + g.ya + S - 1) / S;
  const int i0 = p * S - g.ya;      // < 0 for strip 0 of a slid grid: its first ya columns do not exist
  const int jstart = max(i0, 0);    // first step of the strip
  const bool strip_on = p < P;
  const bool has_consumer = p + 1 < P;
  const int nlast = chunk_seq<DY>(g.sy, g.nj - 1);
  const int nfirst = chunk_seq<DY>(g.sy, min(jstart, g.nj - 1));
  double* ring_out = ring_base + ss.w * kRing;
  constexpr bool kMulti = Emit::kMulti;  // sweeping in rounds is compiled in only where the launch shape needs it
  const bool from_prev_round = kMulti && ss.w == 0 && p > 0;
  const double* ring_in = from_prev_round ? ss.stage : ring_base + (ss.w > 0 ? ss.w - 1 : 0) * kRing;
  const int rin_mask = from_prev_round ? kStage - 1 : kRing - 1;
  bool first_refill = true;

  double prev[R], id[R];
  uint64_t ow[R];
  double rv = 0.0;
  int cur_blk = INT32_MIN;
  // the lane's R columns are x-consecutive; xlo = the lowest x among them
  const int icol0 = i0 + R * lane;
  const int xlo = DX > 0 ? g.sx + icol0 : g.sx - icol0 - (R - 1);
  const bool all_cols = strip_on && i0 + S <= cols_total;  // every lane's every column < cols_total
  const bool head = i0 < 0;                                 // ... but columns icol < 0 are not real
  const bool edge_free = strip_on && i0 + S <= g.ni && i0 + S <= g.nj;  // the diagonal cells (k,k) of all this strip's columns exist
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = icol0 + r;
    prev[r] = 0.0;
    id[r] = (double)i;
    ow[r] = 0;
  }
  auto refill = [&](int blk) {
    cur_blk = blk;
    const int yt = blk * 64 + lane;
    const int jt = DY > 0 ? yt - g.sy : g.sy - yt;
    rv = (jt >= 0 && jt < g.nj) ? m.recip[jt] : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = icol0 + r;
      const int x = (strip_on && i >= 0 && i < g.ni) ? g.sx + DX * i : g.sx;
      ow[r] = m.cols[(size_t)x * m.wpc + 1 + blk];
    }
    pin_loaded(rv);
#pragma unroll
    for (int r = 0; r < R; ++r) pin_loaded(ow[r]);
    if (from_prev_round) {
      for (int b = first_refill ? blk - DY : blk; DY > 0 ? b <= blk : b >= blk; b += DY) {
        const int ys = b * 64 + lane;
        const int js = DY > 0 ? ys - g.sy : g.sy - ys;
        double bv = 0.0;
        if (js >= 0 && js < g.nj) bv = load_shared_f64(ss.bnd_in + js);
        ss.stage[ys & (kStage - 1)] = bv;
      }
      first_refill = false;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  auto slow_step = [&](int j) {
    const int y = g.sy + DY * j;
    const int blk = y >> 6, t = y & 63;
    if (blk != cur_blk) refill(blk);
    const double dj = (double)j;
    const double rj = read_lane(rv, t);
    // column j (if it is one of this strip's) is seeded with diag(j), which the x-major strip
    // owning row j published two pipeline slots ago
    const bool own_diag = j >= i0 && j < i0 + S && j < g.ni;
    double dg = 0.0;
    if (own_diag) dg = diag_ring[j & (kRing - 1)];
    double fill = 0.0;
    if (p > 0) fill = ring_in[(y - DY) & rin_mask];
    double v[R];
    {
      const double b0 = shift_up(prev[R - 1], fill);
      v[0] = and_mask(stencil(prev[0], b0, ratio(id[0], dj, rj)), bit_mask(ow[0], t));
    }
#pragma unroll
    for (int r = 1; r < R; ++r) v[r] = and_mask(stencil(prev[r], prev[r - 1], ratio(id[r], dj, rj)), bit_mask(ow[r], t));
    // emit the row segment: the lane's R columns are x-consecutive, pairs of 16 bytes
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      if (R == 1) {
        emit.pair(xlo, y, v[0], 0.0, icol0 >= 0 && icol0 < j && icol0 < cols_total, false);
      } else {
        const int ia = icol0 + r, ib = ia + 1;
        const bool oka = ia >= 0 && ia < j && ia < cols_total, okb = ib >= 0 && ib < j && ib < cols_total;
        if (DX > 0)
          emit.pair(xlo + r, y, v[r], v[r + 1 < R ? r + 1 : r], oka, okb);
        else
          emit.pair(xlo + (R - 2 - r), y, v[r + 1 < R ? r + 1 : r], v[r], okb, oka);
      }
    }
    if (own_diag) {  // seed: the diagonal cell is column j's first "previous"
      const int k = j - i0;
      const int ld = k / R, rd = k - ld * R;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (rd == r && lane == ld) v[r] = dg;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) prev[r] = v[r];
    if (has_consumer && lane == 63) {
      ring_out[y & (kRing - 1)] = v[R - 1];
      if (kMulti && ss.tail) ss.bnd_out[j] = v[R - 1];
    }
  };

  // fast window: 8 steps j..j+7 covering one 8-aligned window of y
  auto fast_window = [&](int j, auto diag_tag, auto prod_tag, auto edge_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;  // triangular start-up: seeding, ragged stores
    constexpr bool PROD = decltype(prod_tag)::value;  // a strip below feeds lane 0 (p > 0)
    // EDGE: not every column of the strip is real -- strip 0 of a slid grid owns columns icol < 0, the last
    // strip of an octant columns >= cols_total.  Those lanes compute on (nothing depends on them: a column
    // only feeds higher ones, and the ragged strip has no consumer) but never store.
    constexpr bool EDGE = decltype(edge_tag)::value;
    auto real = [&](int icol) { return icol >= 0 && icol < cols_total; };
    const int y0 = g.sy + DY * j;
    const int yb = y0 & ~7;
    const int blk = y0 >> 6, t0 = y0 & 63;
    if (blk != cur_blk) refill(blk);
    double rr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rr[k] = read_lane(rv, t0 + DY * k);
    int hs[R];
    const int sh = DY > 0 ? (t0 & 31) : (t0 & 31) - 7;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t half = (t0 & 32) ? (uint32_t)(ow[r] >> 32) : (uint32_t)ow[r];
      hs[r] = (int)(half >> sh);
    }
    double ringv = 0.0;
    if (PROD) ringv = ring_in[(y0 - DY + DY * lane) & rin_mask];
    double dgv = 0.0;  // lane t: diag(j + t), published by the x-major strips
    if (DIAG) dgv = diag_ring[(j + lane) & (kRing - 1)];
    double* ring_w = ring_out + (yb & (kRing - 1));
    double* ring_wl = lane == 63 ? ring_w : ss.dummy;
    double dj = (double)j;
    uint32_t off = (uint32_t)((y0 * m.nx + xlo) * CB);
    const uint32_t stride = (uint32_t)(DY * m.nx * CB);
    int y = y0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bit = DY > 0 ? k : 7 - k;
      double v[R];
      {
        const double b0 = shift_up(prev[R - 1], ringv);
        v[0] = and_mask(stencil(prev[0], b0, ratio(id[0], dj, rr[k])), __builtin_amdgcn_sbfe(hs[0], bit, 1));
      }
#pragma unroll
      for (int r = 1; r < R; ++r)
        v[r] = and_mask(stencil(prev[r], prev[r - 1], ratio(id[r], dj, rr[k])), __builtin_amdgcn_sbfe(hs[r], bit, 1));
      if (DIAG) {
        const int jk = j + k;
        // seed first: column jk's "previous" is diag(jk).  Unguarded on purpose -- a lane matches only
        // if it owns column jk; if that column lies beyond the quadrant its value is never stored.
        const double dg = read_lane(dgv, k);
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = ((icol0 + r) == jk) ? dg : v[r];
        if constexpr (R == 1) {
          if (icol0 < jk && (!EDGE || real(icol0))) emit.single_at(off, xlo, y, v[0]);
        }
        // ragged stores.  For a pair of columns (ia, ia+1) with ia < jk either both are below the
        // diagonal, or ia+1 == jk: then the second cell is the diagonal cell (jk,jk) itself, which the
        // seeding has just put into v -- the x-major strip stores the same value there.  So away from
        // the edges of the octant a pair is stored whole or not at all.
#pragma unroll
        for (int r = 0; r + 1 < R; r += 2) {
          const int ia = icol0 + r;
          if (EDGE) {  // one cell of the pair may not exist
            const bool oka = ia < jk && real(ia), okb = ia < jk && real(ia + 1);
            const double vs = oka ? v[r] : v[r + 1];
            if (DX > 0)
              emit.pair_or_single_at(off + (uint32_t)(r * CB), xlo + r, y, v[r], v[r + 1], oka && okb, oka != okb, oka ? 0 : 1, vs);
            else
              emit.pair_or_single_at(off + (uint32_t)((R - 2 - r) * CB), xlo + (R - 2 - r), y, v[r + 1], v[r], oka && okb, oka != okb,
                                     oka ? 1 : 0, vs);
          } else if (ia < jk) {
            if (DX > 0)
              emit.pair_at(off + (uint32_t)(r * CB), xlo + r, y, v[r], v[r + 1]);
            else
              emit.pair_at(off + (uint32_t)((R - 2 - r) * CB), xlo + (R - 2 - r), y, v[r + 1], v[r]);
          }
        }
      } else {
        if constexpr (R == 1) {
          if (!EDGE || real(icol0)) emit.single_at(off, xlo, y, v[0]);
        }
#pragma unroll
        for (int r = 0; r + 1 < R; r += 2) {
          if (EDGE) {
            const int ia = icol0 + r;
            const bool oka = real(ia), okb = real(ia + 1);
            const double vs = oka ? v[r] : v[r + 1];
            if (DX > 0)
              emit.pair_or_single_at(off + (uint32_t)(r * CB), xlo + r, y, v[r], v[r + 1], oka && okb, oka != okb, oka ? 0 : 1, vs);
            else
              emit.pair_or_single_at(off + (uint32_t)((R - 2 - r) * CB), xlo + (R - 2 - r), y, v[r + 1], v[r], oka && okb, oka != okb,
                                     oka ? 1 : 0, vs);
          } else if (DX > 0)
            emit.pair_at(off + (uint32_t)(r * CB), xlo + r, y, v[r], v[r + 1]);
          else
            emit.pair_at(off + (uint32_t)((R - 2 - r) * CB), xlo + (R - 2 - r), y, v[r + 1], v[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) prev[r] = v[r];
      ring_wl[bit] = v[R - 1];
      if (PROD) ringv = rotate_down(ringv);
      dj += 1.0;
      off += stride;
      y += DY;
    }
    if (kMulti && ss.tail) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < 8) ss.bnd_out[j + lane] = ring_w[DY > 0 ? lane : 7 - lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  for (int T = 0; T <= tmax; ++T) {
    const int n = T - ss.w - kYLag + ss.nbase;
    if (strip_on && n >= nfirst && n <= nlast) {
      int jlo, jhi;
      chunk_steps<DY>(g.sy, n, &jlo, &jhi);
      jlo = max(jlo, jstart);
      jhi = min(jhi, g.nj - 1);
      int j = jlo;
      while (j <= jhi) {
        const int y = g.sy + DY * j;
        const bool aligned = DY > 0 ? (y & 7) == 0 : (y & 7) == 7;
        if (Emit::kFastPath && aligned && j + 7 <= jhi) {
          const bool steady = j >= i0 + S;                      // every column of the strip has started
          const bool edge = head || !all_cols || !edge_free;    // some of its columns are not real
          using T1 = std::true_type;
          using T0 = std::false_type;
          if (p > 0) {
            if (edge) {
              if (steady) fast_window(j, T0(), T1(), T1());
              else fast_window(j, T1(), T1(), T1());
            } else {
              if (steady) fast_window(j, T0(), T1(), T0());
              else fast_window(j, T1(), T1(), T0());
            }
          } else {
            if (edge) {
              if (steady) fast_window(j, T0(), T0(), T1());
              else fast_window(j, T1(), T0(), T1());
            } else {
              if (steady) fast_window(j, T0(), T0(), T0());
              else fast_window(j, T1(), T0(), T0());
            }
          }
          j += 8;
        } else {
          slow_step(j);
          j += 1;
        }
      }
    }
    __syncthreads();
  }
}

// Sweeps whose fronts exceed one round (rows_per_round = W*64*R) need 4 boundary rows per
// workgroup in global memory; grows the caller's scratch as needed and points the map at it.
inline hipError_t attach_round_scratch(DevMap& m, int rows_per_round, size_t n_workgroups, double** scratch, size_t* cap) {
  const int maxdim = m.nx > m.ny ? m.nx : m.ny;
  if (maxdim <= rows_per_round) return hipSuccess;
  const size_t need = n_workgroups * 4 * (size_t)maxdim * sizeof(double);
  if (*cap < need) {
    if (*scratch) (void)hipFree(*scratch);
    *scratch = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc(scratch, need);
    if (e != hipSuccess) return e;
    *cap = need;
  }
  m.bnd = *scratch;
  m.bnd_len = maxdim;
  return hipSuccess;
}

// LDS of one subgroup with W strips per octant: x rings, y rings, the diagonal ring, (multi-round
// shapes) two boundary staging areas, W staging tiles.  In doubles.
__host__ __device__ inline size_t sweep_lds_doubles(int R, int W, bool multi) {
  return (size_t)2 * W * kRing + kRing + (multi ? 2 * kStage : 0) + (size_t)W * 64 * R * tile_stride(R) + (size_t)2 * W * 8;
}
// Dynamic LDS of a launch with W strips per octant per workgroup; with packing, G subgroups of W/G strips must
// fit too.
inline size_t sweep_lds_bytes(int R, int W, bool multi = false, bool pack = false) {
  size_t d = sweep_lds_doubles(R, W, multi);
  if (pack)
    for (int G = 2; G <= 4 && W / G >= 1; G *= 2) d = std::max(d, (size_t)G * sweep_lds_doubles(R, W / G, multi));
  return d * sizeof(double);
}

// One quadrant of one source: called by all 2*W wavefronts of a workgroup.
// Q1 (+,+) Q2 (-,+) Q3 (-,-) Q4 (+,-), reference solver.cpp:575-695.
template <int R, int DX, int DY, typename Emit>
__device__ __forceinline__ void sweep_quadrant_dir(const DevMap& m, Emit& emit, int sx, int sy, double* lds, const SubGroup sg) {
  constexpr int S = 64 * R;
  const int W = sg.W;
  const int wave = sg.wave;
  UnitGeom g;
  g.sx = sx;
  g.sy = sy;
  g.ni = DX > 0 ? m.nx - sx : sx;  // negative directions stop short of the border
  g.nj = DY > 0 ? m.ny - sy : sy;
  if (g.ni <= 0 || g.nj <= 0) {  // uniform for the subgroup: nothing to sweep, but keep the partners' barriers company
    for (int t = 0; t <= sg.tmax_floor; ++t) __syncthreads();
    return;
  }
  const int rows_total = min(g.nj, g.ni);
  const int cols_total = max(min(g.ni, g.nj - 1), 0);
  // Multi-round sweeps keep the plain grid: slid, a round's first y-major strip would own columns whose diagonal cells
  // the PREVIOUS round published, and the 32-entry diagonal ring has been overwritten by then (found by the 4096^2
  // eight-round parity test of round 2: last-bit differences next to a round boundary).
  g.ya = (m.slide && !Emit::kMulti) ? y_grid_slide<DX>(sx, cols_total, S, W) : 0;
  const int Px = (rows_total + S - 1) / S, Py = (cols_total + g.ya + S - 1) / S;
  double* ring_x = lds;
  double* ring_y = lds + (size_t)W * kRing;
  double* diag_ring = lds + (size_t)2 * W * kRing;
  double* stage = diag_ring + kRing;  // 2 * kStage, multi-round shapes only
  double* tiles = stage + (Emit::kMulti ? 2 * kStage : 0);
  // Fronts longer than W strips are swept in rounds of W strips; the last strip of a round leaves
  // its boundary row in global memory for the first strip of the next round.
  const int rounds = Emit::kMulti ? max((max(Px, Py) + W - 1) / W, 1) : 1;
  double* bnd = m.bnd ? m.bnd + (size_t)blockIdx.x * 4 * m.bnd_len : nullptr;
  for (int rho = 0; rho < rounds; ++rho) {
    const int pg0 = rho * W;
    const int nbase_x = chunk_seq<DX>(sx, min(pg0 * S, g.ni - 1));
    const int nbase_y = chunk_seq<DY>(sy, min(max(pg0 * S - g.ya, 0), g.nj - 1));
    const int wx = max(min(Px - pg0, W), 1), wy = max(min(Py - pg0, W), 1);
    StripSlot ss;
    ss.tmax = max(max(chunk_seq<DX>(sx, g.ni - 1) - nbase_x + wx - 1, chunk_seq<DY>(sy, g.nj - 1) - nbase_y + wy - 1 + kYLag),
                  sg.tmax_floor);
    if (wave < W) {
      ss.pg = pg0 + wave;
      ss.w = wave;
      ss.nbase = nbase_x;
      ss.tail = wave == W - 1 && ss.pg + 1 < Px;
      ss.bnd_in = bnd ? bnd + ((rho + 1) & 1) * m.bnd_len : nullptr;
      ss.bnd_out = bnd ? bnd + (rho & 1) * m.bnd_len : nullptr;
      ss.stage = stage;
      ss.dummy = tiles + (size_t)W * S * tile_stride(R) + (size_t)wave * 8;
      x_strip<R, DX, DY>(m, emit, g, ss, ring_x, tiles + (size_t)wave * S * tile_stride(R), diag_ring);
    } else {
      ss.pg = pg0 + wave - W;
      ss.w = wave - W;
      ss.nbase = nbase_y;
      ss.tail = ss.w == W - 1 && ss.pg + 1 < Py;
      ss.bnd_in = bnd ? bnd + (2 + ((rho + 1) & 1)) * m.bnd_len : nullptr;
      ss.bnd_out = bnd ? bnd + (2 + (rho & 1)) * m.bnd_len : nullptr;
      ss.stage = stage + kStage;
      ss.dummy = tiles + (size_t)W * S * tile_stride(R) + (size_t)wave * 8;
      y_strip<R, DX, DY>(m, emit, g, ss, ring_y, diag_ring);
    }
    if (rho + 1 < rounds) {
      // the boundary rows just stored must have landed (in L2) before the next round reads them
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

// Last pipeline slot of quadrant q of source (sx, sy) when swept in one round by W strips per octant
// (-1: the quadrant is empty).  Must agree with sweep_quadrant_dir.
template <int R>
__device__ __forceinline__ int unit_tmax(const DevMap& m, int sx, int sy, int q, int W) {
  constexpr int S = 64 * R;
  const bool px = (q == 0 || q == 3), py = q < 2;
  const int ni = px ? m.nx - sx : sx, nj = py ? m.ny - sy : sy;
  if (ni <= 0 || nj <= 0) return -1;
  const int cols_total = max(min(ni, nj - 1), 0);
  const int ya = !m.slide ? 0 : px ? y_grid_slide<+1>(sx, cols_total, S, W) : y_grid_slide<-1>(sx, cols_total, S, W);
  const int Px = (min(nj, ni) + S - 1) / S, Py = (cols_total + ya + S - 1) / S;
  const int wx = max(min(Px, W), 1), wy = max(min(Py, W), 1);
  const int nx_last = px ? chunk_seq<+1>(sx, ni - 1) : chunk_seq<-1>(sx, ni - 1);
  const int ny_last = py ? chunk_seq<+1>(sy, nj - 1) : chunk_seq<-1>(sy, nj - 1);
  return max(nx_last + wx - 1, ny_last + wy - 1 + kYLag);
}

template <int R, typename Emit>
__device__ __forceinline__ void sweep_quadrant(const DevMap& m, Emit& emit, int sx, int sy, int q, double* lds, const SubGroup sg) {
  if (q == 0) {
    // rows/columns no quadrant covers read as zero; quadrant 1 always exists
    if (sx > 0)
      for (int y = sg.tid; y < m.ny; y += sg.nthreads) emit.zero(0, y);
    if (sy > 0)
      for (int x = sg.tid; x < m.nx; x += sg.nthreads) emit.zero(x, 0);
    sweep_quadrant_dir<R, +1, +1>(m, emit, sx, sy, lds, sg);
  } else if (q == 1) {
    sweep_quadrant_dir<R, -1, +1>(m, emit, sx, sy, lds, sg);
  } else if (q == 2) {
    sweep_quadrant_dir<R, -1, -1>(m, emit, sx, sy, lds, sg);
  } else {
    sweep_quadrant_dir<R, +1, -1>(m, emit, sx, sy, lds, sg);
  }
}

__device__ __forceinline__ SubGroup whole_workgroup() {
  SubGroup sg;
  sg.W = blockDim.x >> 7;
  sg.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably uniform
  sg.tid = threadIdx.x;
  sg.nthreads = blockDim.x;
  sg.tmax_floor = -1;
  return sg;
}

// grid = n_src * 4 workgroups of 128*W threads; dynamic LDS = sweep_lds_bytes(R, W)
// Register budgets: R <= 2 single-round shapes fit 64 VGPRs so two 16-wavefront workgroups share a
// CU; the R = 2 multi-round shape runs 8-wavefront workgroups, three per CU.
// One workgroup slot: slot b sweeps unit b, or (descriptors from vhp_order_units) `count` units
// order[first .. first+count) side by side, each with 1/G of the wavefronts.
template <int R, bool MULTI, typename OutT>
__device__ __forceinline__ void sweep_slot(const DevMap& m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride,
                                           int* __restrict__ err_flag, const int* __restrict__ order, const int4* __restrict__ wg_desc,
                                           int b, double* lds) {
  int first = b, count = 1, G = 1;
  if (wg_desc) {
    const int4 d = wg_desc[b];
    first = d.x;
    count = d.y;
    G = d.z;
    if (count == 0) return;
  }
  // G and the wavefront count are powers of two: shifts only, and everything stays in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably uniform
  const int wsub_shift = __builtin_amdgcn_readfirstlane(__builtin_ctz(blockDim.x >> 6) - __builtin_ctz(G));  // log2(wavefronts per subgroup)
  const int sub = __builtin_amdgcn_readfirstlane(wave >> wsub_shift);
  SubGroup sg;
  sg.W = __builtin_amdgcn_readfirstlane((1 << wsub_shift) >> 1);
  sg.wave = __builtin_amdgcn_readfirstlane(wave & ((1 << wsub_shift) - 1));
  sg.tid = threadIdx.x & ((64 << wsub_shift) - 1);
  sg.nthreads = 64 << wsub_shift;
  sg.tmax_floor = -1;
  if (G > 1) {
    for (int u = 0; u < count; ++u) {
      const int unit_u = order[first + u];
      const int su = unit_u / kUnitsPerSource, qu = unit_u - su * kUnitsPerSource;
      const int ux = src_xy[2 * su], uy = src_xy[2 * su + 1];
      if (ux >= 0 && uy >= 0 && ux < m.nx && uy < m.ny) sg.tmax_floor = max(sg.tmax_floor, unit_tmax<R>(m, ux, uy, qu, sg.W));
    }
  }
  bool live = sub < count;
  int s = 0, q = 0, sx = 0, sy = 0;
  if (live) {
    const int unit = order ? order[first + sub] : first;
    s = unit / kUnitsPerSource;
    q = unit - s * kUnitsPerSource;
    sx = src_xy[2 * s];
    sy = src_xy[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= m.nx || sy >= m.ny) {
      if (sg.tid == 0 && q == 0) atomicOr(err_flag, 1);
      live = false;
    }
  }
  if (!live) {  // vacant subgroup or rejected source: only attend the partners' barriers
    for (int t = 0; t <= sg.tmax_floor; ++t) __syncthreads();
    return;
  }
#ifndef VHP_FRONT_NT
#define VHP_FRONT_NT false
#endif
  StoreEmit<OutT, MULTI, VHP_FRONT_NT> emit(out + (size_t)s * field_stride, m.nx, m.ny);
  sweep_quadrant<R>(m, emit, sx, sy, q, lds + (size_t)sub * sweep_lds_doubles(R, sg.W, MULTI), sg);
}

// grid = slots: workgroup b sweeps slot b; slots are handed out longest first, which with in-order dispatch is LPT
// scheduling over the CUs.  (Persistent workgroups pulling slots from a two-ended queue -- what the streaming sweep
// does -- were tried here in round 2 and lost 3-8 % at 256 sources; the wrapper also cost the kernel its registers:
// keeping the arguments live across the sweep pushed 26 VGPRs into scratch, whose reloads wait on vmcnt, i.e. for every
// global store in flight.)
template <int R, bool MULTI, typename OutT>
__global__ void __launch_bounds__((R == 2 && MULTI) ? 512 : 1024, R >= 4 ? 4 : ((MULTI && R == 2) ? 6 : 8))
vhp_sweep_fronts(DevMap m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride,
                 int* __restrict__ err_flag, const int* __restrict__ order, const int4* __restrict__ wg_desc) {
  extern __shared__ double lds[];
  sweep_slot<R, MULTI, OutT>(m, src_xy, out, field_stride, err_flag, order, wg_desc, (int)blockIdx.x, lds);
}

// ---------------------------------------------------------------------------
// Launch order of the (source, quadrant) units: a quadrant's sweep time grows with the length
// of its longest front, max(ni, nj).  One workgroup counting-sorts the units by that length,
// longest first.  order[k] = unit index; units of out-of-range sources sort last.
// ---------------------------------------------------------------------------
constexpr int kOrderBuckets = 1024;
constexpr int kOrderClasses = 3;  // G = 1, 2, 4 units per workgroup
// `pack_w` = strips per octant a full workgroup offers (W) when packing is wanted, 0 = no packing.
// Outputs: order[] (unit indices: class-major, longest first inside a class) and one descriptor
// per workgroup slot {first, count, G} (count 0 = vacant slot).
__global__ void __launch_bounds__(1024) vhp_order_units(const int32_t* __restrict__ src_xy, int n_src, int nx, int ny, int rows_per_strip,
                                                        int pack_w, int* __restrict__ order, int4* __restrict__ wg_desc) {
  __shared__ int hist[kOrderClasses * kOrderBuckets];
  __shared__ int start[kOrderClasses * kOrderBuckets];
  __shared__ int cls_n[kOrderClasses], cls_off[kOrderClasses], wg_off[kOrderClasses + 1];
  const int n_units = n_src * kUnitsPerSource;
  const int maxdim = max(nx, ny);
  auto bucket_of = [&](int u) {
    const int s = u / kUnitsPerSource, q = u - s * kUnitsPerSource;
    const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return (pack_w ? kOrderClasses : 1) * kOrderBuckets - 1;  // rejected by the sweep kernel: last
    const int ni = (q == 0 || q == 3) ? nx - sx : sx;
    const int nj = (q < 2) ? ny - sy : sy;
    const int len = (ni <= 0 || nj <= 0) ? 0 : max(ni, nj);
    const int strips = (ni <= 0 || nj <= 0) ? 0 : (min(ni, nj) + rows_per_strip - 1) / rows_per_strip;
    int cls = 0;  // G = 1
    if (pack_w >= 4 && strips <= pack_w / 4) cls = 2;       // four units per workgroup
    else if (pack_w >= 2 && strips <= pack_w / 2) cls = 1;  // two
    // bucket 0 of a class = longest
    return cls * kOrderBuckets + (kOrderBuckets - 1) - (int)(((long long)len * (kOrderBuckets - 1)) / maxdim);
  };
  for (int b = threadIdx.x; b < (pack_w ? kOrderClasses : 1) * kOrderBuckets; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  for (int u = threadIdx.x; u < n_units; u += blockDim.x) atomicAdd(&hist[bucket_of(u)], 1);
  __syncthreads();
  // exclusive prefix sums of the histogram, class after class: a wavefront scan by DPP-free shuffles, the 16
  // wavefront totals by the first wavefront (blockDim.x == kOrderBuckets: thread t owns bucket t of every class)
  __shared__ int wave_tot[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int acc = 0, wacc = 0;  // identical in every thread
  for (int c = 0; c < (pack_w ? kOrderClasses : 1); ++c) {
    const int v = hist[c * kOrderBuckets + threadIdx.x];
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int t = wave_tot[k];
      before += k < wv ? t : 0;
      total += t;
    }
    start[c * kOrderBuckets + threadIdx.x] = acc + before + inc - v;
    if (threadIdx.x == 0) {
      cls_off[c] = acc;
      cls_n[c] = total;
      wg_off[c] = wacc;
    }
    acc += total;
    wacc += (total + (1 << c) - 1) >> c;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int c = pack_w ? kOrderClasses : 1; c < kOrderClasses; ++c) {
      cls_off[c] = acc;
      cls_n[c] = 0;
      wg_off[c] = wacc;
    }
    wg_off[kOrderClasses] = wacc;
  }
  __syncthreads();
  for (int u = threadIdx.x; u < n_units; u += blockDim.x) order[atomicAdd(&start[bucket_of(u)], 1)] = u;
  for (int w = threadIdx.x; w < n_units; w += blockDim.x) {
    int4 d = make_int4(0, 0, 1, 0);
    for (int c = 0; c < kOrderClasses; ++c) {
      if (w >= wg_off[c] && w < wg_off[c + 1]) {
        const int k = (w - wg_off[c]) << c;
        d = make_int4(cls_off[c] + k, min(1 << c, cls_n[c] - k), 1 << c, 0);
      }
    }
    wg_desc[w] = d;
  }
}

// ---------------------------------------------------------------------------
// map packing: one wavefront per 64 cells, ballot -> one word
// ---------------------------------------------------------------------------
__global__ void vhp_pack_rows(const uint8_t* __restrict__ occ, uint64_t* __restrict__ rows, int nx, int ny, int wpr) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int words = wpr - 2;
  if (wave >= words * ny) return;
  const int y = wave / words, w = wave - y * words;
  const int x = w * 64 + lane;
  const bool f = x < nx && occ[(size_t)y * nx + x] != 0;
  const uint64_t b = __ballot(f);
  if (lane == 0) rows[(size_t)y * wpr + 1 + w] = b;
}

__global__ void vhp_pack_cols(const uint8_t* __restrict__ occ, uint64_t* __restrict__ cols, int nx, int ny, int wpc) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int words = wpc - 2;
  if (wave >= words * nx) return;
  const int x = wave / words, w = wave - x * words;
  const int y = w * 64 + lane;
  const bool f = y < ny && occ[(size_t)y * nx + x] != 0;
  const uint64_t b = __ballot(f);
  if (lane == 0) cols[(size_t)x * wpc + 1 + w] = b;
}

}  // namespace vhp
