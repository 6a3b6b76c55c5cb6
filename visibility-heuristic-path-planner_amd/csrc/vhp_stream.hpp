// vhp_stream.hpp -- the streaming visibility sweep: the throughput shape of computeVisibility()
// (reference src/visibilityBasedSolver.cpp:570-696) for large batches of sources.
//
// Same mathematics as the front sweep of vhp_sweep.hip.h (tests/schedule_model.py states it: a quadrant is an
// x-major octant whose fronts are columns and a y-major octant whose fronts are rows; a cell needs only the
// previous front: itself and its neighbour one row / column below), organised for a store-bound launch of a
// thousand quadrants instead of for the latency of one:
//
//   * the unit of scheduling is one OCTANT of one quadrant of one source (8 per source), taken by a workgroup of 2W
//     wavefronts: W wavefronts sweep strips of an x-major octant (64 rows each, one row per lane) and W more flush what
//     they produce (below); or 2W-1 wavefronts sweep strips of a y-major octant (128 columns each, two adjacent columns
//     per lane, so that a lane stores 16 bytes per row and a wavefront 1 KB) and one computes the diagonal.  The two
//     octants of a quadrant share only the diagonal cells, and the y-major unit recomputes those itself (DiagWave: the
//     two-term recurrence sub(k) = V(k,k-1), diag(k) = sub(k)*occ(k,k) of tests/schedule_model.py, running ahead of the
//     strips), so units never talk to each other: a full-size quadrant is two independent 4 MB units instead of one
//     8 MB unit that sets the length of the launch;
//   * the unit of work is one 64-cell BLOCK of the marching coordinate of one strip -- exactly one word of the
//     bit-packed occupancy maps, so a lane loads one word per owned row/column and unit.  A wavefront takes the strips
//     w, w+W, w+2W, ... one after the other, block by block.  There is no workgroup barrier: the workgroup is a small
//     dataflow machine.  Every wavefront counts the units it has finished in an LDS word, and before a unit it polls
//     (once per 64 steps) the counters of the one or two wavefronts it depends on: the strip below must have swept the
//     same block (boundary row, through a four-block LDS ring, or through a full-length LDS row from the last strip of
//     a round to the first strip of the next), the reader of its output ring must not lag more than the ring holds, and
//     a y-major strip needs the diagonal cells of its columns.  A wavefront that is ahead simply runs ahead (the
//     barrier-per-slot form of this kernel left 70 % of its wavefront slots idle: DESIGN.md section 4b);
//   * x-major strips stage their columns in a wave-private LDS tile of two or three 8-column windows; whenever a
//     window completes 128-byte lines of the field, the strip's FLUSHER wavefront -- told so by a 16-byte descriptor in
//     an LDS ring -- reads them out of the tile and stores them, 8 rows per store instruction.  With a row pitch that is
//     an odd multiple of 64 bytes (1000 columns!) odd and even rows are half a line apart, so the two row classes flush
//     alternately, each every 16 steps;
//   * the stale diagonal of the reference (cell (k,k) = cell (k,k-1) * occ) is produced inside the x-major
//     strip that owns row k; the y-major unit seeds column k with the same value from its own DiagWave, through an LDS array.
//
// The code is written against vhp_lanes.hpp and is compiled for gfx950 (vhp_stream.hip) and, unchanged, for the CPU
// simulator of tests/sim (parity against the oracle without a GPU).  Requires nx % 8 == 0 (whole 64-byte sectors per
// row piece); vhp_capi.hip falls back to the front sweep otherwise.
#pragma once
#include "vhp_lanes.hpp"

namespace vhp {
namespace stream {

using namespace vhp::lanes;

#if defined(VHP_DIAG_WGTIME) && !defined(VHP_SIM)  // diagnostic builds only: where a wavefront's cycles go
#define VHP_PROF_DECL unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};
#define VHP_PROF_T0(var) const unsigned long long var = __builtin_readcyclecounter()
#define VHP_PROF_ADD(slot, var) prof[slot] += __builtin_readcyclecounter() - var
#define VHP_PROF_COUNT(slot) prof[slot] += 1
#else
#define VHP_PROF_DECL
#define VHP_PROF_T0(var)
#define VHP_PROF_ADD(slot, var)
#define VHP_PROF_COUNT(slot)
#endif
#ifdef VHP_SIM
#define VHP_FN inline
#define VHP_HD inline
#else
#define VHP_FN __device__ __forceinline__
#define VHP_HD __host__ __device__ __forceinline__
#endif

constexpr int kBlock = 64;       // steps per slot: one word of the packed occupancy maps
constexpr int kRing = 256;       // entries of a strip-to-strip boundary ring: four blocks, indexed by the absolute marching coordinate
                                 // (the reader of block n also needs the last entry of block n-1 while the writer is in block n+1)
constexpr int kDiagRing = 2048;  // entries of the diagonal array of a y-major unit (a ring on grids larger than that)
constexpr int kDescRing = 16;    // flush descriptors in flight between an x-major wavefront and its flusher (at most tile slots + 3)
constexpr int kXRows = 64;       // rows per x-major strip (one per lane)
constexpr int kYCols = 128;      // columns per y-major strip (two per lane)
constexpr int kMaxStrips = 136;  // 8192 / 64 + slack

struct Map {
  const uint64_t* rows;  // bit x&63 of rows[y*wpr + 1 + (x>>6)] = occ(x,y)
  const uint64_t* cols;  // bit y&63 of cols[x*wpc + 1 + (y>>6)] = occ(x,y)
  const double* recip;   // recip[k] = RN(1/k), recip[0] = 0, readable up to max(nx,ny)+8
  int wpr, wpc, nx, ny;
};

VHP_HD int imin(int a, int b) { return a < b ? a : b; }
VHP_HD int imax(int a, int b) { return a > b ? a : b; }
VHP_HD int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// LDS of one workgroup, in doubles.  A workgroup sweeps one octant at a time: the x-major kind needs W staging tiles,
// the y-major kind the diagonal array; rings, round row, slabs and progress words are common.  One launch holds both kinds
// of units, so its dynamic LDS is sized for the larger (lds_doubles).
struct Layout {
  int ring;              // W rings of kRing each (ring w = output of wavefront w)
  int round;             // boundary row from the last strip of a round to the first strip of the next, indexed by the absolute
                         // marching coordinate
  int diag;              // y-major units: diag(k), k = quadrant-local index, written by the DiagWave
  int dummy;             // 8 doubles per wavefront: where lanes that are not the boundary lane "write" theirs
  int slab;              // 64 doubles per wavefront (+ the DiagWave): the reciprocals of the block it sweeps
  int tiles;             // x-major units: W staging tiles of kXRows rows, tile_slots windows of 8 columns (+1 double) per row
  int tile_slots;        // 2: a 128-byte line exactly; 3: one window of slack between the sweeping wavefront and its flusher
  int sched;             // ints: the progress words (kDoneSlots), base[strips], then W rings of kDescRing flush descriptors
  int strips;            // capacity of the base array: the most strips an octant of this grid can have
  int total;
  int round_mask;
};
constexpr int kDoneSlots = 32;
// W = sweeping wavefronts of this kind of unit (x-major and y-major units of one launch may differ)
VHP_HD Layout make_layout(int W, int nx, int ny, bool x_major, int tile_slots) {
  Layout L;
  const int rr = next_pow2(x_major ? nx : ny);
  int o = 0;
  L.ring = o; o += W * kRing;
  L.round = o; o += rr;
  L.diag = o; if (!x_major) o += imin(imax(nx, ny) + 72, kDiagRing);
  L.dummy = o; o += (W + 1) * 8;
  L.slab = o; o += (W + 1) * kBlock;
  L.tile_slots = tile_slots;
  L.tiles = o; if (x_major) o += W * kXRows * (8 * tile_slots + 1);
  L.strips = ((imin(nx, ny) + kXRows - 1) / kXRows + 2 + 3) & ~3;  // (a multiple of 4: the descriptors after it are 16-byte aligned)
  o = (o + 1) & ~1;
  L.sched = o; o += (kDoneSlots + L.strips + (x_major ? W * kDescRing * 4 : 0)) / 2 + 1;
  L.total = o;
  L.round_mask = rr - 1;
  return L;
}
VHP_HD int lds_doubles(int Wx, int Wy, int nx, int ny, int tile_slots) {
  return imax(make_layout(Wx, nx, ny, true, tile_slots).total, make_layout(Wy, nx, ny, false, tile_slots).total);
}

// Geometry of one quadrant.  x = sx + DX*i, y = sy + DY*j; negative directions stop one cell short of the border
// (solver.cpp:607-610,638-642).
template <int DX, int DY>
struct Quad {
  int sx, sy, ni, nj;
  int rows_total;  // x-major rows j in [0, rows_total)
  int cols_total;  // y-major columns i in [0, cols_total) have computed cells (j > i)
  int ya;          // y-major strips own columns [128q - ya, 128q - ya + 128): slid so that a strip starts on a 128-byte line
  int Px, Py;      // strips per octant
  int bx0, by0;    // block (>> 6) of the source
  int Nbx, Nby;    // blocks per march

  VHP_FN void init(int nx, int ny, int sx_, int sy_) {
    sx = sx_; sy = sy_;
    ni = DX > 0 ? nx - sx : sx;
    nj = DY > 0 ? ny - sy : sy;
    rows_total = imin(ni, nj);
    cols_total = imax(imin(ni, nj - 1), 0);
    ya = DX > 0 ? (sx & 15) : ((-(sx + 1)) & 15);
    Px = (rows_total + kXRows - 1) / kXRows;
    Py = cols_total > 0 ? (cols_total + ya + kYCols - 1) / kYCols : 0;
    bx0 = sx >> 6; by0 = sy >> 6;
    Nbx = ni > 0 ? nbx(ni - 1) + 1 : 0;
    Nby = nj > 0 ? nby(nj - 1) + 1 : 0;
  }
  VHP_FN bool empty() const { return ni <= 0 || nj <= 0; }
  VHP_FN int X(int i) const { return sx + DX * i; }
  VHP_FN int Y(int j) const { return sy + DY * j; }
  // block sequence number of a step
  VHP_FN int nbx(int i) const { const int b = X(i) >> 6; return DX > 0 ? b - bx0 : bx0 - b; }
  VHP_FN int nby(int j) const { const int b = Y(j) >> 6; return DY > 0 ? b - by0 : by0 - b; }
  // steps of block n, clipped to the march
  VHP_FN void xsteps(int n, int& lo, int& hi) const {
    const int b = DX > 0 ? bx0 + n : bx0 - n;
    if (DX > 0) { lo = 64 * b - sx; hi = 64 * b + 63 - sx; } else { lo = sx - (64 * b + 63); hi = sx - 64 * b; }
    lo = imax(lo, 0); hi = imin(hi, ni - 1);
  }
  VHP_FN void ysteps(int n, int& lo, int& hi) const {
    const int b = DY > 0 ? by0 + n : by0 - n;
    if (DY > 0) { lo = 64 * b - sy; hi = 64 * b + 63 - sy; } else { lo = sy - (64 * b + 63); hi = sy - 64 * b; }
    lo = imax(lo, 0); hi = imin(hi, nj - 1);
  }
  VHP_FN int ycol0(int q) const { return kYCols * q - ya; }           // first column of y-major strip q (may be < 0)
  VHP_FN int ystart(int q) const { return imax(ycol0(q), 0); }        // its first step
};

// Progress bookkeeping of a workgroup (ints in LDS at Layout::sched):
//   done[0..W-1]   units (strip, block) the sweeping wavefronts have finished;
//   done[16+w]     x-major units: flush descriptors the flusher of wavefront w has read out of the tile (the columns
//                  they covered may be overwritten);
//   done[31]       y-major units: diagonal entries diag(0 .. done[31]-1) have been produced (DiagWave);
//   base[p]        index of strip p's first block in the unit sequence of its wavefront (p % W): the wavefront has
//                  finished block n of strip p once done >= base[p] + (n - first block of p) + 1.
constexpr int kDrainedSlot = 16, kDiagDoneSlot = 31;
template <int DX, int DY>
struct Progress {
  volatile int* done;
  int* base;
  int W;

  VHP_FN void bind(double* lds, const Layout& L, int W_) {
    int* p = reinterpret_cast<int*>(lds + L.sched);
    done = p;
    base = p + kDoneSlots;
    W = W_;
  }
  VHP_FN int* descriptors(const Layout& L, int w) const { return base + L.strips + w * (kDescRing * 4); }
  // Before any wavefront starts: every thread calls clear() (tid of nthreads), then -- after a barrier -- one thread
  // calls setup().  (DS writes through lds_set_int: the generic pointer would make each of these a FLAT store that is
  // waited for, 10 us per unit.)
  VHP_FN void clear(bool x_major, const Layout& L, int tid, int nthreads) {
    int* d = const_cast<int*>(done);
    for (int k = tid; k < kDoneSlots; k += nthreads) lds_set_int(d + k, 0);
    if (x_major)
      for (int k = tid; k < W * kDescRing * 4; k += nthreads) lds_set_int(base + L.strips + k, 0);  // no descriptor carries sequence number 0
  }
  VHP_FN void setup(const Quad<DX, DY>& g, bool x_major) {
    if (x_major) {
      for (int p = 0; p < g.Px; ++p) lds_set_int(base + p, p >= W ? lds_int_at(base + p - W) + (g.Nbx - g.nbx(kXRows * (p - W))) : 0);
    } else {
      for (int q = 0; q < g.Py; ++q) lds_set_int(base + q, q >= W ? lds_int_at(base + q - W) + (g.Nby - g.nby(g.ystart(q - W))) : 0);
    }
  }
  // has x-major strip p swept block n?  (blocks before the strip's first are nothing to wait for, blocks past the
  // march mean "the whole strip")
  VHP_FN bool x_done(const Quad<DX, DY>& g, int p, int n) const {
    const int nf = g.nbx(kXRows * p);
    if (n < nf) return true;
    return lds_poll(done + p % W) >= lds_int_at(base + p) + (imin(n, g.Nbx - 1) - nf) + 1;
  }
  VHP_FN bool y_done(const Quad<DX, DY>& g, int q, int n) const {
    const int nf = g.nby(g.ystart(q));
    if (n < nf) return true;
    return lds_poll(done + q % W) >= lds_int_at(base + q) + (imin(n, g.Nby - 1) - nf) + 1;
  }
  VHP_FN int diag_ready() const { return lds_poll(done + kDiagDoneSlot); }
  // the same tests with the constants of a strip fetched once (XWave / YWave::load_strip): strip p' has swept block n'
  // (n' not before its first block nf') once done[p' % W] >= off(p') + min(n', last block), off = base[p'] - nf' + 1
  VHP_FN int x_off(const Quad<DX, DY>& g, int p) const { return lds_int_at(base + p) - g.nbx(kXRows * p) + 1; }
  VHP_FN int y_off(const Quad<DX, DY>& g, int q) const { return lds_int_at(base + q) - g.nby(g.ystart(q)) + 1; }
};

// ---------------------------------------------------------------------------------------------------------------
// Both kinds of wavefront keep every VALU instruction they can off the step: on gfx950 a wave64 vector instruction
// occupies its SIMD for 4 cycles whatever its width, so the count of vector instructions per step IS the speed of a
// strip.  Wave-uniform operands of a step (the reciprocal of the step index, the boundary value of the strip below,
// the seed of a column) are therefore not moved between lanes by v_readlane / DPP rotates but fetched by LDS broadcast
// reads (every lane reads the same address: an LDS-queue instruction, no vector ALU slot), a window of 8 steps ahead.
// ---------------------------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------------------------
// x-major wavefront: strips p = w, w+W, ...; rows j = 64p + lane; steps i = 64p .. ni-1; cells (i, j), j <= i.
//
// The field leaves through a second wavefront, the strip's FLUSHER.  The sweeping wavefront only computes: it writes
// its column of 64 values into the wave-private tile every step and, whenever a window of 8 columns completes 128-byte
// lines, posts a four-word descriptor (which line, which rows, how far the march is) in an LDS ring.  The flusher reads
// the lines out of the tile and stores them, 8 rows per instruction.  A store instruction stalls the wavefront that
// issues it for some 300 cycles when the CU's store path is busy -- which in this kernel is always -- and the march of
// an x-major strip is one dependent chain of ~140 cycles per window of 8 steps: with the stores in line, the largest
// octants ran at a third of their compute speed and set the length of the launch.  The tile has three windows of 8
// columns (when the LDS allows: make_layout), so the flusher may lag one window behind; with two it is a plain hand-off.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDescPred = 1 << 17, kDescExit = 1 << 18;
template <int DX, int DY, typename OutT>
struct XWave {
  static constexpr int CB = sizeof(OutT);
  // uniform
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  int w, W;
  double* tile;
  int nslots, tstride;  // windows of 8 columns per tile row; doubles per tile row (8*nslots + 1: spreads the column writes over the banks)
  double* ring_base;
  double* round;
  int round_mask;
  double* slab;   // reciprocals of the current block's 64 steps, indexed by x & 63
  double* dummy;  // where the lanes that are not the boundary lane "write" theirs
  Progress<DX, DY> prog;
  int r_stride;   // 2: the row pitch is an odd multiple of 64 bytes, odd and even rows are half a line apart; else 1
  int p, j0, rows_here, nf;
  int n;        // the block to sweep next
  int my_done;  // units finished
  bool active, has_consumer;
  const double* rin;
  int rin_mask;
  double* rout;
  int rout_mask;
  // the hand-off to the flusher
  volatile int* drained_w;
  int* desc;
  int posted;       // sweeping side: descriptors posted so far ...
  int hist[7];      // ... and when the window 1, 2, ... 7 windows before the current one was entered (hist[0]: the one before)
  int cur_win;      // x >> 3 of the window being swept
  int drained;      // flusher side: descriptors read out of the tile
  int pf_blk;       // the block (x >> 6) whose operands wait in ow_nx / rv_nx, or -1
  // the dependencies of the current strip, fetched once per strip (ready() then reads progress words only)
  int raw_nf, raw_off;            // strip p-1: its first block, Progress::x_off
  int war_kind, war_nf, war_off;  // who reads what I write: 0 nobody, 1 strip p+1 (my ring), 2 strip p-W+1 (the round row)
  int reuse_need;                 // my ring's previous reader (strip p-W+1) must have finished: its progress word at least this, or 0
#ifdef VHP_SIM
  XWave* flusher;   // the simulator runs the flusher inside the sweeping wavefront's calls:
  bool lazy;        // as late as the protocol allows (only when the sweeping side has to wait), or right at the post
#endif
  // lanes
  vi lane;
  VHP_PROF_DECL     // [0] block-start loads, [1] steady windows, [2] diagonal windows, [3] single steps, [4] waits for the flusher, [5] windows
  vi tile_l;        // lane * tstride
  vi fl_t0, fl_hi;  // flush: tile index of the lane's piece in tile slot 0 (row slot 0 .. 7), and whether the piece lies in the
                    // second window of its line
  vu32 fl_off;      // flush: byte offset of the lane's piece from the lowest row of a store instruction, for xa = 0
  vd prev, jd;
  vu64 ow, ow_nx;
  vd rv_nx;

  VHP_FN void init_common(const Map& m_, const Quad<DX, DY>& g_, OutT* out_, int w_, int W_, double* lds, const Layout& L) {
    m = m_; g = g_; out = out_; w = w_; W = W_;
    nslots = L.tile_slots;
    tstride = 8 * nslots + 1;
    tile = lds + L.tiles + w * kXRows * tstride;
    ring_base = lds + L.ring;
    round = lds + L.round;
    round_mask = L.round_mask;
    slab = lds + L.slab + w * kBlock;
    dummy = lds + L.dummy + w * 8;
    prog.bind(lds, L, W);
    drained_w = prog.done + kDrainedSlot + w;
    desc = prog.descriptors(L, w);
    posted = drained = 0;
    for (int k = 0; k < 7; ++k) hist[k] = 0;
    cur_win = -1;
    pf_blk = -1;
    r_stride = ((m.nx >> 3) & 1) ? 2 : 1;
    lane = lane_id();
    tile_l = lane * tstride;
    {
      // flush geometry: lane -> (row slot = lane >> 3, piece = lane & 7 = cells xa + 2*piece, +1); the row slots of a
      // store instruction are counted upward in y, so that byte offsets from its lowest row are never negative
      const vi rslot = lane >> 3, pc = lane & 7;
      const vi rs = DY > 0 ? rslot : 7 - rslot;
      fl_t0 = rslot * (r_stride * tstride) + ((pc * 2) & 7);
      fl_hi = pc >> 2;
      fl_off = to_u32((rs * (r_stride * m.nx) + pc * 2) * CB);
    }
    active = true;
    my_done = 0;
    j0 = 0;
    rows_here = 0;
  }
  // the sweeping wavefront w of an x-major unit
  VHP_FN void init(const Map& m_, const Quad<DX, DY>& g_, OutT* out_, int w_, int W_, double* lds, const Layout& L) {
    init_common(m_, g_, out_, w_, W_, lds, L);
    load_strip(w);
  }
  // the flusher of sweeping wavefront w
  VHP_FN void init_flusher(const Map& m_, const Quad<DX, DY>& g_, OutT* out_, int w_, int W_, double* lds, const Layout& L) {
    init_common(m_, g_, out_, w_, W_, lds, L);
  }

  // the tile slot of window `win` (>= 0); slot counts: 2, 3, 4, 6, 8
  VHP_FN int slot_of(int win) const { return nslots == 3 ? win % 3 : nslots == 6 ? win % 6 : win & (nslots - 1); }

  VHP_FN void load_strip(int pn) {
    p = pn;
    if (p >= g.Px) { active = false; return; }
    j0 = kXRows * p;
    rows_here = imin(kXRows, g.rows_total - j0);
    has_consumer = p + 1 < g.Px;
    nf = g.nbx(j0);
    n = nf;
    prev = vd(0.0);
    jd = to_f64(lane + j0);
    if (p % W == 0) { rin = round; rin_mask = round_mask; } else { rin = ring_base + (w - 1) * kRing; rin_mask = kRing - 1; }
    if ((p + 1) % W == 0) { rout = round; rout_mask = round_mask; } else { rout = ring_base + w * kRing; rout_mask = kRing - 1; }
    // a new strip starts anywhere in the tile: everything posted has to be out first
    cur_win = -1;
    for (int k = 0; k < 7; ++k) hist[k] = posted;
    pf_blk = -1;
    // dependencies (see ready())
    raw_nf = raw_off = 0;
    if (p > 0) { raw_nf = g.nbx(kXRows * (p - 1)); raw_off = prog.x_off(g, p - 1); }
    war_kind = 0; war_nf = war_off = 0;
    if (has_consumer) {
      if ((p + 1) % W != 0) { war_kind = 1; war_nf = g.nbx(kXRows * (p + 1)); war_off = prog.x_off(g, p + 1); }
      else if (p >= 2 * W - 1) { war_kind = 2; war_nf = g.nbx(kXRows * (p - W + 1)); war_off = prog.x_off(g, p - W + 1); }
    }
    reuse_need = (p >= W && (p - W + 1) % W != 0) ? prog.x_off(g, p - W + 1) + (g.Nbx - 1) : 0;
  }

  // May the next unit (strip p, block n) run?  Everything it reads from other wavefronts has been produced, and
  // nothing it overwrites is still needed.  (Strip p' has swept block n' once the progress word of its wavefront is at
  // least x_off(p') + min(n', last block); blocks before its first are nothing to wait for.)
  VHP_FN bool ready() const {
    // the two progress words this can depend on, read together: strip p-1's wavefront and strip p+1's (= strip p-W+1's)
    const int w_below = w == 0 ? W - 1 : w - 1, w_above = w == W - 1 ? 0 : w + 1;
    const int d_below = lds_poll(prog.done + w_below), d_above = lds_poll(prog.done + w_above);
    const int last = g.Nbx - 1;
    // the boundary row: strip p-1 has swept this block
    if (p > 0 && n >= raw_nf && d_below < raw_off + imin(n, last)) return false;
    if (war_kind == 1) {
      // my output ring holds four blocks; writing block n overwrites block n-4, whose last entry the reader needs
      // while it sweeps block n-3
      if (n - 3 >= war_nf && d_above < war_off + imin(n - 3, last)) return false;
    } else if (war_kind == 2) {
      // I write the round-to-round row: the strip that still reads the previous round's entries from it (strip
      // p-W+1, the first of my round) needs the last entry of block n while it sweeps block n+1
      if (n + 1 >= war_nf && d_above < war_off + imin(n + 1, last)) return false;
    }
    // a new strip reuses my output ring: the reader of my previous strip must be through with it
    if (n == nf && reuse_need != 0 && d_above < reuse_need) return false;
    return true;
  }

  // sweeps the next unit and publishes it
  VHP_FN void run_unit() {
    sweep_block(n);
    const bool last = n == g.Nbx - 1;
    if (last) end_of_march();
    ++my_done;
    lds_publish(prog.done + w, my_done);
    if (last) load_strip(p + W); else ++n;
  }

  // ---- sweeping side of the hand-off ----
  // One 128-byte line (16 cells from xa on; for fp32 fields one 64-byte sector) of the rows r_first, r_first + r_stride, ...
  // of this strip is complete in the tile (flags & kDescPred: only in part -- the cells with step index j <= i' <= i_now).
  VHP_FN void post(int xa, int r_first, int i_now, int flags) {
    int* d = desc + (posted & (kDescRing - 1)) * 4;
    ++posted;
    // One 16-byte LDS write: the sequence number -- what the flusher polls -- in the first and the last word, the
    // descriptor between them (a reader that sees both numbers equal has not read a half-written descriptor).
    lds_post4(d, posted, (xa + 16) | (r_first << 16) | flags | (rows_here << 19), i_now | (j0 << 14), posted);
#ifdef VHP_SIM
    if (!lazy) flusher->drain_one();
#endif
  }
  VHP_FN void wait_drained(int need) {
#ifdef VHP_SIM
    while (*drained_w < need) flusher->drain_one();
#else
    while (lds_poll(drained_w) < need) backoff();
    lds_acquire();
#endif
  }
  // the sweeping wavefront has no strip left: its flusher may leave too
  VHP_FN void finish() {
    post(0, 0, 0, kDescExit);
#ifdef VHP_SIM
    while (flusher->drain_one()) {}
#endif
  }

  // ---- flusher side ----
  // takes the next descriptor (waiting for it) and stores its lines; false once the sweeping wavefront has left
  VHP_FN bool drain_one() {
    const int* d = desc + (drained & (kDescRing - 1)) * 4;
    int seq0, d0, d1, seq1;
    lds_read4(d, seq0, d0, d1, seq1);
#ifndef VHP_SIM
    while (seq0 != drained + 1 || seq1 != drained + 1) { backoff(); lds_read4(d, seq0, d0, d1, seq1); }
    lds_acquire();
#endif
    if (d0 & kDescExit) return false;
    const int i_now = d1 & 0x3fff;
    j0 = d1 >> 14;
    rows_here = (d0 >> 19) & 0x7f;
    const int xa = (d0 & 0xffff) - 16, r_first = (d0 >> 16) & 1;
    // flush() publishes `drained` as soon as its tile reads are issued -- before the stores, whose issue may stall for
    // hundreds of cycles each: the sweeping wavefront gets its tile columns back that much earlier
    if (d0 & kDescPred) flush<true>(xa, r_first, i_now); else flush<false>(xa, r_first, i_now);
    return true;
  }

  // Emits one line of the rows r = r_first, r_first + r_stride, ... of strip j0 from the tile, 8 rows per store instruction.
  template <bool PRED>
  VHP_FN void flush(int xa, int r_first, int i_now) {
    wave_sync();
    const int win = (xa >> 3) + 24;  // (xa may be -8 at the end of a march; 24 is a multiple of every slot count)
    const int sA = slot_of(win), sB = slot_of(win + 1);
    // (arithmetic, not a select between members: a select of two loads becomes a load through a selected address, which
    // keeps the whole wavefront object in scratch memory)
    const vi t0 = fl_t0 + fl_hi * (8 * (sB - sA)) + (8 * sA + r_first * tstride);
    const vu32 off = fl_off + (uint32_t)(xa * CB);
    const int t_step = 8 * r_stride * tstride;  // per store instruction: 8 row slots further
    const long y_low = DY > 0 ? g.Y(j0 + r_first) : g.Y(j0 + r_first + 7 * r_stride);
    OutT* base = out + y_low * (long)m.nx;          // uniform: the row term lives in scalar registers
    const long base_step = (long)(8 * r_stride * DY) * m.nx;
    if (!PRED && rows_here == kXRows) {
      // all the lines into registers, the tile handed back, then the stores
      vd a[8], b[8];
      if (r_stride == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = lds_load(tile, t0 + u * t_step); b[u] = lds_load(tile, t0 + (u * t_step + 1)); }
        ++drained;
        lds_publish(drained_w, drained);  // (issued after the tile reads: the LDS keeps the order)
#pragma unroll
        for (int u = 0; u < 4; ++u) { g_store2(base, off, a[u], b[u]); base += base_step; }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = lds_load(tile, t0 + u * t_step); b[u] = lds_load(tile, t0 + (u * t_step + 1)); }
        ++drained;
        lds_publish(drained_w, drained);
#pragma unroll
        for (int u = 0; u < 8; ++u) { g_store2(base, off, a[u], b[u]); base += base_step; }
      }
    } else {
      const vi xc = (lane & 7) * 2 + xa;                             // x of the pair's first cell
      const vi i0c = (xc - g.sx) * DX, i1c = (xc + 1 - g.sx) * DX;  // step indices of the two cells
      const vi r0 = (lane >> 3) * r_stride + r_first;
      for (int u = 0; r_first + r_stride * 8 * u < rows_here; ++u) {
        const vi r = r0 + 8 * r_stride * u;
        const vb row_ok = r < rows_here;
        const vi tix = select(row_ok, t0 + u * t_step, vi(0));
        const vd a = lds_load(tile, tix);
        const vd b = lds_load(tile, tix + 1);
        if (!PRED) {
          g_store2_if(row_ok, vb(false), vb(false), base, off, a, b);
        } else {
          const vi jr = r + j0;
          const vb ok0 = row_ok && (i0c >= jr) && (i0c <= i_now);
          const vb ok1 = row_ok && (i1c >= jr) && (i1c <= i_now);
          g_store2_if(ok0 && ok1, ok0, ok1, base, off, a, b);
        }
        base += base_step;
      }
      ++drained;
      lds_publish(drained_w, drained);
    }
    wave_sync();
  }

  // After the step at x_b, which ends an 8-cell window of x: the rows whose 128-byte line this completes leave.
  // (y*nx + x) % 16 == 0 marks a line start; with nx = 8*m that is x % 16 == 8 * ((y*m) & 1).
  VHP_FN void flush_completed(int x_b, int i_now) {
    const int edge = DX > 0 ? x_b + 1 : x_b;  // marching up, the line ends below `edge`; marching down it starts at `edge`
    const int hbit = (edge >> 3) & 1;
    const int xa = DX > 0 ? x_b - 15 : x_b;
    const bool steady = i_now - 15 >= j0 + kXRows - 1;  // the line's first-marched cell is past every row's diagonal
    int r_first = 0;
    if (r_stride == 1) {
      if (hbit != 0) return;
    } else {
      r_first = (hbit ^ g.sy ^ j0) & 1;  // rows with (y & 1) == hbit
    }
    post(xa, r_first, i_now, steady ? 0 : kDescPred);
  }

  // the march of this strip is over: what is still in the tile leaves as partial lines
  VHP_FN void end_of_march() {
    const int i_now = g.ni - 1;
    const int xe = g.X(i_now);
    if (r_stride == 1) {
      post(xe & ~15, 0, i_now, kDescPred);
    } else {
      for (int ph = 0; ph < 2; ++ph) {  // rows whose lines start at x % 16 == 8*ph
        const int xa = 8 * ph + (((xe - 8 * ph) >> 4) << 4);
        post(xa, (ph ^ g.sy ^ j0) & 1, i_now, kDescPred);
      }
    }
  }

  // one generic step
  VHP_FN void step1(int i) {
    const int x = g.X(i);
    const int t = x & 63;
    const double ri = slab[t];
    const double di = (double)i;
    double fill = 0.0, dsrc = 1.0;  // OLD / NEW value of the row just below lane 0's (1.0 = light strength at the origin)
    if (p > 0) { fill = rin[(x - DX) & rin_mask]; dsrc = rin[x & rin_mask]; }
    const vd b = shift_up(prev, vd(fill));
    const vi mk = bit_mask(ow, t);
    vd v = and_mask(stencil(prev, b, ratio(jd, di, ri)), mk);
    if (i < j0 + kXRows) {
      // the diagonal cell (i,i) inherits the NEW value of the row below it times its own occupancy
      const vd up = shift_up(v, vd(dsrc));
      const vb isd = lane == (i - j0);
      const vd dcell = and_mask(up, mk);
      v = select(isd, dcell, v);
    }
    prev = v;
    lds_store(tile, tile_l + (slot_of(x >> 3) * 8 + (x & 7)), v);
    if (has_consumer) lds_store_if(lane == 63, rout, vi(x & rout_mask), v);
  }

  // eight steps covering one aligned window of x; DIAG: the strip's diagonal may fall into it
  template <bool DIAG>
  VHP_FN void window8(int i0) {
    const int x0 = g.X(i0);
    const int t0 = x0 & 63;
    const int xw = x0 & ~7;  // lowest x of the window
    // uniform operands by LDS broadcast: the reciprocals of the 8 step indices ...
    vd rr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rr[k] = lds_bcast(slab, (xw & 63) + (DX > 0 ? k : 7 - k));
    // ... and the boundary row of the strip below: rb[k] = its value at x(i0 + k) - DX, the OLD neighbour of lane 0 at
    // step k (and rb[k + 1] the NEW one, which the diagonal cell of row j0 takes)
    vd rb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) rb[k] = vd(0.0);
    if (p > 0) {
      rb[0] = lds_bcast(rin, (x0 - DX) & rin_mask);
      const int xb = xw & rin_mask;  // the window itself never wraps in the ring
#pragma unroll
      for (int k = 1; k < 9; ++k) rb[k] = lds_bcast(rin, xb + (DX > 0 ? k - 1 : 8 - k));
    }
    const vu32 hs = half_shifted(ow, t0, DX > 0 ? (t0 & 31) : (t0 & 31) - 7);  // step k's bit at position (x & 7)
    const vi tidx = tile_l + slot_of(xw >> 3) * 8;
    // every lane writes "its boundary value" each step -- lane 63 into the ring, the others into a dummy slot: one
    // ds_write instead of an exec-masked region per step
    double* wbase = has_consumer ? rout + (xw & rout_mask) : dummy;
    const vi widx = select(lane == 63, vi(0), vi((int)(dummy - wbase)));
    vd di = vd((double)i0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int col = DX > 0 ? k : 7 - k;
      const vd b = shift_up(prev, rb[k]);
      const vi mk = sbfe1(hs, col);
      vd v = and_mask(stencil(prev, b, ratio(jd, di, rr[k])), mk);
      if (DIAG) {
        const vd up = shift_up(v, p > 0 ? rb[k + 1] : vd(1.0));  // strip 0: 1.0 = light strength at the origin
        const vb isd = lane == (i0 + k - j0);
        const vd dcell = and_mask(up, mk);
        v = select(isd, dcell, v);
      }
      prev = v;
      lds_store(tile, tidx + col, v);
      lds_store(wbase, widx + col, v);
      di = di + 1.0;
    }
  }

  // the occupancy word of every lane's row and the reciprocals of the 64 step indices of block blk (x >> 6)
  VHP_FN void load_block(int blk, vu64& o, vd& rv) {
    const vi yl = (vmin(lane + j0, g.rows_total - 1)) * DY + g.sy;
    o = g_load_u64(m.rows, yl * m.wpr + (1 + blk));
    const vi it = (lane + (blk * 64 - g.sx)) * DX;
    const vb ok = (it >= 0) && (it < g.ni);
    rv = select(ok, g_load_f64(m.recip, select(ok, it, vi(0))), vd(0.0));
  }

  VHP_FN void sweep_block(int nb) {
    int lo, hi;
    g.xsteps(nb, lo, hi);
    lo = imax(lo, j0);
    if (lo > hi) return;
    const int blk = g.X(lo) >> 6;
    VHP_PROF_T0(tl0);
    {
      vd rv;
      if (pf_blk == blk) { ow = ow_nx; rv = rv_nx; } else { load_block(blk, ow, rv); }
      pin(ow);
      pin(rv);
      lds_store(slab, lane, rv);
      wave_sync();
      // the operands of the strip's next block are fetched a block ahead: this wavefront issues no stores, so waiting for
      // them later waits for nothing else
      if (nb + 1 < g.Nbx) { pf_blk = blk + DX; load_block(pf_blk, ow_nx, rv_nx); } else { pf_blk = -1; }
    }
    VHP_PROF_ADD(0, tl0);
    int i = lo;
    while (i <= hi) {
      const int x = g.X(i);
      if ((x >> 3) != cur_win) {
        // This window's tile columns were last read by the flushes posted up to two windows ago (three tile slots; one
        // window ago with two): they must have left the tile.
        VHP_PROF_T0(tf0);
        // (with S tile slots: the flushes posted before the window S-2 windows back was entered)
        int need = posted;
#pragma unroll
        for (int k = 0; k < 6; ++k) need = nslots - 3 == k ? hist[k] : need;
        wait_drained(need);
        VHP_PROF_ADD(4, tf0);
#pragma unroll
        for (int k = 6; k > 0; --k) hist[k] = hist[k - 1];
        hist[0] = posted;
        cur_win = x >> 3;
      }
      const bool aligned = DX > 0 ? (x & 7) == 0 : (x & 7) == 7;
      int i_last;
      VHP_PROF_T0(tw0);
      if (aligned && i + 7 <= hi) {
        if (i < j0 + kXRows) { window8<true>(i); VHP_PROF_ADD(2, tw0); } else { window8<false>(i); VHP_PROF_ADD(1, tw0); }
        VHP_PROF_COUNT(5);
        i_last = i + 7;
      } else {
        step1(i);
        VHP_PROF_ADD(3, tw0);
        i_last = i;
      }
      i = i_last + 1;
      const int xl = g.X(i_last);
      const bool boundary = DX > 0 ? (xl & 7) == 7 : (xl & 7) == 0;
      if (boundary && i_last != g.ni - 1) flush_completed(xl, i_last);  // (the last step of the march is end_of_march's)
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// y-major wavefront: strips q = w, w+W, ...; columns i = 128q - ya + 2*lane + {0,1}; steps j = max(i0,0) .. nj-1;
// cells (i, j), i <= j (the diagonal cell is the seed diag(j), stored again with its neighbour).
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY, typename OutT>
struct YWave {
  static constexpr int CB = sizeof(OutT);
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  int w, W;
  double* ring_base;
  double* round;
  int round_mask;
  const double* diag;  // diag(k) at k & (kDiagRing - 1)
  double* slab;   // reciprocals of the current block's 64 steps, indexed by y & 63
  double* dummy;
  Progress<DX, DY> prog;
  int q, i0, jstart, nf;
  int raw_nf, raw_off, war_kind, war_nf, war_off, reuse_need;  // the strip's dependencies, as in XWave
  int n;        // the block to sweep next
  int my_done;  // units finished
  bool active, has_consumer, interior;
  const double* rin;
  int rin_mask;
  double* rout;
  int rout_mask;
  vi lane, ia, ib;
  vd prev0, prev1, id0, id1;
  vu64 ow0, ow1;
  vu32 xoff;  // byte offset of the lane's pair inside a row
  VHP_PROF_DECL  // [0] block-start loads, [1] steady windows, [2] diagonal / predicated windows, [3] single steps, [5] windows

  VHP_FN void init(const Map& m_, const Quad<DX, DY>& g_, OutT* out_, int w_, int W_, double* lds, const Layout& L) {
    m = m_; g = g_; out = out_; w = w_; W = W_;
    ring_base = lds + L.ring;
    round = lds + L.round;
    round_mask = L.round_mask;
    diag = lds + L.diag;
    slab = lds + L.slab + w * kBlock;
    dummy = lds + L.dummy + w * 8;
    prog.bind(lds, L, W);
    lane = lane_id();
    active = true;
    my_done = 0;
    load_strip(w);
  }

  VHP_FN void load_strip(int qn) {
    q = qn;
    if (q >= g.Py) { active = false; return; }
    i0 = g.ycol0(q);
    jstart = g.ystart(q);
    has_consumer = q + 1 < g.Py;
    interior = i0 >= 0 && i0 + kYCols - 1 < g.ni;  // every column of the strip is a column of the grid
    nf = g.nby(jstart);
    n = nf;
    ia = lane * 2 + i0;
    ib = ia + 1;
    prev0 = vd(0.0);
    prev1 = vd(0.0);
    id0 = to_f64(ia);
    id1 = to_f64(ib);
    // the pair's lower x: x(ia) marching up, x(ib) marching down
    const vi xlo = DX > 0 ? ia + g.sx : (-ib) + g.sx;
    xoff = to_u32(xlo * CB);
    if (q % W == 0) { rin = round; rin_mask = round_mask; } else { rin = ring_base + (w - 1) * kRing; rin_mask = kRing - 1; }
    if ((q + 1) % W == 0) { rout = round; rout_mask = round_mask; } else { rout = ring_base + w * kRing; rout_mask = kRing - 1; }
    // dependencies (see ready())
    raw_nf = raw_off = 0;
    if (q > 0) { raw_nf = g.nby(g.ystart(q - 1)); raw_off = prog.y_off(g, q - 1); }
    war_kind = 0; war_nf = war_off = 0;
    if (has_consumer) {
      if ((q + 1) % W != 0) { war_kind = 1; war_nf = g.nby(g.ystart(q + 1)); war_off = prog.y_off(g, q + 1); }
      else if (q >= 2 * W - 1) { war_kind = 2; war_nf = g.nby(g.ystart(q - W + 1)); war_off = prog.y_off(g, q - W + 1); }
    }
    reuse_need = (q >= W && (q - W + 1) % W != 0) ? prog.y_off(g, q - W + 1) + (g.Nby - 1) : 0;
  }

  // (the same rules as XWave::ready, plus the seeds)
  VHP_FN bool ready() const {
    const int w_below = w == 0 ? W - 1 : w - 1, w_above = w == W - 1 ? 0 : w + 1;
    const int d_below = lds_poll(prog.done + w_below), d_above = lds_poll(prog.done + w_above), d_diag = lds_poll(prog.done + kDiagDoneSlot);
    const int last = g.Nby - 1;
    if (q > 0 && n >= raw_nf && d_below < raw_off + imin(n, last)) return false;
    if (war_kind == 1) {
      if (n - 3 >= war_nf && d_above < war_off + imin(n - 3, last)) return false;
    } else if (war_kind == 2) {
      if (n + 1 >= war_nf && d_above < war_off + imin(n + 1, last)) return false;
    }
    if (n == nf && reuse_need != 0 && d_above < reuse_need) return false;
    // the seeds of this block: diag(k) for the strip's columns k among the block's steps, from the DiagWave
    int lo, hi;
    g.ysteps(n, lo, hi);
    const int kmax = imin(imin(hi, i0 + kYCols - 1), g.rows_total - 1);
    if (kmax >= imax(lo, jstart) && kmax >= d_diag) return false;
    return true;
  }

  VHP_FN void run_unit() {
    sweep_block(n);
    const bool last = n == g.Nby - 1;
    ++my_done;
    lds_publish(prog.done + w, my_done);
    if (last) load_strip(q + W); else ++n;
  }

  // stores the lane's two cells of row y (step j): predicated on the cells being columns of the grid at or below
  // the diagonal
  VHP_FN void store_pred(OutT* row, int j, vd v0, vd v1) {
    const vb ok0 = (ia >= 0) && (ia < g.ni) && (ia <= j);
    const vb ok1 = (ib >= 0) && (ib < g.ni) && (ib <= j);
    if (DX > 0) g_store2_if(ok0 && ok1, ok0, ok1, row, xoff, v0, v1);
    else g_store2_if(ok0 && ok1, ok1, ok0, row, xoff, v1, v0);
  }
  // the row of the field as a uniform pointer: the row term stays in scalar registers, the lane term (xoff) in one VGPR
  VHP_FN OutT* row_ptr(int y) const { return out + (size_t)y * (size_t)m.nx; }

  VHP_FN void step1(int j) {
    const int y = g.Y(j);
    const int t = y & 63;
    const double rj = slab[t];
    const double dj = (double)j;
    double fill = 0.0;
    if (q > 0) fill = rin[(y - DY) & rin_mask];
    const vd b0 = shift_up(prev1, vd(fill));
    vd v0 = and_mask(stencil(prev0, b0, ratio(id0, dj, rj)), bit_mask(ow0, t));
    vd v1 = and_mask(stencil(prev1, prev0, ratio(id1, dj, rj)), bit_mask(ow1, t));
    if (j <= i0 + kYCols - 1) {  // column j (if this strip owns it) is seeded with diag(j)
      const double dg = diag[j & (kDiagRing - 1)];
      v0 = select(ia == j, vd(dg), v0);
      v1 = select(ib == j, vd(dg), v1);
    }
    store_pred(row_ptr(y), j, v0, v1);
    prev0 = v0;
    prev1 = v1;
    if (has_consumer) lds_store_if(lane == 63, rout, vi(y & rout_mask), v1);
  }

  // eight steps covering one aligned window of y.  DIAG: seeding may happen (implies PRED); PRED: predicated stores
  template <bool DIAG, bool PRED>
  VHP_FN void window8(int j0w) {
    const int y0 = g.Y(j0w);
    const int t0 = y0 & 63;
    const int yb = y0 & ~7;
    vd rr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rr[k] = lds_bcast(slab, (yb & 63) + (DY > 0 ? k : 7 - k));
    vd rb[8];  // the boundary column of the strip below at y(j0w + k) - DY
#pragma unroll
    for (int k = 0; k < 8; ++k) rb[k] = vd(0.0);
    if (q > 0) {
      rb[0] = lds_bcast(rin, (y0 - DY) & rin_mask);
      const int yq = yb & rin_mask;
#pragma unroll
      for (int k = 1; k < 8; ++k) rb[k] = lds_bcast(rin, yq + (DY > 0 ? k - 1 : 8 - k));
    }
    vd dg[8];  // diag(j0w + k)
    if (DIAG) {
#pragma unroll
      for (int k = 0; k < 8; ++k) dg[k] = lds_bcast(diag, (j0w + k) & (kDiagRing - 1));
    }
    const int sh = DY > 0 ? (t0 & 31) : (t0 & 31) - 7;
    const vu32 hs0 = half_shifted(ow0, t0, sh), hs1 = half_shifted(ow1, t0, sh);
    // every lane writes "its boundary value" each step -- lane 63 into the ring, the others into a dummy slot: one
    // ds_write instead of an exec-masked region per step
    double* wbase = has_consumer ? rout + (yb & rout_mask) : dummy;
    const vi widx = select(lane == 63, vi(0), vi((int)(dummy - wbase)));
    vd dj = vd((double)j0w);
    OutT* row = row_ptr(y0);
    const long rowstep = (long)DY * m.nx;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bit = DY > 0 ? k : 7 - k;
      const vd b0 = shift_up(prev1, rb[k]);
      vd v0 = and_mask(stencil(prev0, b0, ratio(id0, dj, rr[k])), sbfe1(hs0, bit));
      vd v1 = and_mask(stencil(prev1, prev0, ratio(id1, dj, rr[k])), sbfe1(hs1, bit));
      if (DIAG) {
        v0 = select(ia == j0w + k, dg[k], v0);
        v1 = select(ib == j0w + k, dg[k], v1);
      }
      if (PRED) store_pred(row, j0w + k, v0, v1);
      else if (DX > 0) g_store2(row, xoff, v0, v1);
      else g_store2(row, xoff, v1, v0);
      prev0 = v0;
      prev1 = v1;
      lds_store(wbase, widx + bit, v1);
      dj = dj + 1.0;
      row += rowstep;
    }
  }

  VHP_FN void sweep_block(int nb) {
    int lo, hi;
    g.ysteps(nb, lo, hi);
    lo = imax(lo, jstart);
    if (lo > hi) return;
    const int blk = g.Y(lo) >> 6;
    VHP_PROF_T0(tl0);
    {
      const vi xa = vmin(vmax(ia, 0), g.ni - 1) * DX + g.sx;
      const vi xb = vmin(vmax(ib, 0), g.ni - 1) * DX + g.sx;
      ow0 = g_load_u64(m.cols, xa * m.wpc + (1 + blk));
      ow1 = g_load_u64(m.cols, xb * m.wpc + (1 + blk));
      const vi jt = (lane + (blk * 64 - g.sy)) * DY;
      const vb ok = (jt >= 0) && (jt < g.nj);
      vd rv = select(ok, g_load_f64(m.recip, select(ok, jt, vi(0))), vd(0.0));
      pin(ow0);
      pin(ow1);
      pin(rv);
      lds_store(slab, lane, rv);
      wave_sync();
    }
    VHP_PROF_ADD(0, tl0);
    int j = lo;
    while (j <= hi) {
      const int y = g.Y(j);
      const bool aligned = DY > 0 ? (y & 7) == 0 : (y & 7) == 7;
      VHP_PROF_T0(tw0);
      if (aligned && j + 7 <= hi) {
        if (j <= i0 + kYCols - 1) { window8<true, true>(j); VHP_PROF_ADD(2, tw0); }
        else if (!interior) { window8<false, true>(j); VHP_PROF_ADD(2, tw0); }
        else { window8<false, false>(j); VHP_PROF_ADD(1, tw0); }
        VHP_PROF_COUNT(5);
        j += 8;
      } else {
        step1(j);
        VHP_PROF_ADD(3, tw0);
        j += 1;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// The diagonal of a quadrant for its y-major unit.  diag(0) = occ(source); for k >= 1 (tests/schedule_model.py):
//   sub(k)  = V(k, k-1) = (a - c*(a - b)) * occ(k, k-1),  a = diag(k-1), b = sub(k-1), c = (k-1)/k
//   diag(k) = sub(k) * occ(k, k)                                            (the stale diagonal)
// exactly the operations the x-major strip performs on row k-1 at step k and on its diagonal lane.  The chain is serial:
// one wavefront computes it (every lane the same value), 64 entries per unit, ahead of the strips that need it.
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY>
struct DiagWave {
  Map m;
  Quad<DX, DY> g;
  double* diag;
  volatile int* ready_word;
  Progress<DX, DY> prog;
  int k;        // next entry
  bool active;
  vi lane;
  vd dprev, sprev;  // diag(k-1), sub(k-1): the same value in every lane

  VHP_FN void init(const Map& m_, const Quad<DX, DY>& g_, int W, double* lds, const Layout& L) {
    m = m_; g = g_;
    diag = lds + L.diag;
    prog.bind(lds, L, W);
    ready_word = prog.done + kDiagDoneSlot;
    lane = lane_id();
    k = 0;
    active = g.rows_total > 0;
    dprev = vd(0.0);
    sprev = vd(0.0);
  }
  // On grids larger than the array the entries live in a ring: entry k overwrites entry k - kDiagRing, the seed of a
  // column whose strip (and, through the chain of boundary dependencies, every strip before it) must have finished.
  VHP_FN bool ready() const {
    const int reused = imin(k + kBlock, g.rows_total) - 1 - kDiagRing;
    if (reused < 0) return true;
    return prog.y_done(g, (reused + g.ya) / kYCols, g.Nby - 1);
  }

  // entries k .. k+63
  VHP_FN void run_unit() {
    const int k0 = k, k1 = imin(k0 + kBlock, g.rows_total);
    // lane l: the occupancy bits of cells (kk, kk-1) and (kk, kk), kk = k0 + l, and 1/kk
    const vi kk = vmin(lane + k0, g.rows_total - 1);
    const vi x = kk * DX + g.sx;
    const vi ya = vmax(kk - 1, 0) * DY + g.sy, yb = kk * DY + g.sy;
    const vu64 wa = g_load_u64(m.rows, ya * m.wpr + ((x >> 6) + 1));
    const vu64 wb = g_load_u64(m.rows, yb * m.wpr + ((x >> 6) + 1));
    const vd rk = g_load_f64(m.recip, kk);
    const vi ma = bit_mask_lane(wa, x & 63), mb = bit_mask_lane(wb, x & 63);
    for (int kq = k0; kq < k1; ++kq) {
      const int l = kq - k0;
      vd dcur;
      if (kq == 0) {
        dcur = and_mask(vd(1.0), vi(read_lane_i(mb, l)));  // the origin: light strength 1 times its occupancy
        sprev = vd(0.0);
      } else {
        const vd c = ratio(vd((double)(kq - 1)), (double)kq, read_lane(rk, l));
        const vd sub = and_mask(stencil(dprev, sprev, c), vi(read_lane_i(ma, l)));
        dcur = and_mask(sub, vi(read_lane_i(mb, l)));
        sprev = sub;
      }
      dprev = dcur;
      lds_store(diag, vi(kq & (kDiagRing - 1)), dcur);  // (every lane holds the same value: one write, no exec masking)
    }
    k = k1;
    lds_publish(ready_word, k1);
    if (k >= g.rows_total) active = false;
  }
};

// rows / columns no quadrant covers read as zero: done by the x-major unit of quadrant 1
template <typename OutT>
VHP_FN void zero_fill_cell(OutT* out, size_t idx) { out[idx] = OutT(0); }

}  // namespace stream
}  // namespace vhp
