// vhp_pool.hpp -- the pool sweep: computeVisibility() (reference src/visibilityBasedSolver.cpp:570-696) for large
// batches of sources, organised as a pool of wavefronts per CU that pull STRIPS.
//
// A quadrant is an x-major octant whose fronts are columns and a y-major octant whose fronts are rows (vhp_geom.hpp); a unit
// is one octant; a strip is 64 rows / 128 columns of it; a block is 64 steps of a strip.  (Round 2's streaming sweep, retired --
// DESIGN.md appendix A.4b --, swept ONE unit per workgroup with a fixed team: wavefront 0 of a full-size octant had 2424
// dependent steps, and a launch without stores took 0.46 ms for 0.1 ms of arithmetic.)  Here
//
//   * ONE workgroup of W wavefronts per CU, persistent, holds up to C units at once (contexts in LDS);
//   * every wavefront is a worker: it claims the next strip of a context whose predecessor strip has got far enough
//     (compare-and-swap on the context's claim word), sweeps it from its first step to the end of the march, and looks for the
//     next.  A full-size octant has all of its strips in flight at once, a window apart; small units fill the wavefronts the
//     large ones leave.  When no strip is ready and a context is free, the wavefront installs a unit of the launch's queue
//     (vhp_pool_order: units sorted by the length of their march): the head contexts take the LONGEST unit left, the filler
//     context the SHORTEST -- a strip occupies its wavefront until its march ends, so what runs beside a large unit must be
//     short.  The first unit of every context is handed out by workgroup index, no pull (Args::static_round);
//   * the boundary line of strip p goes to strip p+1 twice.  DURABLY through global memory (the L2), a block of 64 entries
//     at a time: 16-byte entries {value, tag}, the tag being the launch's epoch, so that a reader can tell an entry of
//     this launch from whatever the scratch held before -- no fence, no wait for the producer's stores.  And FAST through a
//     256-entry ring in LDS that belongs to the writing wavefront and that it overwrites without ever waiting for anybody: a
//     reader that follows closely (the normal case) reads the values of its next window straight out of the ring and checks
//     the writer's header word -- which strip the ring belongs to, how far it has got -- before and after; if the writer is
//     more than ~240 steps ahead, or has moved on to another strip, the reader takes the whole block from global memory
//     instead.  So a strip follows the strip below it by one window instead of a 64-step block, and a strip still waits only
//     for strips claimed before it: the machine cannot deadlock, whatever the number of wavefronts, contexts and strips.
//     (A first version had rings that were never overwritten: with the reader of a ring not yet claimed its writer stalls,
//     and on marches longer than ~4 W blocks every wavefront can end up stalled behind an unclaimed reader -- the simulator
//     found it);
//   * the strips (round 5): windows of SIXTEEN steps, XStrip16 / YStrip16 below -- the latency sweep's window machinery
//     (vhp_lat.hpp) with whole-line stores: an x-major strip flushes its own 16-column staging tile, 8 rows x 128 bytes per
//     store instruction, the rows that lie half a line off the grid out of a read in mid-window.  The build for widths that are
//     not a multiple of 8, for padded or unaligned fields (ANYW) keeps the 8-step strips of rounds 3-4, XStrip / YStrip: the lines
//     of a row start anywhere there, the tile is a ring of three 8-column windows (XStrip::flush_rows);
//   * the stale diagonal of a y-major unit is produced by the wavefront that installed the unit -- the serial
//     two-term recurrence of DiagTask -- into a scratch line in global memory, 64 entries at a time; a strip
//     loads the seeds of its own columns into registers when it starts.
//
// Written against vhp_lanes.hpp: compiled for gfx950 (vhp_pool.hip) and, unchanged, for the CPU simulator of tests/sim,
// where every wavefront is a coroutine and backoff() switches to the simulator's scheduler.
#pragma once
#include "vhp_geom.hpp"

namespace vhp {
namespace pool {

using namespace vhp::lanes;
using geom::imax;
using geom::imin;
using geom::Map;
using geom::Quad;

constexpr int kBlock = geom::kBlock;
constexpr int kXRows = geom::kXRows;
constexpr int kYCols = geom::kYCols;
constexpr int kLW16 = 16;     // steps of a window of the latency sweep (vhp_lat.hpp kLW), for the hand-over across workgroups (Link::store_window)
constexpr int kTStride = 17;  // doubles per tile row: two windows of 8 columns + 1 (spreads the column writes over the banks)
constexpr int kTStrideAny = 25;  // ... of the build for widths that are not a multiple of 8: three windows + 1 (XStrip::flush_rows)
constexpr int kUnits = 8;     // units per source: 4 quadrants x {x-major, y-major}
VHP_DIAG_TL_DECLARE

// ---- LDS of a workgroup -------------------------------------------------------------------------------------------
// doubles per wavefront: the staging tile, the reciprocal slab, the boundary line of the strip below for the current
// block (66 entries: the block's 64 and the neighbour of its first step on either side), the ring of the boundary values
// this strip produces (its last 256 steps), a dummy slot; then the scheduler's ints:
//   [0] queue empty   [2] units installed so far (sequence numbers)   [3] wavefronts sweeping a strip right now
//   [4] the late contexts are open   [5] bit c: context c has taken its first unit (Args::static_round)
//   [8 + w] header of wavefront w's ring: tag of the strip it belongs to << 14 | steps of it that are in the ring
//   per context: state (0 free, 1 being installed, 2 active), claim word (seq << 8 | next strip; -1 while not active),
//   unit, strips, strips not yet finished, sx | sy << 16, diagonal entries ready, first block of the unit's boundary lines
//   (the latency sweep keeps another word there: vhp_lat.hpp kDiagZero), then progress[strip] (steps swept) and owner[strip]
//   (the wavefront that sweeps it)
constexpr int kSchedHead = 8 + 16, kCtxHead = 8;
constexpr int kHdr = 8;
constexpr int kBin = 72;   // doubles of a boundary-in slab (66 used)
constexpr int kRing = 256; // entries of a wavefront's boundary ring, indexed by the marching coordinate & 255
constexpr int kRingSafe = 232;  // a reader trusts ring entries only while the writer is at most this many steps past them
enum { kQEmpty = 0, kSeq = 2, kBusy = 3, kLateOpen = 4, kFirstDone = 5 };
enum { kState = 0, kWord = 1, kUnit = 2, kNStrips = 3, kLeft = 4, kSxSy = 5, kDiagReady = 6, kLineBase = 7 };
struct Layout {
  int W, C, S;
  int tiles, slabs, bins, rings, dummies, sched, ctx_stride, total;
};
VHP_HD Layout make_layout(int W, int C, int nx, int ny, int tstride = kTStride) {
  Layout L;
  L.W = W; L.C = C;
  L.S = ((imax(nx, ny) + 63) / 64 + 2 + 3) & ~3;  // most strips a unit can have (+ slack)
  int o = 0;
  L.tiles = o; o += W * kXRows * tstride;
  L.slabs = o; o += W * 2 * kBlock;  // (two blocks of reciprocals per wavefront: the current one and the next)
  L.bins = o; o += W * kBin;
  L.rings = o; o += W * kRing;
  L.dummies = o; o += W * 16;  // (8 per wavefront here; the latency sweep, vhp_lat.hpp, uses 16)
  L.sched = o;
  L.ctx_stride = kCtxHead + 2 * L.S;
  o += (kSchedHead + C * L.ctx_stride + 1) / 2;
  L.total = o;
  return L;
}

struct Shared {
  double* lds;
  Layout L;
  VHP_FN int* sched() const { return reinterpret_cast<int*>(lds + L.sched); }
  VHP_FN int* ctx(int c) const { return sched() + kSchedHead + c * L.ctx_stride; }
  VHP_FN int* prog(int c) const { return ctx(c) + kCtxHead; }
  VHP_FN int* owner(int c) const { return ctx(c) + kCtxHead + L.S; }
  VHP_FN int* hdr(int w) const { return sched() + kHdr + w; }
  VHP_FN double* ring(int w) const { return lds + L.rings + w * kRing; }
  VHP_FN double* bin(int w) const { return lds + L.bins + w * kBin; }
};

template <typename OutT>
struct Args {
  Map m;
  OutT* out;
  long long field_stride;
  int* err_flag;
  // The launch order (vhp_pool_order): record k = {unit, sx | sy << 16 (-1: a source outside the grid), first 64-entry block of
  // the unit's boundary lines, 0} of the k-th unit, longest march first -- one 16-byte load tells a wavefront all it needs to install
  // a unit (the unit id, then the source, then the line base used to be three dependent loads)
  const int* recs;
  unsigned long long* queue;  // units taken so far: from the head (low word) and from the tail (high word)
  int n_units;
  // The first unit of every context is handed out by workgroup index instead of by the queue: context c of workgroup g takes
  // record c * n_groups + g (c < n_head) or n_units - 1 - ((c - n_head) * n_groups + g), and the queue starts behind them
  // (n_head * n_groups | (C - n_head) * n_groups << 32: vhp_pool_order writes it).  Only when n_units >= C * n_groups and every
  // context is open from the start.  Without it a launch opens with C * n_groups atomics on one word, the slowest of which
  // returns ~10 us after the first.
  bool static_round, static_snake;
  int n_groups;
  double* diag;       // scratch: diag(k) of the y-major unit of (source s, quadrant q) at (4 s + q) * diag_stride + k
  int diag_stride;
  Tagged* lines;      // scratch: the boundary lines; strip p of a unit at 64 * (its line base + p * blocks of its march) entries
  uint64_t epoch;     // the tag of this launch (never 0, never repeated on this scratch)
  int busy_cap;       // a workgroup takes another unit only while fewer than this many of its wavefronts are sweeping
  int n_head;         // contexts 0 .. n_head-1 take the largest unit left, the others the smallest ...
  int tail_limit;     // ... while fewer than this many units have been taken from the small end; then the largest left, too
  int early_ctx;      // contexts >= this one open only once `late_after` units have been taken: towards the end of a launch a
  int late_after;     // workgroup holds more, shorter units at once (what is left then has nothing large to get in the way of)
  int claim_ahead;    // a strip may be claimed this many steps before the strip below has swept its first window (find_work)
};

// Geometry of a unit without the direction templates: what the scheduler needs to tell whether a strip may start.
struct UnitGeo {
  int ni, nj, rows_total, cols_total, ya, n_strips, ph, nb;
  bool x_major;
  VHP_FN void init(int nx, int ny, int qo, int sx, int sy) {
    const int q = qo >> 1;
    const int dx = (q == 0 || q == 3) ? 1 : -1, dy = q < 2 ? 1 : -1;
    x_major = (qo & 1) == 0;
    ni = dx > 0 ? nx - sx : sx;
    nj = dy > 0 ? ny - sy : sy;
    rows_total = imin(ni, nj);
    cols_total = imax(imin(ni, nj - 1), 0);
    ya = dx > 0 ? (sx & 15) : ((-(sx + 1)) & 15);
    ph = dy > 0 ? (sy & 63) : 63 - (sy & 63);  // block number of y-major step j: (ph + j) >> 6
    const int phx = dx > 0 ? (sx & 63) : 63 - (sx & 63);
    nb = x_major ? (ni > 0 ? ((phx + ni - 1) >> 6) + 1 : 0) : (nj > 0 ? ((ph + nj - 1) >> 6) + 1 : 0);  // blocks of the march (Quad::Nbx / Nby)
    if (ni <= 0 || nj <= 0) n_strips = 0;
    else if (x_major) n_strips = (rows_total + kXRows - 1) / kXRows;
    else n_strips = cols_total > 0 ? (cols_total + ya + kYCols - 1) / kYCols : 0;
  }
  // first step of strip p
  VHP_FN int first_step(int p) const { return x_major ? kXRows * p : imax(kYCols * p - ya, 0); }
  // diagonal entries a y-major strip needs before it starts
  VHP_FN int diag_need(int p) const { return x_major ? 0 : imin(kYCols * p - ya + kYCols, rows_total); }
  // 64-entry blocks of boundary-line scratch: one line of nb blocks per strip that has a reader
  VHP_FN int line_blocks() const { return n_strips > 1 ? (n_strips - 1) * nb : 0; }
};

// ---------------------------------------------------------------------------------------------------------------
// The boundary line between a strip and its neighbours: what a strip reads from the strip below it and what it hands to
// the strip above.  D = direction of the marching coordinate, c0 = the source's coordinate: step i is at c0 + D i.
// A strip's tag: claim sequence number of its unit (9 bits) << 8 | strip.
// ---------------------------------------------------------------------------------------------------------------
template <int D>
struct Link {
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
  unsigned long long* pp;  // diagnostic builds (tools/lat_timeline.py): the wavefront's cycle accounts
#endif
  // reading side (nothing of it is used by the first strip of a unit)
  double* bin;             // entry of coordinate x of the current block at 1 + (x & 63); 0 / 65: the neighbours of the block
  const double* rd_ring;   // the ring of the wavefront that sweeps the strip below
  const int* rd_hdr;
  int rd_tag;
  const Tagged* line_in;   // the strip below's line in global memory: entry of x in block n at 64 n + (x & 63)
  int bin_block;           // the block that is in `bin` as a whole (from global memory), or -1
  int first_step;
  // The strip below is swept by ANOTHER workgroup (the latency sweep's long octants, vhp_band.hpp BandWorker: halves): no ring, no
  // header, no word of death in this workgroup's LDS -- its line in global memory is all there is, written window by window
  // (store_window) and read the same way (vhp_lat.hpp Below, remote), and beside the lines a record of its death {the step from
  // which its values are +0.0, epoch}, which the reader's wait for a window looks at (a dead writer stores no further entry).
  bool remote;
  const Tagged* death_in;  // the writer's record (one entry)
  int remote_dead;         // ... as last seen: 0 alive, else 1 + the step from which all its values are +0.0
  // writing side
  double* ring;            // mine: entry of x at x & 255
  int* hdr;
  int my_tag;
  Tagged* line_out;        // nullptr without a reader
  int* prog;               // my progress word in the context: steps swept
  uint64_t epoch;
  int c0;
  vi lane;

  VHP_FN int coord(int i) const { return c0 + D * i; }
  VHP_FN int block_of(int i) const { const int b = coord(i) >> 6; return D > 0 ? b - (c0 >> 6) : (c0 >> 6) - b; }

  VHP_FN void init(const Shared& sh, int w, int c0_, int first_step_, int tag, int* prog_, const Tagged* line_in_, Tagged* line_out_, uint64_t epoch_,
                   int below_w, int below_tag) {
    bin = sh.bin(w);
    ring = sh.ring(w);
    hdr = sh.hdr(w);
    my_tag = tag;
    prog = prog_;
    line_in = line_in_; line_out = line_out_; epoch = epoch_;
    c0 = c0_;
    first_step = first_step_;
    bin_block = -1;
    lane = lane_id();
    rd_ring = below_w >= 0 ? sh.ring(below_w) : nullptr;
    rd_hdr = below_w >= 0 ? sh.hdr(below_w) : nullptr;
    rd_tag = below_tag;
    remote = false;
    death_in = nullptr;
    remote_dead = 0;
    lds_publish(hdr, my_tag << 14);  // the ring is this strip's from here on (before its first entry is written)
  }
  // steps 0 .. steps-1 of this strip are swept: their boundary values are in the ring
  VHP_FN void publish(int steps) {
    lds_publish(hdr, (my_tag << 14) | steps);
    lds_publish(prog, steps);
  }
  // The sixteen entries of the window whose lowest coordinate is cw (march block nb) go to my line in global memory, out of the ring: what
  // a reader in another workgroup waits for, window by window (Below, remote) -- a block at a time it would start 64 steps late.
  VHP_FN void store_window(int nb, int cw) {
    wave_sync();
    const vi l = lane & (kLW16 - 1);
    g_store_tagged_if(lane < kLW16, line_out, l + (64 * nb + (cw & 63)), lds_load(ring, l + (cw & (kRing - 1))), epoch);
  }
  // the entry of step s in a line: its march block and its coordinate in the block (lane vector of steps)
  VHP_FN vi line_index(const vi& s) const {
    const vi x = s * D + c0;
    const vi nbv = D > 0 ? (x >> 6) - (c0 >> 6) : (-(x >> 6)) + (c0 >> 6);
    return nbv * 64 + (x & 63);
  }
  // the whole block nb (coordinate block blk = x >> 6) goes to my line in global memory, out of the ring
  VHP_FN void store_block(int nb, int blk) {
    wave_sync();
    g_store_tagged(line_out, lane + 64 * nb, lds_load(ring, lane + 64 * (blk & 3)), epoch);
  }
  // block nb of the strip below from global memory into the slab (it has been stored, or is about to be; the tag tells)
  // step index of the first coordinate of block nb >= 1
  VHP_FN int block_first_step(int nb) const { return D > 0 ? 64 * ((c0 >> 6) + nb) - c0 : c0 - (64 * ((c0 >> 6) - nb) + 63); }
  // the remote writer's record, read afresh: has it died, and at or before step `need`?
  VHP_FN bool remote_died_by(int need) {
    vd r;
    if (wave_all(g_load_tagged(death_in, vi(0), epoch, r))) remote_dead = lane0_int(r);
    return remote_dead != 0 && remote_dead - 1 <= need;
  }
  VHP_FN void load_block(int nb) {
    // the neighbour of the block's first step is the last entry of the line's previous block -- wanted (and written by the strip
    // below) if that step is one of mine.  (Out of memory, not out of what the slab held before: the latency sweep's windows
    // read the writer's ring directly and leave the slab alone.)
    vd carry = vd(0.0);
    if (nb >= 1 && block_first_step(nb) >= imax(first_step, 1))
      while (!wave_all(g_load_tagged(line_in, vi(64 * (nb - 1) + (D > 0 ? 63 : 0)), epoch, carry))) backoff();
    vd v;
    while (!wave_all(g_load_tagged(line_in, lane + 64 * nb, epoch, v))) backoff();
    wave_sync();
    lds_store(bin, lane + 1, v);
    lds_store(bin, vi(D > 0 ? 0 : 65), carry);
    wave_sync();
    bin_block = nb;
  }
  // Makes the boundary values of steps ia-1 .. ib of the strip below (all of block nb but possibly the first) readable in
  // the slab.  Fast: out of the writer's ring, with its header read before and after the copy.  Otherwise the whole
  // block from global memory.
  VHP_FN void fetch(int ia, int ib, int nb) {
    if (bin_block == nb) return;
    VHP_DIAG_NOWAIT_RETURN  // (vhp_diag.h: diagnostic builds in which no strip waits for another)
    for (;;) {
      const int h = lds_poll(rd_hdr);
      if ((h >> 14) != rd_tag) break;                 // the writer has finished that strip: its line is (being) stored
      const int steps = h & 0x3fff;
      if (steps <= ib) { ready_backoff(); continue; }  // not swept yet
      if (steps - (ia - 1) > kRingSafe) { sim_count(2); break; }  // far ahead: the ring may be overwritten any moment, the block is stored
      lds_acquire();
      sim_point();
      const vi st = vmin(lane + (ia - 1), ib);        // lane l: step ia - 1 + l (clamped: the same entry again)
      const vi x = st * D + c0;
      const vd v = lds_load(rd_ring, x & (kRing - 1));
      // the neighbour entry (step ia - 1) goes next to step ia's slot; the others to their own
      const vi slot = select(lane == 0, vi(1 + (coord(ia) & 63) - D), (x & 63) + 1);
      const int h2 = lds_poll(rd_hdr);
      if ((h2 >> 14) != rd_tag || (h2 & 0x3fff) - (ia - 1) > kRingSafe + 8) { sim_count(3); break; }  // (the writer may be 8 steps past what it has published)
      lds_store(bin, slot, v);
      wave_sync();
      sim_count(0);
      return;
    }
    sim_count(1);
    load_block(nb);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// x-major strip p of a unit: rows j = 64p + lane; steps i = 64p .. ni-1; cells (i, j), j <= i.  The step code is the
// streaming sweep's, round 2; the tile has two windows and the wavefront flushes it itself.
// ---------------------------------------------------------------------------------------------------------------
// ANYW: the build for widths that are not a multiple of 8 (flush_rows below); the build for the others contains none of it
template <int DX, int DY, typename OutT, bool ANYW>
struct XStrip {
  static constexpr int CB = sizeof(OutT);
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  double* tile;
  double* slab;   // reciprocals of the current block's 64 steps, indexed by x & 63
  double* dummy;  // where the lanes that are not the boundary lane "write" theirs
  int r_stride;   // 2: the row pitch is an odd multiple of 64 bytes, odd and even rows are half a line apart; 1: every row starts on a line
  static constexpr int kTS = ANYW ? kTStrideAny : kTStride;
  // ANYW: nx mod 16; the index of the field's first cell in memory mod 16; rows r and r + n_phase start at the same place in a line
  int nxm, cell0, n_phase;
  vi rr_a[4], rr_b[4];  // ANYW: the rows whose lines a window with (x & 8) == 0 completes / the other rows: row slot (lane >> 3) of store instruction q
  bool both_sets;       // ANYW: every row starts at the same place (a field off the line grid on a width that is a multiple of 16): ...
  bool both_on_even;    // ... both sets leave after the same windows, those with (x & 8) == 0 or the others
  int p, j0, rows_here;
  bool has_consumer;
  Link<DX> lk;        // the boundary lines: strip p-1's (read) and mine (written)
  double* bin;        // = lk.bin
  int pf_blk;         // the block (x >> 6) whose operands wait in ow_nx / rv_nx, or -1
  vi lane, tile_l, fl_t0, fl_hi;
  vu32 fl_off;
  vd prev, jd;
  vu64 ow, ow_nx;
  vd rv_nx;

  // (the caller has initialised lk)
  VHP_FN void init(const Map& m_, int sx, int sy, OutT* out_, const Shared& sh, int w, int p_) {
    m = m_; out = out_;
    g.init(m.nx, m.ny, sx, sy);
    tile = sh.lds + sh.L.tiles + w * kXRows * kTS;
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);  // (two blocks per wavefront: XStrip16 and the latency sweep keep the next block's too)
    bin = lk.bin;
    dummy = sh.lds + sh.L.dummies + w * 8;
    r_stride = ((m.nx >> 3) & 1) ? 2 : 1;
    nxm = m.nx & 15;
    cell0 = (int)((reinterpret_cast<uintptr_t>(out) / CB) & 15);
    n_phase = (nxm & 1) ? 16 : (nxm & 2) ? 8 : (nxm & 4) ? 4 : (nxm & 8) ? 2 : 1;
    lane = lane_id();
    tile_l = lane * kTS;
    {
      // flush geometry: lane -> (row slot = lane >> 3, piece = lane & 7 = cells xa + 2*piece, +1); the row slots of a
      // store instruction are counted upward in y, so that byte offsets from its lowest row are never negative
      const vi rslot = lane >> 3, pc = lane & 7;
      const vi rs = DY > 0 ? rslot : 7 - rslot;
      fl_t0 = rslot * (r_stride * kTStride) + ((pc * 2) & 7);
      fl_hi = pc >> 2;
      fl_off = to_u32((rs * (r_stride * m.nx) + pc * 2) * CB);
    }
    p = p_;
    j0 = kXRows * p;
    rows_here = imin(kXRows, g.rows_total - j0);
    has_consumer = p + 1 < g.Px;
    prev = vd(0.0);
    jd = to_f64(lane + j0);
    pf_blk = -1;
    if (ANYW) init_sets();
  }

  // the place of row r's first cell in its line
  VHP_FN int row_phase(int r) const { return (cell0 + (g.sy + DY * (j0 + r)) * nxm) & 15; }
  // A window [xw, xw + 8) completes the line of a row iff ((psi + xw) & 15) >= 8, psi = the row's phase marching up (the line's last
  // cell is marched last), the phase - 1 marching down (its first cell is).  The windows alternate, so the rows fall into two sets.
  static VHP_FN int psi_of(int phase) { return DX > 0 ? phase : (phase - 1) & 15; }
  VHP_FN void init_sets() {
    const vi rslot = lane >> 3;
    both_sets = n_phase == 1;
    both_on_even = psi_of(row_phase(0)) >= 8;
    if (both_sets) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { rr_a[q] = rslot + 8 * q; rr_b[q] = rslot + (32 + 8 * q); }
      return;
    }
    // the residues mod n_phase of the rows of either set, a nibble each
    int pack_a = 0, pack_b = 0, n_a = 0, n_b = 0;
    for (int c = 0; c < n_phase; ++c) {
      if (psi_of(row_phase(c)) >= 8) pack_a |= c << (4 * n_a++); else pack_b |= c << (4 * n_b++);
    }
    const int half = n_phase >> 1, lg = half == 8 ? 3 : half == 4 ? 2 : half == 2 ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const vi n = rslot + 8 * q;  // the n-th row of its set: residue n % half, period n / half
      const vi sh = (n & (half - 1)) * 4;
      const vi hi = (n >> lg) * n_phase;
      rr_a[q] = ((vi(pack_a) >> sh) & 15) + hi;
      rr_b[q] = ((vi(pack_b) >> sh) & 15) + hi;
    }
  }

  // the tile slot of column x (x >= -48): a ring of two windows, of three in the ANYW build
  static VHP_FN int ring_slot(int x) { return ANYW ? (x + 48) % 24 : (x & 15); }
  VHP_FN vi slot_of(int x) const { return tile_l + ring_slot(x); }

  // Emits one line of the rows r = r_first, r_first + r_stride, ... of this strip from the tile, 8 rows per store instruction.
  // (PRED: only the cells with step index j <= i' <= i_now.)
  template <bool PRED>
  VHP_FN void flush(int xa, int r_first, int i_now, int i_extra = 0) {
    VHP_DIAG_NOXSTORE_RETURN
    wave_sync();
    const int win = (xa >> 3) + 24;  // (xa may be -8 at the end of a march)
    const int sA = win & 1, sB = sA ^ 1;
    const vi t0 = fl_t0 + fl_hi * (8 * (sB - sA)) + (8 * sA + r_first * kTStride);
    const vu32 off = fl_off + (uint32_t)(xa * CB);
    const int t_step = 8 * r_stride * kTStride;  // per store instruction: 8 row slots further
    const long y_low = DY > 0 ? g.Y(j0 + r_first) : g.Y(j0 + r_first + 7 * r_stride);
    OutT* base = out + y_low * (long)m.nx;
    const long base_step = (long)(8 * r_stride * DY) * m.nx;
    if (!PRED && rows_here == kXRows) {
      vd a[8], b[8];
      if (r_stride == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = lds_load(tile, t0 + u * t_step); b[u] = lds_load(tile, t0 + (u * t_step + 1)); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { g_store2(base, off, a[u], b[u]); base += base_step; }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = lds_load(tile, t0 + u * t_step); b[u] = lds_load(tile, t0 + (u * t_step + 1)); }
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
          const vi jlo = jr;
          const vb ok0 = row_ok && (i0c >= jlo) && (i0c <= i_now + i_extra);
          const vb ok1 = row_ok && (i1c >= jlo) && (i1c <= i_now + i_extra);
          g_store2_if(ok0 && ok1, ok0, ok1, base, off, a, b);
        }
        base += base_step;
      }
    }
    wave_sync();
  }

  // ANYW build (a width that is not a multiple of 8, a field off the line grid): the 128-byte lines of a row (64-byte lines of fp32
  // cells: 16 cells either way) start anywhere, and each row's somewhere else.  The tile is a ring of THREE windows: a line that a
  // window completes is up to 7 columns behind the window's end and 16 long.  After every window the rows whose line it completed
  // -- every other window the same half of the rows: init_sets() -- store that line whole, 8 rows x 128 bytes per store instruction
  // like the other build, each row slot of an instruction with its own x.  A line is aligned in memory, so its pairs of cells are,
  // whatever the width's parity.
  // MODE 0: the line the window [xw, xw + 8) completed.  At the end of a march (xw = the window of its last step xe): MODE 1 the line
  // the march ended in; 2 the line before that, if the march's last window completed it (nobody has flushed it); 3 marching down, the
  // line of x = 0 if it is not the line of xe.
  template <bool PRED, int MODE>
  VHP_FN void flush_rows(const vi (&rr)[4], int xw, int xe, int i_now, int i_extra = 0) {
    VHP_DIAG_NOXSTORE_RETURN
    const vi pc2 = (lane & 7) * 2;
    const int sw = ring_slot(xw);
    OutT* base = out - 32;  // (offsets from 32 cells before the field: a line may begin before row 0's first cell)
    // (the arithmetic of a row slot is redone at every flush from the row's index alone: kept across the march for the 8 row slots
    // of a lane it would cost 40 registers; pin() keeps the compiler from doing just that)
    vd a[4], b[4];
    vu32 off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vi r = rr[q];
      pin(r);
      vb row_ok = r < rows_here;
      const vi y = (r + j0) * DY + g.sy;
      const vi ph = (y * nxm + cell0) & 15;
      vi xa;  // x of the line's first cell
      if (MODE == 0) {
        xa = DX > 0 ? ((ph + (xw + 8)) & ~15) - ph - 16 : ((ph + (xw + 7)) & ~15) - ph;
      } else {
        const vi cur = vi(xe) - ((ph + xe) & 15);
        if (MODE == 1) xa = cur;
        if (MODE == 2) { xa = cur - 16 * DX; row_ok = row_ok && (DX > 0 ? cur - 1 >= xw : cur + 16 <= xw + 7); }
        if (MODE == 3) { xa = cur - 16; row_ok = row_ok && (cur > xe - 1); }
      }
      vi s0 = (xa - xw) + sw;  // the ring slot of xa: -32 <= xa - xw < 24
      if (MODE != 0) { s0 = select(s0 < 0, s0 + 24, s0); s0 = select(s0 >= 24, s0 - 24, s0); }
      if (MODE != 0 || DX > 0) s0 = select(s0 < 0, s0 + 24, s0);
      vi sl0 = s0 + pc2;
      sl0 = select(sl0 >= 24, sl0 - 24, sl0);
      vi sl1 = sl0 + 1;
      sl1 = select(sl1 >= 24, sl1 - 24, sl1);
      const vi tix = select(row_ok, r * kTS, vi(0));
      a[q] = lds_load(tile, tix + sl0);
      b[q] = lds_load(tile, tix + sl1);
      const vi xc = xa + pc2;
      off[q] = to_u32((y * m.nx + xc + 32) * CB);
      if (PRED) {  // (one row slot at a time: the start of a strip and the end of a march)
        const vi jr = r + j0;
        const vi i0c = (xc - g.sx) * DX, i1c = (xc + 1 - g.sx) * DX;
        const vb ok0 = row_ok && (i0c >= jr) && (i0c <= i_now + i_extra);
        const vb ok1 = row_ok && (i1c >= jr) && (i1c <= i_now + i_extra);
        g_store2_if(ok0 && ok1, ok0, ok1, base, off[q], a[q], b[q]);
      }
    }
    if (!PRED) {  // (every row exists, every cell counts: four loads in flight, then four stores)
#pragma unroll
      for (int q = 0; q < 4; ++q) g_store2(base, off[q], a[q], b[q]);
    }
  }

  // After the step at x_b, which ends an 8-cell window of x: the rows whose 128-byte line this completes leave.
  // (y*nx + x) % 16 == 0 marks a line start; with nx = 8*m that is x % 16 == 8 * ((y*m) & 1).
  VHP_FN void flush_completed(int x_b, int i_now) {
    const int edge = DX > 0 ? x_b + 1 : x_b;
    const int hbit = (edge >> 3) & 1;
    const int xa = DX > 0 ? x_b - 15 : x_b;
    if (ANYW) {
      const int xw = x_b & ~7;  // the window [xw, xw + 8)
      const bool even = (xw & 8) == 0;
      if (both_sets && even != both_on_even) return;
      // (steady: the first-marched cell of a line that ends in the window is past every row's diagonal, and every row exists)
      const bool st = i_now - 22 >= j0 + kXRows - 1 && rows_here == kXRows;
      wave_sync();
      if (both_sets || even) { if (st) flush_rows<false, 0>(rr_a, xw, 0, i_now); else flush_rows<true, 0>(rr_a, xw, 0, i_now); }
      if (both_sets || !even) { if (st) flush_rows<false, 0>(rr_b, xw, 0, i_now); else flush_rows<true, 0>(rr_b, xw, 0, i_now); }
      wave_sync();
      return;
    }
    const bool steady = i_now - 15 >= j0 + kXRows - 1;  // the line's first-marched cell is past every row's diagonal
    int r_first = 0;
    if (r_stride == 1) {
      if (hbit != 0) return;
    } else {
      r_first = (hbit ^ g.sy ^ j0) & 1;  // rows with (y & 1) == hbit
    }
    if (steady) flush<false>(xa, r_first, i_now); else flush<true>(xa, r_first, i_now);
  }

  // the march of this strip is over: what is still in the tile leaves as partial lines
  VHP_FN void end_of_march() {
    const int i_now = g.ni - 1;
    const int xe = g.X(i_now);
    // Marching down, the march ends at x = 1: column 0 is never swept and reads as zero.  The zero leaves with
    // the last line of every row (one step "past" the march) instead of as a lone 8-byte store some other time.
    const int extra = DX < 0 ? 1 : 0;
    if (DX < 0) {
      wave_sync();
      lds_store(tile, slot_of(0), vd(0.0));  // x = 0: slot 0, column 0 of every row's ring
      wave_sync();
    }
    if (ANYW) {
      const int xw = xe & ~7;
      wave_sync();
      flush_rows<true, 1>(rr_a, xw, xe, i_now, extra);
      flush_rows<true, 1>(rr_b, xw, xe, i_now, extra);
      flush_rows<true, 2>(rr_a, xw, xe, i_now, extra);
      flush_rows<true, 2>(rr_b, xw, xe, i_now, extra);
      if (DX < 0) {
        flush_rows<true, 3>(rr_a, xw, xe, i_now, extra);
        flush_rows<true, 3>(rr_b, xw, xe, i_now, extra);
      }
      wave_sync();
    } else if (r_stride == 1) {
      flush<true>(xe & ~15, 0, i_now, extra);
    } else {
      for (int ph = 0; ph < 2; ++ph) {  // rows whose lines start at x % 16 == 8*ph
        const int xa = 8 * ph + (((xe - 8 * ph) >> 4) << 4);
        flush<true>(xa, (ph ^ g.sy ^ j0) & 1, i_now, extra);
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
    if (p > 0) { fill = bin[1 + (x & 63) - DX]; dsrc = bin[1 + (x & 63)]; }
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
    lds_store(tile, slot_of(x), v);
    if (has_consumer) lds_store_if(lane == 63, lk.ring, vi(x & (kRing - 1)), v);
  }

  // eight steps covering one aligned window of x; DIAG: the strip's diagonal may fall into it
  template <bool DIAG>
  VHP_FN void window8(int i0) {
    const int x0 = g.X(i0);
    const int t0 = x0 & 63;
    const int xw = x0 & ~7;  // lowest x of the window
    vd rr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rr[k] = lds_bcast(slab, (xw & 63) + (DX > 0 ? k : 7 - k));
    // the boundary row of the strip below: rb[k] = its value at x(i0 + k) - DX, the OLD neighbour of lane 0 at step k
    // (and rb[k + 1] the NEW one, which the diagonal cell of row j0 takes)
    vd rb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) rb[k] = vd(0.0);
    if (p > 0) {
      rb[0] = lds_bcast(bin, 1 + (x0 & 63) - DX);
      const int xb = 1 + (xw & 63);
#pragma unroll
      for (int k = 1; k < 9; ++k) rb[k] = lds_bcast(bin, xb + (DX > 0 ? k - 1 : 8 - k));
    }
    const vu32 hs = half_shifted(ow, t0, DX > 0 ? (t0 & 31) : (t0 & 31) - 7);  // step k's bit at position (x & 7)
    const vi tidx = tile_l + ring_slot(xw);  // (a window never straddles the ring's wrap)
    // every lane writes "its boundary value" each step -- lane 63 into the block's out slab, the others into a dummy
    // slot: one ds_write instead of an exec-masked region per step
    double* wbase = has_consumer ? lk.ring + (xw & (kRing - 1)) : dummy;
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
    VHP_EXP_PRIO_SET(lo < j0 + kXRows)
    const int blk = g.X(lo) >> 6;
    {
      vd rv;
      if (pf_blk == blk) { ow = ow_nx; rv = rv_nx; } else { load_block(blk, ow, rv); }
      pin(ow);
      pin(rv);
      lds_store(slab, lane, rv);
      wave_sync();
      if (nb + 1 < g.Nbx) { pf_blk = blk + DX; load_block(pf_blk, ow_nx, rv_nx); } else { pf_blk = -1; }
    }
    int i = lo;
    while (i <= hi) {
      const int x = g.X(i);
      const bool aligned = DX > 0 ? (x & 7) == 0 : (x & 7) == 7;
      int i_last;
      if (aligned && i + 7 <= hi) {
        if (p > 0) lk.fetch(i, i + 7, nb);
        if (i < j0 + kXRows) window8<true>(i); else window8<false>(i);
        i_last = i + 7;
      } else {
        if (p > 0) lk.fetch(i, i, nb);
        step1(i);
        i_last = i;
      }
      if (has_consumer) lk.publish(i_last + 1);
      i = i_last + 1;
      const int xl = g.X(i_last);
      const bool boundary = DX > 0 ? (xl & 7) == 7 : (xl & 7) == 0;
      if (boundary && i_last != g.ni - 1) flush_completed(xl, i_last);  // (the last step of the march is end_of_march's)
    }
    if (has_consumer) lk.store_block(nb, blk);
    VHP_DIAG_TL_XBLOCK(lo, hi, rows_here, j0, CB)
  }
};

// ---------------------------------------------------------------------------------------------------------------
// What a 16-step window reads from the strip below: v[0] = its value one step BEFORE the window's first-marched cell, v[k]
// (k >= 1) at step k - 1 of the window.  Straight out of the writer's ring, the writer's header read before and after (Link::fetch's
// check without its copy into the slab); the whole block from global memory if the writer is gone or too far ahead.
// (cw = lowest coordinate of the window, c_first = its first-marched coordinate.)
// ---------------------------------------------------------------------------------------------------------------
template <int D>
struct Below16 {
  static constexpr int NB = 17;
  bool ring;   // the values came out of the writer's ring and are still to be checked against h1 / h2
  int h1, h2;
  vd v[NB];

  VHP_FN void from_ring(const Link<D>& lk, int cw, int c_first) {
    h1 = lds_peek(lk.rd_hdr);
    sim_point();
    v[0] = lds_bcast(lk.rd_ring, (c_first - D) & (kRing - 1));
    const int rw = cw & (kRing - 1);
#pragma unroll
    for (int k = 1; k < NB; ++k) v[k] = lds_bcast(lk.rd_ring, rw + (D > 0 ? k - 1 : 16 - k));
    sim_point();
    h2 = lds_peek(lk.rd_hdr);
    ring = true;
  }
  VHP_FN void from_slab(const double* bin, int cw, int c_first) {
    v[0] = lds_bcast(bin, 1 + (c_first & 63) - D);
    const int b = 1 + (cw & 63);
#pragma unroll
    for (int k = 1; k < NB; ++k) v[k] = lds_bcast(bin, b + (D > 0 ? k - 1 : 16 - k));
    ring = false;
  }
  // the values of steps ia - 1 .. last_needed of the strip below (block nb), waiting for the writer if it has not got there
  VHP_FN void get(Link<D>& lk, const double* bin, int cw, int c_first, int ia, int last_needed, int nb) {
    if (!VHP_DIAG_WAITS) { from_slab(bin, cw, c_first); return; }
    if (lk.bin_block == nb) { from_slab(bin, cw, c_first); return; }
#ifndef VHP_POOL_NO_HDR_POLL   // (-DVHP_POOL_NO_HDR_POLL: every poll reads the 17 values as well, for A/B)
    {
      // a strip that follows its writer closely waits here window after window: on the header alone (one LDS read a poll), the
      // values only once they are there
      int h = lds_poll(lk.rd_hdr);
      while ((h >> 14) == lk.rd_tag && (h & 0x3fff) <= last_needed) { ready_backoff(); sim_point(); h = lds_poll(lk.rd_hdr); }
      lds_acquire();
    }
#endif
    from_ring(lk, cw, c_first);
    for (;;) {
      const int ha = uniform(h1), hb = uniform(h2);
      if ((ha >> 14) != lk.rd_tag) break;                         // the writer has finished that strip: its line is (being) stored
      if ((ha & 0x3fff) <= last_needed) {                          // not swept yet
        ready_backoff();
        from_ring(lk, cw, c_first);
        continue;
      }
      // (a writer is at most one window past what it has published: an entry of step s is safe while published - s <= kRingSafe)
      if ((hb >> 14) != lk.rd_tag || (hb & 0x3fff) - (ia - 1) > kRingSafe) { sim_count(3); break; }
      sim_count(0);
      ring = false;
      return;
    }
    lk.fetch(ia, last_needed, nb);
    from_slab(bin, cw, c_first);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// x-major strip p of a unit in windows of SIXTEEN steps (round 5; widths that are a multiple of 8): rows j = 64p + lane; steps
// i = 64p .. ni-1; cells (i, j), j <= i.  The window machinery is the latency sweep's (vhp_lat.hpp LatX): a window is 16 adjacent
// cells of every row, aligned to 16 cells of x; heads, tails and ragged ends are the same window code -- steps that do not
// exist leave garbage where garbage does no harm (a row is garbage until its diagonal cell switches it on; cells outside
// j <= i <= i_last are never stored; the reciprocal of a step that does not exist is 0) --, the boundary values of the strip below
// come straight out of its writer's ring, the ones this strip produces go to its ring once per window out of the tile's last row,
// and they are published BEFORE the window's own cells are stored.  A steady window is ~ 20 instructions a step all told where
// the 8-step windows of XStrip took ~ 35 (half of them around the steps: fetch, publish, flush decisions, loop), and a strip that
// is growing along its diagonal -- the chain of its unit -- hands over every 16 steps at 2/3 of the cost.
// What the latency sweep does not need and this does: WHOLE LINES.  On a pitch that is an odd multiple of 64 bytes every other row
// starts half a 128-byte line off the grid: the lines of such a row ("O" rows) are the cells [xw - 8, xw + 8) of two adjacent
// windows.  The tile stays 16 columns: an O row's line is read out of the tile in the MIDDLE of a window -- after step 7, when the
// window's first-marched half is new and the other half still holds the window before (the LDS executes a wavefront's
// instructions in order: the reads are issued before step 8's write) -- and stored with the others at the end; "E" rows (lines on
// the window grid) are read after step 15.  Every row leaves as whole lines, 8 rows x 128 bytes per store instruction.
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY, typename OutT>
struct XStrip16 {
  static constexpr int CB = sizeof(OutT);
  static constexpr int kW = 16;
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  double* tile;   // 64 rows x 16 columns (pitch kTStride): column c = x - (lowest x of the window)
  double* slab;   // reciprocals of the step indices of two blocks of 64 coordinates (the current one and the next), indexed by x & 127
  double* bin;    // = lk.bin
  Link<DX> lk;
  int p, j0, rows_here, i_first, i_last;
  bool below, has_consumer;
  // The two sets of rows: E (lines on the window grid: n_e = 4 or 8 store instructions a window) and O (lines half a window off:
  // four store instructions, if there are any): first row and row step.  (Where EVERY row is half a window off -- a pitch that is a
  // multiple of 128 bytes under fields that start half a line into one: a caller's buffer -- the rows go as E rows, in half lines.)
  int e_first, o_first, rstep, n_e;
  bool has_o;
  int blk;        // the current block (x >> 6): its occupancy words are in ow
  int pf_blk;     // the block whose operands wait in ow_nx / rv_nx (requested when the current block began), or -1 ...
  int staged_blk; // ... and the block whose reciprocals were put into the slab last
  bool pf_wait;   // the loads of ow_nx / rv_nx have not been waited for yet
  int tl_lo;      // (timeline builds: the first step of the current block that this strip sweeps)
  vi lane, tile_l, fl_e, fl_o;
  vu32 fl_off;
  vd prev, jd;
  vu64 ow, ow_nx;
  vd rv_nx;

  // (the caller has initialised lk)
  VHP_FN void init(const Map& m_, int sx, int sy, OutT* out_, const Shared& sh, int w, int p_) {
    m = m_; out = out_;
    g.init(m.nx, m.ny, sx, sy);
    tile = sh.lds + sh.L.tiles + w * kXRows * kTStride;
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);
    bin = lk.bin;
    lane = lane_id();
    tile_l = lane * kTStride;
    p = p_;
    j0 = kXRows * p;
    rows_here = imin(kXRows, g.rows_total - j0);
    i_first = j0;
    i_last = g.ni - 1;
    below = p > 0;
    has_consumer = p + 1 < g.Px;
    {
      // Which rows start on the line grid?  Row r's first cell is cell (cell0 + y nx) of the buffer; nx = 8 m: bit 3 of that is
      // (cell0 >> 3) + y m mod 2.  (cell0 is a multiple of 8 wherever the fields start on a line or half a line -- everywhere but in
      // a caller's buffer at an odd 16-byte offset, where no row has whole lines on any grid.)
      const int cell0 = (int)((reinterpret_cast<uintptr_t>(out) / CB) & 15);
      const int mm = (m.nx >> 3) & 1;
      const int ph0 = ((cell0 >> 3) + (g.sy + DY * j0) * mm) & 1;   // row 0 of the strip: 0 = E, 1 = O
      if (mm) { rstep = 2; n_e = 4; has_o = true; e_first = ph0; o_first = ph0 ^ 1; }
      else { rstep = 1; n_e = 8; has_o = false; e_first = o_first = 0; }
      // flush geometry: lane -> (row slot = lane >> 3, piece = lane & 7 = cells 2 * piece, + 1 of the line's 16); the row slots of a
      // store instruction are counted upward in y, so that byte offsets from its lowest row are never negative
      const vi rslot = lane >> 3, pc = lane & 7;
      const vi rs = DY > 0 ? rslot : 7 - rslot;
      fl_e = rslot * (rstep * kTStride) + pc * 2;
      fl_o = rslot * (rstep * kTStride) + ((pc * 2 + 8) & 15);
      fl_off = to_u32((rs * (rstep * m.nx) + pc * 2) * CB);
    }
    prev = vd(0.0);
    jd = to_f64(lane + j0);
    pf_blk = -1;
    pf_wait = false;
    staged_blk = -0x7fffffff;
    blk = -0x7fffffff;
    tl_lo = i_first;
  }

  // the occupancy word of every lane's row and the reciprocals of the step indices of the 64 coordinates of block blk (x >> 6)
  VHP_FN void load_ops(int b, vu64& o, vd& rv) {
    const vi yl = (vmin(lane + j0, g.rows_total - 1)) * DY + g.sy;
    o = g_load_u64(m.rows, yl * m.wpr + (1 + b));
    const vi it = (lane + (b * 64 - g.sx)) * DX;
    const vb ok = (it >= 0) && (it < g.ni);
    rv = select(ok, g_load_f64(m.recip, select(ok, it, vi(0))), vd(0.0));
  }
  VHP_FN bool block_in_march(int b) const { const int xe = g.X(i_last); return DX > 0 ? 64 * b <= xe : 64 * b + 63 >= xe; }
  VHP_FN void prefetch_ops(int b) { pf_blk = b; pf_wait = true; load_ops(b, ow_nx, rv_nx); }
  VHP_FN void stage(int b, vd rv) {
    wave_sync();
    lds_store(slab, lane + kBlock * (b & 1), rv);
    wave_sync();
    staged_blk = b;
  }
  VHP_FN void settle() {
    if (pf_wait) { pin(ow_nx); pin(rv_nx); pf_wait = false; }
  }
  VHP_FN void stage_next() {
    if (pf_blk != -1 && staged_blk != pf_blk) { settle(); stage(pf_blk, rv_nx); }
  }
  // block b becomes the current one (the first block of the strip, or the one after the current)
  VHP_FN void enter_block(int b) {
    if (pf_blk == b) { stage_next(); ow = ow_nx; }
    else { vd rv; load_ops(b, ow, rv); pin(ow); pin(rv); stage(b, rv); }
    blk = b;
    if (block_in_march(b + DX)) prefetch_ops(b + DX); else pf_blk = -1;
  }

  // the lane's pairs of FOUR store instructions' worth of lines out of the tile: instruction u = the rows first + rstep (8 u + row slot);
  // fl = fl_e (the line = the window's 16 columns) or fl_o (the line = columns 8 .. 15, then 0 .. 7)
  VHP_FN void load_lines4(int first, const vi& fl, vd (&fa)[4], vd (&fb)[4]) {
    const vi t0 = fl + first * kTStride;
#pragma unroll
    for (int u = 0; u < 4; ++u) { fa[u] = lds_load(tile, t0 + u * (8 * rstep * kTStride)); fb[u] = lds_load(tile, t0 + (u * (8 * rstep * kTStride) + 1)); }
  }
  // ... and into memory: s_base = the step index of the line's lowest-step cell, xa = its lowest x.  A cell (i', j) exists for
  // j <= i' <= lim.  An instruction is whole (every cell a cell of a row of this strip: one 16-byte store per lane), skipped (no row of
  // it has a cell in this line), or goes cell by cell.
  VHP_FN void store_lines4(int first, int s_base, int xa, int lim, const vd (&fa)[4], const vd (&fb)[4]) {
    VHP_DIAG_NOXSTORE_RETURN
    OutT* base = out + (long)(DY > 0 ? g.Y(j0 + first) : g.Y(j0 + first + rstep * 7)) * (long)m.nx + xa;
    const long base_step = (long)(8 * rstep * DY) * m.nx;
    if (rows_here == kXRows && s_base >= j0 + kXRows - 1 && s_base + kW - 1 <= lim) {
      // past every row's diagonal, inside the march, all 64 rows: every instruction whole
#pragma unroll
      for (int u = 0; u < 4; ++u) { g_store2(base, fl_off, fa[u], fb[u]); base += base_step; }
      return;
    }
    const int s_hi = imin(s_base + kW - 1, lim);
    const vi cc = (lane & 7) * 2;
    const vi s0 = DX > 0 ? cc + s_base : (-cc) + (kW - 1 + s_base), s1 = s0 + DX;  // step indices of the pair's two cells
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r_lo = first + rstep * 8 * u, r_hi = r_lo + rstep * 7;
      if (r_lo < rows_here && j0 + r_lo <= s_hi) {          // (else: no row of the instruction has a cell in this line)
        if (r_hi < rows_here && j0 + r_hi <= s_base && s_base + kW - 1 <= lim) {
          g_store2(base, fl_off, fa[u], fb[u]);             // whole
        } else {
          const vi r = (lane >> 3) * rstep + r_lo;
          const vb row_ok = r < rows_here;
          const vi jr = r + j0;
          const vb ok0 = row_ok && (s0 >= jr) && (s0 <= lim);
          const vb ok1 = row_ok && (s1 >= jr) && (s1 <= lim);
          g_store2_if(ok0 && ok1, ok0, ok1, base, fl_off, fa[u], fb[u]);
        }
      }
      base += base_step;
    }
  }

  // One window: steps ia + k, k = 0 .. 15, at x = xw + (k marching up, 15 - k marching down).  DIAG: the strip's diagonal may fall
  // into it (rows switch on one by one: the diagonal cell of row j takes the NEW value of the row below it times its own
  // occupancy).
  template <bool DIAG>
  VHP_FN void window(int ia, int xw, int nb) {
    const int k_hi = imin(kW - 1, i_last - ia);
    vd rr[kW];
#pragma unroll
    for (int k = 0; k < kW; ++k) rr[k] = lds_bcast(slab, (xw & (2 * kBlock - 1)) + (DX > 0 ? k : kW - 1 - k));
    Below16<DX> bl;
    if (below) bl.get(lk, bin, xw, DX > 0 ? xw : xw + kW - 1, imax(ia, i_first), ia + k_hi, nb);
    const vu32 hs = half_shifted(ow, xw & 63, xw & 31);  // the window's 16 occupancy bits: bit c = the cell at x = xw + c
    vd di = vd((double)ia);
    vd oa[4], ob[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { oa[u] = vd(0.0); ob[u] = vd(0.0); }
#pragma unroll
    for (int k = 0; k < kW; ++k) {
      const int c = DX > 0 ? k : kW - 1 - k;
      if (k == kW / 2 && has_o) {
        // the O rows' line [xw - 8, xw + 8) (marching down: [xw + 8, xw + 24)): this window's first-marched half is new, the other
        // half of the tile still holds the window before
        wave_sync();
        load_lines4(o_first, fl_o, oa, ob);
        wave_sync();
      }
      // bl.v[k] = the row below at x(step k) - DX, the OLD neighbour of lane 0 (bl.v[k + 1] the NEW one); strip 0 has none
      const vd b = shift_up(prev, below ? bl.v[k] : vd(0.0));
      const vi mk = sbfe1(hs, c);
      vd v = and_mask(stencil(prev, b, ratio(jd, di, rr[k])), mk);
      if (DIAG) {
        const vd up = shift_up(v, below ? bl.v[k + 1] : vd(1.0));  // strip 0: 1.0 = light strength at the origin
        const vb isd = lane == (ia + k - j0);
        const vd dcell = and_mask(up, mk);
        v = select(isd, dcell, v);
      }
      prev = v;
      lds_store(tile, tile_l + c, v);
      di = di + 1.0;
    }
    // ---- the window's boundary values go to the ring; its cells leave ----
    int lim = i_last;
    if (DX < 0 && xw == 0) {
      // Marching down, the march ends at x = 1: column 0 is never swept and reads as zero.  The zero leaves with
      // the last cells of every row (one step "past" the march) instead of as a lone 8-byte store some other time.
      wave_sync();
      lds_store(tile, tile_l, vd(0.0));
      lim = i_last + 1;
    }
    wave_sync();
    // The boundary values first, the window's own stores after: the strip above -- the one that is growing, the chain of the unit --
    // is let past its gate a window's worth of loads and stores earlier.
    if (has_consumer) {
      const vd bv = lds_load(tile, (lane & (kW - 1)) + (kXRows - 1) * kTStride);  // the last row: what the strip above reads
      wave_sync();
      lds_store(lk.ring, (lane & (kW - 1)) + (xw & (kRing - 1)), bv);  // (every lane: the four lanes of an entry write the same value)
      lk.publish(ia + k_hi + 1);
    }
    if (has_o) store_lines4(o_first, ia - kW / 2, DX > 0 ? xw - kW / 2 : xw + kW / 2, i_last, oa, ob);
    for (int h = 0; h < n_e; h += 4) {   // (four store instructions at a time: four loads in flight, then four stores)
      vd fa[4], fb[4];
      const int first = e_first + rstep * 8 * h;
      load_lines4(first, fl_e, fa, fb);
      store_lines4(first, ia, xw, lim, fa, fb);
    }
  }

  // After the last window (lowest step ia, lowest x xw): the half line of the O rows that no later window completes.
  VHP_FN void final_flush(int ia, int xw) {
    if (!has_o) return;
    int lim = i_last;
    if (DX < 0 && xw == 0) lim = i_last + 1;   // (the zero of x = 0 is in the tile's column 0: window())
    if (ia + kW / 2 > lim) return;              // (the march ended in the window's first-marched half: the mid-window lines had it all)
    wave_sync();
    vd oa[4], ob[4];
    load_lines4(o_first, fl_o, oa, ob);
    store_lines4(o_first, ia + kW / 2, DX > 0 ? xw + kW / 2 : xw - kW / 2, lim, oa, ob);
  }

  // the window at xw (lowest step ia) is next: if it opens a block, that block's operands become the current ones
  VHP_FN void open_block(int xw, int ia) {
    const int b = xw >> 6;
    if (b == blk) return;
    if (has_consumer) lk.store_block(g.nbx(ia - 1), blk);
    VHP_DIAG_TL_XBLOCK(tl_lo, ia - 1, rows_here, j0, CB)
    tl_lo = ia;
    VHP_EXP_PRIO_SET_LEFT(ia <= j0 + kXRows - 1, i_last - ia)
    enter_block(b);
  }
  VHP_FN void run() {
    int xw = g.X(i_first) & ~(kW - 1);
    int ia = DX > 0 ? xw - g.sx : g.sx - (xw + kW - 1);
    VHP_EXP_PRIO_SET(true)
    enter_block(xw >> 6);
    int ia_l = ia, xw_l = xw;
    while (ia <= i_last) {
      open_block(xw, ia);
      const int nb = DX > 0 ? blk - g.bx0 : g.bx0 - blk;
      if (ia <= j0 + kXRows - 1) window<true>(ia, xw, nb); else window<false>(ia, xw, nb);
      ia_l = ia; xw_l = xw;
      ia += kW; xw += kW * DX;
      sim_progress();
      sim_point();
    }
    if (has_consumer) lk.store_block(imax(g.nbx(imin(imax(ia - 1, 0), i_last)), 0), blk);
    VHP_DIAG_TL_XBLOCK(tl_lo, i_last, rows_here, j0, CB)
    final_flush(ia_l, xw_l);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// y-major strip q of a unit: columns i = 128q - ya + 2*lane + {0,1}; steps j = max(i0,0) .. nj-1; cells (i, j), i <= j
// (the diagonal cell is the seed diag(j), stored again with its neighbour).  the streaming sweep's step code (round 2); the seeds of the
// lane's two columns wait in registers.
// ---------------------------------------------------------------------------------------------------------------
#ifndef VHP_POOL_SPLIT_ENDS   // (the ends of a row piece that starts inside a sector stored plain: -DVHP_POOL_SPLIT_ENDS=0 for A/B)
#define VHP_POOL_SPLIT_ENDS 1
#endif
template <int DX, int DY, typename OutT, bool ANYW>
struct YStrip {
  static constexpr int CB = sizeof(OutT);
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  double* slab;   // reciprocals of the current block's 64 steps, indexed by y & 63
  double* dummy;
  int q, i0, jstart;
  int piece_lo;       // the lowest x of the strip's columns
  bool has_consumer, interior, odd_pitch;
  Link<DY> lk;        // the boundary lines, along y
  double* bin;        // = lk.bin
  vi lane, ia, ib;
  vd prev0, prev1, id0, id1, dg0, dg1;
  vu64 ow0, ow1;
  vu32 xoff;  // byte offset of the lane's pair inside a row

  // (the caller has initialised lk)
  VHP_FN void init(const Map& m_, int sx, int sy, OutT* out_, const Shared& sh, int w, int q_, const double* diag) {
    m = m_; out = out_;
    odd_pitch = ANYW && ((m.nx & 1) != 0 || (reinterpret_cast<uintptr_t>(out) & (2 * CB - 1)) != 0);
    g.init(m.nx, m.ny, sx, sy);
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);
    bin = lk.bin;
    dummy = sh.lds + sh.L.dummies + w * 8;
    lane = lane_id();
    q = q_;
    i0 = g.ycol0(q);
    jstart = g.ystart(q);
    has_consumer = q + 1 < g.Py;
    interior = i0 >= 0 && i0 + kYCols - 1 < g.ni;  // every column of the strip is a column of the grid
    ia = lane * 2 + i0;
    ib = ia + 1;
    prev0 = vd(0.0);
    prev1 = vd(0.0);
    id0 = to_f64(ia);
    id1 = to_f64(ib);
    // the seeds of my columns (columns that have none are never seeded: the index is only kept inside the line)
    dg0 = g_load_f64(diag, vmin(vmax(ia, 0), g.rows_total - 1));
    dg1 = g_load_f64(diag, vmin(vmax(ib, 0), g.rows_total - 1));
    pin(dg0);
    pin(dg1);
    const vi xlo = DX > 0 ? ia + g.sx : (-ib) + g.sx;  // the pair's lower x: x(ia) marching up, x(ib) marching down
    xoff = to_u32(xlo * CB);
    piece_lo = DX > 0 ? i0 + g.sx : g.sx - (i0 + kYCols - 1);
  }

  // is column c a column of the grid at or below the diagonal at step j?
  VHP_FN vb col_ok(const vi& c, int j) const { return (c >= 0) && (c < g.ni) && (c <= j); }
  // ... the lower-x cell of the pair of columns (ca, ca + 1)?  Marching down that is column ca + 1 -- or x = 0, which is never swept
  // (the march stops at x = 1) and reads as zero: whoever stores x = 1 stores that zero with it.
  VHP_FN vb lo_ok(const vi& ca, int j) const {
    if (DX > 0) return col_ok(ca, j);
    return col_ok(ca + 1, j) || ((ca + 1 == g.ni) && col_ok(ca, j));
  }

  // stores the lane's two cells of row y (step j).  PRED: only the cells that are columns of the grid at or below the diagonal
  template <bool PRED>
  VHP_FN void store_row(OutT* row, int j, vd v0, vd v1) {
    VHP_DIAG_NOYSTORE_RETURN
    if (DX < 0 && PRED) v1 = select((ib == g.ni), vd(0.0), v1);  // x = 0
    const vd lo = DX > 0 ? v0 : v1, hi = DX > 0 ? v1 : v0;        // in memory order
    vb oklo = vb(true), okhi = vb(true);
    if (PRED) { oklo = lo_ok(ia, j); okhi = col_ok(DX > 0 ? ib : ia, j); }
    if (ANYW && odd_pitch && !row_aligned(row)) {
      // An odd width: in every other row the pairs of the lanes are off the 16-byte grid, and the pairs that are on it are a
      // lane's higher cell with the next lane's lower one.  That is what leaves; the strip's first cell in memory goes alone.
      const vd nb = DX > 0 ? shift_down(lo, vd(0.0)) : shift_up(lo, vd(0.0));
      const vi nlane = DX > 0 ? lane + 1 : lane - 1;
      vb oknb = (nlane >= 0) && (nlane < 64);
      if (PRED) oknb = oknb && lo_ok(DX > 0 ? ia + 2 : ia - 2, j);
      // (offsets from two cells before the row: marching down, the lane past x = 1 stores the x = 0 of its neighbour, and its own
      // xoff is "negative" -- the 32-bit sum wraps back)
      if (!PRED && VHP_POOL_SPLIT_ENDS && CB == 8) {
        // (the 63 pairs between the piece's first and last cell: those of whole sectors with the nt bit -- see below; fp32
        // fields: 1 % slower that way at 1001^2, left as they were)
        const uint32_t kk = (static_cast<uint32_t>(reinterpret_cast<uintptr_t>(row)) + static_cast<uint32_t>((piece_lo + 1) * CB)) & 63u;
        const int head = (int)(((64u - kk) & 63u) / (2 * CB)), tail = 63 - (int)(((kk + 63u * (2 * CB)) & 63u) / (2 * CB));
        const vi mp = DX > 0 ? lane : 63 - lane;   // the pair's index in memory order; 63: the lane without a neighbour
        const vb end = (mp < head) || (mp >= tail);
        g_store2_mask(!end, row - 2, xoff + (uint32_t)(3 * CB), hi, nb);
        g_store2_if(end && oknb, end && !oknb, vb(false), row - 2, xoff + (uint32_t)(3 * CB), hi, nb);
      } else
        g_store2_if(okhi && oknb, okhi, oknb, row - 2, xoff + (uint32_t)(3 * CB), hi, nb);
      g_store1_if((lane == (DX > 0 ? 0 : 63)) && oklo, row, xoff, lo);
      return;
    }
    if (PRED) g_store2_if(oklo && okhi, oklo, okhi, row, xoff, lo, hi);
    // (on the other widths a row piece starts anywhere in a sector, so its two ends are parts of sectors as well: every unpredicated
    // y-major store of that build plain, measured: 1002^2 0.658 -> 0.670 ms, 1004^2 0.648 -> 0.668 -- the interior wants the nt bit)
    else if (ANYW && VHP_POOL_SPLIT_ENDS) {
      // The piece starts k bytes into a sector and ends k bytes into one (it is 8 or 16 sectors long): the lanes of those two
      // sectors store plain, as parts of sectors do everywhere else, the whole sectors between them with the nt bit.  256 sources,
      // fp64: 1002^2 0.670 -> 0.649 ms, 1500^2 1.274 -> 1.233, 1001^2 0.674 -> 0.658, 1501^2 1.278 -> 1.227
      // (profiles/r05_ab_row_piece_ends_plain.txt).
      const uint32_t k = (static_cast<uint32_t>(reinterpret_cast<uintptr_t>(row)) + static_cast<uint32_t>(piece_lo * CB)) & 63u;
      if (k == 0) { g_store2(row, xoff, lo, hi); return; }
      const int head = (int)((64u - k) / (2 * CB)), tail = 64 - (int)(k / (2 * CB));  // lanes, in memory order
      const vi ml = DX > 0 ? lane : 63 - lane;
      const vb end = (ml < head) || (ml >= tail);
      g_store2_mask(!end, row, xoff, lo, hi);
      g_store2_if(end, vb(false), vb(false), row, xoff, lo, hi);
    }
    else g_store2(row, xoff, lo, hi);
  }
  // (a strip's pairs start on an even x: a row is aligned or not as a whole)
  VHP_FN bool row_aligned(const OutT* row) const { return (reinterpret_cast<uintptr_t>(row) & (2 * CB - 1)) == 0; }
  VHP_FN OutT* row_ptr(int y) const { return out + (size_t)y * (size_t)m.nx; }

  VHP_FN void step1(int j) {
    const int y = g.Y(j);
    const int t = y & 63;
    const double rj = slab[t];
    const double dj = (double)j;
    double fill = 0.0;
    if (q > 0) fill = bin[1 + (y & 63) - DY];
    const vd b0 = shift_up(prev1, vd(fill));
    vd v0 = and_mask(stencil(prev0, b0, ratio(id0, dj, rj)), bit_mask(ow0, t));
    vd v1 = and_mask(stencil(prev1, prev0, ratio(id1, dj, rj)), bit_mask(ow1, t));
    if (j <= i0 + kYCols - 1) {  // column j (if this strip owns it) is seeded with diag(j)
      v0 = select(ia == j, dg0, v0);
      v1 = select(ib == j, dg1, v1);
    }
    store_row<true>(row_ptr(y), j, v0, v1);
    prev0 = v0;
    prev1 = v1;
    if (has_consumer) lds_store_if(lane == 63, lk.ring, vi(y & (kRing - 1)), v1);
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
      rb[0] = lds_bcast(bin, 1 + (y0 & 63) - DY);
      const int yq = 1 + (yb & 63);
#pragma unroll
      for (int k = 1; k < 8; ++k) rb[k] = lds_bcast(bin, yq + (DY > 0 ? k - 1 : 8 - k));
    }
    const int sh = DY > 0 ? (t0 & 31) : (t0 & 31) - 7;
    const vu32 hs0 = half_shifted(ow0, t0, sh), hs1 = half_shifted(ow1, t0, sh);
    double* wbase = has_consumer ? lk.ring + (yb & (kRing - 1)) : dummy;
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
        v0 = select(ia == j0w + k, dg0, v0);
        v1 = select(ib == j0w + k, dg1, v1);
      }
      store_row<PRED>(row, j0w + k, v0, v1);
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
    VHP_EXP_PRIO_SET(lo <= i0 + kYCols - 1)
    const int blk = g.Y(lo) >> 6;
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
    int j = lo;
    while (j <= hi) {
      const int y = g.Y(j);
      const bool aligned = DY > 0 ? (y & 7) == 0 : (y & 7) == 7;
      if (aligned && j + 7 <= hi) {
        if (q > 0) lk.fetch(j, j + 7, nb);
        if (j <= i0 + kYCols - 1) window8<true, true>(j);
        else if (!interior) window8<false, true>(j);
        else window8<false, false>(j);
        j += 8;
      } else {
        if (q > 0) lk.fetch(j, j, nb);
        step1(j);
        j += 1;
      }
      if (has_consumer) lk.publish(j);
    }
    if (has_consumer) lk.store_block(nb, blk);
    VHP_DIAG_TL_YBLOCK(lo, hi, i0, kYCols, g.ni, CB)
  }
};

// ---------------------------------------------------------------------------------------------------------------
// y-major strip q of a unit in windows of SIXTEEN steps (round 5; widths that are a multiple of 8): columns i = 128q - ya + 2 lane
// + {0, 1}; steps j = max(i0, 0) .. nj - 1; cells (i, j), i <= j.  XStrip16's machinery along y: windows aligned to 16 rows, heads and
// tails as ordinary windows (a column is garbage until its seed switches it on; a cell is stored from its column's seed step on and
// never past the march), the boundary values of the strip below straight out of its writer's ring, the next block's occupancy words
// and reciprocals requested a block ahead.  What a cell's store costs in the growing phase: one compare per column of the lane
// against a per-lane "first step" (the column's index; never for a column that is not one of the grid) instead of three range
// checks per cell.
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY, typename OutT>
struct YStrip16 {
  static constexpr int CB = sizeof(OutT);
  static constexpr int kW = 16;
  static constexpr int kNever = 0x7fffffff;
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  double* slab;   // reciprocals of the step indices of two blocks of 64 coordinates, indexed by y & 127
  double* bin;    // = lk.bin
  double* dummy;
  Link<DY> lk;
  int q, i0, j_first, j_last;
  bool below, has_consumer, interior;
  int blk, pf_blk, staged_blk;
  bool pf_wait;
  int tl_lo;
  vi lane, ia, ib;
  vi f0, f1;      // the first step at which the lane's cell of column ia / ib is stored (kNever: not a column of the grid)
  vd prev0, prev1, id0, id1, dg0, dg1;
  vu64 ow0, ow1, ow0_nx, ow1_nx;
  vd rv_nx;
  vu32 xoff;      // byte offset of the lane's pair inside a row

  // (the caller has initialised lk; the seeds of the strip's columns are in `diag`: find_work claims a strip only then)
  VHP_FN void init(const Map& m_, int sx, int sy, OutT* out_, const Shared& sh, int w, int q_, const double* diag) {
    m = m_; out = out_;
    g.init(m.nx, m.ny, sx, sy);
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);
    bin = lk.bin;
    dummy = sh.lds + sh.L.dummies + w * 8;
    lane = lane_id();
    q = q_;
    i0 = g.ycol0(q);
    j_first = g.ystart(q);
    j_last = g.nj - 1;
    below = q > 0;
    has_consumer = q + 1 < g.Py;
    interior = i0 >= 0 && i0 + kYCols - 1 < g.ni;  // every column of the strip is a column of the grid
    ia = lane * 2 + i0;
    ib = ia + 1;
    prev0 = vd(0.0);
    prev1 = vd(0.0);
    id0 = to_f64(ia);
    id1 = to_f64(ib);
    const vb real0 = (ia >= 0) && (ia < g.ni), real1 = (ib >= 0) && (ib < g.ni);
    // Marching down in x, the march stops at x = 1: column 0 is never swept and reads as zero.  "Column ni" (x = 0) is
    // the higher column of its lane's pair -- the pairs start on an even x --: it computes zeros (occupancy 0, seed 0) and is stored
    // with every row that stores x = 1.
    const vb zero1 = (DX < 0) ? ((ib == g.ni) && real0) : vb(false);
    f0 = select(real0, ia, vi(kNever));
    f1 = select(real1, ib, select(zero1, ia, vi(kNever)));
    // the seeds of my columns (columns that have none are never seeded: the index is only kept inside the line)
    dg0 = g_load_f64(diag, vmin(vmax(ia, 0), g.rows_total - 1));
    dg1 = g_load_f64(diag, vmin(vmax(ib, 0), g.rows_total - 1));
    pin(dg0);
    pin(dg1);
    dg1 = select(zero1, vd(0.0), dg1);
    const vi xlo = DX > 0 ? ia + g.sx : (-ib) + g.sx;  // the pair's lower x: x(ia) marching up, x(ib) marching down
    xoff = to_u32(xlo * CB);
    pf_blk = -1;
    pf_wait = false;
    staged_blk = -0x7fffffff;
    blk = -0x7fffffff;
    tl_lo = j_first;
  }

  // the occupancy words of the lane's two columns and the reciprocals of the step indices of the 64 coordinates of block b (y >> 6)
  VHP_FN void load_ops(int b, vu64& o0, vu64& o1, vd& rv) {
    const vi xa = vmin(vmax(ia, 0), g.ni - 1) * DX + g.sx;
    const vi xb = vmin(vmax(ib, 0), g.ni - 1) * DX + g.sx;
    o0 = g_load_u64(m.cols, xa * m.wpc + (1 + b));
    o1 = g_load_u64(m.cols, xb * m.wpc + (1 + b));
    if (DX < 0) o1 = select(ib == g.ni, vu64(0), o1);  // ("column ni" computes zeros: blocked all the way)
    const vi jt = (lane + (b * 64 - g.sy)) * DY;
    const vb ok = (jt >= 0) && (jt < g.nj);
    rv = select(ok, g_load_f64(m.recip, select(ok, jt, vi(0))), vd(0.0));
  }
  VHP_FN bool block_in_march(int b) const { const int ye = g.Y(j_last); return DY > 0 ? 64 * b <= ye : 64 * b + 63 >= ye; }
  VHP_FN void prefetch_ops(int b) { pf_blk = b; pf_wait = true; load_ops(b, ow0_nx, ow1_nx, rv_nx); }
  VHP_FN void stage(int b, vd rv) {
    wave_sync();
    lds_store(slab, lane + kBlock * (b & 1), rv);
    wave_sync();
    staged_blk = b;
  }
  VHP_FN void settle() {
    if (pf_wait) { pin(ow0_nx); pin(ow1_nx); pin(rv_nx); pf_wait = false; }
  }
  VHP_FN void stage_next() {
    if (pf_blk != -1 && staged_blk != pf_blk) { settle(); stage(pf_blk, rv_nx); }
  }
  VHP_FN void enter_block(int b) {
    if (pf_blk == b) { stage_next(); ow0 = ow0_nx; ow1 = ow1_nx; }
    else { vd rv; load_ops(b, ow0, ow1, rv); pin(ow0); pin(ow1); pin(rv); stage(b, rv); }
    blk = b;
    if (block_in_march(b + DY)) prefetch_ops(b + DY); else pf_blk = -1;
  }

  // One window: steps ja + k, k = 0 .. 15, at y = yw + (k marching up, 15 - k marching down).  DIAG: columns may be seeded in it
  // (implies PRED); PRED: predicated stores (columns that do not exist, or not yet; steps past the march).
  template <bool DIAG, bool PRED>
  VHP_FN void window(int ja, int yw, int nb) {
    const int k_hi = imin(kW - 1, j_last - ja);
    vd rr[kW];
#pragma unroll
    for (int k = 0; k < kW; ++k) rr[k] = lds_bcast(slab, (yw & (2 * kBlock - 1)) + (DY > 0 ? k : kW - 1 - k));
    Below16<DY> bl;
    if (below) bl.get(lk, bin, yw, DY > 0 ? yw : yw + kW - 1, imax(ja, j_first), ja + k_hi - 1, nb);
    const vu32 hs0 = half_shifted(ow0, yw & 63, yw & 31), hs1 = half_shifted(ow1, yw & 63, yw & 31);
    // every lane writes "its boundary value" each step -- lane 63 into the ring, the others into a dummy slot: one ds_write instead
    // of an exec-masked region per step
    double* wbase = has_consumer ? lk.ring + (yw & (kRing - 1)) : dummy;
    const vi widx = select(lane == 63, vi(0), vi((int)(dummy - wbase)));
    vd dj = vd((double)ja);
    OutT* row = out + (long)g.Y(ja) * (long)m.nx;
    const long rowstep = (long)DY * m.nx;
#pragma unroll
    for (int k = 0; k < kW; ++k) {
      const int c = DY > 0 ? k : kW - 1 - k;
      const vd b0 = shift_up(prev1, below ? bl.v[k] : vd(0.0));
      vd v0 = and_mask(stencil(prev0, b0, ratio(id0, dj, rr[k])), sbfe1(hs0, c));
      vd v1 = and_mask(stencil(prev1, prev0, ratio(id1, dj, rr[k])), sbfe1(hs1, c));
      if (DIAG) {
        v0 = select(ia == ja + k, dg0, v0);
        v1 = select(ib == ja + k, dg1, v1);
      }
      VHP_DIAG_NOYSTORE_GUARD
      {
        const vd lo = DX > 0 ? v0 : v1, hi = DX > 0 ? v1 : v0;  // in memory order
        if (PRED) {
          const int j = ja + k <= j_last ? ja + k : -1;         // (past the march: nothing)
          const vb ok0 = f0 <= j, ok1 = f1 <= j;
          const vb oklo = DX > 0 ? ok0 : ok1, okhi = DX > 0 ? ok1 : ok0;
          g_store2_if(oklo && okhi, oklo, okhi, row, xoff, lo, hi);
        } else {
          g_store2(row, xoff, lo, hi);
        }
      }
      prev0 = v0;
      prev1 = v1;
      lds_store(wbase, widx + c, v1);
      dj = dj + 1.0;
      row += rowstep;
    }
    if (has_consumer) lk.publish(ja + k_hi + 1);
  }

  VHP_FN void open_block(int yw, int ja) {
    const int b = yw >> 6;
    if (b == blk) return;
    if (has_consumer) lk.store_block(g.nby(ja - 1), blk);
    VHP_DIAG_TL_YBLOCK(tl_lo, ja - 1, i0, kYCols, g.ni, CB)
    tl_lo = ja;
    VHP_EXP_PRIO_SET_LEFT(ja <= i0 + kYCols - 1, j_last - ja)
    enter_block(b);
  }
  VHP_FN void run() {
    int yw = g.Y(j_first) & ~(kW - 1);
    int ja = DY > 0 ? yw - g.sy : g.sy - (yw + kW - 1);
    VHP_EXP_PRIO_SET(true)
    enter_block(yw >> 6);
    while (ja <= j_last) {
      open_block(yw, ja);
      const int nb = DY > 0 ? blk - g.by0 : g.by0 - blk;
      if (ja <= i0 + kYCols - 1) window<true, true>(ja, yw, nb);
      else if (!interior || ja + kW - 1 > j_last) window<false, true>(ja, yw, nb);
      else window<false, false>(ja, yw, nb);
      ja += kW; yw += kW * DY;
      sim_progress();
      sim_point();
    }
    if (has_consumer) lk.store_block(imax(g.nby(imin(imax(ja - 1, 0), j_last)), 0), blk);
    VHP_DIAG_TL_YBLOCK(tl_lo, j_last, i0, kYCols, g.ni, CB)
  }
};

// ---------------------------------------------------------------------------------------------------------------
// The diagonal of a quadrant for its y-major unit (the two-term recurrence below), into the unit's scratch line in global
// memory, 64 entries per call.  diag(0) = occ(source); for k >= 1:
//   sub(k)  = V(k, k-1) = (a - c*(a - b)) * occ(k, k-1),  a = diag(k-1), b = sub(k-1), c = (k-1)/k
//   diag(k) = sub(k) * occ(k, k)                                            (the stale diagonal)
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY>
struct DiagTask {
  Map m;
  Quad<DX, DY> g;
  double* diag;
  int k;
  vi lane;
  vd dprev, sprev;

  VHP_FN void init(const Map& m_, int sx, int sy, double* diag_) {
    m = m_;
    g.init(m.nx, m.ny, sx, sy);
    diag = diag_;
    lane = lane_id();
    k = 0;
    dprev = vd(0.0);
    sprev = vd(0.0);
  }
  VHP_FN bool done() const { return k >= g.rows_total; }
  // entries k .. k+63; returns the number of entries ready afterwards
  VHP_FN int run_chunk() {
    const int k0 = k, k1 = imin(k0 + kBlock, g.rows_total);
    const vi kk = vmin(lane + k0, g.rows_total - 1);
    const vi x = kk * DX + g.sx;
    const vi ya = vmax(kk - 1, 0) * DY + g.sy, yb = kk * DY + g.sy;
    const vu64 wa = g_load_u64(m.rows, ya * m.wpr + ((x >> 6) + 1));
    const vu64 wb = g_load_u64(m.rows, yb * m.wpr + ((x >> 6) + 1));
    const vd rk = g_load_f64(m.recip, kk);
    const vi ma = bit_mask_lane(wa, x & 63), mb = bit_mask_lane(wb, x & 63);
    const vd cv = ratio(to_f64(vmax(kk - 1, 0)), to_f64(kk), rk);  // (k-1)/k of every entry of the chunk: lane work, off the chain
    vd acc = vd(0.0);
    for (int kq = k0; kq < k1; ++kq) {
      const int l = kq - k0;
      vd dcur;
      if (kq == 0) {
        dcur = and_mask(vd(1.0), vi(read_lane_i(mb, l)));  // the origin: light strength 1 times its occupancy
        sprev = vd(0.0);
      } else {
        const vd c = vd(read_lane(cv, l));
        const vd sub = and_mask(stencil(dprev, sprev, c), vi(read_lane_i(ma, l)));
        dcur = and_mask(sub, vi(read_lane_i(mb, l)));
        sprev = sub;
      }
      dprev = dcur;
      acc = select(lane == l, dcur, acc);
    }
    g_store_f64_if(lane < (k1 - k0), diag, lane + k0, acc);
    k = k1;
    return k1;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// One wavefront of the pool.
// ---------------------------------------------------------------------------------------------------------------
enum { kFound = 0, kRetry = 1, kIdle = 2, kExit = 3 };


// the launch needs the ANYW build of the kernel: rows that do not start on a 64-byte half line (a width that is not a multiple of 8),
// or fields whose pairs of cells are not aligned to their size
template <typename OutT>
VHP_HD bool pool_needs_anyw(int nx, long long field_stride, const OutT* out) {
  return (nx & 7) != 0 || (field_stride & 1) != 0 || (reinterpret_cast<uintptr_t>(out) & (2 * sizeof(OutT) - 1)) != 0;
}

#ifndef VHP_POOL_X8   // (-DVHP_POOL_X8: the 8-step x-major strips of rounds 3-4 in every build, for A/B)
#define VHP_POOL_X16 1
#else
#define VHP_POOL_X16 0
#endif

template <typename OutT, bool ANYW = false>
struct Worker {
  static constexpr bool kUseX16 = !ANYW && VHP_POOL_X16 != 0;
#ifdef VHP_POOL_Y8   // (-DVHP_POOL_Y8: the 8-step y-major strips beside the 16-step x-major ones, for A/B)
  static constexpr bool kUseY16 = false, kMayY16 = false;
#else
  static constexpr bool kUseY16 = kUseX16;
  // The build for the other widths keeps the 8-step strips.  (YStrip16 stores aligned pairs of cells and nothing else about a row's
  // place in its lines, so it could serve every even width there; measured, 256 sources, 8-step / 16-step y-major strips: 1002^2
  // 0.686 / 0.700 ms, 1004^2 0.647 / 0.665 -- those launches are bound by the memory's handling of their row pieces, not by
  // instructions: -DVHP_POOL_ANYW_Y16 builds it.)
#ifdef VHP_POOL_ANYW_Y16
  static constexpr bool kMayY16 = ANYW && VHP_POOL_X16 != 0;
#else
  static constexpr bool kMayY16 = false;
#endif
#endif
  Args<OutT> a;
  Shared sh;
  int w, group;
  bool pairs_aligned;   // every pair of cells (x even, x + 1) of every field is aligned to its size
  vi lane;

  // (group: the index of the workgroup, 0 .. Args::n_groups - 1)
  VHP_FN void init(const Args<OutT>& a_, double* lds, const Layout& L, int w_, int group_) {
    a = a_;
    pairs_aligned = (a_.m.nx & 1) == 0 && (a_.field_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(a_.out) & (2 * sizeof(OutT) - 1)) == 0;
    sh.lds = lds;
    sh.L = L;
    w = w_;
    group = group_;
    lane = lane_id();
  }

  // Before any wavefront runs: every thread of the workgroup calls this (tid of nthreads), then a barrier.
  static VHP_FN void clear(double* lds, const Layout& L, int tid, int nthreads) {
    Shared s;
    s.lds = lds;
    s.L = L;
    int* sc = s.sched();
    const int n = kSchedHead + L.C * L.ctx_stride;
    for (int k = tid; k < n; k += nthreads) lds_set_int(sc + k, 0);
  }

  // Finds a strip to sweep, or installs a unit, or tells that there is nothing left.  A strip may be claimed when its
  // predecessor has finished the strip's first block (and, y-major, its seeds are there); the oldest context goes first.
  // A running strip waits only for strips that were claimed before it (nothing is ever overwritten: see the header), so
  // whatever the wavefronts do, the oldest running strip of every context can always finish.
  VHP_FN int find_work(int& c_out, int& p_out, int& qo_out, int& sx_out, int& sy_out) {
    int* sc = sh.sched();
    int best_c = -1, best_seq = 0x7fffffff, best_p = 0, best_word = 0, best_qo = 0, best_sx = 0, best_sy = 0;
    int free_c = -1, free_first = -1;
    bool unclaimed = false, installing = false;
    const bool late_open = a.early_ctx >= sh.L.C || lds_poll(sc + kLateOpen) != 0;  // (raised by this workgroup's own pulls: install)
    const int first_done = a.static_round ? lds_poll(sc + kFirstDone) : -1;  // (a context's first unit does not come out of the queue)
    for (int c = 0; c < sh.L.C; ++c) {
      int* cx = sh.ctx(c);
      const int st = lds_poll(cx + kState);
      if (st == 0) {
        if (((first_done >> c) & 1) == 0) free_first = c;
        else if (c < a.early_ctx || late_open) free_c = c;
        continue;
      }
      if (st == 1) { installing = true; continue; }
      const int word = lds_poll(cx + kWord);
      if (word < 0) continue;  // being recycled
      const int p = word & 255, seq = word >> 8;
      const int ns = lds_poll(cx + kNStrips);
      if (p >= ns) continue;
      unclaimed = true;
      const int rank = c < a.n_head ? c : seq + 16;  // the large units of the head contexts first, then the oldest
      if (rank >= best_seq) continue;
      const int unit = lds_poll(cx + kUnit), sxsy = lds_poll(cx + kSxSy);
      const int qo = unit & 7, sx = sxsy & 0xffff, sy = sxsy >> 16;
      UnitGeo ug;
      ug.init(a.m.nx, a.m.ny, qo, sx, sy);
      // the strip below has (all but) swept my first window.  claim_ahead > 0: claimed that many steps EARLIER -- the claim, the
      // strip's set-up and the loads of its first block (a round trip to memory) then run beside the strip below instead of
      // behind it; the new strip waits for its first boundary values inside fetch(), for a strip claimed before it as ever
      if (VHP_DIAG_WAITS && p > 0 && lds_poll(sh.prog(c) + (p - 1)) < ug.first_step(p) + 8 + 1 - a.claim_ahead) continue;
      if (VHP_DIAG_WAITS && !ug.x_major && lds_poll(cx + kDiagReady) < ug.diag_need(p)) continue;
      best_c = c; best_seq = rank; best_p = p; best_word = word; best_qo = qo; best_sx = sx; best_sy = sy;
    }
    if (best_c >= 0) {
      sim_point();
      if (lds_cas(sh.ctx(best_c) + kWord, best_word, best_word + 1) != best_word) return kRetry;
      lds_set_int(sh.owner(best_c) + best_p, w);
      c_out = best_c; p_out = best_p; qo_out = best_qo; sx_out = best_sx; sy_out = best_sy;
      sim_progress();
      return kFound;
    }
    const bool q_empty = lds_poll(sc + kQEmpty) != 0;
    // Another unit only while the workgroup is short of work: a CU that holds a large unit (many strips in flight) keeps
    // its whole share of the store path for it -- the march of a full-size octant is the longest dependent chain of the
    // launch, and every other unit on its CU slows each of its steps down.
    if (free_first >= 0) free_c = free_first;
    if (free_c >= 0 && (!q_empty || free_first >= 0) && lds_poll(sc + kBusy) < a.busy_cap) {
      sim_point();
      if (lds_cas(sh.ctx(free_c) + kState, 0, 1) == 0) {
        install(free_c);
      }
      return kRetry;
    }
    if (q_empty && !unclaimed && !installing && free_first < 0) return kExit;
    return kIdle;
  }

  // Takes the next unit of the queue into context c (state 1: mine).
  VHP_FN void install(int c) {
    int* sc = sh.sched();
    int* cx = sh.ctx(c);
    int idx;
    if (a.static_round && ((lds_poll(sc + kFirstDone) >> c) & 1) == 0) {
      // the context's first unit: by workgroup index (only the wavefront that holds the context in state 1 gets here)
      lds_or(sc + kFirstDone, 1 << c);
      // (odd head contexts count down: workgroup g's second unit is the (2 G - 1 - g)-th longest -- the longest units share their CU,
      // and its bandwidth, with the shortest of the round, and the round's bytes are spread evenly over the CUs)
      const int slot = (a.static_snake && (c & 1)) ? a.n_groups - 1 - group : group;
      idx = c < a.n_head ? c * a.n_groups + slot : a.n_units - 1 - ((c - a.n_head) * a.n_groups + group);
      sim_progress();
    } else {
      // (the small end only feeds the gaps beside the large units: once its share is gone, what is left leaves in size order,
      // largest first, so that the launch ends on its smallest units and not on whatever the two ends met at)
      bool from_tail = c >= a.n_head;
      if (from_tail && (int)(g_peek_u64(a.queue) >> 32) >= a.tail_limit) from_tail = false;
      const unsigned long long old = g_add_u64(a.queue, from_tail ? (1ull << 32) : 1ull);
      const unsigned taken_head = (unsigned)old, taken_tail = (unsigned)(old >> 32);
      idx = from_tail ? a.n_units - 1 - (int)taken_tail : (int)taken_head;
      if ((int)(taken_head + taken_tail) >= a.late_after) lds_publish(sc + kLateOpen, 1);
      sim_progress();
      if (taken_head + taken_tail >= (unsigned)a.n_units) {
        lds_publish(sc + kQEmpty, 1);
        lds_publish(cx + kState, 0);
        return;
      }
    }
    int unit, sxsy, lbase, rsvd;
    g_load_rec4(a.recs, idx, unit, sxsy, lbase, rsvd);
    const int s = unit / kUnits, qo = unit - s * kUnits;
    if (sxsy < 0) {  // units of a rejected source do nothing (-2: the launch's boundary lines do not fit its scratch -- vhp_pool_order has said so in the error flag -- and no unit is swept)
      if (qo == 0 && sxsy == -1) g_store_scalar_if(lane == 0, a.err_flag, vi(0), 1);
      lds_publish(cx + kState, 0);
      return;
    }
    const int sx = sxsy & 0xffff, sy = sxsy >> 16;
    VHP_DIAG_TL_UNIT(unit, 0)
    UnitGeo ug;
    ug.init(a.m.nx, a.m.ny, qo, sx, sy);
    OutT* field = a.out + (size_t)s * a.field_stride;
    if (qo == 0) {
      // rows/columns no quadrant covers read as zero; the x-major unit of quadrant 1 always exists
      // (column 0, x = 0, y >= 1: written as zero by the units that march down to x = 1, with their last store of the row)
      if (sy > 0)
        for (int x0 = 0; x0 < a.m.nx; x0 += kLanes) g_store_scalar_if(lane + x0 < a.m.nx, field, lane + x0, OutT(0));
    }
    if (ug.n_strips == 0) {
      lds_publish(cx + kState, 0);
      return;
    }
    lds_set_int(cx + kUnit, unit);
    lds_set_int(cx + kNStrips, ug.n_strips);
    lds_set_int(cx + kLeft, ug.n_strips);
    lds_set_int(cx + kSxSy, sxsy);
    lds_set_int(cx + kDiagReady, 0);
    lds_set_int(cx + kLineBase, lbase);
    for (int k0 = 0; k0 < ug.n_strips; k0 += kLanes) lds_store_i_if(lane + k0 < ug.n_strips, sh.prog(c), lane + k0, 0);
    const int seq = lds_add(sc + kSeq, 1) + 1;
    lds_publish(cx + kWord, seq << 8);
    lds_publish(cx + kState, 2);
    if (!ug.x_major) {
      double* dline = a.diag + (size_t)(4 * s + (qo >> 1)) * a.diag_stride;
      switch (qo >> 1) {
        case 0: run_diag<+1, +1>(cx, sx, sy, dline); break;
        case 1: run_diag<-1, +1>(cx, sx, sy, dline); break;
        case 2: run_diag<-1, -1>(cx, sx, sy, dline); break;
        default: run_diag<+1, -1>(cx, sx, sy, dline); break;
      }
    }
  }

  template <int DX, int DY>
  VHP_FN void run_diag(int* cx, int sx, int sy, double* dline) {
    DiagTask<DX, DY> dt;
    dt.init(a.m, sx, sy, dline);
    VHP_EXP_PRIO_TASK_BEGIN
    while (!dt.done()) {
      const int ready = dt.run_chunk();
      stores_done();  // the entries are in memory (L2) before the count says so: the strips that load them run on this CU
      lds_publish(cx + kDiagReady, ready);
      sim_progress();
      sim_point();
    }
    VHP_EXP_PRIO_END
  }

  // A strip of context c is finished: count down, free the context after the last.
  VHP_FN void strip_done(int c) {
    int* cx = sh.ctx(c);
    if (lds_add(cx + kLeft, -1) == 1) {
      VHP_DIAG_TL_UNIT(lds_int_at(cx + kUnit), 1)
      lds_publish(cx + kWord, -1);
      lds_publish(cx + kState, 0);
    }
    sim_progress();
  }

  // the boundary line of strip p of the unit in context c
  VHP_FN Tagged* line_of(int c, int p, int nb) const { return a.lines + (size_t)64 * ((size_t)lds_int_at(sh.ctx(c) + kLineBase) + (size_t)p * nb); }
  // tag of strip p of the unit in context c (claim sequence number of the unit, strip)
  VHP_FN int tag_of(int c, int p) const { return (((lds_int_at(sh.ctx(c) + kWord) >> 8) & 0x1ff) << 8) | p; }  // 17 bits: tag << 14 stays positive

  template <int DX, int DY>
  VHP_FN void run_x(int c, int unit, int p, int sx, int sy, OutT* field) {
    Quad<DX, DY> g;
    g.init(a.m.nx, a.m.ny, sx, sy);
    int* mine = sh.prog(c) + p;
    if (kUseX16) {
      // (widths that are a multiple of 8: windows of 16 steps)
      XStrip16<DX, DY, OutT> xs;
      xs.lk.init(sh, w, sx, kXRows * p, tag_of(c, p), mine, p > 0 ? line_of(c, p - 1, g.Nbx) : nullptr, p + 1 < g.Px ? line_of(c, p, g.Nbx) : nullptr,
                 a.epoch, p > 0 ? lds_int_at(sh.owner(c) + (p - 1)) : -1, p > 0 ? tag_of(c, p - 1) : 0);
      xs.init(a.m, sx, sy, field, sh, w, p);
      VHP_DIAG_TL_STRIPS(1)
      xs.run();
    } else {
      XStrip<DX, DY, OutT, ANYW> xs;
      xs.lk.init(sh, w, sx, kXRows * p, tag_of(c, p), mine, p > 0 ? line_of(c, p - 1, g.Nbx) : nullptr, p + 1 < g.Px ? line_of(c, p, g.Nbx) : nullptr,
                 a.epoch, p > 0 ? lds_int_at(sh.owner(c) + (p - 1)) : -1, p > 0 ? tag_of(c, p - 1) : 0);
      xs.init(a.m, sx, sy, field, sh, w, p);
      VHP_DIAG_TL_STRIPS(1)
      for (int n = p; n < xs.g.Nbx; ++n) {
        xs.sweep_block(n);
        if (n == xs.g.Nbx - 1) xs.end_of_march();
        sim_progress();
        sim_point();
      }
    }
    lds_publish(mine, 0x3fff);  // finished: whatever the strip above needs to start is there (a march can end before its first window does)
    VHP_DIAG_TL_STRIPS(-1)
    strip_done(c);
  }

  template <int DX, int DY>
  VHP_FN void run_y(int c, int unit, int q, int sx, int sy, OutT* field, const double* dline) {
    Quad<DX, DY> g;
    g.init(a.m.nx, a.m.ny, sx, sy);
    int* mine = sh.prog(c) + q;
    if (kUseY16 || (kMayY16 && pairs_aligned)) {
      YStrip16<DX, DY, OutT> ys;
      ys.lk.init(sh, w, sy, g.ystart(q), tag_of(c, q), mine, q > 0 ? line_of(c, q - 1, g.Nby) : nullptr, q + 1 < g.Py ? line_of(c, q, g.Nby) : nullptr,
                 a.epoch, q > 0 ? lds_int_at(sh.owner(c) + (q - 1)) : -1, q > 0 ? tag_of(c, q - 1) : 0);
      ys.init(a.m, sx, sy, field, sh, w, q, dline);
      VHP_DIAG_TL_STRIPS(1)
      ys.run();
    } else {
      YStrip<DX, DY, OutT, ANYW> ys;
      ys.lk.init(sh, w, sy, g.ystart(q), tag_of(c, q), mine, q > 0 ? line_of(c, q - 1, g.Nby) : nullptr, q + 1 < g.Py ? line_of(c, q, g.Nby) : nullptr,
                 a.epoch, q > 0 ? lds_int_at(sh.owner(c) + (q - 1)) : -1, q > 0 ? tag_of(c, q - 1) : 0);
      ys.init(a.m, sx, sy, field, sh, w, q, dline);
      VHP_DIAG_TL_STRIPS(1)
      for (int n = ys.g.nby(ys.jstart); n < ys.g.Nby; ++n) {
        ys.sweep_block(n);
        sim_progress();
        sim_point();
      }
    }
    lds_publish(mine, 0x3fff);
    VHP_DIAG_TL_STRIPS(-1)
    strip_done(c);
  }

  VHP_FN void run_strip(int c, int p, int qo, int sx, int sy) {
    const int unit = lds_int_at(sh.ctx(c) + kUnit);
    const int s = unit / kUnits;
    OutT* field = a.out + (size_t)s * a.field_stride;
    const double* dline = a.diag + (size_t)(4 * s + (qo >> 1)) * a.diag_stride;
    switch (qo) {
      case 0: run_x<+1, +1>(c, unit, p, sx, sy, field); break;
      case 1: run_y<+1, +1>(c, unit, p, sx, sy, field, dline); break;
      case 2: run_x<-1, +1>(c, unit, p, sx, sy, field); break;
      case 3: run_y<-1, +1>(c, unit, p, sx, sy, field, dline); break;
      case 4: run_x<-1, -1>(c, unit, p, sx, sy, field); break;
      case 5: run_y<-1, -1>(c, unit, p, sx, sy, field, dline); break;
      case 6: run_x<+1, -1>(c, unit, p, sx, sy, field); break;
      default: run_y<+1, -1>(c, unit, p, sx, sy, field, dline); break;
    }
  }

  VHP_FN void run() {
    for (;;) {
      int c = 0, p = 0, qo = 0, sx = 0, sy = 0;
      const int r = find_work(c, p, qo, sx, sy);
      if (r == kExit) break;
      if (r == kIdle) { backoff(); continue; }
      if (r == kRetry) continue;
      lds_add(sh.sched() + kBusy, 1);
      run_strip(c, p, qo, sx, sy);
      VHP_EXP_PRIO_END
      lds_add(sh.sched() + kBusy, -1);
    }
  }
};

}  // namespace pool
}  // namespace vhp
