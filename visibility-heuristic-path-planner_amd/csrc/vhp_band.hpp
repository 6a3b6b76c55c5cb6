// vhp_band.hpp -- the latency sweep in BANDS: a few sources as fast as their dependency chain allows (round 6).
//
// Replaces computeVisibility() (/root/reference/src/visibilityBasedSolver.cpp:570-696) for launches too small to fill the chip,
// like vhp_lat.hpp before it, whose protocol (Link, Below, words of death, windows of 16 steps) and whose launch shape (one
// workgroup per octant, a wavefront per strip) it keeps.  What changes is WHICH cells a wavefront owns.
//
// The strips of vhp_lat.hpp are 64 ROWS of an octant: lane = row j, step = column i.  Row j exists from column j on, so a strip
// spends its first 64 steps "growing" along the diagonal, one row switching on per step, with the diagonal rule in
// every step and ragged stores in every window -- and the strip above cannot start before that is over.  Measured on BASELINE
// config 2 (profiles/r06_a_lat_timeline_c2.txt): 40 of those growing windows in a row, 4.3 k cycles each against 2.7 k for a
// steady one, are 72 of the launch's 98 us.
//
// A BAND is 64 DIAGONALS of an octant: lane = distance from the diagonal, d = i - j (x-major) or j - i (y-major), step t = the
// marching coordinate (column i / row j).  Cell (d, t) reads (d - 1, t - 1) -- the lane below, one step ago: a DPP shift -- and
// (d, t - 1) -- its own previous value --, with the ratio (t - d) / t in both kinds of octant:
//     v = a - c (a - b),  a = lane d - 1 at step t - 1,  b = lane d at step t - 1,  c = (t - d) / t        (solver.cpp:592-600)
// Every lane runs this from the band's first step: a lane whose cell does not exist yet (t < d) computes with c = 0, i.e. copies
// its neighbour -- finite garbage that the first real step (c = 0 again: the axis cell, a copy; solver.cpp:586-590) does not look
// at -- and a lane whose cell lies past the octant's last row computes zeros (the packed map holds no cell there).  So a band has
// NO growing phase: all its windows are the steady window, and the diagonal rule is two fixed lanes of band 0 (diag(t) = the new
// sub-diagonal cell of the same step times its occupancy), which both octants of a quadrant run for themselves -- the y-major
// octants need no seeds from anybody, the diagonal's wavefront and its LDS line are gone.
//
// Price: a lane's cells run along a diagonal of the grid, so their occupancy bits are contiguous in neither the row-packed nor
// the column-packed map.  vhp_set_map packs the grid along its diagonals as well (DiagMaps: both directions, indexed by x and by
// y: 4 maps of (nx + ny - 1) runs), one 64-bit word per lane per block of 64 steps as before.  And the cells a window produces
// form a parallelogram: an x-major window's tile is read out along its diagonals (79 rows of up to 16 cells: 10 store
// instructions, 6 of them whole); a y-major step stores 64 adjacent cells of its row, one column further every step.
//
// Written against vhp_lanes.hpp: compiled for gfx950 (vhp_lat.hip) and for the CPU simulator (tests/sim), bit-exact against the
// oracle in both (tests/test_lat_sim.py, tests/test_gpu_lat.py).
#pragma once
#include <type_traits>

#include "vhp_lat.hpp"

namespace vhp {
namespace pool {

// The occupancy packed along diagonals.  A "main" run holds the cells with y - x constant (id = y - x + nx - 1), an "anti" run
// those with y + x constant (id = y + x); nx + ny - 1 runs of either kind.  Packed by x: bit x & 63 of word 1 + (x >> 6) of the
// run; packed by y likewise (a run has one cell per x and one per y).  Cells outside the grid read as 0.
// Layout in one allocation: [main by x][anti by x][main by y][anti by y].
// ---- sweepers and storers --------------------------------------------------------------------------------------------------------
// A wavefront that runs alone on its SIMD pays 4-6 cycles for every instruction it issues, and a window's cells leaving -- the tile
// read out along its diagonals, ten store instructions with their addresses and lane masks -- are as many instructions as its
// sixteen steps (measured: 2.2-2.8 k cycles of a window's 4.1-5.2 k, profiles/r06_b_lat_timeline_c2.txt).  So the workgroup has a
// second set of wavefronts: SWEEPER w computes, leaves a window's values in its LDS tile and posts a 16-byte record; STORER w (wavefront
// W + w) reads the tile out and stores.  One tile per sweeper: the storer says when it has read a tile out (its LDS reads are ordered
// before that word by the LDS itself), and the sweeper looks at the word after the arithmetic of its next window's first step, just
// before that step's tile write -- by then the storer has long been through.  The zeros of a dead band's cells are the storer's, too.
// instruction-arbitration priorities (s_setprio): the sweeper of a band in its first 64 steps / of a band in its steady state; storers run at 0.
// Measured (profiles/r06_exp_band_priorities.txt): 3 / 0 takes 2 µs off C2 and 1 µs off one source at 512²; resetting the priority after a
// march costs 512² 3 µs (the end of a launch is the last bands' zeros and set-ups); all sweepers at 3 is the same as all at 0.
#ifndef VHP_BAND_PRIO_CHAIN
#define VHP_BAND_PRIO_CHAIN 3
#endif
#ifndef VHP_BAND_PRIO_STEADY
#define VHP_BAND_PRIO_STEADY 0
#endif
// The tile's pitch.  The storer of an x-major band reads the tile out along the grid's rows, i.e. along the tile's DIAGONALS: the cell at
// column c of row j sits in lane c - j (marching up) or -c - j (marching down) + const, at pitch P word (P + 1) c - P j or -(P - 1) c - P j.
// At the pool sweep's pitch of 17 the second form is -16 c - 17 j: the eight pairs of a row in ONE bank, an eight-way conflict in every
// read-out of the two x-major units that march down (their windows 3.4-4.7 k cycles against 2.6-3.5 k, profiles/r06_b_lat_timeline_c2.txt).
// 19 is conflict-free for them (and two-way marching up, where 17 is conflict-free): the pitch goes by direction, the space by the larger.
constexpr int kBandTStrideDown = 19;
constexpr int kBandTileDoubles = kXRows * kBandTStrideDown;   // a sweeper's tile
constexpr int kPostRec = 4;    // ints into the sweeper's dummy slots (16-byte aligned): {seq, lowest step, lowest coordinate, band | flags << 16}
constexpr int kPostTaken = 8;  // the storer's word: the last seq whose tile has been read out
enum { kPostZero = 1, kPostDone = 2 };

struct DiagMaps {
  static VHP_HD int runs(int nx, int ny) { return nx + ny - 1; }
  static VHP_HD int wpdx(int nx) { return (nx + 63) / 64 + 2; }
  static VHP_HD int wpdy(int ny) { return (ny + 63) / 64 + 2; }
  static VHP_HD size_t words(int nx, int ny) { return (size_t)2 * runs(nx, ny) * (size_t)(wpdx(nx) + wpdy(ny)); }
  // kind: 0 main by x, 1 anti by x, 2 main by y, 3 anti by y
  static VHP_HD size_t offset(int nx, int ny, int kind) {
    const size_t n = (size_t)runs(nx, ny);
    return kind < 2 ? (size_t)kind * n * wpdx(nx) : 2 * n * wpdx(nx) + (size_t)(kind - 2) * n * wpdy(ny);
  }
};

// a sweeper's words for its storer (in the sweeper's dummy slots)
VHP_FN int* post_of(const Shared& sh, int w) { return reinterpret_cast<int*>(sh.lds + sh.L.dummies + w * kLatDummy); }

// What a sweeper of either kind does with its storer: a record per window (or per run of zeros, or the end), one at a time.
struct Poster {
  int* post;
  int seq;  // the next record's number (the first is 1)
  VHP_FN void init(const Shared& sh, int w) { post = post_of(sh, w); seq = 1; }
  // the storer has taken record seq - 1 (read its tile out, or copied the record of a run of zeros)
  VHP_FN bool taken(int tk) const { return tk >= seq - 1; }
  VHP_FN void wait_taken() {
    int tk = lds_poll(post + kPostTaken);
    while (!taken(tk)) { ready_backoff(); sim_point(); tk = lds_poll(post + kPostTaken); }
    lds_acquire();
  }
  VHP_FN void send(int ta, int cw, int band, int flags) {
    lds_post4(post + kPostRec, seq, ta, cw, band | (flags << 16));
    seq += 1;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// x-major band b of a unit: lanes d = 64 b + lane (distance from the diagonal), steps t = column index i; the lane's cell of
// step t is (i, j) = (t, t - d).  Cells exist for 0 <= j < rows_total, t < ni.  BandX sweeps; BandXStore stores.
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY>
struct BandXGeo {
  static VHP_FN int n_bands(const Quad<DX, DY>& q) { return (q.ni > 0 && q.nj > 0) ? (q.ni + kBlock - 1) / kBlock : 0; }
  // the lowest step of the band's first window (windows are aligned to 16 cells of x)
  static VHP_FN int first_window(const Quad<DX, DY>& q, int b_) {
    const int xw = q.X(kBlock * b_) & ~(kLW - 1);
    return DX > 0 ? xw - q.sx : q.sx - (xw + kLW - 1);
  }
  // the last step at which a lane of band b_ has a cell: lane 63's cell of the octant's last row
  static VHP_FN int t_last(const Quad<DX, DY>& q, int b_) { return imin(q.ni - 1, kBlock * b_ + kBlock - 2 + q.rows_total); }
};

template <int DX, int DY, bool MULTI>
struct BandX {
  static constexpr int kTS = DX > 0 ? kTStride : kBandTStrideDown;   // the tile's pitch (see kBandTStrideDown)
  static constexpr bool kMain = DX * DY > 0;  // the lanes' runs: y - x constant (main) or y + x constant (anti)
  using Geo = BandXGeo<DX, DY>;
  Map m;
  const uint64_t* dm;  // the packed runs of this quadrant's kind, by x
  int wpd;
  Quad<DX, DY> g;
  double* tile;   // 64 lanes x 16 columns (pitch kTS): column c = x - (lowest x of the window)
  double* slab;   // reciprocals of the step indices of two blocks of 64 coordinates, indexed by x & 127
  double* bin;
  Link<DX> lk;
  Poster* po;
  int b, D0, t_last, t_end, i_last;
  bool below, has_consumer, retires;
  int blk, pf_blk, staged_blk;
  bool pf_wait;
  int* dead_mine;
  const int* dead_below;
  bool skip_fill;
  vi lane, tile_l;
  vi dgw;      // word 0 of the lane's run in dm (clamped into the map)
  vb dg_ok;    // ... which exists
  vd prev, jd; // jd = t - d of the next step
  uint64_t dg_ow;              // band 0: the occupancy word of the DIAGONAL's run for the current block (see window)
  int dg_w0;                   // ... whose word 0 in dm this is
  int m_prev;                  // ... and the diagonal cell's occupancy at the step before the next one, as a mask (0 / -1)
  vb lane0;
  int ring_rel;                // where lane 63 writes instead of into the tile: the band's ring, as an index from `tile`
  vu64 ow, ow_nx;
  vd rv_nx;
  Tagged* death_out;           // this band's record of death beside the lines, if the band above is swept by another workgroup (else null)
  Below<DX, kLW, MULTI> nx;

  // (the caller has initialised lk)
  VHP_FN void init(const Map& m_, const uint64_t* dmap, int sx, int sy, const Shared& sh, int w, int b_, Poster* po_) {
    m = m_;
    po = po_;
    g.init(m.nx, m.ny, sx, sy);
    dm = dmap + DiagMaps::offset(m.nx, m.ny, kMain ? 0 : 1);
    wpd = DiagMaps::wpdx(m.nx);
    tile = sh.lds + sh.L.tiles + w * kBandTileDoubles;
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);
    bin = lk.bin;
    lane = lane_id();
    tile_l = lane * kTS;
    b = b_;
    D0 = kBlock * b;
    i_last = g.ni - 1;
    t_last = Geo::t_last(g, b);
    below = b > 0;
    has_consumer = b + 1 < Geo::n_bands(g);
    // A band whose lanes leave the octant at its last row before the march ends (a quadrant wider than high) RETIRES: the band above
    // still needs one value of its last lane's last cell, and zeros after that -- it marches one step further (all zeros) and then
    // says it is dead (the reader's zeros: Below::accept).
    retires = has_consumer && t_last < i_last;
    t_end = retires ? t_last + 1 : t_last;
    lane0 = lane == 0;
    {
      // the lane's run: y - s x = sy - s sx - DY d.  (Band 0's lane 0 carries the SUB-diagonal cell, distance 1 like lane 1: window)
      const vi d = b == 0 ? vmax(lane, 1) : lane + D0;
      const vi id = kMain ? (-d) * DY + (sy - sx + m.nx - 1) : (-d) * DY + (sy + sx);
      dg_ok = (id >= 0) && (id < DiagMaps::runs(m.nx, m.ny));
      dgw = vmin(vmax(id, 0), DiagMaps::runs(m.nx, m.ny) - 1) * wpd;
      dg_w0 = (kMain ? sy - sx + m.nx - 1 : sy + sx) * wpd;   // (the diagonal's run, d = 0: the source lies on it)
    }
    ring_rel = has_consumer ? (int)(lk.ring - tile) : (kBlock - 1) * kTS;
    dead_mine = sh.owner(0) + b;
    dead_below = sh.owner(0) + (b > 0 ? b - 1 : b);
    skip_fill = false;
    death_out = nullptr;
    nx.hd = 0;
    nx.h1 = nx.h2 = 0;
    nx.ring = false;
    // (defined on every path -- band 0 reads no band, a band that is dead when it starts requests nothing --, so that the compiler
    // sees that no register of the band before lives on into this one)
#pragma unroll
    for (int k = 0; k < kLW; ++k) nx.v[k] = vd(0.0);
    prev = vd(0.0);
    dg_ow = 0;
    m_prev = -1;
    pf_blk = -1;
    pf_wait = false;
    staged_blk = -0x7fffffff;
    blk = -0x7fffffff;
  }

  // the occupancy word of the diagonal's run for block blk_ (band 0 only; a uniform load, once per 64 steps)
  VHP_FN uint64_t load_diag(int blk_) const { return b == 0 ? dm[(long)dg_w0 + (1 + blk_)] : 0; }
  // the occupancy word of every lane's run and the reciprocals of the step indices of the 64 coordinates of block blk_ (x >> 6)
  VHP_FN void load_ops(int blk_, vu64& o, vd& rv) {
    o = select(dg_ok, g_load_u64(dm, dgw + (1 + blk_)), vu64(0));
    const vi it = (lane + (blk_ * 64 - g.sx)) * DX;
    const vb ok = (it >= 0) && (it < g.ni);
    rv = select(ok, g_load_f64(m.recip, select(ok, it, vi(0))), vd(0.0));
  }
  VHP_FN bool block_in_march(int b_) const { const int xe = g.X(t_end); return DX > 0 ? 64 * b_ <= xe : 64 * b_ + 63 >= xe; }
  VHP_FN void prefetch_ops(int b_) { pf_blk = b_; pf_wait = true; load_ops(b_, ow_nx, rv_nx); }
  VHP_FN void stage(int b_, vd rv) {
    wave_sync();
    lds_store(slab, lane + kBlock * (b_ & 1), rv);
    wave_sync();
    staged_blk = b_;
  }
  VHP_FN void settle() {
    if (pf_wait) { pin(ow_nx); pin(rv_nx); pf_wait = false; }
  }
  VHP_FN void stage_next() {
    if (pf_blk != -1 && staged_blk != pf_blk) { settle(); stage(pf_blk, rv_nx); }
  }
  VHP_FN void enter_block(int b_) {
    if (pf_blk == b_) { stage_next(); ow = ow_nx; }
    else { vd rv; load_ops(b_, ow, rv); pin(ow); pin(rv); stage(b_, rv); }
    dg_ow = load_diag(b_);
    blk = b_;
    if (block_in_march(b_ + DX)) prefetch_ops(b_ + DX); else pf_blk = -1;
  }

  // requests the operands of the window whose lowest x is xw in block nb; nothing is waited for
  VHP_FN void request(int xw, int nb) {
    if (below) {
      if (VHP_DIAG_WAITS) nx.request(lk, dead_below, bin, xw, DX > 0 ? xw : xw + kLW - 1, nb); else nx.ring = false;
    }
  }

  // One window: steps ta + k, k = 0 .. 15, at x = xw + (k marching up, 15 - k marching down).  A lane whose cell does not exist yet
  // has a negative numerator: clamped to 0, it copies its neighbour.  A window that sticks out of the march (before step 0, past the
  // last) is swept like any other: the reciprocal of a step that does not exist is 0.  The source itself needs no step of its own
  // either (origin_bits).  more: the next window belongs to the march.
  // B0: band 0.  The diagonal cell is the sub-diagonal cell of the same step, S(t) = V(t, t - 1), times its own occupancy (the stale
  // diagonal), and S(t) = stencil(a = diag(t - 1), b = S(t - 1), c = (t - 1) / t) times ITS occupancy: a two-term recurrence
  // in S alone, because diag(t - 1) = S(t - 1) times an occupancy bit.  Lane 0 of band 0 therefore carries S, not the diagonal: it is a
  // lane at distance 1 like lane 1 (same run, same ratio) whose neighbour `a` is its own last value under the diagonal's occupancy bit
  // of that step (a scalar word) -- and the same masked value is what lane 1 takes from it.  Three more instructions per step, one of
  // them in the chain.  The diagonal cells themselves are the storer's: lane 0's row of the tile under the diagonal's occupancy bits.
  // (ONE loop of windows per march, whatever the window's kind: a loop per kind, as the strips had, keeps a copy of the operand
  // registers per loop and spills them where the loops meet.  The window's operands have been requested: by run() for the band's
  // first window, by the window before for every other -- unconditionally, so that none of those registers outlives its band.)
  template <bool B0>
  VHP_FN void window(int ta, int xw, int nb, bool more) {
    const int k_hi = imin(kLW - 1, t_end - ta);
    VHP_WP_T0(tw0);
    if (below) nx.accept(lk, dead_below, bin, xw, DX > 0 ? xw : xw + kLW - 1, ta, ta + k_hi - 1, nb);
    VHP_WP_ADDP(lk.pp, 8, tw0);
    VHP_WP_T0(tw1);
    const vu32 hs = half_shifted(ow, xw & 63, xw & 31);  // the window's 16 occupancy bits: bit c = the lane's cell at x = xw + c
    const uint32_t hd = (uint32_t)(dg_ow >> (xw & 63));  // ... and the diagonal's (band 0)
    int tk = lds_peek(po->post + kPostTaken);            // has the storer read the tile of the window before out?  (looked at below)
    // lane 63's values are what the band above reads: that lane writes them into the band's ring instead of into the tile (the storer
    // looks there for them), so the ring costs the sweeper no read-back of the tile
    const vi tl = select(lane == kLanes - 1, vi(ring_rel + (has_consumer ? (xw & (kRing - 1)) : 0)), tile_l);
    vd di = vd((double)ta);
    // the reciprocals of the step indices come out of the slab a pair at a time, two steps ahead of their use (sixteen of them held in
    // registers from the request on were 32 registers that the workgroup's sixteen wavefronts do not have)
    const double* rslab = slab + (xw & (2 * kBlock - 1));
    vd rr[kLW];
    rr[0] = lds_bcast(rslab, DX > 0 ? 0 : kLW - 1);
    rr[1] = lds_bcast(rslab, DX > 0 ? 1 : kLW - 2);
    vd cc = ratio(vmaxd(jd, 0.0), di, rr[0]);
#pragma unroll
    for (int k = 0; k < kLW; ++k) {
      const int c = DX > 0 ? k : kLW - 1 - k;
      if ((k & 1) == 0 && k + 2 < kLW) {
        rr[k + 2] = lds_bcast(rslab, DX > 0 ? k + 2 : kLW - 3 - k);
        rr[k + 3] = lds_bcast(rslab, DX > 0 ? k + 3 : kLW - 4 - k);
      }
      vd a;
      if (B0) {
        const vd gm = and_mask(prev, select(lane0, vi(m_prev), vi(-1)));  // lane 0: S(t - 1) under the diagonal's bit = diag(t - 1)
        a = shift_up(gm, gm);
        m_prev = ((hd >> c) & 1u) ? -1 : 0;
      } else {
        a = shift_up_into(nx.v[k], prev);  // (the band below's value of this step, in lane 0; read for this step only)
      }
      const vi mk = sbfe1(hs, c);
      const vd v = and_mask(stencil(a, prev, cc), mk);
      prev = v;
      if (k == 0) {  // (the first write to the tile: after the step's arithmetic, which has hidden the look at the storer's word)
        VHP_WP_T0(tw4);
        while (!po->taken(uniform(tk))) { taken_backoff(); sim_point(); tk = lds_peek(po->post + kPostTaken); }
        lds_acquire();
        VHP_WP_ADDP(lk.pp, 11, tw4);
      }
      lds_store(tile, tl + c, v);
      di = di + 1.0;
      jd = jd + 1.0;
      if (k + 1 < kLW) cc = ratio(vmaxd(jd, 0.0), di, rr[k + 1]);  // the next step's ratio beside this step's chain, and no further ahead
      sched_fence();
    }
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
    pin(prev);
#endif
    VHP_WP_ADDP(lk.pp, 9, tw1);
    VHP_WP_T0(tw2);
    // ---- the next window's operands are asked for, the band above is told, the tile goes to the storer ----
    if (more) {
      const int xn = xw + kLW * DX;
      const bool other = (xn >> 6) != (xw >> 6);  // the next window opens the next block
      if (other) stage_next();
      request(xn, other ? nb + 1 : nb);
    }
    if (has_consumer) lk.publish(retires ? ta + kLW : ta + k_hi + 1);  // (a band that retires: its reader's window reaches past its own last step)
    if (MULTI && death_out) lk.store_window(nb, xw);  // (the band above is swept by another workgroup: it reads this band's line, window by window)
    po->send(ta, xw, b, 0);
    sim_progress();
    VHP_WP_ADDP(lk.pp, 10, tw2);
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
    VHP_WP_ADDP(lk.pp, 14, tw0); lk.pp[13] += 1;
#endif
  }

  VHP_FN void open_block(int xw, int ta) {
    const int b_ = xw >> 6;
    if (b_ == blk) return;
    if (has_consumer && !(MULTI && death_out)) lk.store_block(g.nbx(ta - 1), blk);
    enter_block(b_);
  }

  template <bool B0>
  VHP_FN void march(int& ta, int& xw, bool& dead) {
    while (ta <= t_end) {
      if (dies_at(ta - 1)) { dead = true; break; }
      open_block(xw, ta);
      // (a band's first 64 steps and the window that hands over are what the band above waits for: the chain of the unit issues ahead of
      // the bands in their steady state and of the storers that share its SIMD)
      wave_priority(ta < D0 + kBlock + kLW ? VHP_BAND_PRIO_CHAIN : VHP_BAND_PRIO_STEADY);
      window<B0>(ta, xw, DX > 0 ? blk - g.bx0 : g.bx0 - blk, ta + kLW <= t_end);
      ta += kLW; xw += kLW * DX;
      sim_progress();
      sim_point();
    }
  }
  // Band 0 starts at the source, whose value is 1.0 times its occupancy (solver.cpp:583-585) -- by data, not by a step of its own:
  // lane 0 starts from S = 1.0, and in the steps of the first window that lie before the source (they do not exist) the diagonal's
  // and the sub-diagonal's cells read as free, and so does the sub-diagonal's cell of step 0 (it does not exist either): S is still
  // 1.0 after step 0, the storer's diagonal cell of step 0 is 1.0 under the source's own occupancy bit, and so is diag(0) in step 1.
  VHP_FN void origin_bits(int xw) {
    const int s0 = g.sx & 63, w0 = xw & 63;
    // bits of the coordinates of the steps before step 0 in the window (marching up: below the source; marching down: above it)
    const uint64_t before = DX > 0 ? ((1ull << s0) - 1) & ~((1ull << w0) - 1) : ((w0 + kLW == 64 ? 0ull : 1ull << (w0 + kLW)) - 1) & ~((2ull << s0) - 1);
    ow = ow | select(lane0, vu64(before | (1ull << s0)), vu64(0));
    dg_ow |= before;
    prev = select(lane0, vd(1.0), vd(0.0));
  }

  VHP_FN void run() {
    int xw = g.X(D0) & ~(kLW - 1);
    int ta = DX > 0 ? xw - g.sx : g.sx - (xw + kLW - 1);
    jd = to_f64((-(b == 0 ? vmax(lane, 1) : lane + D0)) + ta);
    if (below) nx.hd = (MULTI && lk.remote) ? (lk.remote_died_by(0), lk.remote_dead) : lds_poll(dead_below);  // (a band that starts above a dead band need not sweep a window to find out)
    enter_block(xw >> 6);
    bool dead = below && lds_poll(dead_mine) != 0;  // (a band below died before this one could start, and said so for it)
    // (a band below in another workgroup says it for nobody: this band's first step lies past that band's death, as in announce_death)
    if (MULTI && lk.remote && lk.remote_dead != 0 && D0 > lk.remote_dead) dead = true;
    if (!dead) {
      request(xw, DX > 0 ? blk - g.bx0 : g.bx0 - blk);
      if (b == 0) { origin_bits(xw); march<true>(ta, xw, dead); } else march<false>(ta, xw, dead);
    }
    // (a reader in another workgroup sees this band's death in a record beside the lines; the line itself it has window by window,
    // Link::store_window, and never as whole blocks -- the entries of a block past a band's death were never swept)
    if (MULTI && (dead || retires) && death_out) g_store_tagged_device(death_out, vi(0), vd((double)ta), lk.epoch);
    if (has_consumer && !(MULTI && death_out)) lk.store_block(imax(g.nbx(imin(imax(ta - 1, 0), i_last)), 0), blk);
    if (dead || retires) {
      // Everything from step ta - 1 on is +0.0, in this band and below it (or: the band has left the octant): the word of death
      // first, then the progress word that lets the band above past its gate.
      lds_publish(dead_mine, ta);
      lds_publish(lk.prog, 0x3fff);
      if (dead) {
        announce_death(ta - 1);
        sim_count(4);
      }
      sim_progress();
      if (!dead || skip_fill || ta > t_last) return;
      // the zeros of what is left of the march are the storer's
      po->wait_taken();
      po->send(ta, xw, b, kPostZero);
      sim_progress();
    }
  }
  // If the band above cannot have started yet (its first step lies past s + 1), neither has any band above that one, and all of
  // them are dead from their first step on (a band's light comes from the band below it and from nowhere else): they are told at
  // once, instead of one waking the next.  (A band above that HAS started may still hold light of its own.)
  VHP_FN void announce_death(int s) {
    if ((b + 1) * kBlock <= s + 1) return;
    const int n = Geo::n_bands(g);
    int* dead_base = dead_mine - b;
    int* prog_base = lk.prog - b;
    for (int q0 = b + 1; q0 < n; q0 += kLanes) {
      const vi q = lane + q0;
      const vb up = q < n;
      lds_store_i_if(up, dead_base, q, s + 1);
      lds_acquire();
      lds_store_i_if(up, prog_base, q, 0x3fff);
    }
    lds_acquire();
  }
  // Is the band dead from step te on?  (Asked between windows, te = the last step swept.)  Every lane holds +0.0 -- lanes whose cells
  // do not exist yet hold copies of the lanes below them -- and the band below has been dead since te or before (band 0: the diagonal
  // and the sub-diagonal are lanes of its own): then all this band will ever compute is the stencil of zeros.
  VHP_FN bool dies_at(int te) {
#ifdef VHP_DIAG_NODEATH
    return false;
#endif
    if (te < 0) return false;
    if (below && nx.dead_from() > te) return false;
    return wave_all(is_pos_zero(prev));
  }
};

// The storing side of an x-major unit: the tile of a window read out along its diagonals (79 rows of up to 16 cells: 10 groups of 8
// rows, the lane's pair of cells per group), the six groups in the middle whole.  ODD: pairs of cells are not 16-byte aligned in
// every row (an odd width, or fields that start off the pair grid): every cell leaves by itself.
template <int DX, int DY, typename OutT, bool ODD>
struct BandXStore {
  static constexpr int kTS = DX > 0 ? kTStride : kBandTStrideDown;
  static constexpr int CB = sizeof(OutT);
  using Geo = BandXGeo<DX, DY>;
  Quad<DX, DY> g;
  OutT* out;
  const double* tile;
  int nxm, rows_total, i_last, ring_rel;
  const uint64_t* dg_run;   // the diagonal's run in the packed map (by x): band 0's lane 0 carries the sub-diagonal cell, the diagonal
                            // cell is that value under the diagonal's own occupancy bit (BandX::window)
  vi lane, fl_a, fl_l, fl_rj, fl_c;
  vu32 fl_off;

  VHP_FN void init(const Map& m, const uint64_t* dmap, int sx, int sy, OutT* out_, const double* tile_, const double* ring_) {
    g.init(m.nx, m.ny, sx, sy);
    out = out_;
    tile = tile_;
    ring_rel = (int)(ring_ - tile_);
    dg_run = dmap + DiagMaps::offset(m.nx, m.ny, DX * DY > 0 ? 0 : 1) + (long)(DX * DY > 0 ? sy - sx + m.nx - 1 : sy + sx) * DiagMaps::wpdx(m.nx);
    nxm = m.nx;
    rows_total = g.rows_total;
    i_last = g.ni - 1;
    lane = lane_id();
    // read-out geometry: lane -> (memory row slot rs = lane >> 3, pair pc = lane & 7: the cells at columns 2 pc, 2 pc + 1 of the
    // window).  Group u of a window holds the rows j = jA + 8 u + rj, jA = ta - D0 - 63 (the row of lane 63 at the window's first
    // step), rj = the slot counted in j.  The lane that computed cell (column c, row j) is t - D0 - j.
    const vi rs = lane >> 3, pc = lane & 7;
    fl_rj = DY > 0 ? rs : (-rs) + 7;
    fl_c = pc * 2;
    fl_l = DX > 0 ? (fl_c - fl_rj) + 63 : (-fl_c - fl_rj) + 78;  // the lane of the pair's first cell in group 0; - 8 per group
    // (the lower of the pair's two addresses in group 7, the last whole group: every other whole group at a positive immediate
    // offset from it -- a DS instruction encodes no negative one, the compiler would keep a register per group)
    fl_a = (fl_l - 56) * kTS + fl_c + (DX > 0 ? 0 : 1 - kTS);
    fl_off = to_u32((rs * nxm + pc * 2) * CB);
  }
  VHP_FN void store_group(OutT* base, const vb& ok0, const vb& ok1, const vd& a, const vd& c) {
    if (!ODD) { g_store2_if(ok0 && ok1, ok0, ok1, base, fl_off, a, c); return; }
    g_store2_if(vb(false), ok0, vb(false), base, fl_off, a, c);
    g_store2_if(vb(false), vb(false), ok1, base, fl_off, a, c);
  }
  // every row and step of band b's window at ta exists
  VHP_FN bool interior(int D0, int ta) const { return ta - D0 - (kBlock - 1) >= 0 && ta + kLW - 1 - D0 <= rows_total - 1 && ta + kLW - 1 <= i_last; }

  // Which groups of a window hold a cell of the lane's pair?  A RANGE of groups: cell (column c, group u, row slot rj) was computed by
  // lane l = fl_l - 8 u of the band (0 .. 63 for u in [(fl_l - 56) >> 3, fl_l >> 3]: the parallelogram's two ends) and lies in row
  // jA + 8 u + rj (0 .. rows_total - 1 for another range of u: the rows the octant has); its step must be a step of the march.  The
  // ranges are intersected once per window, and a group's predicate is one subtraction and one unsigned compare per cell -- per
  // group, the same conditions spelt out were fourteen vector and two dozen scalar instructions, in the windows of a band's first 64
  // steps, the ones the band above waits for.
  VHP_FN void group_range(const vi& l0, const vb& step_ok, int jA, bool interior_, vi& lo, vi& w) const {
    vi a = (l0 - (kBlock - 8)) >> 3, b = l0 >> 3;      // 0 <= l0 - 8 u <= 63
    if (!interior_) {
      a = select((-fl_rj - jA + 7) >> 3 > a, (-fl_rj - jA + 7) >> 3, a);                           // jA + 8 u + rj >= 0
      b = select((-fl_rj + (rows_total - 1 - jA)) >> 3 < b, (-fl_rj + (rows_total - 1 - jA)) >> 3, b);  // ... <= rows_total - 1
    }
    const vb none = (b < a) || !step_ok;
    lo = select(none, vi(100), a);
    w = select(none, vi(0), b - a);
  }
  // INTERIOR: only the two ends of the parallelogram (the groups 0, 1, 8, 9) need a lane mask
  template <bool INTERIOR>
  VHP_FN void store_window(int D0, int ta, int xw, const vd (&fa)[10], const vd (&fb)[10]) {
    const int jA = ta - D0 - (kBlock - 1);
    const long base_step = (long)(8 * DY) * nxm;
    OutT* base = out + (long)(DY > 0 ? g.Y(jA) : g.Y(jA + 7)) * (long)nxm + xw;
    vi lo0, w0, lo1, w1;
    {
      const vi t0 = DX > 0 ? fl_c + ta : (-fl_c) + (ta + kLW - 1), t1 = t0 + DX;   // the steps of the pair's two cells
      group_range(fl_l, INTERIOR ? vb(true) : (t0 <= i_last), jA, INTERIOR, lo0, w0);
      group_range(DX > 0 ? fl_l + 1 : fl_l - 1, INTERIOR ? vb(true) : (t1 <= i_last), jA, INTERIOR, lo1, w1);
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (INTERIOR && !ODD && u >= 2 && u <= 7) {
        g_store2(base, fl_off, fa[u], fb[u]);
      } else if (INTERIOR || (jA + 8 * u + 7 >= 0 && jA + 8 * u < rows_total)) {
        const vb ok0 = to_u32((-lo0) + u) <= to_u32(w0), ok1 = to_u32((-lo1) + u) <= to_u32(w1);
        store_group(base, ok0, ok1, fa[u], fb[u]);
      }
      base += base_step;
    }
  }
  // a window of band b: the tile read out (the word `taken` tells the sweeper so: an LDS write issued after the reads, which the
  // LDS executes in that order), then stored
  VHP_FN void flush(int b, int ta, int xw, int* taken, int seq) {
    const int D0 = kBlock * b;
    // (the last lane's row is in the band's ring if a band above reads it: BandX::window)
    const int row63 = b + 1 < Geo::n_bands(g) ? ring_rel + (xw & (kRing - 1)) : (kBlock - 1) * kTS;
    wave_sync();
    vd fa[10], fb[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (u >= 2 && u <= 7) {
        fa[u] = lds_load(tile, fl_a + ((DX > 0 ? 0 : kTS - 1) + (7 - u) * (8 * kTS)));
        fb[u] = lds_load(tile, fl_a + ((DX > 0 ? kTS + 1 : 0) + (7 - u) * (8 * kTS)));
      } else {  // (the lanes of the two ends of the parallelogram that have no cell read the nearest row of the tile instead of what lies beside it)
        const vi la = vmin(vmax(fl_l - 8 * u, 0), kBlock - 1), lb = vmin(vmax(fl_l + ((DX > 0 ? 1 : -1) - 8 * u), 0), kBlock - 1);
        fa[u] = lds_load(tile, select(la == kBlock - 1, vi(row63), la * kTS) + fl_c);
        fb[u] = lds_load(tile, select(lb == kBlock - 1, vi(row63), lb * kTS) + (fl_c + 1));
      }
    }
    lds_publish(taken, seq);
    if (b == 0) {
      // the diagonal cells: lane 0's values (the groups 7 .. 9 hold them) under the diagonal's occupancy bits of the window
      const vu64 dbits = vu64(dg_run[1 + (xw >> 6)] >> (xw & 63));
      const vi bm0 = bit_mask_lane(dbits, fl_c), bm1 = bit_mask_lane(dbits, fl_c + 1);
#pragma unroll
      for (int u = 7; u < 10; ++u) {
        const vi la = fl_l - 8 * u, lb = DX > 0 ? la + 1 : la - 1;
        fa[u] = and_mask(fa[u], select(la == 0, bm0, vi(-1)));
        fb[u] = and_mask(fb[u], select(lb == 0, bm1, vi(-1)));
      }
    }
    if (interior(D0, ta)) store_window<true>(D0, ta, xw, fa, fb); else store_window<false>(D0, ta, xw, fa, fb);
  }
  // The zeros of what is left of band b's march from step ta on: row by row -- a row of the band is at most 64 adjacent cells, the
  // steps max(ta, j + D0) .. min(t_last, j + D0 + 63) of row j --, one store instruction per row (window by window, through the
  // read-out's predicates, it was forty-odd instructions per group of eight rows: 18 us for a band of the maze).
  VHP_FN void zero_fill(int b, int ta, int) {
    const int D0 = kBlock * b, t_last = Geo::t_last(g, b);
    const int j_lo = imax(0, ta - D0 - (kBlock - 1)), j_hi = imin(rows_total - 1, t_last - D0);
    // the rows whose piece is whole (64 cells: the steps j + D0 .. j + D0 + 63 all lie in [ta, t_last]): one store and one pointer
    // increment per row -- the piece slides by one column from row to row
    const int jf_lo = imax(j_lo, ta - D0), jf_hi = imin(j_hi, t_last - D0 - (kBlock - 1));
    auto ragged = [&](int ja, int jb) {
      for (int j = ja; j <= jb; ++j) {
        const int t_lo = imax(ta, j + D0), t_hi = imin(t_last, j + D0 + kBlock - 1);
        const int x_lo = DX > 0 ? g.X(t_lo) : g.X(t_hi);   // lane l: the cell l columns above the row piece's lowest x
        g_store_scalar_if(lane <= t_hi - t_lo, out, lane + (g.Y(j) * nxm + x_lo), OutT(0));
        if (((j - ja) & 15) == 15) sim_point();
      }
    };
    if (jf_lo > jf_hi) { ragged(j_lo, j_hi); return; }
    ragged(j_lo, jf_lo - 1);
    {
      const vu32 off = to_u32(lane * CB);
      OutT* row = out + ((long)g.Y(jf_lo) * nxm + (DX > 0 ? g.X(jf_lo + D0) : g.X(jf_lo + D0 + kBlock - 1)));
      const long step = (long)DY * nxm + DX;
      for (int j = jf_lo; j <= jf_hi; ++j) {
        g_store1_if(vb(true), row, off, vd(0.0));
        row += step;
        if (((j - jf_lo) & 15) == 15) sim_point();
      }
    }
    ragged(jf_hi + 1, j_hi);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// y-major band b of a unit: lanes d' = 64 b + lane (distance j - i from the diagonal), steps t = row index j; the lane's cell of
// step t is (i, j) = (t - d', t): the 64 cells of a step are adjacent cells of one row.  Band 0's lane 0 is the diagonal (stored
// again here), run by the same private two-term recurrence as in BandX: both octants of a quadrant compute their diagonal for
// themselves, nobody seeds anybody.  Cells exist for d' <= t <= min(nj - 1, d' + ni - 1).
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY>
struct BandYGeo {
  static VHP_FN int n_bands(const Quad<DX, DY>& q) { return (q.ni > 0 && q.nj > 1) ? (q.nj + kBlock - 1) / kBlock : 0; }
  static VHP_FN int first_window(const Quad<DX, DY>& q, int b_) {
    const int yw = q.Y(kBlock * b_) & ~(kLW - 1);
    return DY > 0 ? yw - q.sy : q.sy - (yw + kLW - 1);
  }
  // lane 63's cell of the octant's last column
  static VHP_FN int t_last(const Quad<DX, DY>& q, int b_) { return imin(q.nj - 1, kBlock * b_ + kBlock - 1 + q.ni - 1); }
};

template <int DX, int DY, bool MULTI>
struct BandY {
  static constexpr bool kMain = DX * DY > 0;
  using Geo = BandYGeo<DX, DY>;
  Map m;
  const uint64_t* dm;  // the packed runs of this quadrant's kind, by y
  int wpd;
  Quad<DX, DY> g;
  double* tile;   // 64 lanes x 16 rows (pitch kTStride): column c = y - (lowest y of the window)
  double* slab;
  double* bin;
  Link<DY> lk;
  Poster* po;
  int b, D0, t_last, t_end, j_last;
  bool below, has_consumer, retires;
  int blk, pf_blk, staged_blk;
  bool pf_wait;
  int* dead_mine;
  const int* dead_below;
  bool skip_fill;
  vi lane, tile_l;
  vi dgw;
  vb dg_ok;
  vd prev, jd;
  uint64_t dg_ow;              // band 0: the occupancy word of the diagonal's run (see BandX)
  int sub_w0, dg_w0;           // ... and where lane 0's run (the sub-diagonal's: load_sub) and the diagonal's start in dm
  int m_prev;
  vb lane0;
  int ring_rel;
  vu64 ow, ow_nx;
  vd rv_nx;
  Tagged* death_out;           // this band's record of death beside the lines, if the band above is swept by another workgroup (else null)
  Below<DY, kLW, MULTI> nx;

  VHP_FN void init(const Map& m_, const uint64_t* dmap, int sx, int sy, const Shared& sh, int w, int b_, Poster* po_) {
    m = m_;
    po = po_;
    g.init(m.nx, m.ny, sx, sy);
    dm = dmap + DiagMaps::offset(m.nx, m.ny, kMain ? 2 : 3);
    wpd = DiagMaps::wpdy(m.ny);
    tile = sh.lds + sh.L.tiles + w * kBandTileDoubles;
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);
    bin = lk.bin;
    lane = lane_id();
    tile_l = lane * kTStride;
    b = b_;
    D0 = kBlock * b;
    j_last = g.nj - 1;
    t_last = Geo::t_last(g, b);
    below = b > 0;
    has_consumer = b + 1 < Geo::n_bands(g);
    retires = has_consumer && t_last < j_last;   // (see BandX::init)
    t_end = retires ? t_last + 1 : t_last;
    {
      // the lane's run: y - s x = sy - s sx + DY d'
      const vi d = lane + D0;
      const vi id = kMain ? d * DY + (sy - sx + m.nx - 1) : d * DY + (sy + sx);
      dg_ok = (id >= 0) && (id < DiagMaps::runs(m.nx, m.ny));
      dgw = vmin(vmax(id, 0), DiagMaps::runs(m.nx, m.ny) - 1) * wpd;
      // (the sub-diagonal's run, d' = -1: the cells (t, t - 1) of the x-major octant -- what band 0's lane 0 carries, BandX::window)
      const int id1 = kMain ? -DY + (sy - sx + m.nx - 1) : -DY + (sy + sx);
      sub_w0 = imin(imax(id1, 0), DiagMaps::runs(m.nx, m.ny) - 1) * wpd;
      if (id1 < 0 || id1 >= DiagMaps::runs(m.nx, m.ny)) sub_w0 = -1;
      dg_w0 = (kMain ? sy - sx + m.nx - 1 : sy + sx) * wpd;
    }
    lane0 = lane == 0;
    ring_rel = has_consumer ? (int)(lk.ring - tile) : (kBlock - 1) * kTStride;
    dead_mine = sh.owner(0) + b;
    dead_below = sh.owner(0) + (b > 0 ? b - 1 : b);
    skip_fill = false;
    death_out = nullptr;
    nx.hd = 0;
    nx.h1 = nx.h2 = 0;
    nx.ring = false;
    // (defined on every path -- band 0 reads no band, a band that is dead when it starts requests nothing --, so that the compiler
    // sees that no register of the band before lives on into this one)
#pragma unroll
    for (int k = 0; k < kLW; ++k) nx.v[k] = vd(0.0);
    prev = vd(0.0);
    dg_ow = 0;
    m_prev = -1;
    pf_blk = -1;
    pf_wait = false;
    staged_blk = -0x7fffffff;
    blk = -0x7fffffff;
  }

  // The occupancy word of the sub-diagonal's run for block blk_ (band 0 only; uniform loads).  The sub-diagonal's cell of step t lies
  // in the row of step t - 1 -- one bit behind in its run, where every lane's cell is at the step's own y: the word is shifted by it.
  VHP_FN uint64_t load_sub(int blk_) const {
    if (b != 0 || sub_w0 < 0) return 0;
    const uint64_t o = dm[(long)sub_w0 + (1 + blk_)], o2 = dm[(long)sub_w0 + (1 + blk_ - DY)];
    return DY > 0 ? (o << 1) | (o2 >> 63) : (o >> 1) | (o2 << 63);
  }
  VHP_FN uint64_t load_diag(int blk_) const { return b == 0 ? dm[(long)dg_w0 + (1 + blk_)] : 0; }
  VHP_FN void load_ops(int blk_, vu64& o, vd& rv) {
    o = select(dg_ok, g_load_u64(dm, dgw + (1 + blk_)), vu64(0));
    const vi jt = (lane + (blk_ * 64 - g.sy)) * DY;
    const vb ok = (jt >= 0) && (jt < g.nj);
    rv = select(ok, g_load_f64(m.recip, select(ok, jt, vi(0))), vd(0.0));
  }
  VHP_FN bool block_in_march(int b_) const { const int ye = g.Y(t_end); return DY > 0 ? 64 * b_ <= ye : 64 * b_ + 63 >= ye; }
  VHP_FN void prefetch_ops(int b_) { pf_blk = b_; pf_wait = true; load_ops(b_, ow_nx, rv_nx); }
  VHP_FN void stage(int b_, vd rv) {
    wave_sync();
    lds_store(slab, lane + kBlock * (b_ & 1), rv);
    wave_sync();
    staged_blk = b_;
  }
  VHP_FN void settle() {
    if (pf_wait) { pin(ow_nx); pin(rv_nx); pf_wait = false; }
  }
  VHP_FN void stage_next() {
    if (pf_blk != -1 && staged_blk != pf_blk) { settle(); stage(pf_blk, rv_nx); }
  }
  VHP_FN void enter_block(int b_) {
    if (pf_blk == b_) { stage_next(); ow = ow_nx; }
    else { vd rv; load_ops(b_, ow, rv); pin(ow); pin(rv); stage(b_, rv); }
    if (b == 0) {  // (lane 0 carries the sub-diagonal cell; uniform loads, once per 64 steps)
      ow = select(lane0, vu64(load_sub(b_)), ow);
      dg_ow = load_diag(b_);
    }
    blk = b_;
    if (block_in_march(b_ + DY)) prefetch_ops(b_ + DY); else pf_blk = -1;
  }
  VHP_FN void request(int yw, int nb) {
    if (below) {
      if (VHP_DIAG_WAITS) nx.request(lk, dead_below, bin, yw, DY > 0 ? yw : yw + kLW - 1, nb); else nx.ring = false;
    }
  }

  // One window: steps ta + k, k = 0 .. 15, at y = yw + (k marching up, 15 - k marching down): BandX::window with x and y exchanged.
  template <bool B0>
  VHP_FN void window(int ta, int yw, int nb, bool more) {
    const int k_hi = imin(kLW - 1, t_end - ta);
    if (below) nx.accept(lk, dead_below, bin, yw, DY > 0 ? yw : yw + kLW - 1, ta, ta + k_hi - 1, nb);
    const vu32 hs = half_shifted(ow, yw & 63, yw & 31);
    const uint32_t hd = (uint32_t)(dg_ow >> (yw & 63));
    int tk = lds_peek(po->post + kPostTaken);
    const vi tl = select(lane == kLanes - 1, vi(ring_rel + (has_consumer ? (yw & (kRing - 1)) : 0)), tile_l);
    vd dj = vd((double)ta);
    const double* rslab = slab + (yw & (2 * kBlock - 1));
    vd rr[kLW];
    rr[0] = lds_bcast(rslab, DY > 0 ? 0 : kLW - 1);
    rr[1] = lds_bcast(rslab, DY > 0 ? 1 : kLW - 2);
    vd cc = ratio(vmaxd(jd, 0.0), dj, rr[0]);
#pragma unroll
    for (int k = 0; k < kLW; ++k) {
      const int c = DY > 0 ? k : kLW - 1 - k;
      if ((k & 1) == 0 && k + 2 < kLW) {
        rr[k + 2] = lds_bcast(rslab, DY > 0 ? k + 2 : kLW - 3 - k);
        rr[k + 3] = lds_bcast(rslab, DY > 0 ? k + 3 : kLW - 4 - k);
      }
      vd a;
      if (B0) {
        const vd gm = and_mask(prev, select(lane0, vi(m_prev), vi(-1)));
        a = shift_up(gm, gm);
        m_prev = ((hd >> c) & 1u) ? -1 : 0;
      } else {
        a = shift_up_into(nx.v[k], prev);
      }
      const vi mk = sbfe1(hs, c);
      const vd v = and_mask(stencil(a, prev, cc), mk);
      prev = v;
      if (k == 0) {
        while (!po->taken(uniform(tk))) { taken_backoff(); sim_point(); tk = lds_peek(po->post + kPostTaken); }
        lds_acquire();
      }
      lds_store(tile, tl + c, v);
      dj = dj + 1.0;
      jd = jd + 1.0;
      if (k + 1 < kLW) cc = ratio(vmaxd(jd, 0.0), dj, rr[k + 1]);
      sched_fence();
    }
    if (more) {
      const int yn = yw + kLW * DY;
      const bool other = (yn >> 6) != (yw >> 6);
      if (other) stage_next();
      request(yn, other ? nb + 1 : nb);
    }
    if (has_consumer) lk.publish(retires ? ta + kLW : ta + k_hi + 1);
    if (MULTI && death_out) lk.store_window(nb, yw);  // (see BandX::window)
    po->send(ta, yw, b, 0);
    sim_progress();
  }

  VHP_FN void open_block(int yw, int ta) {
    const int b_ = yw >> 6;
    if (b_ == blk) return;
    if (has_consumer && !(MULTI && death_out)) lk.store_block(g.nby(ta - 1), blk);
    enter_block(b_);
  }

  template <bool B0>
  VHP_FN void march(int& ta, int& yw, bool& dead) {
    while (ta <= t_end) {
      if (dies_at(ta - 1)) { dead = true; break; }
      open_block(yw, ta);
      wave_priority(ta < D0 + kBlock + kLW ? VHP_BAND_PRIO_CHAIN : VHP_BAND_PRIO_STEADY);   // (see BandX::march)
      window<B0>(ta, yw, DY > 0 ? blk - g.by0 : g.by0 - blk, ta + kLW <= t_end);
      ta += kLW; yw += kLW * DY;
      sim_progress();
      sim_point();
    }
  }
  // (see BandX::origin_bits)
  VHP_FN void origin_bits(int yw) {
    const int s0 = g.sy & 63, w0 = yw & 63;
    const uint64_t before = DY > 0 ? ((1ull << s0) - 1) & ~((1ull << w0) - 1) : ((w0 + kLW == 64 ? 0ull : 1ull << (w0 + kLW)) - 1) & ~((2ull << s0) - 1);
    ow = ow | select(lane0, vu64(before | (1ull << s0)), vu64(0));
    dg_ow |= before;
    prev = select(lane0, vd(1.0), vd(0.0));
  }

  VHP_FN void run() {
    int yw = g.Y(D0) & ~(kLW - 1);
    int ta = DY > 0 ? yw - g.sy : g.sy - (yw + kLW - 1);
    jd = to_f64((-(b == 0 ? vmax(lane, 1) : lane + D0)) + ta);
    if (below) nx.hd = (MULTI && lk.remote) ? (lk.remote_died_by(0), lk.remote_dead) : lds_poll(dead_below);
    enter_block(yw >> 6);
    bool dead = below && lds_poll(dead_mine) != 0;
    if (MULTI && lk.remote && lk.remote_dead != 0 && D0 > lk.remote_dead) dead = true;   // (see BandX::run)
    if (!dead) {
      request(yw, DY > 0 ? blk - g.by0 : g.by0 - blk);
      if (b == 0) { origin_bits(yw); march<true>(ta, yw, dead); } else march<false>(ta, yw, dead);
    }
    if (MULTI && (dead || retires) && death_out) g_store_tagged_device(death_out, vi(0), vd((double)ta), lk.epoch);   // (see BandX::run)
    if (has_consumer && !(MULTI && death_out)) lk.store_block(imax(g.nby(imin(imax(ta - 1, 0), j_last)), 0), blk);
    if (dead || retires) {  // (see BandX::run)
      lds_publish(dead_mine, ta);
      lds_publish(lk.prog, 0x3fff);
      if (dead) {
        announce_death(ta - 1);
        sim_count(4);
      }
      sim_progress();
      if (!dead || skip_fill || ta > t_last) return;
      po->wait_taken();
      po->send(ta, yw, b, kPostZero);
      sim_progress();
    }
  }
  VHP_FN void announce_death(int s) {
    if ((b + 1) * kBlock <= s + 1) return;
    const int n = Geo::n_bands(g);
    int* dead_base = dead_mine - b;
    int* prog_base = lk.prog - b;
    for (int q0 = b + 1; q0 < n; q0 += kLanes) {
      const vi q = lane + q0;
      const vb up = q < n;
      lds_store_i_if(up, dead_base, q, s + 1);
      lds_acquire();
      lds_store_i_if(up, prog_base, q, 0x3fff);
    }
    lds_acquire();
  }
  VHP_FN bool dies_at(int te) {
#ifdef VHP_DIAG_NODEATH
    return false;
#endif
    if (te < 0) return false;
    if (below && nx.dead_from() > te) return false;
    return wave_all(is_pos_zero(prev));
  }
};

// The storing side of a y-major unit: a step's 64 cells are adjacent cells of its row, one column further every step.
template <int DX, int DY, typename OutT>
struct BandYStore {
  static constexpr int CB = sizeof(OutT);
  using Geo = BandYGeo<DX, DY>;
  Quad<DX, DY> g;
  OutT* out;
  const double* tile;
  int nxm, j_last, ring_rel;
  const uint64_t* dg_run;   // the diagonal's run in the packed map (by y): see BandXStore
  vi lane;
  vu32 xoff;

  VHP_FN void init(const Map& m, const uint64_t* dmap, int sx, int sy, OutT* out_, const double* tile_, const double* ring_) {
    g.init(m.nx, m.ny, sx, sy);
    out = out_;
    tile = tile_;
    ring_rel = (int)(ring_ - tile_);
    dg_run = dmap + DiagMaps::offset(m.nx, m.ny, DX * DY > 0 ? 2 : 3) + (long)(DX * DY > 0 ? sy - sx + m.nx - 1 : sy + sx) * DiagMaps::wpdy(m.ny);
    nxm = m.nx;
    j_last = g.nj - 1;
    lane = lane_id();
    // the lane's cell of step t is at x = X(t) - DX d': byte offsets from the row's lowest x of the band, never negative
    xoff = to_u32((DX > 0 ? (-lane) + (kLanes - 1) : lane) * CB);
  }
  // where lane 0's (DX < 0) or lane 63's (DX > 0) cell of step t of the band whose lane 0 is at distance D0 is stored: xoff counts from there
  VHP_FN OutT* row_base(int D0, int t) const {
    const long x0 = DX > 0 ? (long)g.X(t) - (D0 + kLanes - 1) : (long)g.X(t) + D0;
    return out + (long)g.Y(t) * (long)nxm + x0;
  }
  // every lane has a cell in every step of the window at ta
  VHP_FN bool interior(int D0, int ta) const { return ta >= D0 + kLanes - 1 && ta + kLW - 1 <= imin(j_last, D0 + g.ni - 1); }

  VHP_FN void flush(int b, int ta, int yw, int* taken, int seq) {
    const int D0 = kBlock * b;
    const int row63 = b + 1 < Geo::n_bands(g) ? ring_rel + (yw & (kRing - 1)) : (kBlock - 1) * kTStride;
    const vi tl = select(lane == kLanes - 1, vi(row63), lane * kTStride);
    wave_sync();
    vd fv[kLW];
#pragma unroll
    for (int k = 0; k < kLW; ++k) fv[k] = lds_load(tile, tl + (DY > 0 ? k : kLW - 1 - k));
    lds_publish(taken, seq);
    if (b == 0) {  // the diagonal cells: lane 0's values under the diagonal's occupancy bits of the window
      const uint32_t dbits = (uint32_t)(dg_run[1 + (yw >> 6)] >> (yw & 63));
      const vb lane0 = lane == 0;
#pragma unroll
      for (int k = 0; k < kLW; ++k) fv[k] = and_mask(fv[k], select(lane0, vi(((dbits >> (DY > 0 ? k : kLW - 1 - k)) & 1u) ? -1 : 0), vi(-1)));
    }
    const long rowstep = (long)DY * nxm + DX;
    OutT* row = row_base(D0, ta);
    if (interior(D0, ta)) {
#pragma unroll
      for (int k = 0; k < kLW; ++k) {
        g_store1_if(vb(true), row, xoff, fv[k]);
        row += rowstep;
      }
    } else {
      const vi d = lane + D0;
      const vi t_hi = vmin(d + (g.ni - 1), j_last);  // the lane's cell exists at the steps d' .. t_hi
#pragma unroll
      for (int k = 0; k < kLW; ++k) {
        const int t = ta + k;
        g_store1_if((d <= t) && (t_hi >= t), row, xoff, fv[k]);
        row += rowstep;
      }
    }
  }
  VHP_FN void zero_fill(int b, int ta, int) {
    const int D0 = kBlock * b, t_last = Geo::t_last(g, b);
    const vi d = lane + D0;
    const vi t_hi = vmin(d + (g.ni - 1), j_last);
    const long rowstep = (long)DY * nxm + DX;
    OutT* row = row_base(D0, ta);
    // (the rows in which every lane has a cell: no predicate)
    const int tf_lo = imax(ta, D0 + kLanes - 1), tf_hi = imin(t_last, imin(j_last, D0 + g.ni - 1));
    for (int t = ta; t <= t_last; ++t) {
      if (t >= tf_lo && t <= tf_hi) g_store1_if(vb(true), row, xoff, vd(0.0)); else g_store1_if((d <= t) && (t_hi >= t), row, xoff, vd(0.0));
      row += rowstep;
      if (((t - ta) & (kLW - 1)) == kLW - 1) sim_point();
    }
  }
};

// One wavefront of a unit's workgroup: sweeper w (w < W) or the storer of sweeper w - W.
// MULTI: the build for launches with more than one workgroup per unit (LatArgs::halves > 1); the build for one holds none of that code
template <typename OutT, bool ODD = false, bool MULTI = false>
struct BandWorker {
  static constexpr int kRoles = 2;  // wavefronts per sweeper (the simulator and the launcher size the workgroup by it)
  static constexpr int kTilePitch = kBandTStrideDown;  // ... and the LDS layout's tiles by this
  LatArgs<OutT> a;
  Shared sh;
  int w, W;
  int H, half;   // workgroups per unit, and which of them this is (run)
  vi lane;
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
  unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

  VHP_FN void init(const LatArgs<OutT>& a_, double* lds, const Layout& L, int w_) {
    a = a_;
    sh.lds = lds;
    sh.L = L;
    w = w_;
    W = L.W;
    H = (MULTI && a.halves > 1) ? a.halves : 1;
    half = 0;
    lane = lane_id();
  }
  // Before any wavefront runs: every thread of the workgroup calls this (tid of nthreads), then a barrier.
  static VHP_FN void clear(double* lds, const Layout& L, int tid, int nthreads) {
    Worker<OutT>::clear(lds, L, tid, nthreads);
    Shared s;
    s.lds = lds;
    s.L = L;
    for (int k = tid; k < L.W * 2; k += nthreads) lds_set_int(post_of(s, k >> 1) + ((k & 1) ? kPostTaken : kPostRec), 0);
  }

  VHP_FN Tagged* line_of(int unit, int p, int nb) const { return a.lines + (size_t)64 * ((size_t)unit * (size_t)a.unit_blocks + (size_t)p * nb); }
  static VHP_FN int tag_of(int p) { return (1 << 8) | p; }  // (never 0: a cleared header belongs to no band)
  // More than one workgroup per unit (LatArgs::halves): band p is swept by workgroup (p / W) % H of its unit, sweeper p % W -- band W's
  // sweeper no longer waits for band 0's, and an octant of 16 bands has them all in flight.  Where a band and the band below it live in
  // different workgroups (p % W == 0) the reader takes the writer's LINE in global memory, block by block (Link::remote; the lines are
  // the durable copy that every band with a reader writes anyway), and the writer's death out of a record beside the lines: entry p of
  // the unit's last two blocks of scratch (the lines of at most 128 bands leave them free: lat_unit_blocks).
  VHP_FN Tagged* death_rec(int unit, int p) const { return a.lines + ((size_t)64 * ((size_t)unit * (size_t)a.unit_blocks + (size_t)(a.unit_blocks - 2)) + (size_t)p); }
  VHP_FN bool remote_below(int p) const { return MULTI && H > 1 && p > 0 && p % W == 0; }
  VHP_FN bool remote_above(int p, int n) const { return MULTI && H > 1 && p + 1 < n && (p + 1) % W == 0; }

  // a coarse gate ahead of a band's first window (which then checks exactly what it reads): one word per poll
  VHP_FN void wait_for(const int* word, int at_least) {
    while (lds_poll(word) < at_least) { ready_backoff(); sim_point(); }
    lds_acquire();
  }

  template <int DX, int DY>
  VHP_FN void sweep_x(int unit, int sx, int sy, Poster& po) {
    Quad<DX, DY> g;
    g.init(a.m.nx, a.m.ny, sx, sy);
    int* prog = sh.prog(0);
    const int n = BandXGeo<DX, DY>::n_bands(g);
    for (int p = half * W + w; p < n; p += H * W) {
      BandX<DX, DY, MULTI> xs;
      xs.lk.init(sh, w, sx, imax(BandXGeo<DX, DY>::first_window(g, p), 0), tag_of(p), prog + p, p > 0 ? line_of(unit, p - 1, g.Nbx) : nullptr,
                 p + 1 < n ? line_of(unit, p, g.Nbx) : nullptr, a.epoch, p > 0 ? (p - 1) % W : -1, p > 0 ? tag_of(p - 1) : 0);
      if (remote_below(p)) { xs.lk.remote = true; xs.lk.death_in = death_rec(unit, p - 1); }
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
      xs.lk.pp = prof;
#endif
      xs.init(a.m, a.dmap, sx, sy, sh, w, p, &po);
      if (remote_above(p, n)) xs.death_out = death_rec(unit, p);
      xs.skip_fill = a.dead_cells_are_zero;
      xs.prefetch_ops(g.X(kBlock * p) >> 6);
      VHP_LAT_STAMP(unit, p, 0);
      if (VHP_DIAG_WAITS && p > 0 && !remote_below(p)) wait_for(prog + (p - 1), imin(kBlock * p + 2, g.ni));  // the band below has got to my first step
      VHP_LAT_STAMP(unit, p, 1);
      xs.run();
      VHP_LAT_STAMP(unit, p, 3);
      lds_publish(prog + p, 0x3fff);  // finished
      sim_progress();
    }
  }

  template <int DX, int DY>
  VHP_FN void sweep_y(int unit, int sx, int sy, Poster& po) {
    Quad<DX, DY> g;
    g.init(a.m.nx, a.m.ny, sx, sy);
    int* prog = sh.prog(0);
    const int n = BandYGeo<DX, DY>::n_bands(g);
    const int Nby = g.Nby;
    for (int q = half * W + w; q < n; q += H * W) {
      BandY<DX, DY, MULTI> ys;
      ys.lk.init(sh, w, sy, imax(BandYGeo<DX, DY>::first_window(g, q), 0), tag_of(q), prog + q, q > 0 ? line_of(unit, q - 1, Nby) : nullptr,
                 q + 1 < n ? line_of(unit, q, Nby) : nullptr, a.epoch, q > 0 ? (q - 1) % W : -1, q > 0 ? tag_of(q - 1) : 0);
      if (remote_below(q)) { ys.lk.remote = true; ys.lk.death_in = death_rec(unit, q - 1); }
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
      ys.lk.pp = prof;
#endif
      VHP_LAT_STAMP(unit, q, 0);
      ys.init(a.m, a.dmap, sx, sy, sh, w, q, &po);
      if (remote_above(q, n)) ys.death_out = death_rec(unit, q);
      ys.skip_fill = a.dead_cells_are_zero;
      ys.prefetch_ops(g.Y(kBlock * q) >> 6);
      if (VHP_DIAG_WAITS && q > 0 && !remote_below(q)) wait_for(prog + (q - 1), imin(kBlock * q + 2, g.nj));
      VHP_LAT_STAMP(unit, q, 1);
      ys.run();
      VHP_LAT_STAMP(unit, q, 3);
      lds_publish(prog + q, 0x3fff);
      sim_progress();
    }
  }

  // the storer of sweeper ws: record after record until the sweeper says it has swept its last band
  template <int DX, int DY, bool XM>
  VHP_FN void store(int ws, int sx, int sy, OutT* field) {
    int* post = post_of(sh, ws);
    const double* tile = sh.lds + sh.L.tiles + ws * kBandTileDoubles;
    typename std::conditional<XM, BandXStore<DX, DY, OutT, ODD>, BandYStore<DX, DY, OutT>>::type st;
    st.init(a.m, a.dmap, sx, sy, field, tile, sh.ring(ws));
    for (int n = 1;; ++n) {
      int seq, ta, cw, bf;
      VHP_WP_T0(ts0);
      lds_read4(post + kPostRec, seq, ta, cw, bf);
      while (seq < n) { short_backoff(); sim_point(); lds_read4(post + kPostRec, seq, ta, cw, bf); }
      lds_acquire();
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
      VHP_WP_ADDP(prof, 6, ts0);
#endif
      VHP_WP_T0(ts1);
      const int band = bf & 0xffff, flags = bf >> 16;
      if (flags & kPostDone) return;
      if (flags & kPostZero) {
        lds_publish(post + kPostTaken, n);  // (the record is in registers: the sweeper may go on)
        st.zero_fill(band, ta, cw);
      } else {
        st.flush(band, ta, cw, post + kPostTaken, n);
      }
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
      VHP_WP_ADDP(prof, 15, ts1); prof[12] += 1;
#endif
      sim_progress();
    }
  }

  // the whole life of this wavefront in unit `unit` (8 * source + 2 * quadrant + {0: x-major, 1: y-major})
  // (with H workgroups per unit: workgroup g is workgroup g % H of unit g / H -- the workgroups of a unit NEXT to each other in the
  // launch.  A workgroup waits for workgroups of its own unit only, and the dispatcher hands out a launch's workgroups in order: whatever
  // part of the chip the launch gets -- another stream's kernels beside it, fewer CUs than it was sized for --, at most one unit at
  // the frontier of the dispatch is incomplete, every unit before it can finish, and the CUs they free go to that one)
  VHP_FN void run(int wg) {
    half = H > 1 ? wg % H : 0;
    const int unit = H > 1 ? wg / H : (a.order ? uniform(a.order[wg]) : wg);
    const int s = unit / kUnits, qo = unit - s * kUnits;
    // (the planners' control words together, before any of them is looked at: one trip to memory, then the source's)
    int sx, sy;
    const int slot0 = a.slot_base ? *a.slot_base : 0;
    if (a.pivot_rec) {  // the planner's loop: {done, nb, x, y} in one load
      int done, nb;
      g_load_rec4(a.pivot_rec, 0, done, nb, sx, sy);
      if (done != 0) return;
    } else {
      const int skip = a.skip ? *a.skip : 0, run = a.run_if ? *a.run_if : 1, si0 = a.src_index ? *a.src_index : s;
      if (uniform(skip) != 0 || uniform(run) == 0) return;
      const int si = uniform(si0);
      sx = uniform(a.src_xy[2 * si]);
      sy = uniform(a.src_xy[2 * si + 1]);
    }
    if (a.slot_base && sx < 0) return;
    if (sx < 0 || sy < 0 || sx >= a.m.nx || sy >= a.m.ny) {  // units of a rejected source do nothing
      if (qo == 0 && w == 0 && half == 0) g_store_scalar_if(lane == 0, a.err_flag, vi(0), 1);
      return;
    }
    OutT* field = a.out + (size_t)(s + uniform(slot0)) * a.field_stride;
    if (w < W) {
      Poster po;
      po.init(sh, w);
      switch (qo) {
        case 0: sweep_x<+1, +1>(unit, sx, sy, po); break;
        case 1: sweep_y<+1, +1>(unit, sx, sy, po); break;
        case 2: sweep_x<-1, +1>(unit, sx, sy, po); break;
        case 3: sweep_y<-1, +1>(unit, sx, sy, po); break;
        case 4: sweep_x<-1, -1>(unit, sx, sy, po); break;
        case 5: sweep_y<-1, -1>(unit, sx, sy, po); break;
        case 6: sweep_x<+1, -1>(unit, sx, sy, po); break;
        default: sweep_y<+1, -1>(unit, sx, sy, po); break;
      }
      po.wait_taken();
      po.send(0, 0, 0, kPostDone);
      sim_progress();
      return;
    }
    const int ws = w - W;
    // Row 0 and column 0 are swept only from a source that lies on them and read as zero otherwise.  (A field that is
    // known to hold +0.0 wherever the launch does not write -- the planner's loop -- holds it there as well.)
    if (!a.dead_cells_are_zero && ws == W - 1 && half == 0) {
      if (qo == 0 && sy > 0)
        for (int x0 = 0; x0 < a.m.nx; x0 += kLanes) g_store_scalar_if(lane + x0 < a.m.nx, field, lane + x0, OutT(0));
      if (qo == 1 && sx > 0)
        for (int y0 = 0; y0 < a.m.ny; y0 += kLanes) g_store_scalar_if(lane + y0 < a.m.ny, field, (lane + y0) * a.m.nx, OutT(0));
    }
    switch (qo) {
      case 0: store<+1, +1, true>(ws, sx, sy, field); break;
      case 1: store<+1, +1, false>(ws, sx, sy, field); break;
      case 2: store<-1, +1, true>(ws, sx, sy, field); break;
      case 3: store<-1, +1, false>(ws, sx, sy, field); break;
      case 4: store<-1, -1, true>(ws, sx, sy, field); break;
      case 5: store<-1, -1, false>(ws, sx, sy, field); break;
      case 6: store<+1, -1, true>(ws, sx, sy, field); break;
      default: store<+1, -1, false>(ws, sx, sy, field); break;
    }
  }
};

}  // namespace pool
}  // namespace vhp
