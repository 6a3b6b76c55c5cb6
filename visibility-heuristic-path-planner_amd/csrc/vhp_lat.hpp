// vhp_lat.hpp -- the latency sweep: a FEW sources (one, in the planner's loop) as fast as their dependency chain allows.
//
// Replaces computeVisibility() (/root/reference/src/visibilityBasedSolver.cpp:570-696) for launches too small to fill the
// chip.  There the time of a launch is the time of its longest chain -- the ni steps of a quadrant's march, one after the
// other, each of them in the strip that is just growing along the diagonal -- and a wavefront that runs alone on its SIMD
// pays for every instruction it issues (4-5 cycles whatever its kind, ~250 cycles for a round trip to the LDS, ~11 per
// dependent fp64 operation: tools/ldsbench.hip).  So this kernel is built around instruction count per step:
//   * ONE workgroup per unit (octant), strips bound to wavefronts statically (strip p to wavefront p mod W);
//   * windows of 16 steps, aligned to 16 cells of the marching coordinate: an x-major window is 16 adjacent cells of every
//     row, so the tile is 16 columns, flushed by its own wavefront right after the window -- no pending lines, no line
//     phases; heads, tails and ragged ends are the same window (steps that do not exist leave garbage where it does no harm),
//     not a step-by-step path;
//   * the operands of window n+1 (the boundary values of the strip below STRAIGHT OUT OF ITS WRITER'S RING, the writer's
//     header read before and after them; the reciprocals of the step indices) are requested while window n's cells leave,
//     and checked when they are needed;
//   * the boundary values an x-major strip produces go to its ring once per window (out of the tile's last row);
//   * y-major strips own one column per lane (64 columns), and the diagonal they start from comes from a wavefront of
//     their workgroup that runs the two-term recurrence ahead of them into LDS (and then sweeps strips like the others);
//   * a strip whose values are all +0.0, below a strip that has said the same, is DEAD: everything it would still compute is
//     +0.0 (the stencil of zeros, times an occupancy), so it says so, stops sweeping and stores the zeros of its remaining
//     cells.  In a maze that is most of a sweep: from the pivots of the reference's maze_6 run the light of the longest-lived
//     octant is gone after 59 steps on average, of marches of 551.
// The boundary protocol is the pool sweep's (vhp_pool.hpp Link: LDS ring of the writing wavefront first, tagged lines in
// global memory as the durable copy), so a strip waits only for the strip below it and a wavefront sweeps its strips in
// rising order: whatever the number of strips and wavefronts, the lowest unfinished strip can always run.
// Grids: any height and width (on an odd width every other row of a field starts 8 bytes off the 16-byte grid: the x-major cells
// of those rows leave one by one instead of in pairs -- store_group --, everything else stores single cells anyway).
//
// Written against vhp_lanes.hpp: compiled for gfx950 (vhp_lat.hip) and for the CPU simulator (tests/sim), bit-exact against
// the oracle in both (tests/test_lat_sim.py, tests/test_gpu_lat.py).
#pragma once
#include "vhp_pool.hpp"

namespace vhp {
namespace pool {

constexpr int kLW = 16;        // steps per window
static_assert(kLW == kLW16, "Link::store_window hands over windows of the latency sweep");
constexpr int kLatDummy = 16;  // doubles of a wavefront's dummy slots (Layout::dummies holds 16 per wavefront)
constexpr int kDiagZero = 7;   // the diagonal's "all +0.0 from here" word: a free word of the context's head (vhp_pool.hpp kCtxHead = 8)

template <typename OutT>
struct LatArgs {
  Map m;
  const int32_t* src_xy;
  OutT* out;
  long long field_stride;
  int* err_flag;
  Tagged* lines;       // scratch: the boundary lines; strip p of unit u at 64 * (u * unit_blocks + p * blocks of the march)
  long long unit_blocks;
  uint64_t epoch;      // the tag of this launch (never 0, never repeated on this scratch)
  // the planner's loop (vhp_planner.hip.h): the launch sweeps source number *src_index of src_xy (the current pivot) into field 0,
  // or nothing at all if *skip is set (the loop has ended: launches are enqueued ahead of the host's polls); both may be null
  const int* src_index;
  const int* skip;
  // ... or both and the pivot itself as one 16-byte record {done, nb, x, y} that the loop's epilogue keeps beside its control block: one
  // trip to memory before the sweep can start instead of two (the band sweep; null: the two words above)
  const int* pivot_rec = nullptr;
  // the speculative planner's launches: field s of the launch is field *slot_base + s of `out`, a source with a negative x is no
  // source (its units do nothing, no error), and the launch sweeps nothing unless *run_if is set; both may be null
  const int* slot_base = nullptr;
  const int* run_if = nullptr;
  // the field is known to hold +0.0 wherever this launch does not write: dead strips store nothing (the planner's loop keeps two
  // local fields and clears the one that is not in use while it reads the other: vhp_planner.hip.h)
  bool dead_cells_are_zero;
  unsigned long long* strip_times;  // diagnostic builds (tools/lat_timeline.py): [unit][48][4] wall-clock stamps, or nullptr
  const uint64_t* dmap = nullptr;   // the occupancy packed along the grid's diagonals (vhp_band.hpp DiagMaps): what the band sweep reads
  // The band sweep's long octants: `halves` workgroups per unit (1 or 2), workgroup g = half (g / n_units) of unit g % n_units; half h
  // sweeps the bands p with (p / sweepers) % halves == h, so that 16 bands of an octant are in flight instead of 8 (vhp_band.hpp BandWorker)
  int halves = 1;
  int n_units = 0;
  // Launches with more units than the chip holds workgroups at once: workgroup g sweeps unit order[g] -- the units by falling length of
  // their march (vhp_lat.hip vhp_lat_order), so that the long ones start first and the short ones fill the CUs they leave; null: unit g
  const int* order = nullptr;
};

#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)  // diagnostic builds only: cycle accounts inside the x-major windows (they cost a few hundred cycles per window themselves)
#define VHP_WP_T0(var) const unsigned long long var = __builtin_readcyclecounter()
#define VHP_WP_ADDP(pp, slot, var) { (pp)[slot] += __builtin_readcyclecounter() - var; }
#else
#define VHP_WP_T0(var)
#define VHP_WP_ADDP(pp, slot, var) {}
#endif
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
#define VHP_LAT_STAMP(unit, p, k) do { if (a.strip_times && (p) < 48 && (unit) < 64 && (threadIdx.x & 63) == 0) a.strip_times[(((unit) * 48) + (p)) * 4 + (k)] = wall_clock64(); } while (0)
#else
#define VHP_LAT_STAMP(unit, p, k)
#endif

// 64-entry blocks of boundary-line scratch a unit can need on an nx * ny grid (an upper bound: the sources are device data)
VHP_HD long long lat_unit_blocks(int nx, int ny) {
  const int side = imax(nx, ny);
  return (long long)((side + 63) / 64 + 2) * ((side + 63) / 64 + 2);
}

// What the strips of both kinds read from the strip below for one window.  D = direction of the marching coordinate;
// v[0] = the strip below at the coordinate one step BEFORE the window's first-marched cell, v[k] (k >= 1) at the cell of
// step k - 1 of the window.
// MULTI: the build whose writer may live in another workgroup (Link::remote; the band sweep's launches with more than one workgroup
// per unit) -- the other build holds none of that code
template <int D, int NB, bool MULTI = false>
struct Below {
  bool ring;   // the values came out of the writer's ring and are still to be checked against h1 / h2
  int h1, h2;
  // a writer in another workgroup (Link::remote): the window's entries of its line in global memory as loaded when the window was
  // asked for -- lane l: step ia - 1 + l, l = 0 .. NB - 1 --, tags not looked at yet (accept), and the window's first step
  vu64 gtag;
  vd gval;
  int g_ia;
  int hd;      // the strip below's word of death, read with the values: 0 alive, else 1 + the step from which all its values are +0.0
  vd v[NB];

  // cw = lowest coordinate of the window, c_first = its first-marched coordinate (cw marching up, cw + 15 marching down)
  VHP_FN void from_ring(const Link<D>& lk, const int* dead_below, int cw, int c_first) {
    h1 = lds_peek(lk.rd_hdr);
    sim_point();
    v[0] = lds_bcast(lk.rd_ring, (c_first - D) & (kRing - 1));
    const int rw = cw & (kRing - 1);
#pragma unroll
    for (int k = 1; k < NB; ++k) v[k] = lds_bcast(lk.rd_ring, rw + (D > 0 ? k - 1 : kLW - k));
    sim_point();
    h2 = lds_peek(lk.rd_hdr);
    hd = lds_peek(dead_below);
    ring = true;
  }
  VHP_FN void from_slab(const double* bin, int cw, int c_first) {
    v[0] = lds_bcast(bin, 1 + (c_first & 63) - D);
    const int b = 1 + (cw & 63);
#pragma unroll
    for (int k = 1; k < NB; ++k) v[k] = lds_bcast(bin, b + (D > 0 ? k - 1 : kLW - k));
    ring = false;
  }
  VHP_FN void request(const Link<D>& lk, const int* dead_below, const double* bin, int cw, int c_first, int nb) {
    if (MULTI && lk.remote) {
      // the writer is in another workgroup: its word of death as the link last saw it beside the lines, its values out of its line in
      // global memory -- the loads go out now, a window ahead like the reads of a ring, and accept looks at the tags
      hd = lk.remote_dead;
      g_ia = (c_first - lk.c0) * D;
      remote_load(lk);
      ring = true;
      return;
    }
    if (lk.bin_block == nb) { from_slab(bin, cw, c_first); hd = lds_peek(dead_below); } else from_ring(lk, dead_below, cw, c_first);
  }
  VHP_FN void remote_load(const Link<D>& lk) {
    const vi st = vmin(lane_id(), NB - 1) + (g_ia - 1);
    g_load_tagged_raw(lk.line_in, lk.line_index(st), gtag, gval);
  }
  // the step from which the strip below is dead, as of the last request (0x7fffffff: alive)
  VHP_FN int dead_from() const { const int d = uniform(hd); return d != 0 ? d - 1 : 0x7fffffff; }
  // Makes sure the values are those of steps ia - 1 .. last_needed of the strip below: waits for the writer (asking again), or
  // takes the block from global memory if the writer is gone or too far ahead for its ring.
  VHP_FN void accept(Link<D>& lk, const int* dead_below, const double* bin, int cw, int c_first, int ia, int last_needed, int nb) {
    // (a strip dies at the end of a window, and all strips cut their windows alike: a window is before the death of the strip
    // below or after it, never across)
    if (dead_from() <= ia - 1) { zeros(); return; }
    if (!ring) return;
    if (MULTI && lk.remote) {
      // every entry of the window carries this launch's tag, or the writer has died at or before the window (its record beside the
      // lines: it stores no further entry) -- asked again until one of the two
      while (!wave_all(tags_are(gtag, lk.epoch))) {
        if (lk.remote_died_by(ia - 1)) { hd = lk.remote_dead; zeros(); return; }
        short_backoff();   // (every look is a trip to memory already)
        sim_point();
        remote_load(lk);
      }
      // (a window's entries are stored when the window has been swept and the record after the last of them: a window whose entries
      // are all there lies before the writer's death)
      const vi ln = lane_id();
      const vi x = (vmin(ln, NB - 1) + (ia - 1)) * D + lk.c0;
      lds_store_if(ln < NB, lk.bin, select(ln == 0, vi(1 + (c_first & 63) - D), (x & 63) + 1), gval);
      wave_sync();
      from_slab(bin, cw, c_first);
      return;
    }
    int ha = uniform(h1);
    if ((ha >> 14) == lk.rd_tag && (ha & 0x3fff) <= last_needed) {
      // Not swept yet.  The wait looks at the writer's header and at its word of death only, and the values are read ONCE after it:
      // a loop that re-read them carried the sixteen values around its back edge, and the compiler paid for that with two sets of
      // sixteen register copies in EVERY window, the ones that never wait included.
      for (;;) {
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
        lk.pp[7] += 1;
#endif
        ready_backoff();
        sim_point();
        const int d = lds_poll(dead_below);
        if (d != 0 && d - 1 <= ia - 1) { hd = d; zeros(); return; }
        ha = lds_poll(lk.rd_hdr);
        if ((ha >> 14) != lk.rd_tag || (ha & 0x3fff) > last_needed) break;
      }
      from_ring(lk, dead_below, cw, c_first);
      ha = uniform(h1);
    }
    const int hb = uniform(h2);
    // (the writer has finished that strip -- its line is (being) stored --, or it is too far ahead for its ring: a writer is at most
    // one window past what it has published, an entry of step s is safe while published - s <= kRingSafe)
    if ((ha >> 14) == lk.rd_tag && (hb >> 14) == lk.rd_tag && (hb & 0x3fff) - (ia - 1) <= kRingSafe) {
      sim_count(0);
      ring = false;
      return;
    }
    sim_count(3);
    lk.fetch(ia, last_needed, nb);
    from_slab(bin, cw, c_first);
  }
  VHP_FN void zeros() {
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = vd(0.0);
    ring = false;
  }
};

// +0.0 into the cells [x0, x1] x [y0, y1] of a field of even pitch nx (what a dead strip leaves behind): pairs of cells as 16-byte
// stores, two rows of 64 columns per instruction; an odd first or last column as 8-byte stores, 64 rows per instruction.
template <typename OutT>
VHP_FN void lat_zero_rect(OutT* out, int nx, int x0, int x1, int y0, int y1) {
  if (x0 > x1 || y0 > y1) return;
  constexpr int CB = sizeof(OutT);
  const vi lane = lane_id();
  if ((nx & 1) != 0 || (reinterpret_cast<uintptr_t>(out) & (2 * CB - 1)) != 0) {
    // an odd pitch (or a field that starts off the pair grid): no pair is aligned in every row -- single cells, 64 columns of a row
    // per instruction
    for (int y = y0; y <= y1; ++y) {
      for (int xc = x0; xc <= x1; xc += kLanes) g_store_scalar_if((lane + xc) <= x1, out, lane + (y * nx + xc), OutT(0));
      if (((y - y0) & 63) == 63) sim_point();
    }
    return;
  }
  const int xe = x0 + (x0 & 1);            // the first even column: (y * nx + x) is even there, whatever the row
  const int np = (x1 + 1 - xe) >> 1;       // whole pairs
  const vi pi = lane & 31, ro = lane >> 5;
  for (int pc = 0; pc < np; pc += 32) {
    const vb col_ok = (pi + pc) < np;
    const vu32 off = to_u32((ro * nx + (pi + pc) * 2 + xe) * CB);
    OutT* row = out + (long)y0 * (long)nx;
    for (int y = y0; y <= y1; y += 2) {
      const vb ok = col_ok && ((ro + y) <= y1);
      g_store2_if(ok, vb(false), vb(false), row, off, vd(0.0), vd(0.0));
      row += 2 * (long)nx;
      if (((y - y0) & 62) == 62) sim_point();
    }
  }
  for (int e = 0; e < 2; ++e) {
    const int x = e == 0 ? x0 : x1;
    if (e == 0 ? (x0 & 1) == 0 : ((x1 & 1) != 0 || (x1 == x0 && (x0 & 1) != 0))) continue;  // (part of a pair; or the same odd column twice)
    for (int y = y0; y <= y1; y += kLanes) g_store_scalar_if((lane + y) <= y1, out, (lane + y) * nx + x, OutT(0));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// x-major strip p of a unit: rows j = 64 p + lane; steps i = 64 p .. ni - 1; cells (i, j), j <= i.
// ---------------------------------------------------------------------------------------------------------------
// ODD: pairs of cells are not 16-byte aligned in every row (an odd width, or fields that start off the pair grid): a build of its
// own, so that the even-width kernel carries none of it (the kernel sits at its register limit: one more live mask spills).
template <int DX, int DY, typename OutT, bool ODD = false>
struct LatX {
  static constexpr int CB = sizeof(OutT);
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  double* tile;   // 64 rows x 16 columns (pitch kTStride): column c = x - (lowest x of the window)
  double* slab;   // reciprocals of the step indices of two blocks of 64 coordinates (the current one and the next), indexed by x & 127
  double* bin;    // = lk.bin
  Link<DX> lk;
  int p, j0, rows_here, i_first, i_last;
  bool below, has_consumer;
  int blk;        // the current block (x >> 6): its occupancy words are in ow
  int pf_blk;     // the block whose operands wait in ow_nx / rv_nx (requested when the current block began), or -1 ...
  int staged_blk; // ... and the block whose reciprocals were put into the slab last
  bool pf_wait;   // the loads of ow_nx / rv_nx have not been waited for yet
  int* dead_mine;          // my word of death (0: alive; else 1 + the step from which every value of the strip is +0.0) ...
  const int* dead_below;   // ... and the strip below's
  bool skip_fill;          // a dead strip stores nothing (LatArgs::dead_cells_are_zero)
  vi lane, tile_l, fl_t;
  vu32 fl_off;
  static constexpr bool odd_pitch = ODD;
  vd prev, jd;
  vu64 ow, ow_nx;
  vd rv_nx;
  Below<DX, kLW + 1> nx;   // the next window's operands from the strip below ...
  vd nx_rr[kLW];           // ... and the reciprocals of its step indices
  int nx_ia;               // that window's lowest step (-0x7fffffff: none)

  // (the caller has initialised lk)
  VHP_FN void init(const Map& m_, int sx, int sy, OutT* out_, const Shared& sh, int w, int p_) {
    m = m_; out = out_;
    g.init(m.nx, m.ny, sx, sy);
    tile = sh.lds + sh.L.tiles + w * kXRows * kTStride;
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);
    bin = lk.bin;
    lane = lane_id();
    tile_l = lane * kTStride;
    {
      // flush geometry: lane -> (row slot = lane >> 3, piece = lane & 7 = cells 2 * piece, + 1 of the window's 16); the row slots
      // of a store instruction are counted upward in y, so that byte offsets from its lowest row are never negative
      const vi rslot = lane >> 3, pc = lane & 7;
      const vi rs = DY > 0 ? rslot : 7 - rslot;
      fl_t = rslot * kTStride + pc * 2;
      fl_off = to_u32((rs * m.nx + pc * 2) * CB);
    }
    p = p_;
    j0 = kXRows * p;
    rows_here = imin(kXRows, g.rows_total - j0);
    i_first = j0;
    i_last = g.ni - 1;
    below = p > 0;
    has_consumer = p + 1 < g.Px;
    dead_mine = sh.owner(0) + p;
    dead_below = sh.owner(0) + (p > 0 ? p - 1 : p);
    skip_fill = false;
    nx.hd = 0;
    prev = vd(0.0);
    jd = to_f64(lane + j0);
    pf_blk = -1;
    pf_wait = false;
    staged_blk = -0x7fffffff;
    blk = -0x7fffffff;
    nx_ia = -0x7fffffff;
  }

  // the occupancy word of every lane's row and the reciprocals of the step indices of the 64 coordinates of block blk (x >> 6)
  VHP_FN void load_ops(int blk, vu64& o, vd& rv) {
    const vi yl = (vmin(lane + j0, g.rows_total - 1)) * DY + g.sy;
    o = g_load_u64(m.rows, yl * m.wpr + (1 + blk));
    const vi it = (lane + (blk * 64 - g.sx)) * DX;
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
  // the reciprocals of the block after the current one into the other half of the slab (its loads were issued when the current
  // block began: a window ago at the least), so that the windows of that block can be requested before it begins
  // (waiting for a global load waits for every global store issued before it as well: one counter, out of order between the two
  // kinds.  Settling the next block's loads right after a window's compute, when the last stores are oldest, was measured: +3 %)
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

  // requests the operands of the window whose lowest x is xw (lowest step ia) in block nb; nothing is waited for
  VHP_FN void request(int ia, int xw, int nb) {
#pragma unroll
    for (int k = 0; k < kLW; ++k) nx_rr[k] = lds_bcast(slab, (xw & (2 * kBlock - 1)) + (DX > 0 ? k : kLW - 1 - k));
    if (below) {
      if (VHP_DIAG_WAITS) nx.request(lk, dead_below, bin, xw, DX > 0 ? xw : xw + kLW - 1, nb); else nx.ring = false;
    }
    nx_ia = ia;
  }

  // The 16 cells of every row of the window at xw (lowest step ia) leave: fa[u], fb[u] = the lane's pair of cells of the rows of
  // group u (8 rows).  A cell (i', j) exists for j <= i' <= lim.
  // One group of 8 rows of a window: the lane's pair (cells ok0, ok1).  On an odd pitch the pairs of every other row lie 8 bytes
  // off the 16-byte grid: those rows' cells leave one by one.
  VHP_FN void store_group(OutT* base, const vb& ok0, const vb& ok1, const vd& a, const vd& b) {
    if (!odd_pitch) { g_store2_if(ok0 && ok1, ok0, ok1, base, fl_off, a, b); return; }
    // (every row's cells one by one: telling the rows whose pairs ARE aligned apart costs a live lane mask, and this kernel sits
    // at its register limit -- with the mask the fp64 build spilled 10 registers)
    g_store2_if(vb(false), ok0, vb(false), base, fl_off, a, b);
    g_store2_if(vb(false), vb(false), ok1, base, fl_off, a, b);
  }
  template <bool DIAG>
  VHP_FN void store_window(int ia, int xw, int lim, const vd (&fa)[8], const vd (&fb)[8]) {
    {
      // Groups of 8 rows (u): whole (every cell a computed cell of a row of this strip: 16-byte stores), none (skipped), or cell
      // by cell.  A cell (i', j) exists for j <= i' <= lim; the window's steps are ia .. ia + 15.
      OutT* base = out + (long)(DY > 0 ? g.Y(j0) : g.Y(j0 + 7)) * (long)m.nx + xw;
      const long base_step = (long)(8 * DY) * m.nx;
      if (!DIAG && !odd_pitch && rows_here == kXRows && ia + kLW - 1 <= i_last) {  // past the diagonal, inside the march, all 64 rows: every group whole
#pragma unroll
        for (int u = 0; u < 8; ++u) { g_store2(base, fl_off, fa[u], fb[u]); base += base_step; }
      } else {
      const int top = imin(ia + kLW - 1, lim) - j0;                      // rows up to j0 + top have cells in this window
      const int u_end = top >= 0 ? imin(top / 8 + 1, (rows_here + 7) >> 3) : 0;
      int u_full = 0;
      if (ia + kLW - 1 <= lim && ia - j0 >= 7) u_full = imin((ia - j0 - 7) / 8 + 1, rows_here >> 3);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < u_full) {
          if (odd_pitch) store_group(base, vb(true), vb(true), fa[u], fb[u]); else g_store2(base, fl_off, fa[u], fb[u]);
        } else if (u < u_end) {
          const vi cc = (lane & 7) * 2;
          const vi s0 = DX > 0 ? cc + ia : (-cc) + (kLW - 1 + ia), s1 = s0 + DX;  // step indices of the pair's two cells
          const vi r = (lane >> 3) + 8 * u;
          const vb row_ok = r < rows_here;
          const vi jr = r + j0;
          const vb ok0 = row_ok && (s0 >= jr) && (s0 <= lim);
          const vb ok1 = row_ok && (s1 >= jr) && (s1 <= lim);
          store_group(base, ok0, ok1, fa[u], fb[u]);
        }
        base += base_step;
      }
      }
    }
  }

  // One window: steps ia + k, k = 0 .. 15, at x = xw + (k marching up, 15 - k marching down).  DIAG: the strip's diagonal may fall
  // into it (rows switch on one by one: the diagonal cell of row j takes the NEW value of the row below it times its own
  // occupancy).  A window that sticks out of the march (before the strip's first step, past the last) is swept
  // like any other: the steps that do not exist leave garbage where garbage does no harm -- a row is garbage until its diagonal
  // cell switches it on, cells of steps outside j <= i <= i_last are never stored, and no strip reads another's values of such
  // steps (reciprocal 0 for a step that does not exist: the ratio is 0 and nothing overflows).  more: the next window belongs
  // to the same block (its operands are requested while this one's cells leave).
  template <bool DIAG>
  VHP_FN void window(int ia, int xw, int nb, bool more) {
    const int k_hi = imin(kLW - 1, i_last - ia);
    VHP_WP_T0(tw0);
    if (nx_ia != ia) request(ia, xw, nb);
    nx_ia = -0x7fffffff;
    if (below) nx.accept(lk, dead_below, bin, xw, DX > 0 ? xw : xw + kLW - 1, imax(ia, i_first), ia + k_hi, nb);
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
    pin(nx_rr[kLW - 1]);
#endif
    if (DIAG) VHP_WP_ADDP(lk.pp, 8, tw0);
    VHP_WP_T0(tw1);
    const vu32 hs = half_shifted(ow, xw & 63, xw & 31);  // the window's 16 occupancy bits: bit c = the cell at x = xw + c
    vd di = vd((double)ia);
#pragma unroll
    for (int k = 0; k < kLW; ++k) {
      const int c = DX > 0 ? k : kLW - 1 - k;
      {
        // nx.v[k] = the row below at x(step k) - DX, the OLD neighbour of lane 0 (nx.v[k + 1] the NEW one); strip 0 has none
        const vd b = shift_up(prev, below ? nx.v[k] : vd(0.0));
        const vi mk = sbfe1(hs, c);
        vd v = and_mask(stencil(prev, b, ratio(jd, di, nx_rr[k])), mk);
        if (DIAG) {
          const vd up = shift_up(v, below ? nx.v[k + 1] : vd(1.0));  // strip 0: 1.0 = light strength at the origin
          const vb isd = lane == (ia + k - j0);
          const vd dcell = and_mask(up, mk);
          v = select(isd, dcell, v);
        }
        prev = v;
        lds_store(tile, tile_l + c, v);
        di = di + 1.0;
      }
    }
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
    pin(prev);
#endif
    if (DIAG) VHP_WP_ADDP(lk.pp, 9, tw1);
    VHP_WP_T0(tw2);
    // ---- the window's 16 cells of every row leave; its boundary values go to the ring ----
    int lim = i_last;
    if (DX < 0 && xw == 0) {
      // Marching down, the march ends at x = 1: column 0 is never swept and reads as zero.  The zero leaves with
      // the last cells of every row (one step "past" the march) instead of as a lone 8-byte store some other time.
      wave_sync();
      lds_store(tile, tile_l, vd(0.0));
      lim = i_last + 1;
    }
    wave_sync();
    vd fa[8], fb[8];
#ifndef VHP_DIAG_LAT_NOFLUSH  // diagnostic builds only (WRONG results): the sweep without tile read-out and field stores -- the bound of handing both to another wavefront
#pragma unroll
    for (int u = 0; u < 8; ++u) { fa[u] = lds_load(tile, fl_t + u * (8 * kTStride)); fb[u] = lds_load(tile, fl_t + (u * (8 * kTStride) + 1)); }
#endif
    vd bv = vd(0.0);
    if (has_consumer) bv = lds_load(tile, (lane & (kLW - 1)) + (kXRows - 1) * kTStride);  // the last row: what the strip above reads
    wave_sync();
    if (more) {
      const int xn = xw + kLW * DX;
      const bool other = (xn >> 6) != (xw >> 6);  // the next window opens the next block
      if (other) stage_next();
      request(ia + kLW, xn, other ? nb + 1 : nb);
    }
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
    pin(fa[7]);
#endif
    if (DIAG) VHP_WP_ADDP(lk.pp, 10, tw2);
    VHP_WP_T0(tw3);
    // The boundary values first, the window's own stores after: the strip above -- the one that is growing, the chain of the launch --
    // is let past its gate a window's worth of store instructions earlier (C2 100.4 -> 98.0 us, 8 sources at 1000^2 178.5 -> 172.0).
    if (has_consumer) {
      lds_store(lk.ring, (lane & (kLW - 1)) + (xw & (kRing - 1)), bv);  // (every lane: the four lanes of an entry write the same value)
      lk.publish(ia + k_hi + 1);
    }
#ifdef VHP_DIAG_NODIAGSTORE  // diagnostic builds only (WRONG results): what the stores of a strip that is growing cost its chain
    if (!DIAG)
#endif
#ifndef VHP_DIAG_LAT_NOFLUSH
    store_window<DIAG>(ia, xw, lim, fa, fb);
#endif
    if (DIAG) VHP_WP_ADDP(lk.pp, 11, tw3);
#if defined(VHP_DIAG_WINPROF) && !defined(VHP_SIM)
    if (DIAG) { VHP_WP_ADDP(lk.pp, 14, tw0); lk.pp[13] += 1; } else { VHP_WP_ADDP(lk.pp, 15, tw0); lk.pp[12] += 1; }
#endif
  }

  // the window at xw (lowest step ia) is next: if it opens a block, that block's operands become the current ones
  VHP_FN void open_block(int xw, int ia) {
    const int b = xw >> 6;
    if (b == blk) return;
    if (has_consumer) lk.store_block(g.nbx(ia - 1), blk);
    enter_block(b);
  }
  VHP_FN void run() {
    int xw = g.X(i_first) & ~(kLW - 1);
    int ia = DX > 0 ? xw - g.sx : g.sx - (xw + kLW - 1);
    if (below) nx.hd = lds_poll(dead_below);  // (a strip that starts below a dead strip need not sweep a window to find out)
    enter_block(xw >> 6);
    // runs of windows of one kind (a loop of its own per kind, so that the operands requested for the next window stay where
    // they are from one window to the next); every window requests the next one's operands, whatever its kind and block
    bool dead = below && lds_poll(dead_mine) != 0;  // (a strip below died before this one could start, and said so for it)
    while (ia <= i_last && !dead) {
      if (ia <= j0 + kXRows - 1) {
        do {
          if (dies_at(ia - 1)) { dead = true; break; }
          open_block(xw, ia);
          window<true>(ia, xw, DX > 0 ? blk - g.bx0 : g.bx0 - blk, ia + kLW <= i_last);
          ia += kLW; xw += kLW * DX;
          sim_progress();
          sim_point();
        } while (ia <= i_last && ia <= j0 + kXRows - 1);
      } else {
        do {
          if (dies_at(ia - 1)) { dead = true; break; }
          open_block(xw, ia);
          window<false>(ia, xw, DX > 0 ? blk - g.bx0 : g.bx0 - blk, ia + kLW <= i_last);
          ia += kLW; xw += kLW * DX;
          sim_progress();
          sim_point();
        } while (ia <= i_last);
      }
    }
    if (has_consumer) lk.store_block(imax(g.nbx(imin(imax(ia - 1, 0), i_last)), 0), blk);
    if (dead) {
      // Everything from step ia - 1 on is +0.0, in this strip and below it: say so (the word of death first, then the progress
      // word that lets the strip above past its gate), then store the zeros of what is left of the march.
      lds_publish(dead_mine, ia);
      lds_publish(lk.prog, 0x3fff);
      announce_death(ia - 1, g.Px);
      sim_progress();
      sim_count(4);
      if (skip_fill) return;
      vd z[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = vd(0.0);
      // the windows that the strip's diagonal runs through cell by cell as usual; what lies past it is a rectangle
      for (; ia <= i_last && ia <= j0 + kXRows - 1; ia += kLW, xw += kLW * DX) {
        store_window<true>(ia, xw, (DX < 0 && xw == 0) ? i_last + 1 : i_last, z, z);
        sim_point();
      }
      if (ia <= i_last) {
        const int xa = DX > 0 ? xw : 0, xb = DX > 0 ? g.X(i_last) : xw + kLW - 1;  // (marching down, column 0 -- never swept, SURVEY Q2 -- with it)
        const int ya = DY > 0 ? g.Y(j0) : g.Y(j0 + rows_here - 1), yb = DY > 0 ? g.Y(j0 + rows_here - 1) : g.Y(j0);
        lat_zero_rect(out, m.nx, xa, xb, ya, yb);
      }
    }
  }
  // If the strip above cannot have started yet (its first step lies past s + 1: this strip has not got there), neither has any
  // strip above that one, and all of them are dead from their first step on: they are told at once, instead of one waking the
  // next.  (A strip above that HAS started may still hold light of its own, and so may every strip above it.)
  VHP_FN void announce_death(int s, int n_strips) {
    if ((p + 1) * kXRows <= s + 1) return;
    int* dead_base = dead_mine - p;
    int* prog_base = lk.prog - p;
    for (int q0 = p + 1; q0 < n_strips; q0 += kLanes) {
      const vi q = lane + q0;
      const vb up = q < n_strips;
      lds_store_i_if(up, dead_base, q, s + 1);
      lds_acquire();
      lds_store_i_if(up, prog_base, q, 0x3fff);
    }
    lds_acquire();
  }
  // Is the strip dead from step ie on?  (Asked between windows, ie = the last step swept.)  Its rows that are switched on hold
  // +0.0 -- the others are switched on by the row below them, with its value -- and the strip below has been dead since ie or
  // before: then all this strip will ever compute is the stencil of zeros.  (The strip below's word is as old as the last
  // request: a window late at most.)
  VHP_FN bool dies_at(int ie) {
#ifdef VHP_DIAG_NODEATH  // diagnostic builds only: what the early exits are worth
    return false;
#endif
    if (ie < 0) return false;
    if (below && nx.dead_from() > ie) return false;
    return wave_all(!((lane < rows_here) && ((lane + j0) <= ie)) || is_pos_zero(prev));
  }
};

// ---------------------------------------------------------------------------------------------------------------
// y-major strip q of a unit: columns i = 64 q + lane; steps j = 64 q .. nj - 1; cells (i, j), i <= j (the diagonal cell is
// the seed diag(i), stored again with its row).  Marching down in x, the lane of "column ni" (x = 0, never swept: SURVEY
// Q2) stores the zero that column reads as, with every row that stores x = 1.
// ---------------------------------------------------------------------------------------------------------------
template <int DX>
VHP_HD int lat_ycols(int ni, int nj) {
  const int cols = imax(imin(ni, nj - 1), 0);  // columns with computed cells (j > i)
  return cols + ((DX < 0 && cols == ni && cols > 0) ? 1 : 0);
}

template <int DX, int DY, typename OutT>
struct LatY {
  static constexpr int CB = sizeof(OutT);
  // Lanes are laid along x, lowest x in lane 0 (a row's 64 cells leave as one ascending 512-byte piece whatever the marching
  // direction): marching down in x the columns run against the lanes, REV, and the neighbour of a column sits one lane up.
  static constexpr bool REV = DX < 0;
  static constexpr int kEdge = REV ? 0 : kLanes - 1;  // the lane of the strip's last column: what the strip above reads
  Map m;
  Quad<DX, DY> g;
  OutT* out;
  double* slab;   // reciprocals of the step indices of two blocks of 64 coordinates (the current one and the next), indexed by y & 127
  double* bin;
  double* dummy;
  Link<DY> lk;
  int q, n_q, i0, j_first, j_last;
  bool below, has_consumer, interior;
  int blk, pf_blk, staged_blk;  // (as in LatX)
  bool pf_wait;
  int* dead_mine;
  const int* dead_below;
  bool skip_fill;
  const int* diag_zero;  // the diagonal's word: 0 unknown, else 1 + the entry from which all of it is +0.0
  const int* diag_ready; // entries of the diagonal that are in diag_lds ...
  int diag_seen;         // ... as of the last look
  int seeds_total;       // entries the diagonal has in all
  const double* diag_lds;
  vi lane, ic;
  vi first_j;     // the first step at which the lane stores its cell of the row (0x7fffffff: never)
  vb zero_lane;
  vu32 xoff;
  vd prev, id, dg;
  vu64 ow, ow_nx;
  vd rv_nx;
  Below<DY, kLW> nx;
  vd nx_rr[kLW];
  int nx_ja;

  // (the caller has initialised lk; the seeds of the strip's columns are in diag_lds)
  VHP_FN void init(const Map& m_, int sx, int sy, OutT* out_, const Shared& sh, int w, int q_, int n_strips, const double* diag_lds_) {
    m = m_; out = out_;
    g.init(m.nx, m.ny, sx, sy);
    slab = sh.lds + sh.L.slabs + w * (2 * kBlock);
    bin = lk.bin;
    dummy = sh.lds + sh.L.dummies + w * kLatDummy;
    lane = lane_id();
    q = q_;
    n_q = n_strips;
    i0 = kBlock * q;
    j_first = i0;
    j_last = g.nj - 1;
    below = q > 0;
    has_consumer = q + 1 < n_strips;
    dead_mine = sh.owner(0) + q;
    dead_below = sh.owner(0) + (q > 0 ? q - 1 : q);
    diag_zero = sh.ctx(0) + kDiagZero;
    diag_ready = sh.ctx(0) + kDiagReady;
    skip_fill = false;
    diag_seen = 0;
    diag_lds = diag_lds_;
    nx.hd = 0;
    ic = REV ? (-lane) + (i0 + kLanes - 1) : lane + i0;
    zero_lane = (ic == g.ni) && (DX < 0);
    // a column stores from its seed on (i <= j); the lane of "column ni" stores its zero with every row that stores x = 1
    first_j = select(ic < g.ni, ic, select(zero_lane, vi(g.ni - 1), vi(0x7fffffff)));
    interior = i0 + kBlock - 1 < g.ni;  // every lane a column of the quadrant
    prev = vd(0.0);
    id = to_f64(ic);
    seeds_total = g.rows_total;
    dg = vd(0.0);  // (the seeds are taken window by window, as the diagonal's wavefront delivers them: seeds_through)
    xoff = to_u32((vmin(ic, g.ni) * DX + g.sx) * CB);  // (lanes past "column ni" store nothing)
    pf_blk = -1;
    pf_wait = false;
    staged_blk = -0x7fffffff;
    blk = -0x7fffffff;
    nx_ja = -0x7fffffff;
  }

  VHP_FN void load_ops(int blk, vu64& o, vd& rv) {
    const vi xl = vmin(ic, DX < 0 ? g.ni : g.ni - 1) * DX + g.sx;
    o = g_load_u64(m.cols, xl * m.wpc + (1 + blk));
    if (DX < 0) o = select(zero_lane, vu64(0), o);  // ("column ni" computes zeros: blocked all the way)
    const vi jt = (lane + (blk * 64 - g.sy)) * DY;
    const vb ok = (jt >= 0) && (jt < g.nj);
    rv = select(ok, g_load_f64(m.recip, select(ok, jt, vi(0))), vd(0.0));
  }
  VHP_FN bool block_in_march(int b) const { const int ye = g.Y(j_last); return DY > 0 ? 64 * b <= ye : 64 * b + 63 >= ye; }
  VHP_FN void prefetch_ops(int b) { pf_blk = b; pf_wait = true; load_ops(b, ow_nx, rv_nx); }
  VHP_FN void stage(int b, vd rv) {
    wave_sync();
    lds_store(slab, lane + kBlock * (b & 1), rv);
    wave_sync();
    staged_blk = b;
  }
  // (waiting for a global load waits for every global store issued before it as well: one counter, out of order between the two
  // kinds.  Settling the next block's loads right after a window's compute, when the last stores are oldest, was measured: +3 %)
  VHP_FN void settle() {
    if (pf_wait) { pin(ow_nx); pin(rv_nx); pf_wait = false; }
  }
  VHP_FN void stage_next() {
    if (pf_blk != -1 && staged_blk != pf_blk) { settle(); stage(pf_blk, rv_nx); }
  }
  VHP_FN void enter_block(int b) {
    if (pf_blk == b) { stage_next(); ow = ow_nx; }
    else { vd rv; load_ops(b, ow, rv); pin(ow); pin(rv); stage(b, rv); }
    blk = b;
    if (block_in_march(b + DY)) prefetch_ops(b + DY); else pf_blk = -1;
  }
  VHP_FN void request(int ja, int yw, int nb) {
#pragma unroll
    for (int k = 0; k < kLW; ++k) nx_rr[k] = lds_bcast(slab, (yw & (2 * kBlock - 1)) + (DY > 0 ? k : kLW - 1 - k));
    if (below) {
      if (VHP_DIAG_WAITS) nx.request(lk, dead_below, bin, yw, DY > 0 ? yw : yw + kLW - 1, nb); else nx.ring = false;
    }
    nx_ja = ja;
  }

  // The seeds of the columns i <= upto are wanted: waits for the diagonal's wavefront if it has not got there, and takes every
  // lane's seed afresh (lanes whose seed is not there yet get whatever the LDS holds; they are not seeded before a later call).
  VHP_FN void seeds_through(int upto) {
    const int need = imin(upto + 1, seeds_total);
    if (diag_seen < need) {
      int have = lds_poll(diag_ready);
      while (have < need) { ready_backoff(); sim_point(); have = lds_poll(diag_ready); }
      lds_acquire();
      diag_seen = have;
    }
    dg = select(zero_lane, vd(0.0), lds_load(diag_lds, vmin(ic, seeds_total - 1)));
  }

  // One window: steps ja + k, k = 0 .. 15, at y = yw + (k marching up, 15 - k marching down).  DIAG: columns may be seeded in it
  // (implies PRED); PRED: predicated stores (columns that do not exist, or not yet; steps past the march).  Windows that stick
  // out of the march are swept like the others (see LatX::window).
  template <bool DIAG, bool PRED>
  VHP_FN void window(int ja, int yw, int nb, bool more) {
    const int k_hi = imin(kLW - 1, j_last - ja);
    if (DIAG) seeds_through(ja + kLW - 1);
    if (nx_ja != ja) request(ja, yw, nb);
    nx_ja = -0x7fffffff;
    if (below) nx.accept(lk, dead_below, bin, yw, DY > 0 ? yw : yw + kLW - 1, imax(ja, j_first), ja + k_hi - 1, nb);
    const vu32 hs = half_shifted(ow, yw & 63, yw & 31);
    double* wbase = has_consumer ? lk.ring + (yw & (kRing - 1)) : dummy;
    const vi widx = select(lane == kEdge, vi(0), vi((int)(dummy - wbase)));
    vd dj = vd((double)ja);
    const long rowstep = (long)DY * m.nx;
    OutT* row = out + (long)g.Y(ja) * (long)m.nx;
#pragma unroll
    for (int k = 0; k < kLW; ++k) {
      const int c = DY > 0 ? k : kLW - 1 - k;
      {
        const vd fill = below ? nx.v[k] : vd(0.0);
        const vd b = REV ? shift_down(prev, fill) : shift_up(prev, fill);
        vd v = and_mask(stencil(prev, b, ratio(id, dj, nx_rr[k])), sbfe1(hs, c));
        if (DIAG) v = select(ic == ja + k, dg, v);
#ifdef VHP_DIAG_NODIAGSTORE
        if (!DIAG)
#endif
#ifdef VHP_DIAG_LAT_NOFLUSH
        if (m.nx == 0x7fffffff)
#endif
        if (PRED) {
          const int j = ja + k;
          g_store1_if(first_j <= (j <= j_last ? j : -1), row, xoff, v);
        } else {
          g_store1_if(vb(true), row, xoff, v);
        }
        prev = v;
        lds_store(wbase, widx + c, v);
        dj = dj + 1.0;
      }
      row += rowstep;
    }
    if (more) {
      const int yn = yw + kLW * DY;
      const bool other = (yn >> 6) != (yw >> 6);
      if (other) stage_next();
      request(ja + kLW, yn, other ? nb + 1 : nb);
    }
    if (has_consumer) lk.publish(ja + k_hi + 1);
  }

  VHP_FN void open_block(int yw, int ja) {
    const int b = yw >> 6;
    if (b == blk) return;
    if (has_consumer) lk.store_block(g.nby(ja - 1), blk);
    enter_block(b);
  }
  VHP_FN int kind_of(int ja) const { return ja <= i0 + kBlock - 1 ? 0 : (!interior || ja + kLW - 1 > j_last) ? 1 : 2; }
  VHP_FN void run() {
    int yw = g.Y(j_first) & ~(kLW - 1);
    int ja = DY > 0 ? yw - g.sy : g.sy - (yw + kLW - 1);
    if (below) nx.hd = lds_poll(dead_below);
    enter_block(yw >> 6);
    // runs of windows of one kind (see LatX::run)
    bool dead = below && lds_poll(dead_mine) != 0;
    while (ja <= j_last && !dead) {
      const int kind = kind_of(ja);
      if (kind == 0) {
        do {
          if (dies_at(ja - 1)) { dead = true; break; }
          open_block(yw, ja);
          window<true, true>(ja, yw, DY > 0 ? blk - g.by0 : g.by0 - blk, ja + kLW <= j_last);
          ja += kLW; yw += kLW * DY;
          sim_progress();
          sim_point();
        } while (ja <= j_last && kind_of(ja) == 0);
      } else if (kind == 1) {
        do {
          if (dies_at(ja - 1)) { dead = true; break; }
          open_block(yw, ja);
          window<false, true>(ja, yw, DY > 0 ? blk - g.by0 : g.by0 - blk, ja + kLW <= j_last);
          ja += kLW; yw += kLW * DY;
          sim_progress();
          sim_point();
        } while (ja <= j_last && kind_of(ja) == 1);
      } else {
        do {
          if (dies_at(ja - 1)) { dead = true; break; }
          open_block(yw, ja);
          window<false, false>(ja, yw, DY > 0 ? blk - g.by0 : g.by0 - blk, ja + kLW <= j_last);
          ja += kLW; yw += kLW * DY;
          sim_progress();
          sim_point();
        } while (ja <= j_last && kind_of(ja) == 2);
      }
    }
    if (has_consumer) lk.store_block(imax(g.nby(imin(imax(ja - 1, 0), j_last)), 0), blk);
    if (dead) {  // (see LatX::run)
      lds_publish(dead_mine, ja);
      lds_publish(lk.prog, 0x3fff);
      announce_death(ja - 1);
      sim_progress();
      sim_count(4);
      if (skip_fill) return;
      const long rowstep = (long)DY * m.nx;
      OutT* row = out + (long)g.Y(ja) * (long)m.nx;
      int j = ja;
      // row by row while columns are still being seeded; below that every column of the strip is stored in every row: a rectangle
      for (; j <= j_last && j <= i0 + kBlock - 1; ++j) {
        g_store1_if(first_j <= j, row, xoff, vd(0.0));
        row += rowstep;
        if (((j - ja) & (kLW - 1)) == kLW - 1) sim_point();
      }
      if (j <= j_last) {
        const int c_hi = imin(i0 + kBlock - 1, DX < 0 ? g.ni : g.ni - 1);  // the strip's last column (marching down: "column ni", x = 0, if it is here)
        const int xa = DX > 0 ? g.X(i0) : g.sx - c_hi, xb = DX > 0 ? g.X(c_hi) : g.X(i0);
        const int ya = DY > 0 ? g.Y(j) : g.Y(j_last), yb = DY > 0 ? g.Y(j_last) : g.Y(j);
        lat_zero_rect(out, m.nx, xa, xb, ya, yb);
      }
    }
  }
  // (see LatX::announce_death; a y-major strip is seeded by the diagonal as well: only strips whose every seed is +0.0 -- the
  // diagonal's wavefront says from where on, if it has found out -- are dead before they start)
  VHP_FN void announce_death(int s) {
    int* dead_base = dead_mine - q;
    int* prog_base = lk.prog - q;
    const int dz = lds_poll(diag_zero);
    if (dz == 0 || (q + 1) * kBlock <= s + 1 || (q + 1) * kBlock < dz - 1) return;
    for (int q0 = q + 1; q0 < n_q; q0 += kLanes) {
      const vi qq = lane + q0;
      const vb up = qq < n_q;
      lds_store_i_if(up, dead_base, qq, s + 1);
      lds_acquire();
      lds_store_i_if(up, prog_base, qq, 0x3fff);
    }
    lds_acquire();
  }
  // Is the strip dead from step je on?  Its columns that have been seeded hold +0.0, the seeds of the others are +0.0, and the
  // strip below has been dead since je or before.
  VHP_FN bool dies_at(int je) {
#ifdef VHP_DIAG_NODEATH
    return false;
#endif
    if (je < 0) return false;
    if (below && nx.dead_from() > je) return false;
    if (je < i0 + kBlock - 1) {  // columns still to be seeded: their seeds have to be there, and +0.0
      if (lds_poll(diag_ready) < imin(i0 + kBlock, seeds_total)) return false;
      seeds_through(i0 + kBlock - 1);
    }
    return wave_all((first_j == 0x7fffffff) || is_pos_zero(select(ic <= je, prev, dg)));
  }
};

// ---------------------------------------------------------------------------------------------------------------
// The diagonal of a quadrant for its y-major unit, into LDS, 64 entries per call.  diag(0) = occ(source); for k >= 1:
//   sub(k)  = V(k, k-1) = (a - c*(a - b)) * occ(k, k-1),  a = diag(k-1), b = sub(k-1), c = (k-1)/k
//   diag(k) = sub(k) * occ(k, k)                                            (the stale diagonal)
// The ratios and the occupancy bits of a chunk are lane work; the chain is one value after the other.
// ---------------------------------------------------------------------------------------------------------------
template <int DX, int DY>
struct LatDiag {
  Map m;
  Quad<DX, DY> g;
  double* diag;
  int k;
  vi lane;
  vd dprev, sprev;
  bool zero_rest;  // diag(k-1) and sub(k-1) are +0.0: so is everything after them ...
  int zero_from;   // ... i.e. the entries from this one on

  VHP_FN void init(const Map& m_, int sx, int sy, double* diag_lds) {
    m = m_;
    g.init(m.nx, m.ny, sx, sy);
    diag = diag_lds;
    lane = lane_id();
    zero_rest = false;
    zero_from = 0;
    k = 0;
    dprev = vd(0.0);
    sprev = vd(0.0);
  }
  VHP_FN bool done() const { return k >= g.rows_total; }
  // entries k .. k+63; returns the number of entries ready afterwards
  VHP_FN int run_chunk(int* ready_word) {
    const int k0 = k, k1 = imin(k0 + kBlock, g.rows_total);
    if (zero_rest) {
      wave_sync();
      lds_store_if(lane < (k1 - k0), diag, lane + k0, vd(0.0));
      k = k1;
      return k1;
    }
    const vi kk = vmin(lane + k0, g.rows_total - 1);
    const vi x = kk * DX + g.sx;
    const vi ya = vmax(kk - 1, 0) * DY + g.sy, yb = kk * DY + g.sy;
    const vu64 wa = g_load_u64(m.rows, ya * m.wpr + ((x >> 6) + 1));
    const vu64 wb = g_load_u64(m.rows, yb * m.wpr + ((x >> 6) + 1));
    const vd rk = g_load_f64(m.recip, kk);
    const vi ma = bit_mask_lane(wa, x & 63), mb = bit_mask_lane(wb, x & 63);
    const vd cv = ratio(to_f64(vmax(kk - 1, 0)), to_f64(kk), rk);  // (k-1)/k of every entry of the chunk
    vd acc = vd(0.0);
    for (int kq = k0; kq < k1; ++kq) {
      const int l = kq - k0;
      vd dcur;
      if (kq == 0) {
        dcur = and_mask(vd(1.0), vi(read_lane_i(mb, l)));  // the origin: light strength 1 times its occupancy
        sprev = vd(0.0);
      } else {
        const vd sub = and_mask(stencil(dprev, sprev, vd(read_lane(cv, l))), vi(read_lane_i(ma, l)));
        dcur = and_mask(sub, vi(read_lane_i(mb, l)));
        sprev = sub;
      }
      dprev = dcur;
      acc = select(lane == l, dcur, acc);
      if ((l & (kLW - 1)) == kLW - 1 && kq + 1 < k1) {  // a window's worth of seeds: the strips need not wait for the whole chunk
        wave_sync();
        lds_store_if((lane <= l) && (lane > l - kLW), diag, lane + k0, acc);
        lds_publish(ready_word, kq + 1);
        sim_progress();
      }
    }
    wave_sync();
    lds_store_if((lane < (k1 - k0)) && (lane >= ((k1 - k0 - 1) & ~(kLW - 1))), diag, lane + k0, acc);
    k = k1;
#ifndef VHP_DIAG_NODEATH
    if (!zero_rest && k1 > 0 && wave_all(is_pos_zero(dprev) && is_pos_zero(sprev))) {
      zero_rest = true;
      zero_from = k1;
    }
#endif
    return k1;
  }
};

// the launch needs the ODD build of the kernel: some pair of cells that an x-major strip stores is not aligned to its size
template <typename OutT>
VHP_HD bool lat_needs_odd(int nx, long long field_stride, const OutT* out) {
  return (nx & 1) != 0 || (field_stride & 1) != 0 || (reinterpret_cast<uintptr_t>(out) & (2 * sizeof(OutT) - 1)) != 0;
}

template <typename OutT, bool ODD = false>
struct LatWorker {
  static constexpr int kRoles = 1;  // wavefronts per strip wavefront (the band sweep, vhp_band.hpp, has a storer beside every sweeper)
  static constexpr int kTilePitch = kTStride;
  LatArgs<OutT> a;
  Shared sh;
  int w, W;
  vi lane;
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
  unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (Worker::prof: [1] waiting for / copying boundary values, [4] block-start loads)
#endif

  VHP_FN void init(const LatArgs<OutT>& a_, double* lds, const Layout& L, int w_) {
    a = a_;
    sh.lds = lds;
    sh.L = L;
    w = w_;
    W = L.W;
    lane = lane_id();
  }
  // Before any wavefront runs: every thread of the workgroup calls this (tid of nthreads), then a barrier.
  static VHP_FN void clear(double* lds, const Layout& L, int tid, int nthreads) { Worker<OutT>::clear(lds, L, tid, nthreads); }

  VHP_FN Tagged* line_of(int unit, int p, int nb) const { return a.lines + (size_t)64 * ((size_t)unit * (size_t)a.unit_blocks + (size_t)p * nb); }
  static VHP_FN int tag_of(int p) { return (1 << 8) | p; }  // (never 0: a cleared header belongs to no strip)

  // a coarse gate ahead of a strip's first window (which then checks exactly what it reads): one word per poll
  VHP_FN void wait_for(const int* word, int at_least) {
    while (lds_poll(word) < at_least) { ready_backoff(); sim_point(); }
    lds_acquire();
  }

  template <int DX, int DY>
  VHP_FN void run_x(int unit, int sx, int sy, OutT* field) {
    Quad<DX, DY> g;
    g.init(a.m.nx, a.m.ny, sx, sy);
    int* prog = sh.prog(0);
    for (int p = w; p < g.Px; p += W) {
      LatX<DX, DY, OutT, ODD> xs;
      xs.lk.init(sh, w, sx, kXRows * p, tag_of(p), prog + p, p > 0 ? line_of(unit, p - 1, g.Nbx) : nullptr, p + 1 < g.Px ? line_of(unit, p, g.Nbx) : nullptr,
                 a.epoch, p > 0 ? (p - 1) % W : -1, p > 0 ? tag_of(p - 1) : 0);
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
      xs.lk.pp = prof;
#endif
      xs.init(a.m, sx, sy, field, sh, w, p);
      xs.skip_fill = a.dead_cells_are_zero;
      xs.prefetch_ops(g.X(xs.i_first) >> 6);
      VHP_LAT_STAMP(unit, p, 0);
      if (VHP_DIAG_WAITS && p > 0) wait_for(prog + (p - 1), imin(kXRows * p + 2, g.ni));  // the strip below has got to my rows
      VHP_LAT_STAMP(unit, p, 1);
      xs.run();
      VHP_LAT_STAMP(unit, p, 3);
      lds_publish(prog + p, 0x3fff);  // finished (a march can end before the first window of the strip above does)
      sim_progress();
    }
  }

  template <int DX, int DY>
  VHP_FN void run_y(int unit, int sx, int sy, OutT* field) {
    Quad<DX, DY> g;
    g.init(a.m.nx, a.m.ny, sx, sy);
    int* prog = sh.prog(0);
    int* cx = sh.ctx(0);
    double* diag_lds = sh.lds + sh.L.tiles;  // (a y-major workgroup stages nothing: the tiles' space holds the diagonal)
    const int n_strips = (lat_ycols<DX>(g.ni, g.nj) + kBlock - 1) / kBlock;
    if (n_strips == 0) return;
    if (w == W - 1) {  // the last wavefront runs the diagonal ahead of the strips, then takes its share of them
      LatDiag<DX, DY> dt;
      dt.init(a.m, sx, sy, diag_lds);
      VHP_LAT_STAMP(unit, 47, 0);
      while (!dt.done()) {
        const int ready = dt.run_chunk(cx + kDiagReady);
        if (dt.zero_rest) lds_publish(cx + kDiagZero, dt.zero_from + 1);
        lds_publish(cx + kDiagReady, ready);
        if (ready <= kBlock) VHP_LAT_STAMP(unit, 47, 1);
        if (ready <= 2 * kBlock) VHP_LAT_STAMP(unit, 47, 2);
        sim_progress();
        sim_point();
      }
      VHP_LAT_STAMP(unit, 47, 3);
    }
    const int Nby = g.Nby;
    for (int q = w; q < n_strips; q += W) {
      LatY<DX, DY, OutT> ys;
      ys.lk.init(sh, w, sy, kBlock * q, tag_of(q), prog + q, q > 0 ? line_of(unit, q - 1, Nby) : nullptr, q + 1 < n_strips ? line_of(unit, q, Nby) : nullptr,
                 a.epoch, q > 0 ? (q - 1) % W : -1, q > 0 ? tag_of(q - 1) : 0);
#if defined(VHP_DIAG_POOLPROF) && !defined(VHP_SIM)
      ys.lk.pp = prof;
#endif
      VHP_LAT_STAMP(unit, q, 0);
      ys.init(a.m, sx, sy, field, sh, w, q, n_strips, diag_lds);
      ys.skip_fill = a.dead_cells_are_zero;
      ys.prefetch_ops(g.Y(ys.j_first) >> 6);
      if (VHP_DIAG_WAITS && q > 0) wait_for(prog + (q - 1), imin(kBlock * q + 1, g.nj));
      VHP_LAT_STAMP(unit, q, 1);
      ys.run();
      VHP_LAT_STAMP(unit, q, 3);
      lds_publish(prog + q, 0x3fff);
      sim_progress();
    }
  }

  // the whole life of this wavefront: its strips of unit `unit` (8 * source + 2 * quadrant + {0: x-major, 1: y-major})
  VHP_FN void run(int unit) {
    const int s = unit / kUnits, qo = unit - s * kUnits;
    // (the planners' control words together, before any of them is looked at: one trip to memory, then the source's)
    const int skip = a.skip ? *a.skip : 0, run = a.run_if ? *a.run_if : 1, si0 = a.src_index ? *a.src_index : s;
    const int slot0 = a.slot_base ? *a.slot_base : 0;
    if (uniform(skip) != 0 || uniform(run) == 0) return;
    const int si = uniform(si0);
    const int sx = uniform(a.src_xy[2 * si]), sy = uniform(a.src_xy[2 * si + 1]);
    if (a.slot_base && sx < 0) return;
    if (sx < 0 || sy < 0 || sx >= a.m.nx || sy >= a.m.ny) {  // units of a rejected source do nothing
      if (qo == 0 && w == 0) g_store_scalar_if(lane == 0, a.err_flag, vi(0), 1);
      return;
    }
    OutT* field = a.out + (size_t)(s + uniform(slot0)) * a.field_stride;
    if (qo == 0 && w == W - 1 && sy > 0) {
      // rows/columns no quadrant covers read as zero; the x-major unit of quadrant 0 always exists
      // (column 0, x = 0, y >= 1: written as zero by the units that march down to x = 1, with their last store of the row)
      for (int x0 = 0; x0 < a.m.nx; x0 += kLanes) g_store_scalar_if(lane + x0 < a.m.nx, field, lane + x0, OutT(0));
    }
    switch (qo) {
      case 0: run_x<+1, +1>(unit, sx, sy, field); break;
      case 1: run_y<+1, +1>(unit, sx, sy, field); break;
      case 2: run_x<-1, +1>(unit, sx, sy, field); break;
      case 3: run_y<-1, +1>(unit, sx, sy, field); break;
      case 4: run_x<-1, -1>(unit, sx, sy, field); break;
      case 5: run_y<-1, -1>(unit, sx, sy, field); break;
      case 6: run_x<+1, -1>(unit, sx, sy, field); break;
      default: run_y<+1, -1>(unit, sx, sy, field); break;
    }
  }
};

}  // namespace pool
}  // namespace vhp
