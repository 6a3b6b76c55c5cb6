// vhp_sweep.cuh -- CDNA4 (gfx950) device code of the visibility-transport sweep.
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
//   runs one pipeline slot (kChunk steps, one workgroup barrier) behind strip p-1,
//   which hands it its boundary lane through a small LDS ring: the LDS-staged front.
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
// Memory: occupancy is read from two bit-packed copies of the map.  A lane keeps one
// 64-bit word per owned row/column, packed along the marching direction: 64 steps of
// occupancy per load, refilled (with the reciprocal table) once per 64 steps by
// prefetched loads, so the steady-state step issues no loads at all.
//
// Arithmetic is IEEE binary64 with contraction off.  The per-cell division
// c = j/i is replaced by Markstein's correction with a host-computed table of
// correctly rounded reciprocals: q = j*y; r = fma(-i,q,j); c = fma(r,y,q), which
// oracle/markstein_check.c proves bit-identical to j/i for all j < i <= 16384.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vhp {

constexpr int kChunk = 16;       // steps per pipeline slot
constexpr int kRing = 64;        // entries of a boundary ring (>= 4*kChunk)
constexpr int kTileCols = 8;     // columns staged per flush = 64 B of fp64
constexpr int kTileStride = 9;   // doubles per staged row (odd: spreads column writes over banks)
constexpr int kUnitsPerSource = 4;

struct DevMap {
  const uint64_t* rows;  // bit x&63 of rows[y*wpr + 1 + (x>>6)] = occ(x,y); word 0 and the last word of a line are zero pads
  const uint64_t* cols;  // bit y&63 of cols[x*wpc + 1 + (y>>6)] = occ(x,y)
  const double* recip;   // recip[k] = RN(1/k), k = 1..max(nx,ny); recip[0] = 0
  int wpr, wpc;
  int nx, ny;
};

// lane l <- lane l-1, lane 0 <- fill.  DPP wave_shr:1 (gfx9 encoding 0x138); with
// bound_ctrl off the lane without a source keeps `old`, which carries the fill.
__device__ __forceinline__ double shift_up(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  const int flo = __double2loint(fill), fhi = __double2hiint(fill);
  lo = __builtin_amdgcn_update_dpp(flo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x138, 0xf, 0xf, false);
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

struct UnitGeom {
  int sx, sy, dirx, diry, ni, nj;
};

// Emit policy of the plain sweep: store cells.  pair(): cells (x, y) and (x+1, y).
template <typename OutT>
struct StoreEmit {
  OutT* __restrict__ out;
  int nx;
  __device__ __forceinline__ void pair(int x, int y, double v0, double v1, bool ok0, bool ok1) {
    OutT* p = out + (size_t)y * nx + x;
    if (ok0 && ok1) {
      struct alignas(2 * sizeof(OutT)) Two { OutT a, b; };
      *reinterpret_cast<Two*>(p) = Two{static_cast<OutT>(v0), static_cast<OutT>(v1)};  // one 16-byte (fp64) store
    } else {
      if (ok0) p[0] = static_cast<OutT>(v0);
      if (ok1) p[1] = static_cast<OutT>(v1);
    }
  }
  __device__ __forceinline__ void zero(int x, int y) { out[(size_t)y * nx + x] = OutT(0); }
  __device__ __forceinline__ void finish() {}
};

// Per-64-step refill state shared by both strip kinds: each lane holds the reciprocal
// of "its" step of the current 64-aligned block of the marching coordinate.
// (marching coordinate = x for x-major strips, y for y-major ones)

// ---------------------------------------------------------------------------
// x-major strip: rows j = j0 + R*lane + r, steps i = j0 .. ni-1, cells (i, j), i >= j.
// ---------------------------------------------------------------------------
template <int R, typename Emit>
__device__ __forceinline__ void x_strip_init(const DevMap& m, const UnitGeom& g, int j0, int rows_total, int lane,
                                             double (&jd)[R], int (&dmask)[R], const uint64_t* (&rowp)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = j0 + R * lane + r;
    const bool on = j < rows_total;
    const int y = on ? g.sy + g.diry * j : g.sy;
    const int xd = on ? g.sx + g.dirx * j : g.sx;  // x of this row's diagonal cell
    jd[r] = (double)j;
    rowp[r] = m.rows + (size_t)y * m.wpr + 1;
    dmask[r] = ((rowp[r][xd >> 6] >> (xd & 63)) & 1ull) ? -1 : 0;
  }
}

template <int R, typename Emit>
__device__ void x_strip(const DevMap& m, Emit& emit, const UnitGeom g, int p, int tmax, double* ring_base, double* tile) {
  constexpr int S = 64 * R;
  const int lane = threadIdx.x & 63;
  const int rows_total = min(g.nj, g.ni);
  const int P = (rows_total + S - 1) / S;
  const int glast = (g.ni - 1) / kChunk;
  const int j0 = p * S;
  const bool strip_on = p < P;
  double* ring_out = ring_base + p * kRing;
  const double* ring_in = ring_base + (p > 0 ? p - 1 : 0) * kRing;

  double prev[R], jd[R];
  int dmask[R];
  const uint64_t* rowp[R];
  uint64_t ow[R], own[R];   // occupancy words of the current / next 64-block of x, one per owned row
  double rv = 0.0, rvn = 0.0;  // reciprocals: lane t holds 1/i of the step whose x is (block, t)
  int cur_blk = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) { prev[r] = 0.0; jd[r] = 0.0; dmask[r] = 0; rowp[r] = m.rows + 1; ow[r] = own[r] = 0; }
  if (strip_on) {
    x_strip_init<R, Emit>(m, g, j0, rows_total, lane, jd, dmask, rowp);
    const int x0 = g.sx + g.dirx * j0;
    cur_blk = x0 >> 6;
    auto recip_of = [&](int blk) {
      const int xt = blk * 64 + lane;
      const int it = g.dirx > 0 ? xt - g.sx : g.sx - xt;
      return (it >= 0 && it < g.ni) ? m.recip[it] : 0.0;
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ow[r] = rowp[r][cur_blk];
      own[r] = rowp[r][cur_blk + g.dirx];
    }
    rv = recip_of(cur_blk);
    rvn = recip_of(cur_blk + g.dirx);
  }

  for (int T = 0; T <= tmax; ++T) {
    const int gch = T - p;
    if (strip_on && gch >= j0 / kChunk && gch <= glast) {
      const int ibeg = max(gch * kChunk, j0);
      const int iend = min(gch * kChunk + kChunk - 1, g.ni - 1);
      // boundary row of the strip below for this chunk: lane t holds its value at step ibeg-1+t
      double ringv = 0.0;
      if (p > 0) ringv = ring_in[(ibeg - 1 + lane) & (kRing - 1)];
      for (int i = ibeg; i <= iend; ++i) {
        const int x = g.sx + g.dirx * i;
        const int blk = x >> 6, t = x & 63;
        if (blk != cur_blk) {  // crossed into the next 64-block of x: rotate the prefetched words in
          cur_blk = blk;
          const int xt = (blk + g.dirx) * 64 + lane;
          const int it = g.dirx > 0 ? xt - g.sx : g.sx - xt;
          rv = rvn;
          rvn = (it >= 0 && it < g.ni) ? m.recip[it] : 0.0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            ow[r] = own[r];
            own[r] = rowp[r][blk + g.dirx];
          }
        }
        const double di = (double)i;
        const double ri = read_lane(rv, t);
        double fill = 0.0;  // OLD value of the row just below lane 0's first row
        double dsrc = 1.0;  // NEW value of that row (feeds the diagonal cell); 1.0 = light strength at the origin
        if (p > 0) {
          fill = read_lane(ringv, i - ibeg);
          dsrc = read_lane(ringv, i - ibeg + 1);
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
              if (lane == ld) v[r] = and_mask(below, dmask[r]);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          prev[r] = v[r];
          tile[(R * lane + r) * kTileStride + (x & (kTileCols - 1))] = v[r];
        }
        if (lane == 63) ring_out[i & (kRing - 1)] = v[R - 1];

        const bool endwin = g.dirx > 0 ? ((x & (kTileCols - 1)) == kTileCols - 1) : ((x & (kTileCols - 1)) == 0);
        if (endwin || i == g.ni - 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // lane <-> (row-in-group = lane>>2, column pair = lane&3): 16 rows x 64 B per pass
          const int xbase = x & ~(kTileCols - 1);
          const int cp = lane & 3, rsub = lane >> 2;
          const int xc = xbase + 2 * cp;
          const int ic0 = g.dirx > 0 ? xc - g.sx : g.sx - xc;
          const int ic1 = g.dirx > 0 ? ic0 + 1 : ic0 - 1;
          const bool c0 = ic0 >= 0 && ic0 <= i, c1 = ic1 >= 0 && ic1 <= i;
          const int rows_here = min(S, rows_total - j0);
          const int rows_live = min(rows_here, i - j0 + 1);  // rows j <= i
          for (int rb = 0; rb < rows_live; rb += 32) {
            double ta[2], tb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const double* q = tile + (rb + 16 * u + rsub) * kTileStride + 2 * cp;
              ta[u] = q[0];
              tb[u] = q[1];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int rl = rb + 16 * u + rsub;
              const int j = j0 + rl;
              const bool rowok = rl < rows_here;
              emit.pair(xc, g.sy + g.diry * j, ta[u], tb[u], rowok && c0 && j <= ic0, rowok && c1 && j <= ic1);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// y-major strip: columns i = i0 + R*lane + r, steps j = i0 .. nj-1, cells (i, j), j > i.
// ---------------------------------------------------------------------------
template <int R, typename Emit>
__device__ void y_strip(const DevMap& m, Emit& emit, const UnitGeom g, int p, int tmax, double* ring_base, double* dstate) {
  constexpr int S = 64 * R;
  const int lane = threadIdx.x & 63;
  const int cols_total = max(min(g.ni, g.nj - 1), 0);
  const int P = (cols_total + S - 1) / S;
  const int glast = (g.nj - 1) / kChunk;
  const int i0 = p * S;
  const bool strip_on = p < P;
  double* ring_out = ring_base + p * kRing;
  const double* ring_in = ring_base + (p > 0 ? p - 1 : 0) * kRing;

  double prev[R], id[R];
  const uint64_t* colp[R];
  uint64_t ow[R], own[R], b1[R], b2[R];
  double rv = 0.0, rvn = 0.0;
  int cur_blk = 0;
  double dg = 0.0, sb = 0.0;  // private diagonal recurrence state
  // the pair this lane stores each step: columns i0 + R*lane + (0..R-1) are x-consecutive
  const int icol0 = i0 + R * lane;
  const int xlo = g.dirx > 0 ? g.sx + icol0 : g.sx - icol0 - (R - 1);  // lowest x of the lane's R columns
#pragma unroll
  for (int r = 0; r < R; ++r) { prev[r] = 0.0; id[r] = 0.0; colp[r] = m.cols + 1; ow[r] = own[r] = b1[r] = b2[r] = 0; }
  if (strip_on) {
    const int y0 = g.sy + g.diry * i0;
    cur_blk = y0 >> 6;
    auto recip_of = [&](int blk) {
      const int yt = blk * 64 + lane;
      const int jt = g.diry > 0 ? yt - g.sy : g.sy - yt;
      return (jt >= 0 && jt < g.nj) ? m.recip[jt] : 0.0;
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = icol0 + r;
      const bool on = i < g.ni;
      const int x = on ? g.sx + g.dirx * i : g.sx;
      id[r] = (double)i;
      colp[r] = m.cols + (size_t)x * m.wpc + 1;
      ow[r] = colp[r][cur_blk];
      own[r] = colp[r][cur_blk + g.diry];
    }
    rv = recip_of(cur_blk);
    rvn = recip_of(cur_blk + g.diry);
    // occupancy of (X(k), Y(k-1)) and (X(k), Y(k)) for k = i0 + 64*q + lane: the two factors of
    // the diagonal recurrence at step k, gathered once and balloted (bit = lane)
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int k = i0 + 64 * q + lane;
      bool f1 = false, f2 = false;
      if (k >= 1 && k < g.ni && k < g.nj) {
        const uint64_t* col = m.cols + (size_t)(g.sx + g.dirx * k) * m.wpc + 1;
        const int yk = g.sy + g.diry * k, ykm = g.sy + g.diry * (k - 1);
        f1 = (col[ykm >> 6] >> (ykm & 63)) & 1ull;
        f2 = (col[yk >> 6] >> (yk & 63)) & 1ull;
      }
      b1[q] = __ballot(f1);
      b2[q] = __ballot(f2);
    }
    if (p == 0) {
      const uint64_t sw = m.cols[(size_t)g.sx * m.wpc + 1 + (g.sy >> 6)];
      dg = ((sw >> (g.sy & 63)) & 1ull) ? 1.0 : 0.0;  // origin = lightStrength * occ(source)
    }
  }

  for (int T = 0; T <= tmax; ++T) {
    const int gch = T - p;
    if (strip_on && gch >= i0 / kChunk && gch <= glast) {
      const int jbeg = max(gch * kChunk, i0);
      const int jend = min(gch * kChunk + kChunk - 1, g.nj - 1);
      double ringv = 0.0;
      if (p > 0) ringv = ring_in[(jbeg - 1 + lane) & (kRing - 1)];
      for (int j = jbeg; j <= jend; ++j) {
        const int y = g.sy + g.diry * j;
        const int blk = y >> 6, t = y & 63;
        if (blk != cur_blk) {
          cur_blk = blk;
          const int yt = (blk + g.diry) * 64 + lane;
          const int jt = g.diry > 0 ? yt - g.sy : g.sy - yt;
          rv = rvn;
          rvn = (jt >= 0 && jt < g.nj) ? m.recip[jt] : 0.0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            ow[r] = own[r];
            own[r] = colp[r][blk + g.diry];
          }
        }
        const double dj = (double)j;
        const double rj = read_lane(rv, t);
        // advance the private diagonal recurrence to diag(j) while j is one of this
        // strip's own columns (the previous strip hands over the state at j = i0 - 1)
        const bool own_diag = j >= i0 && j < i0 + S && j < g.ni;
        if (own_diag) {
          if (j >= 1) {
            if (j == i0) {  // p > 0 here
              dg = dstate[2 * (p - 1)];
              sb = dstate[2 * (p - 1) + 1];
            }
            const int k = j - i0;
            uint64_t m1 = b1[0], m2 = b2[0];
#pragma unroll
            for (int q = 1; q < R; ++q) {
              if ((k >> 6) == q) { m1 = b1[q]; m2 = b2[q]; }
            }
            const double cj = ratio(dj - 1.0, dj, rj);
            const double s = stencil(dg, sb, cj);
            sb = ((m1 >> (k & 63)) & 1ull) ? s : 0.0;
            dg = ((m2 >> (k & 63)) & 1ull) ? sb : 0.0;
          }
          if (j == i0 + S - 1 && lane == 0) {
            dstate[2 * p] = dg;
            dstate[2 * p + 1] = sb;
          }
        }
        double fill = 0.0;
        if (p > 0) fill = read_lane(ringv, j - jbeg);
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
            const int i = icol0;
            emit.pair(xlo, y, v[0], 0.0, i < j && i < cols_total, false);
          } else {
            const int ia = icol0 + r, ib = ia + 1;
            const bool oka = ia < j && ia < cols_total, okb = ib < j && ib < cols_total;
            if (g.dirx > 0)
              emit.pair(xlo + r, y, v[r], v[r + 1], oka, okb);
            else
              emit.pair(xlo + (R - 2 - r), y, v[r + 1], v[r], okb, oka);
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
        if (lane == 63) ring_out[j & (kRing - 1)] = v[R - 1];
      }
    }
    __syncthreads();
  }
}

// geometry of quadrant q of a source
__device__ __forceinline__ UnitGeom unit_geom(const DevMap& m, int sx, int sy, int q) {
  UnitGeom g;
  g.sx = sx;
  g.sy = sy;
  g.dirx = (q == 0 || q == 3) ? 1 : -1;  // Q1 (+,+) Q2 (-,+) Q3 (-,-) Q4 (+,-), solver.cpp:575-695
  g.diry = (q < 2) ? 1 : -1;
  g.ni = g.dirx > 0 ? m.nx - sx : sx;    // negative directions stop short of the border (Q2)
  g.nj = g.diry > 0 ? m.ny - sy : sy;
  return g;
}

inline size_t sweep_lds_bytes(int R, int W) {
  // x rings, y rings, diagonal hand-over state, W staging tiles
  return ((size_t)2 * W * kRing + 2 * W + (size_t)W * 64 * R * kTileStride) * sizeof(double);
}

// One quadrant of one source: called by all 2*W wavefronts of a workgroup.
template <int R, typename Emit>
__device__ void sweep_quadrant(const DevMap& m, Emit& emit, int sx, int sy, int q, double* lds) {
  constexpr int S = 64 * R;
  const int W = blockDim.x >> 7;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably uniform
  const UnitGeom g = unit_geom(m, sx, sy, q);
  if (q == 0) {
    // rows/columns no quadrant covers read as zero; quadrant 1 always exists
    if (sx > 0)
      for (int y = threadIdx.x; y < m.ny; y += blockDim.x) emit.zero(0, y);
    if (sy > 0)
      for (int x = threadIdx.x; x < m.nx; x += blockDim.x) emit.zero(x, 0);
  }
  if (g.ni <= 0 || g.nj <= 0) return;  // uniform for the workgroup
  const int rows_total = min(g.nj, g.ni);
  const int cols_total = max(min(g.ni, g.nj - 1), 0);
  const int Px = (rows_total + S - 1) / S, Py = (cols_total + S - 1) / S;
  const int tmax = max((g.ni - 1) / kChunk + Px - 1, (g.nj - 1) / kChunk + max(Py, 1) - 1);
  double* ring_x = lds;
  double* ring_y = lds + (size_t)W * kRing;
  double* dstate = lds + (size_t)2 * W * kRing;
  double* tiles = dstate + 2 * W;
  if (wave < W)
    x_strip<R>(m, emit, g, wave, tmax, ring_x, tiles + (size_t)wave * S * kTileStride);
  else
    y_strip<R>(m, emit, g, wave - W, tmax, ring_y, dstate);
}

// grid = n_src * 4 workgroups of 128*W threads; dynamic LDS = sweep_lds_bytes(R, W)
template <int R, typename OutT>
__global__ void __launch_bounds__(1024)
vhp_sweep_fronts(DevMap m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride,
                 int* __restrict__ err_flag) {
  extern __shared__ double lds[];
  const int s = blockIdx.x / kUnitsPerSource;
  const int q = blockIdx.x - s * kUnitsPerSource;
  const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
  if (sx < 0 || sy < 0 || sx >= m.nx || sy >= m.ny) {
    if (threadIdx.x == 0 && q == 0) atomicOr(err_flag, 1);
    return;
  }
  StoreEmit<OutT> emit{out + (size_t)s * field_stride, m.nx};
  sweep_quadrant<R>(m, emit, sx, sy, q, lds);
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
