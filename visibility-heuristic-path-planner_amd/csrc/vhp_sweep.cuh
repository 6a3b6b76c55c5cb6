// vhp_sweep.cuh -- CDNA4 (gfx950) device code of the visibility-transport sweep.
//
// Replaces the four loop nests of computeVisibility()
// (reference src/visibilityBasedSolver.cpp:570-696) for a batch of sources.
//
// Decomposition (tests/schedule_model.py is the executable statement of it and is
// checked bit-for-bit against the oracle):
//   source -> 4 quadrants -> 2 octants each = 8 work units (+1 zero-fill unit),
//   one workgroup per unit.  Inside an octant a cell depends only on the previous
//   "front" (the previous column for the x-major octant, the previous row for the
//   y-major one), on itself and on its neighbour one lane below:
//       v = (a - c*(a - b)) * occ,   a = own previous, b = lower neighbour's previous.
//   So a front is swept by the lanes of a wavefront: lane <-> row (x-major) or column
//   (y-major), the neighbour exchange is one DPP wave shift, nothing but registers
//   sits on the dependent chain.  A workgroup is W wavefronts; wavefront p owns strip
//   p (64*R consecutive lanes-worth of rows/columns) and runs one pipeline slot
//   (kChunk steps) behind wavefront p-1, which hands it the boundary lane through a
//   small LDS ring -- the LDS-staged active front.
//
//   x-major octant: a lane produces consecutive x of ONE row, i.e. a wavefront
//   produces a column per step.  Values are staged in a wave-private LDS tile and
//   flushed as 64-byte row segments (coalesced, sector aligned).
//   y-major octant: a wavefront produces 64 consecutive x of one row per step and
//   stores them directly (512 contiguous bytes per instruction).
//
//   The reference's stale diagonal (cell (k,k) = cell (k,k-1) * occ) is
//   produced by the x-major unit (the lane below hands its NEW value up); the
//   y-major unit needs diag(k) as the seed of lane k and recomputes it from the
//   private two-term recurrence sub(k) = V(k,k-1), diag(k) = sub(k)*occ(k,k).
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
constexpr int kTileStride = 9;   // doubles per staged row (odd: conflict-free column writes)
constexpr int kUnitsPerSource = 9;

struct DevMap {
  const uint64_t* rows;  // bit x&63 of rows[y*wpr + 1 + (x>>6)] = occ(x,y); word 0 and the last word of a row are zero pads
  const uint64_t* cols;  // bit y&63 of cols[x*wpc + 1 + (y>>6)] = occ(x,y)
  const double* recip;   // recip[k] = RN(1/k), k = 1..max(nx,ny)
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

__device__ __forceinline__ double lane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
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

// v * occ for occ in {0,1} and finite v >= 0 (solver.cpp:602)
__device__ __forceinline__ double gate(double v, uint64_t word, int bit) {
  return ((word >> bit) & 1ull) ? v : 0.0;
}

template <typename OutT>
__device__ __forceinline__ OutT to_out(double v) { return static_cast<OutT>(v); }

struct UnitGeom {
  int sx, sy, dirx, diry, ni, nj;
};

// ---------------------------------------------------------------------------
// x-major octant: cells (i, j), i > j, plus the diagonal (k, k) and the origin.
// ---------------------------------------------------------------------------
template <int R, typename OutT>
__device__ void x_unit(const DevMap& m, OutT* __restrict__ out, const UnitGeom g, double* lds) {
  constexpr int S = 64 * R;
  const int lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wavefront index, made provably uniform
  const int W = blockDim.x >> 6;
  const int rows_total = min(g.nj, g.ni);
  const int P = (rows_total + S - 1) / S;
  const int glast = (g.ni - 1) / kChunk;
  const int tmax = glast + P - 1;
  const int j0 = p * S;
  const bool strip_on = p < P;

  double* ring = lds;                                    // W rings
  double* tile = lds + (size_t)W * (kRing + 2) + (size_t)p * S * kTileStride;
  double* ring_out = ring + p * kRing;
  const double* ring_in = ring + (p > 0 ? p - 1 : 0) * kRing;

  double prev[R], jd[R];
  uint32_t dmask[R];
  uint64_t ow[R], own[R];
  const uint64_t* orow[R];
  int cur_xw = 0;
  if (strip_on) {
    const int x0 = g.sx + g.dirx * j0;  // x of the strip's first step
    cur_xw = x0 >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = j0 + 64 * r + lane;
      const bool on = j < rows_total;
      const int y = on ? g.sy + g.diry * j : g.sy;
      const int xd = on ? g.sx + g.dirx * j : g.sx;  // x of this row's diagonal cell
      prev[r] = 0.0;
      jd[r] = (double)j;
      orow[r] = m.rows + (size_t)y * m.wpr + 1;
      const uint64_t dw = orow[r][xd >> 6];
      dmask[r] = ((dw >> (xd & 63)) & 1ull) ? 0xffffffffu : 0u;
      ow[r] = orow[r][cur_xw];
      own[r] = orow[r][cur_xw + g.dirx];
    }
  }

  for (int T = 0; T <= tmax; ++T) {
    const int gch = T - p;
    if (strip_on && gch >= j0 / kChunk && gch <= glast) {
      const int ibeg = max(gch * kChunk, j0);
      const int iend = min(gch * kChunk + kChunk - 1, g.ni - 1);
      for (int i = ibeg; i <= iend; ++i) {
        const int x = g.sx + g.dirx * i;
        const int xw = x >> 6, xb = x & 63;
        if (xw != cur_xw) {
          cur_xw = xw;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            ow[r] = own[r];
            own[r] = orow[r][xw + g.dirx];
          }
        }
        const double di = (double)i;
        const double ri = m.recip[i];
        double fill = 0.0;   // OLD value of the row just below this register row's lane 0
        double dsrc = 1.0;   // NEW value of that row (source of the diagonal cell); 1.0 = light strength at the origin
        if (p > 0) {
          fill = ring_in[(i - 1) & (kRing - 1)];
          dsrc = ring_in[i & (kRing - 1)];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int jr0 = j0 + 64 * r;
          if (i >= jr0) {
            const double a = prev[r];
            const double b = shift_up(a, fill);
            const double c = ratio(jd[r], di, ri);
            double v = gate(stencil(a, b, c), ow[r], xb);
            if (i < jr0 + 64) {
              // the diagonal cell (i,i) lives in this register row: it inherits the NEW
              // value of the row below it times its own occupancy
              const double below = shift_up(v, dsrc);
              if (jr0 + lane == i) {
                v = __hiloint2double(__double2hiint(below) & (int)dmask[r], __double2loint(below) & (int)dmask[r]);
              }
            }
            fill = lane63(a);
            dsrc = lane63(v);
            prev[r] = v;
            tile[(64 * r + lane) * kTileStride + (x & (kTileCols - 1))] = v;
          }
        }
        if (lane == 63) ring_out[i & (kRing - 1)] = prev[R - 1];

        const bool endwin = g.dirx > 0 ? ((x & (kTileCols - 1)) == kTileCols - 1) : ((x & (kTileCols - 1)) == 0);
        if (endwin || i == g.ni - 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const int xbase = x & ~(kTileCols - 1);
          const int col = lane & (kTileCols - 1), rsub = lane >> 3;
          const int xc = xbase + col;
          const int ic = g.dirx > 0 ? xc - g.sx : g.sx - xc;
          const bool colok = ic >= 0 && ic <= i;
          const int rows_here = min(S, rows_total - j0);
          for (int rb = 0; rb < rows_here; rb += 8) {
            if (j0 + rb > i) break;
            const int rl = rb + rsub;
            const int j = j0 + rl;
            if (colok && rl < rows_here && j <= ic) {
              const int y = g.sy + g.diry * j;
              out[(size_t)y * m.nx + xc] = to_out<OutT>(tile[rl * kTileStride + col]);
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
// y-major octant: cells (i, j), j > i.
// ---------------------------------------------------------------------------
template <int R, typename OutT>
__device__ void y_unit(const DevMap& m, OutT* __restrict__ out, const UnitGeom g, double* lds) {
  constexpr int S = 64 * R;
  const int lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wavefront index, made provably uniform
  const int W = blockDim.x >> 6;
  const int cols_total = min(g.ni, g.nj - 1);
  if (cols_total <= 0) return;  // uniform for the workgroup
  const int P = (cols_total + S - 1) / S;
  const int glast = (g.nj - 1) / kChunk;
  const int tmax = glast + P - 1;
  const int i0 = p * S;
  const bool strip_on = p < P;

  double* ring = lds;
  double* dstate = lds + (size_t)W * kRing;  // 2 doubles per strip: (diag, sub) handed to the next strip
  double* ring_out = ring + p * kRing;
  const double* ring_in = ring + (p > 0 ? p - 1 : 0) * kRing;

  double prev[R], id[R];
  uint64_t ow[R], own[R], b1[R], b2[R];
  const uint64_t* ocol[R];
  int xs[R];
  int cur_yw = 0;
  // private diagonal recurrence state
  double dg = 0.0, sb = 0.0;
  if (strip_on) {
    const int y0 = g.sy + g.diry * i0;
    cur_yw = y0 >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + 64 * r + lane;
      const bool on = i < g.ni;
      const int x = on ? g.sx + g.dirx * i : g.sx;
      xs[r] = x;
      prev[r] = 0.0;
      id[r] = (double)i;
      ocol[r] = m.cols + (size_t)x * m.wpc + 1;
      ow[r] = ocol[r][cur_yw];
      own[r] = ocol[r][cur_yw + g.diry];
      // occupancy of (X(k), Y(k-1)) and (X(k), Y(k)) for k = this lane's index: the two
      // factors of the diagonal recurrence at step k, gathered once and balloted
      bool f1 = false, f2 = false;
      if (on && i >= 1 && i < g.nj) {
        const int yk = g.sy + g.diry * i, ykm = g.sy + g.diry * (i - 1);
        f1 = (ocol[r][ykm >> 6] >> (ykm & 63)) & 1ull;
        f2 = (ocol[r][yk >> 6] >> (yk & 63)) & 1ull;
      }
      b1[r] = __ballot(f1);
      b2[r] = __ballot(f2);
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
      for (int j = jbeg; j <= jend; ++j) {
        const int y = g.sy + g.diry * j;
        const int yw = y >> 6, yb = y & 63;
        if (yw != cur_yw) {
          cur_yw = yw;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            ow[r] = own[r];
            own[r] = ocol[r][yw + g.diry];
          }
        }
        const double dj = (double)j;
        const double rj = m.recip[j];
        // advance the private diagonal recurrence to diag(j) while j is one of this
        // strip's own lanes (the previous strip hands over the state at j = i0 - 1)
        if (j >= i0 && j < i0 + S && j < g.ni) {
          if (j >= 1) {
            if (j == i0) {  // p > 0 here
              dg = dstate[2 * (p - 1)];
              sb = dstate[2 * (p - 1) + 1];
            }
            const int k = j - i0;
            uint64_t m1 = b1[0], m2 = b2[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
              if ((k >> 6) == r) { m1 = b1[r]; m2 = b2[r]; }
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
        if (p > 0) fill = ring_in[(j - 1) & (kRing - 1)];
        OutT* orow = out + (size_t)y * m.nx;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ir0 = i0 + 64 * r;
          if (j >= ir0) {
            const int i = ir0 + lane;
            const double a = prev[r];
            const double b = shift_up(a, fill);
            const double c = ratio(id[r], dj, rj);
            double v = gate(stencil(a, b, c), ow[r], yb);
            if (i < j && i < cols_total) orow[xs[r]] = to_out<OutT>(v);
            if (i == j) v = dg;  // seed: the diagonal cell is this column's first "previous"
            fill = lane63(a);
            prev[r] = v;
          }
        }
        if (lane == 63) ring_out[j & (kRing - 1)] = prev[R - 1];
      }
    }
    __syncthreads();
  }
}

// rows/columns no quadrant covers read as zero
template <typename OutT>
__device__ void zero_unit(const DevMap& m, OutT* __restrict__ out, int sx, int sy) {
  if (sx > 0)
    for (int y = threadIdx.x; y < m.ny; y += blockDim.x) out[(size_t)y * m.nx] = OutT(0);
  if (sy > 0)
    for (int x = threadIdx.x; x < m.nx; x += blockDim.x) out[x] = OutT(0);
}

// grid = n_src * kUnitsPerSource workgroups of 64*W threads;
// dynamic LDS = sweep_lds_bytes(R, W)
template <int R, typename OutT>
__global__ void __launch_bounds__(1024)
vhp_sweep_fronts(DevMap m, const int32_t* __restrict__ src_xy, OutT* __restrict__ out, long long field_stride,
                 int* __restrict__ err_flag) {
  extern __shared__ double lds[];
  const int s = blockIdx.x / kUnitsPerSource;
  const int unit = blockIdx.x - s * kUnitsPerSource;
  const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
  if (sx < 0 || sy < 0 || sx >= m.nx || sy >= m.ny) {
    if (threadIdx.x == 0 && unit == 0) atomicOr(err_flag, 1);
    return;
  }
  OutT* o = out + (size_t)s * field_stride;
  if (unit == 8) {
    zero_unit<OutT>(m, o, sx, sy);
    return;
  }
  const int q = unit >> 1;
  UnitGeom g;
  g.sx = sx;
  g.sy = sy;
  g.dirx = (q == 0 || q == 3) ? 1 : -1;  // Q1 (+,+) Q2 (-,+) Q3 (-,-) Q4 (+,-), solver.cpp:575-695
  g.diry = (q < 2) ? 1 : -1;
  g.ni = g.dirx > 0 ? m.nx - sx : sx;    // negative directions stop short of the border (Q2)
  g.nj = g.diry > 0 ? m.ny - sy : sy;
  if (g.ni <= 0 || g.nj <= 0) return;
  if (unit & 1)
    y_unit<R, OutT>(m, o, g, lds);
  else
    x_unit<R, OutT>(m, o, g, lds);
}

inline size_t sweep_lds_bytes(int R, int W) {
  return ((size_t)W * (kRing + 2) + (size_t)W * 64 * R * kTileStride) * sizeof(double);
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
