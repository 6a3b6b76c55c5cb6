// vhp_geom.hpp -- what the lane-vector sweep kernels (vhp_pool.hpp, vhp_lat.hpp) share: the bit-packed occupancy maps and the
// geometry of a quadrant of computeVisibility() (reference src/visibilityBasedSolver.cpp:570-696): a quadrant is an x-major
// octant whose fronts are columns and a y-major octant whose fronts are rows; strips of 64 rows / 128 columns; blocks of 64
// steps (one word of the packed maps).  Compiled for gfx950 and, with -DVHP_SIM, for the CPU simulator of tests/sim.
// (Until round 4 this lived in the streaming sweep's header; that kernel was superseded by the pool sweep and is gone.)
#pragma once
#include "vhp_lanes.hpp"

namespace vhp {
namespace geom {

using namespace vhp::lanes;

#ifdef VHP_SIM
#define VHP_FN inline
#define VHP_HD inline
#else
#define VHP_FN __device__ __forceinline__
#define VHP_HD __host__ __device__ __forceinline__
#endif

constexpr int kBlock = 64;       // steps per block: one word of the packed occupancy maps
constexpr int kXRows = 64;       // rows per x-major strip (one per lane)
constexpr int kYCols = 128;      // columns per y-major strip (two per lane)

struct Map {
  const uint64_t* rows;  // bit x&63 of rows[y*wpr + 1 + (x>>6)] = occ(x,y)
  const uint64_t* cols;  // bit y&63 of cols[x*wpc + 1 + (y>>6)] = occ(x,y)
  const double* recip;   // recip[k] = RN(1/k), recip[0] = 0, readable up to max(nx,ny)+8
  int wpr, wpc, nx, ny;
};

VHP_HD int imin(int a, int b) { return a < b ? a : b; }
VHP_HD int imax(int a, int b) { return a > b ? a : b; }

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

}  // namespace geom
}  // namespace vhp
