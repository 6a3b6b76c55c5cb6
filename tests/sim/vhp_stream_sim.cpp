// vhp_stream_sim.cpp -- CPU simulator of the streaming sweep kernel.  TEST INFRASTRUCTURE ONLY.
//
// Compiles csrc/vhp_stream.hpp -- the very source hipcc builds for gfx950 -- with -DVHP_SIM, where a wavefront's
// lane vector is an array of 64 values (csrc/vhp_lanes.hpp), and runs one workgroup at a time.  The workgroup is a
// dataflow machine: a wavefront may sweep its next unit (one 64-step block of one strip) whenever its ready() says so.
// The simulator interleaves the wavefronts unit by unit under several policies -- round robin forward / backward,
// shuffled, and "greedy" (one wavefront runs for as long as it is ready: maximally ahead of everyone else, which is what
// stresses the ring-capacity and buffer-reuse conditions).  Every interleaving must give the same bytes; a result that
// depends on the policy is a missing dependency in ready().  A state in which no wavefront is ready is a deadlock and
// is reported.  LDS starts poisoned (NaN), so a read of a value that was never produced shows up in the field.
//
// Only tests/ loads this library (tests/sim_lib.py).  It is not a CPU fallback of the product: libvhp_hip.so neither
// links nor loads it, and it is three orders of magnitude slower than the oracle.
#define VHP_SIM
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../visibility-heuristic-path-planner_amd/csrc/vhp_stream.hpp"

using namespace vhp::stream;

namespace {

struct HostMap {
  std::vector<uint64_t> rows, cols;
  std::vector<double> recip;
  Map m;
};

// the packed maps and the reciprocal table, as vhp_set_map builds them (vhp_capi.hip finish_set_map)
void build_map(const uint8_t* occ, int nx, int ny, HostMap& h) {
  const int wpr = (nx + 63) / 64 + 2, wpc = (ny + 63) / 64 + 2;
  h.rows.assign((size_t)ny * wpr, 0);
  h.cols.assign((size_t)nx * wpc, 0);
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x)
      if (occ[(size_t)y * nx + x]) {
        h.rows[(size_t)y * wpr + 1 + (x >> 6)] |= 1ull << (x & 63);
        h.cols[(size_t)x * wpc + 1 + (y >> 6)] |= 1ull << (y & 63);
      }
  const int nrec = (nx > ny ? nx : ny) + 1 + 8;
  h.recip.resize(nrec);
  h.recip[0] = 0.0;
  for (int k = 1; k < nrec; ++k) {
    volatile double d = (double)k;
    h.recip[k] = 1.0 / d;
  }
  h.m.rows = h.rows.data();
  h.m.cols = h.cols.data();
  h.m.recip = h.recip.data();
  h.m.wpr = wpr;
  h.m.wpc = wpc;
  h.m.nx = nx;
  h.m.ny = ny;
}

struct SimInfo {
  long long slots = 0, max_slots = 0, lag_violations = 0;
};

uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// one workgroup sweeping one octant: `n_waves` schedulable wavefronts given by the three callbacks
template <class Active, class Ready, class Run>
void interleave(int n_waves, int order_mode, SimInfo& info, uint32_t& rng, Active active, Ready ready, Run run) {
  std::vector<int> order(n_waves);
  long long passes = 0;
  for (;;) {
    bool any_active = false, progress = false;
    for (int k = 0; k < n_waves; ++k) order[k] = k;
    if (order_mode == 1) for (int k = 0; k < n_waves; ++k) order[k] = n_waves - 1 - k;
    if (order_mode >= 2) for (int k = n_waves - 1; k > 0; --k) { const int r = lcg(rng) % (k + 1); std::swap(order[k], order[r]); }
    for (int k = 0; k < n_waves; ++k) {
      const int wv = order[k];
      if (!active(wv)) continue;
      any_active = true;
      // policies 0-2: one unit per visit; 3 (greedy): as many as it can; 4: a random few
      int budget = order_mode == 3 ? (1 << 30) : order_mode == 4 ? 1 + (int)(lcg(rng) % 5) : 1;
      while (budget-- > 0 && active(wv) && ready(wv)) {
        run(wv);
        progress = true;
      }
    }
    ++passes;
    if (!any_active) break;
    if (!progress) { info.lag_violations += 1000000; break; }  // deadlock
  }
  info.slots += passes;
  if (passes > info.max_slots) info.max_slots = passes;
}

// what the launcher would pick (vhp_stream.hip): y-major units sweep with every wavefront but the DiagWave; x-major
// units pair each sweeping wavefront with a flusher
struct SimShape {
  int Wx, Wy, tile_slots;
  bool lazy_flush;
};

// The two units of a quadrant, each a workgroup of its own with its own (poisoned) LDS: they share nothing.
template <int DX, int DY, typename OutT>
void run_quadrant(const HostMap& h, OutT* field, int sx, int sy, const SimShape& sh, int order_mode, SimInfo& info, uint32_t& rng) {
  Quad<DX, DY> g;
  g.init(h.m.nx, h.m.ny, sx, sy);
  if (g.empty()) return;
  const int lds_n = lds_doubles(sh.Wx, sh.Wy, h.m.nx, h.m.ny, sh.tile_slots);
  {  // x-major unit: Wx sweeping wavefronts, each with its flusher (run inside the sweeping wavefront's post / wait calls)
    const int W = sh.Wx;
    const Layout L = make_layout(W, h.m.nx, h.m.ny, true, sh.tile_slots);
    std::vector<double> lds(lds_n, std::numeric_limits<double>::quiet_NaN());
    Progress<DX, DY> prog;
    prog.bind(lds.data(), L, W);
    prog.clear(true, L, 0, 1);
    prog.setup(g, true);
    std::vector<XWave<DX, DY, OutT>> xs(W), fl(W);
    for (int w = 0; w < W; ++w) {
      fl[w].init_flusher(h.m, g, field, w, W, lds.data(), L);
      xs[w].flusher = &fl[w];
      xs[w].lazy = sh.lazy_flush;
      xs[w].init(h.m, g, field, w, W, lds.data(), L);
    }
    interleave(W, order_mode, info, rng, [&](int w) { return xs[w].active; }, [&](int w) { return xs[w].ready(); }, [&](int w) { xs[w].run_unit(); });
    for (int w = 0; w < W; ++w) xs[w].finish();
  }
  if (g.Py > 0) {  // y-major unit: Wy sweeping wavefronts and the DiagWave
    const int W = sh.Wy;
    const Layout L = make_layout(W, h.m.nx, h.m.ny, false, sh.tile_slots);
    std::vector<double> lds(lds_n, std::numeric_limits<double>::quiet_NaN());
    Progress<DX, DY> prog;
    prog.bind(lds.data(), L, W);
    prog.clear(false, L, 0, 1);
    prog.setup(g, false);
    std::vector<YWave<DX, DY, OutT>> ys(W);
    for (int w = 0; w < W; ++w) ys[w].init(h.m, g, field, w, W, lds.data(), L);
    DiagWave<DX, DY> dw;
    dw.init(h.m, g, W, lds.data(), L);
    interleave(W + 1, order_mode, info, rng, [&](int w) { return w < W ? ys[w].active : dw.active; },
               [&](int w) { return w < W ? ys[w].ready() : dw.ready(); }, [&](int w) { if (w < W) ys[w].run_unit(); else dw.run_unit(); });
  }
}

template <typename OutT>
int run_batch(const uint8_t* occ, int nx, int ny, const int32_t* src, int n_src, OutT* out, const SimShape& sh, int order_mode, long long* stats) {
  HostMap h;
  build_map(occ, nx, ny, h);
  SimInfo info;
  uint32_t rng = 12345u;
  vhp::lanes::store_stats() = vhp::lanes::StoreStats();
  for (int s = 0; s < n_src; ++s) {
    const int sx = src[2 * s], sy = src[2 * s + 1];
    if (sx < 0 || sy < 0 || sx >= nx || sy >= ny) return 2;
    OutT* field = out + (size_t)s * nx * ny;
    // rows / columns no quadrant covers read as zero: the workgroup of quadrant 1 stores them
    if (sx > 0) for (int y = 0; y < ny; ++y) field[(size_t)y * nx] = OutT(0);
    if (sy > 0) for (int x = 0; x < nx; ++x) field[x] = OutT(0);
    run_quadrant<+1, +1>(h, field, sx, sy, sh, order_mode, info, rng);
    run_quadrant<-1, +1>(h, field, sx, sy, sh, order_mode, info, rng);
    run_quadrant<-1, -1>(h, field, sx, sy, sh, order_mode, info, rng);
    run_quadrant<+1, -1>(h, field, sx, sy, sh, order_mode, info, rng);
  }
  if (stats) {
    stats[0] = info.slots;
    stats[1] = info.max_slots;
    stats[2] = info.lag_violations;
    stats[3] = vhp::lanes::store_stats().n16;
    stats[4] = vhp::lanes::store_stats().n8;
  }
  return 0;
}

}  // namespace

extern "C" {

// out: n_src fields of nx*ny elements (dtype 0 = double, 1 = float), pre-filled by the caller (e.g. with NaN, to prove
// that every cell is written).  W: sweeping wavefronts of an x-major unit (>= 3); a y-major unit sweeps with 2W - 1 (the
// launcher's shape: 2W wavefronts per workgroup).  order_mode & 7: 0 round robin forward, 1 backward, 2 shuffled,
// 3 greedy (a wavefront runs while it is ready), 4 shuffled with random bursts;  order_mode & 8: the flushers run as late
// as the hand-off allows instead of right at the post;  order_mode & 16: two tile slots (no slack) instead of three;
// order_mode & 32: y-major units swept by W - 1 wavefronts (the team size when two of them share a workgroup).
// stats (5 entries, may be null): scheduler passes in total, most passes of one workgroup, violations (deadlocks),
// 16-byte / 8-byte store instructions.
int vhp_sim_stream_sweep(const uint8_t* occ, int nx, int ny, const int32_t* src, int n_src, int dtype, void* out, int W,
                         int order_mode, long long* stats) {
  if (!occ || !src || !out || nx <= 0 || ny <= 0 || (nx & 7) != 0 || W < 3 || W > 8) return 1;
  SimShape sh;
  sh.Wx = W;
  sh.Wy = (order_mode & 32) ? W - 1 : 2 * W - 1;
  sh.tile_slots = (order_mode >> 8) & 15 ? (order_mode >> 8) & 15 : (order_mode & 16) ? 2 : 3;  // bits 8-11: an explicit slot count (4, 6, 8)
  sh.lazy_flush = (order_mode & 8) != 0;
  if (dtype == 0) return run_batch<double>(occ, nx, ny, src, n_src, static_cast<double*>(out), sh, order_mode & 7, stats);
  return run_batch<float>(occ, nx, ny, src, n_src, static_cast<float*>(out), sh, order_mode & 7, stats);
}

int vhp_sim_lds_bytes(int nx, int ny, int W, int tile_slots) { return lds_doubles(W, 2 * W - 1, nx, ny, tile_slots) * 8; }

}  // extern "C"
