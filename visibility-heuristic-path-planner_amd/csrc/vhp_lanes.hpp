// vhp_lanes.hpp -- the wavefront as a value type.
//
// The pool sweep and the latency sweep (vhp_pool.hpp, vhp_lat.hpp) are written once against the few operations below and compiled twice:
//
//   * by hipcc for gfx950 (the product): a "lane vector" is the plain per-thread scalar of the SIMT model, every
//     operation is the instruction it names (DPP wave shifts, v_readlane, v_bfe_i32, ds_read/ds_write, global
//     stores with a 32-bit byte offset on a scalar base) and all of it inlines away;
//   * by g++ with -DVHP_SIM (tests/sim only): a lane vector is an array of 64 values and every operation is a loop
//     over lanes, so that the kernel's schedule, index arithmetic, predicates and data flow can be checked against
//     the oracle on a machine without a GPU, cell for cell, over thousands of geometries.  That build is TEST
//     INFRASTRUCTURE: nothing in the product loads it, and it is not a CPU fallback (it runs one workgroup at a
//     time, wavefront by wavefront, and is ~1000x slower than the oracle itself).
//
// The discipline this buys: control flow in the kernel is wavefront-uniform by construction -- an `if` on a lane
// vector does not compile in the simulator build -- and everything lane-dependent is a select or a predicated store.
#pragma once
#include <stdint.h>

#include "vhp_diag.h"

#ifdef VHP_SIM
#include <cmath>
#include <cstdint>
#include <cstring>
#define VHP_LANE_FN inline
#else
#include <hip/hip_runtime.h>
#define VHP_LANE_FN __device__ __forceinline__
#endif

namespace vhp {
namespace lanes {

constexpr int kLanes = 64;

#ifdef VHP_SIM
// ------------------------------------------------------------------------------------------------------------
// simulator build: 64 explicit lanes
// ------------------------------------------------------------------------------------------------------------
template <typename T>
struct V {
  T v[kLanes];
  V() = default;
  V(T s) {  // broadcast (implicit on purpose: uniform values mix freely with lane vectors)
    for (int l = 0; l < kLanes; ++l) v[l] = s;
  }
};
using vd = V<double>;
using vi = V<int>;
using vu32 = V<uint32_t>;
using vu64 = V<uint64_t>;
using vb = V<bool>;

#define VHP_SIM_BINOP(OP, RT)                                                                        \
  template <typename T> inline V<RT> operator OP(const V<T>& a, const V<T>& b) {                       \
    V<RT> r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] OP b.v[l]; return r; }                   \
  template <typename T> inline V<RT> operator OP(const V<T>& a, T b) {                                 \
    V<RT> r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] OP b; return r; }                        \
  template <typename T> inline V<RT> operator OP(T a, const V<T>& b) {                                 \
    V<RT> r; for (int l = 0; l < kLanes; ++l) r.v[l] = a OP b.v[l]; return r; }
#define VHP_SIM_ARITH(OP) VHP_SIM_BINOP(OP, T)
#define VHP_SIM_CMP(OP) VHP_SIM_BINOP(OP, bool)
VHP_SIM_ARITH(+) VHP_SIM_ARITH(-) VHP_SIM_ARITH(*) VHP_SIM_ARITH(&) VHP_SIM_ARITH(|) VHP_SIM_ARITH(^)
VHP_SIM_ARITH(>>) VHP_SIM_ARITH(<<)
VHP_SIM_CMP(==) VHP_SIM_CMP(!=) VHP_SIM_CMP(<) VHP_SIM_CMP(<=) VHP_SIM_CMP(>) VHP_SIM_CMP(>=)
#undef VHP_SIM_ARITH
#undef VHP_SIM_CMP
#undef VHP_SIM_BINOP
inline vb operator&&(const vb& a, const vb& b) { vb r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] && b.v[l]; return r; }
inline vb operator&&(const vb& a, bool b) { vb r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] && b; return r; }
inline vb operator&&(bool a, const vb& b) { return b && a; }
inline vb operator||(const vb& a, const vb& b) { vb r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] || b.v[l]; return r; }
inline vb operator!(const vb& a) { vb r; for (int l = 0; l < kLanes; ++l) r.v[l] = !a.v[l]; return r; }
template <typename T> inline V<T> operator-(const V<T>& a) { V<T> r; for (int l = 0; l < kLanes; ++l) r.v[l] = -a.v[l]; return r; }

inline vi lane_id() { vi r; for (int l = 0; l < kLanes; ++l) r.v[l] = l; return r; }
inline vd to_f64(const vi& a) { vd r; for (int l = 0; l < kLanes; ++l) r.v[l] = (double)a.v[l]; return r; }
inline vu32 to_u32(const vi& a) { vu32 r; for (int l = 0; l < kLanes; ++l) r.v[l] = (uint32_t)a.v[l]; return r; }
template <typename T> inline V<T> select(const vb& c, const V<T>& a, const V<T>& b) {
  V<T> r; for (int l = 0; l < kLanes; ++l) r.v[l] = c.v[l] ? a.v[l] : b.v[l]; return r; }
inline vi vmin(const vi& a, int b) { vi r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] < b ? a.v[l] : b; return r; }
inline vi vmax(const vi& a, int b) { vi r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] > b ? a.v[l] : b; return r; }
inline vd vfma(const vd& a, const vd& b, const vd& c) { vd r; for (int l = 0; l < kLanes; ++l) r.v[l] = std::fma(a.v[l], b.v[l], c.v[l]); return r; }
// max(a, b) of a lane vector and a uniform double (neither is ever NaN where this is used)
inline vd vmaxd(const vd& a, double b) { vd r; for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] > b ? a.v[l] : b; return r; }

// lane l <- lane l-1; lane 0 <- fill's lane 0
inline vd shift_up(const vd& v, const vd& fill) { vd r; r.v[0] = fill.v[0]; for (int l = 1; l < kLanes; ++l) r.v[l] = v.v[l - 1]; return r; }
// the same, written over `fill` (whose other lanes are dead after it): the device build shifts into fill's own registers
inline vd shift_up_into(vd& fill, const vd& v) { fill = shift_up(v, fill); return fill; }
// the value is +0.0, bit for bit
inline vb is_pos_zero(const vd& v) { vb r; for (int l = 0; l < kLanes; ++l) { uint64_t u; std::memcpy(&u, &v.v[l], 8); r.v[l] = u == 0; } return r; }
// lane l <- lane l+1; lane 63 <- fill's lane 63
inline vd shift_down(const vd& v, const vd& fill) { vd r; r.v[kLanes - 1] = fill.v[kLanes - 1]; for (int l = 0; l + 1 < kLanes; ++l) r.v[l] = v.v[l + 1]; return r; }
// lane l <- lane l+1; lane 63 <- lane 0
inline vd rotate_down(const vd& v) { vd r; for (int l = 0; l < kLanes; ++l) r.v[l] = v.v[(l + 1) & 63]; return r; }
inline double read_lane(const vd& v, int l) { return v.v[l & 63]; }
// v * occ for occ in {0,1}: AND with 0 / ~0
inline vd and_mask(const vd& v, const vi& m) {
  vd r;
  for (int l = 0; l < kLanes; ++l) { uint64_t u; std::memcpy(&u, &v.v[l], 8); u &= m.v[l] ? ~0ull : 0ull; std::memcpy(&r.v[l], &u, 8); }
  return r;
}
// bit b (uniform, 0..63) of a lane-private word as 0 / ~0
inline vi bit_mask(const vu64& w, int b) { vi r; for (int l = 0; l < kLanes; ++l) r.v[l] = ((w.v[l] >> (b & 63)) & 1ull) ? -1 : 0; return r; }
// bit b (per lane, 0..63) of a lane-private word as 0 / ~0
inline vi bit_mask_lane(const vu64& w, const vi& b) { vi r; for (int l = 0; l < kLanes; ++l) r.v[l] = ((w.v[l] >> (b.v[l] & 63)) & 1ull) ? -1 : 0; return r; }
inline int read_lane_i(const vi& v, int l) { return v.v[l & 63]; }
inline uint64_t read_lane_u64(const vu64& v, int l) { return v.v[l & 63]; }
// the 32-bit half of w that holds bit t (uniform), shifted right by sh (uniform, 0..31)
inline vu32 half_shifted(const vu64& w, int t, int sh) {
  vu32 r; for (int l = 0; l < kLanes; ++l) r.v[l] = (uint32_t)((t & 32) ? (w.v[l] >> 32) : w.v[l]) >> sh; return r; }
// bit `b` (compile time in the kernel) of hs as 0 / ~0: v_bfe_i32
inline vi sbfe1(const vu32& hs, int b) { vi r; for (int l = 0; l < kLanes; ++l) r.v[l] = ((hs.v[l] >> b) & 1u) ? -1 : 0; return r; }

inline vd lds_load(const double* base, const vi& idx) { vd r; for (int l = 0; l < kLanes; ++l) r.v[l] = base[idx.v[l]]; return r; }
// every lane reads the same (uniform) entry
inline vd lds_bcast(const double* base, int idx) { return vd(base[idx]); }
inline void lds_store(double* base, const vi& idx, const vd& v) { for (int l = 0; l < kLanes; ++l) base[idx.v[l]] = v.v[l]; }
inline void lds_store_if(const vb& p, double* base, const vi& idx, const vd& v) { for (int l = 0; l < kLanes; ++l) if (p.v[l]) base[idx.v[l]] = v.v[l]; }
inline vu64 g_load_u64(const uint64_t* base, const vi& idx) { vu64 r; for (int l = 0; l < kLanes; ++l) r.v[l] = base[idx.v[l]]; return r; }
inline vd g_load_f64(const double* base, const vi& idx) { vd r; for (int l = 0; l < kLanes; ++l) r.v[l] = base[idx.v[l]]; return r; }
// two adjacent doubles at an even index (one 16-byte load per lane)
inline void g_load2_f64(const double* base, const vi& idx, vd& a, vd& b) { for (int l = 0; l < kLanes; ++l) { a.v[l] = base[idx.v[l]]; b.v[l] = base[idx.v[l] + 1]; } }

// The simulator records every global store (how many, how wide, which 64-byte sectors) for the tests' coverage
// and store-shape checks.
// ... and, for stores into the registered output [base, base + bytes), how many 128-byte lines an instruction wrote whole and
// how many only in part (a line that leaves in pieces costs the memory several whole ones: DESIGN.md appendix A.7).
struct StoreStats {
  long long n16 = 0, n8 = 0, n4 = 0, lines_whole = 0, lines_part = 0, misaligned = 0;  // misaligned: pairs off their own size's grid (must stay 0)
  const char* base = nullptr;
  size_t bytes = 0;
  // one store instruction: byte ranges [p, p + n) of its active lanes
  long long tmp_line[2 * kLanes];
  int tmp_bytes[2 * kLanes];
  int tmp_n = 0;
  void add(const void* p, int n) {
    const char* c = static_cast<const char*>(p);
    if (!base || c < base || c >= base + bytes) return;
    const long long off = c - base;
    for (long long o = off; o < off + n;) {  // (a lane's piece may straddle a line only if it is misaligned; split it)
      const long long line = o >> 7;
      const int take = (int)((((line + 1) << 7) < off + n ? ((line + 1) << 7) : off + n) - o);
      int k = 0;
      while (k < tmp_n && tmp_line[k] != line) ++k;
      if (k == tmp_n) { if (tmp_n == 2 * kLanes) return; tmp_line[k] = line; tmp_bytes[k] = 0; ++tmp_n; }
      tmp_bytes[k] += take;
      o += take;
    }
  }
  void end_instruction() {
    for (int k = 0; k < tmp_n; ++k) (tmp_bytes[k] >= 128 ? lines_whole : lines_part) += 1;
    tmp_n = 0;
  }
};
inline StoreStats& store_stats() { static StoreStats s; return s; }

// two adjacent cells at byte offset off (16-byte aligned for double, 8 for float)
template <typename OutT> inline void g_store2(OutT* base, const vu32& off, const vd& a, const vd& b) {
  for (int l = 0; l < kLanes; ++l) {
    OutT* p = reinterpret_cast<OutT*>(reinterpret_cast<char*>(base) + off.v[l]);
    p[0] = (OutT)a.v[l]; p[1] = (OutT)b.v[l];
    store_stats().add(p, 2 * (int)sizeof(OutT));
    store_stats().misaligned += (reinterpret_cast<uintptr_t>(p) & (2 * sizeof(OutT) - 1)) != 0;
  }
  store_stats().n16 += 1;
  store_stats().end_instruction();
}
template <typename OutT> inline void g_store2_if(const vb& p2, const vb& p_lo, const vb& p_hi, OutT* base, const vu32& off, const vd& a, const vd& b) {
  // p2: both cells; else p_lo: only the first; else p_hi: only the second
  for (int l = 0; l < kLanes; ++l) {
    OutT* p = reinterpret_cast<OutT*>(reinterpret_cast<char*>(base) + off.v[l]);
    if (p2.v[l]) { p[0] = (OutT)a.v[l]; p[1] = (OutT)b.v[l]; store_stats().add(p, 2 * (int)sizeof(OutT)); store_stats().misaligned += (reinterpret_cast<uintptr_t>(p) & (2 * sizeof(OutT) - 1)) != 0; }
    else if (p_lo.v[l]) { p[0] = (OutT)a.v[l]; store_stats().add(p, (int)sizeof(OutT)); }
    else if (p_hi.v[l]) { p[1] = (OutT)b.v[l]; store_stats().add(p + 1, (int)sizeof(OutT)); }
  }
  store_stats().n16 += 1;
  store_stats().end_instruction();
}
// (the device build's stores without the nt bit: the same thing here)
template <typename OutT> inline void g_store2_plain(OutT* base, const vu32& off, const vd& a, const vd& b) { g_store2(base, off, a, b); }
template <typename OutT> inline void g_store2_if_plain(const vb& p2, const vb& p_lo, const vb& p_hi, OutT* base, const vu32& off, const vd& a, const vd& b) { g_store2_if(p2, p_lo, p_hi, base, off, a, b); }
// is the pair at byte offset off aligned to its own size (2 cells)?  On a grid of odd width every other row is not.
template <typename OutT> inline vb pair_aligned(const OutT* base, const vu32& off) {
  vb r; for (int l = 0; l < kLanes; ++l) r.v[l] = ((reinterpret_cast<uintptr_t>(base) + off.v[l]) & (2 * sizeof(OutT) - 1)) == 0; return r; }
// the pair, from the lanes of p only (nothing of a pair is ever split)
template <typename OutT> inline void g_store2_mask(const vb& p, OutT* base, const vu32& off, const vd& a, const vd& b) {
  g_store2_if(p, vb(false), vb(false), base, off, a, b);
}
template <typename OutT> inline void g_store1_if(const vb& p1, OutT* base, const vu32& off, const vd& a) {
  for (int l = 0; l < kLanes; ++l) if (p1.v[l]) { OutT* p = reinterpret_cast<OutT*>(reinterpret_cast<char*>(base) + off.v[l]); *p = (OutT)a.v[l]; store_stats().add(p, (int)sizeof(OutT)); }
  store_stats().n8 += 1;
  store_stats().end_instruction();
}
inline void wave_sync() {}
inline void sched_fence() {}
template <typename T> inline void pin(T&) {}
inline int uniform(int x) { return x; }
inline int lane0_int(const vd& a) { return (int)a.v[0]; }   // a value that is the same in every lane, as an integer
// release: everything this wavefront has written to LDS becomes visible, then the progress word
inline void lds_publish(volatile int* word, int value) { *word = value; }
inline int lds_poll(const volatile int* word) { return *word; }
inline int lds_peek(const volatile int* word) { return *word; }
inline int lds_int_at(const int* p) { return *p; }
inline void lds_set_int(int* p, int v) { *p = v; }
inline void lds_acquire() {}
inline void lds_post4(int* d, int a, int b, int c, int e) { d[0] = a; d[1] = b; d[2] = c; d[3] = e; }
inline void lds_read4(const int* d, int& a, int& b, int& c, int& e) { a = d[0]; b = d[1]; c = d[2]; e = d[3]; }
// The pool kernel (vhp_pool.hpp) is written as blocking code: a wavefront that has to wait loops over backoff().  The
// simulator runs every wavefront as a coroutine and installs a hook here that switches to its scheduler; sim_progress()
// tells the scheduler that the calling wavefront got something done (deadlock detection), sim_point() marks a place
// where a device wavefront can be overtaken by another (between a read and the compare-and-swap that depends on it).
struct SimHooks { void (*yield)() = nullptr; void (*progress)() = nullptr; void (*point)() = nullptr; };
inline SimHooks& sim_hooks() { static SimHooks h; return h; }
inline void backoff() { if (sim_hooks().yield) sim_hooks().yield(); }
inline void ready_backoff() { backoff(); }
inline void short_backoff() { backoff(); }
inline void taken_backoff() { backoff(); }
inline void wave_priority(int) {}
// event counters of the simulator (which path a hand-off took); the device build counts nothing
struct SimCounts { long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
inline SimCounts& sim_counts() { static SimCounts s; return s; }
inline void sim_count(int k) { sim_counts().c[k] += 1; }
inline void sim_progress() { if (sim_hooks().progress) sim_hooks().progress(); }
inline void sim_point() { if (sim_hooks().point) sim_hooks().point(); }
inline void stores_done() {}
// LDS atomics, executed once per wavefront (every lane sees the returned old value)
inline int lds_cas(int* p, int expected, int desired) { const int old = *p; if (old == expected) *p = desired; return old; }
inline int lds_add(int* p, int v) { const int old = *p; *p = old + v; return old; }
inline int lds_or(int* p, int v) { const int old = *p; *p = old | v; return old; }
inline int lds_and(int* p, int v) { const int old = *p; *p = old & v; return old; }
// global memory: one atomic add per wavefront; a lane vector stored / loaded as 64 consecutive doubles
inline int g_add(int* p, int v) { const int old = *p; *p = old + v; return old; }
inline unsigned long long g_add_u64(unsigned long long* p, unsigned long long v) { const unsigned long long old = *p; *p = old + v; return old; }
inline unsigned long long g_peek_u64(const unsigned long long* p) { return *p; }
// four consecutive ints at a uniform index (a 16-byte record that an earlier kernel wrote), as uniform values
inline void g_load_rec4(const int* base, int idx, int& a, int& b, int& c, int& d) { a = base[4 * idx]; b = base[4 * idx + 1]; c = base[4 * idx + 2]; d = base[4 * idx + 3]; }
inline void g_store_f64(double* base, const vi& idx, const vd& v) { for (int l = 0; l < kLanes; ++l) base[idx.v[l]] = v.v[l]; }
inline void g_store_f64_if(const vb& p, double* base, const vi& idx, const vd& v) { for (int l = 0; l < kLanes; ++l) if (p.v[l]) base[idx.v[l]] = v.v[l]; }
template <typename T> inline void g_store_scalar_if(const vb& p, T* base, const vi& idx, T v) {
  for (int l = 0; l < kLanes; ++l) if (p.v[l]) { base[idx.v[l]] = v; store_stats().add(&base[idx.v[l]], (int)sizeof(T)); }
  store_stats().end_instruction();
}
inline int ffs_u32(uint32_t v) { return v ? __builtin_ctz(v) : -1; }
// Tagged 16-byte entries in global memory ({value, tag}): how a strip hands its boundary line to the strip that reads it
// (vhp_pool.hpp).  The store writes both words at once; the load tells, per lane, whether the entry carries `tag`.
struct Tagged { double v; uint64_t tag; };
inline void g_store_tagged(Tagged* base, const vi& idx, const vd& v, uint64_t tag) { for (int l = 0; l < kLanes; ++l) { base[idx.v[l]].v = v.v[l]; base[idx.v[l]].tag = tag; } }
inline vb g_load_tagged(const Tagged* base, const vi& idx, uint64_t tag, vd& v) {
  vb ok; for (int l = 0; l < kLanes; ++l) { v.v[l] = base[idx.v[l]].v; ok.v[l] = base[idx.v[l]].tag == tag; } return ok; }
inline bool wave_all(const vb& p) { for (int l = 0; l < kLanes; ++l) if (!p.v[l]) return false; return true; }
inline void lds_store_i_if(const vb& p, int* base, const vi& idx, int v) { for (int l = 0; l < kLanes; ++l) if (p.v[l]) base[idx.v[l]] = v; }
// a tagged entry per lane as it is in memory, tag and value -- what g_load_tagged compares at once, for a caller that looks later
inline void g_load_tagged_raw(const Tagged* base, const vi& idx, vu64& tag, vd& v) { for (int l = 0; l < kLanes; ++l) { v.v[l] = base[idx.v[l]].v; tag.v[l] = base[idx.v[l]].tag; } }
inline vb tags_are(const vu64& tag, uint64_t want) { vb ok; for (int l = 0; l < kLanes; ++l) ok.v[l] = tag.v[l] == want; return ok; }
inline void g_store_tagged_if(const vb& p, Tagged* base, const vi& idx, const vd& v, uint64_t tag) { for (int l = 0; l < kLanes; ++l) if (p.v[l]) { base[idx.v[l]].v = v.v[l]; base[idx.v[l]].tag = tag; } }
inline void g_store_tagged_device(Tagged* base, const vi& idx, const vd& v, uint64_t tag) { g_store_tagged(base, idx, v, tag); }

#else
// ------------------------------------------------------------------------------------------------------------
// device build (gfx950): a lane vector is the per-thread scalar
// ------------------------------------------------------------------------------------------------------------
using vd = double;
using vi = int;
using vu32 = uint32_t;
using vu64 = uint64_t;
using vb = bool;

VHP_LANE_FN vi lane_id() { return (int)(threadIdx.x & 63u); }
VHP_LANE_FN vd to_f64(vi a) { return (double)a; }
VHP_LANE_FN vu32 to_u32(vi a) { return (uint32_t)a; }
template <typename T> VHP_LANE_FN T select(bool c, T a, T b) { return c ? a : b; }
VHP_LANE_FN vi vmin(vi a, int b) { return a < b ? a : b; }
VHP_LANE_FN vi vmax(vi a, int b) { return a > b ? a : b; }
VHP_LANE_FN vd vfma(vd a, vd b, vd c) { return __builtin_fma(a, b, c); }
VHP_LANE_FN vd vmaxd(vd a, double b) { return __builtin_fmax(a, b); }

// lane l <- lane l-1, lane 0 keeps `fill`'s lane 0.  DPP wave_shr:1 (gfx9 encoding 0x138); with bound_ctrl off the
// lane without a source keeps `old`.
VHP_LANE_FN vd shift_up(vd v, vd fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  const int flo = __double2loint(fill), fhi = __double2hiint(fill);
  lo = __builtin_amdgcn_update_dpp(flo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// The same, written over `fill` -- a value whose other lanes are dead after this (a boundary value read for this one step): the DPP
// move goes into fill's own registers.  (update_dpp makes the compiler copy `old` first -- two v_mov per step of every window --
// even where the copy's source dies right there; the asm's tied operand leaves it no choice.)
VHP_LANE_FN vd shift_up_into(vd& fill, vd v) {
  int flo = __double2loint(fill), fhi = __double2hiint(fill);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  // (s_nop 1: a DPP read needs two wait states after the vector instruction that wrote its source, and the compiler's hazard
  // recognizer does not look inside an asm; the two sources were written by adjacent instructions, so one s_nop covers both)
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(flo) : "v"(lo));
  asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(fhi) : "v"(hi), "v"(flo));
  fill = __hiloint2double(fhi, flo);
  return fill;
}
// the value is +0.0, bit for bit
VHP_LANE_FN vb is_pos_zero(vd v) { return __double_as_longlong(v) == 0; }
// lane l <- lane l+1, lane 63 keeps `fill`'s lane 63: DPP wave_shl:1 (0x130)
VHP_LANE_FN vd shift_down(vd v, vd fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  const int flo = __double2loint(fill), fhi = __double2hiint(fill);
  lo = __builtin_amdgcn_update_dpp(flo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// lane l <- lane l+1 (lane 63 <- lane 0): DPP wave_rol:1 (0x134).  Every lane has a source, so the destination needs
// no previous value: mov_dpp leaves `old` undefined and the compiler is free to write a fresh register (update_dpp ties
// the destination to `old` and costs two v_mov per use when the source is still live).
VHP_LANE_FN vd rotate_down(vd v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x134, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x134, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// wave-uniform read of lane l (l uniform): v_readlane_b32 x 2
VHP_LANE_FN double read_lane(vd v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
VHP_LANE_FN vd and_mask(vd v, vi m) { return __hiloint2double(__double2hiint(v) & m, __double2loint(v) & m); }
VHP_LANE_FN vi bit_mask(vu64 w, int b) {
  const uint32_t half = (b & 32) ? (uint32_t)(w >> 32) : (uint32_t)w;
  return __builtin_amdgcn_sbfe(half, b & 31, 1);
}
VHP_LANE_FN vi bit_mask_lane(vu64 w, vi b) { return ((w >> (b & 63)) & 1ull) ? -1 : 0; }
VHP_LANE_FN int read_lane_i(vi v, int l) { return __builtin_amdgcn_readlane(v, l); }
VHP_LANE_FN uint64_t read_lane_u64(vu64 v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
VHP_LANE_FN vu32 half_shifted(vu64 w, int t, int sh) { return ((t & 32) ? (uint32_t)(w >> 32) : (uint32_t)w) >> sh; }
VHP_LANE_FN vi sbfe1(vu32 hs, int b) { return __builtin_amdgcn_sbfe(hs, b, 1); }

VHP_LANE_FN vd lds_load(const double* base, vi idx) { return base[idx]; }
// every lane reads the same (uniform) entry: an LDS broadcast read, the value arrives in a VGPR
VHP_LANE_FN vd lds_bcast(const double* base, int idx) { return base[idx]; }
VHP_LANE_FN void lds_store(double* base, vi idx, vd v) { base[idx] = v; }
VHP_LANE_FN void lds_store_if(bool p, double* base, vi idx, vd v) { if (p) base[idx] = v; }
VHP_LANE_FN vu64 g_load_u64(const uint64_t* base, vi idx) { return base[idx]; }
VHP_LANE_FN vd g_load_f64(const double* base, vi idx) { return base[idx]; }
VHP_LANE_FN void g_load2_f64(const double* base, vi idx, vd& a, vd& b) { const double2 v = *reinterpret_cast<const double2*>(base + idx); a = v.x; b = v.y; }

// Field stores are NON-TEMPORAL (the nt bit of global_store).  A field is written once and never read by the launch, and --
// measured, DESIGN.md appendix A.7 -- the memory side charges a plain store that covers only part of a 128-byte line (the two ends
// of a row piece that is half a line off the line grid, the cells next to a diagonal or an axis) several whole lines' worth on
// two thirds of the device's memory; with nt a partially written line costs what it weighs (tools/policybench.hip: the same
// bytes, every other row half a line off: 3.65 -> 4.81 TB/s on a slow buffer, 5.27 -> 5.96 on a fast one).
#ifdef VHP_FIELD_STORE_PLAIN  // the latency sweep (vhp_lat.hip); -DVHP_DIAG_PLAINSTORE builds (vhp_diag.h): what a launch took until round 4
#define VHP_FIELD_STORE(ptr, val) (*(ptr) = (val))
#else
#define VHP_FIELD_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif
template <typename OutT> using Pair = OutT __attribute__((ext_vector_type(2)));
template <typename OutT> VHP_LANE_FN void g_store2(OutT* base, vu32 off, vd a, vd b) {
  VHP_DIAG_STORE_GUARD(a, b, off)
  VHP_FIELD_STORE(reinterpret_cast<Pair<OutT>*>(reinterpret_cast<char*>(base) + off), (Pair<OutT>{static_cast<OutT>(a), static_cast<OutT>(b)}));
}
// The PREDICATED stores -- the ragged ends at diagonals, axes and the ends of a march: 8- and 16-byte pieces of sectors that other
// wavefronts complete later -- are PLAIN stores (round 5): a non-temporal store of part of a 64-byte sector goes to the memory as it
// is and costs it a read-modify-write, a plain one is merged with its neighbours in the L2 first.  (The unpredicated field stores
// cover whole sectors -- 1 KB row pieces, 128-byte lines -- and keep the nt bit, above.)  Measured, 256 sources at 1000^2, all nt /
// predicated plain: slow buffer 0.578 -> 0.554 ms, fast 0.422 -> 0.416; 640^2 0.228 -> 0.222; 1002^2 0.586 -> 0.573; level at 2048^2
// and 4096^2 (profiles/r05_ab_predicated_stores_plain.txt).  -DVHP_PRED_STORE_NT: the round-4 behaviour.
#if defined(VHP_PRED_STORE_NT) && !defined(VHP_FIELD_STORE_PLAIN)
#define VHP_PRED_FIELD_STORE(ptr, val) VHP_FIELD_STORE(ptr, val)
#else
#define VHP_PRED_FIELD_STORE(ptr, val) (*(ptr) = (val))
#endif
template <typename OutT> VHP_LANE_FN void g_store2_if(bool p2, bool p_lo, bool p_hi, OutT* base, vu32 off, vd a, vd b) {
  VHP_DIAG_STORE_GUARD(a, b, off)
  vd single = p_lo ? a : b;
  asm volatile("" : "+v"(single));  // keep the compiler from splitting the 16-byte store to share a half with the single
  if (p2) VHP_PRED_FIELD_STORE(reinterpret_cast<Pair<OutT>*>(reinterpret_cast<char*>(base) + off), (Pair<OutT>{static_cast<OutT>(a), static_cast<OutT>(b)}));
  else if (p_lo || p_hi)
    VHP_PRED_FIELD_STORE(reinterpret_cast<OutT*>(reinterpret_cast<char*>(base) + off + (p_lo ? 0u : (uint32_t)sizeof(OutT))), static_cast<OutT>(single));
}
// Without the nt bit, whatever the translation unit's VHP_FIELD_STORE is: for 64-byte pieces, which a wavefront's L2 merges into
// lines when they are plain stores (3.75 TB/s against 3.07 with nt on the slow kind of memory).
template <typename OutT> VHP_LANE_FN void g_store2_plain(OutT* base, vu32 off, vd a, vd b) {
  VHP_DIAG_STORE_GUARD(a, b, off)
  *reinterpret_cast<Pair<OutT>*>(reinterpret_cast<char*>(base) + off) = Pair<OutT>{static_cast<OutT>(a), static_cast<OutT>(b)};
}
template <typename OutT> VHP_LANE_FN void g_store2_if_plain(bool p2, bool p_lo, bool p_hi, OutT* base, vu32 off, vd a, vd b) {
  VHP_DIAG_STORE_GUARD(a, b, off)
  vd single = p_lo ? a : b;
  asm volatile("" : "+v"(single));
  if (p2) *reinterpret_cast<Pair<OutT>*>(reinterpret_cast<char*>(base) + off) = Pair<OutT>{static_cast<OutT>(a), static_cast<OutT>(b)};
  else if (p_lo || p_hi) *reinterpret_cast<OutT*>(reinterpret_cast<char*>(base) + off + (p_lo ? 0u : (uint32_t)sizeof(OutT))) = static_cast<OutT>(single);
}
template <typename OutT> VHP_LANE_FN bool pair_aligned(const OutT* base, vu32 off) {
  return ((reinterpret_cast<uintptr_t>(base) + off) & (2 * sizeof(OutT) - 1)) == 0;
}
template <typename OutT> VHP_LANE_FN void g_store2_mask(bool p, OutT* base, vu32 off, vd a, vd b) {
  VHP_DIAG_STORE_GUARD(a, b, off)
  if (p) VHP_FIELD_STORE(reinterpret_cast<Pair<OutT>*>(reinterpret_cast<char*>(base) + off), (Pair<OutT>{static_cast<OutT>(a), static_cast<OutT>(b)}));
}
template <typename OutT> VHP_LANE_FN void g_store1_if(bool p1, OutT* base, vu32 off, vd a) {
  if (p1) VHP_PRED_FIELD_STORE(reinterpret_cast<OutT*>(reinterpret_cast<char*>(base) + off), static_cast<OutT>(a));   // (a single cell: part of a sector)
}
// Orders this wavefront's LDS writes before its later LDS reads of other lanes' data.  The LDS executes the DS
// instructions of one wavefront in issue order, so a read issued after a write sees it without an s_waitcnt in between:
// only the compiler must be kept from moving the read up (wave_barrier is a scheduling barrier, it emits no code).
VHP_LANE_FN void wave_sync() {
  __builtin_amdgcn_wave_barrier();
}
// The instruction scheduler moves nothing across this point (no code is emitted).  Between the steps of an unrolled window it keeps
// the compiler from computing all sixteen steps' ratios and masks ahead of the dependent chain -- sixty-odd registers that a lone
// wavefront gains nothing from.
VHP_LANE_FN void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Makes a just-loaded value count as "used here": the compiler then waits for the load at this point instead of at
// the first real use (where the s_waitcnt vmcnt would also drain every store issued in between).
VHP_LANE_FN void pin(double& v) { asm volatile("" : "+v"(v)); }
VHP_LANE_FN void pin(uint64_t& v) { asm volatile("" : "+v"(v)); }
VHP_LANE_FN void pin(int& v) { asm volatile("" : "+v"(v)); }  // (... and nothing derived from it is computed ahead of this point and kept)
VHP_LANE_FN int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
VHP_LANE_FN int lane0_int(vd a) { return __builtin_amdgcn_readfirstlane((int)a); }
typedef __attribute__((address_space(3))) int lds_int;
// Release of LDS data to the other wavefronts of the workgroup: the progress word is an LDS write issued after the data's
// LDS writes by the same wavefront, and the LDS executes one wavefront's instructions in order -- nothing to wait for.
// (A workgroup-scope release FENCE would also drain vmcnt: every global store the wavefront has in flight, 5 us per unit
// under load -- measured: it doubled the time of the largest quadrants.)  Only the compiler must keep the order.
VHP_LANE_FN void lds_publish(volatile int* word, int value) {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  // Every lane writes the same value to the same address: no exec masking around it.  The pointer is cast to the LDS
  // address space: through the generic pointer the compiler emits FLAT instructions, which are slow, are not ordered
  // with the DS instructions that wrote the data, and count on vmcnt -- a poll would wait for every global store the
  // wavefront has in flight.
  *(volatile lds_int*)word = value;
  asm volatile("" ::: "memory");
}
// a (uniform) int that was written to LDS before the wavefronts started
VHP_LANE_FN int lds_int_at(const int* p) { return __builtin_amdgcn_readfirstlane(*(const lds_int*)p); }
VHP_LANE_FN void lds_set_int(int* p, int v) { *(lds_int*)p = v; }
// a progress word of another wavefront, read afresh, as a uniform value (a DS read: see lds_publish)
VHP_LANE_FN int lds_poll(const volatile int* word) { return __builtin_amdgcn_readfirstlane(*(const volatile lds_int*)word); }
// The same read without waiting for it: the value (the same in every lane) stays in a vector register until uniform() is
// taken of it, so that the read joins a batch of LDS reads issued around it -- the LDS serves one wavefront's reads in the
// order they were issued, which is what a "header, data, header again" check needs; only the compiler must keep that order.
VHP_LANE_FN int lds_peek(const volatile int* word) {
  asm volatile("" ::: "memory");
  const int v = *(const volatile lds_int*)word;
  asm volatile("" ::: "memory");
  return v;
}
// Four uniform words for another wavefront, as ONE 16-byte LDS write (d is 16-byte aligned): every lane writes the same
// values to the same address, so there is no exec masking, and a reader sees all four words or none.  Ordered after
// this wavefront's earlier LDS writes like lds_publish.
VHP_LANE_FN void lds_post4(int* d, int a, int b, int c, int e) {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  *reinterpret_cast<int4*>(d) = make_int4(a, b, c, e);
  asm volatile("" ::: "memory");
}
// the four words of lds_post4, read afresh (one 16-byte LDS read), as uniform values
VHP_LANE_FN void lds_read4(const int* d, int& a, int& b, int& c, int& e) {
  asm volatile("" ::: "memory");
  const int4 v = *reinterpret_cast<const int4*>(d);
  a = __builtin_amdgcn_readfirstlane(v.x);
  b = __builtin_amdgcn_readfirstlane(v.y);
  c = __builtin_amdgcn_readfirstlane(v.z);
  e = __builtin_amdgcn_readfirstlane(v.w);
}
// instruction-arbitration priority of this wavefront (0 lowest .. 3): s_setprio takes an immediate
VHP_LANE_FN void wave_priority(int p) {
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
VHP_LANE_FN void sim_count(int) {}
VHP_LANE_FN void sim_progress() {}
VHP_LANE_FN void sim_point() {}
// LDS atomics executed by one lane of the wavefront, the old value returned to all of them as a uniform.  The pointers
// are cast to the LDS address space: through a generic pointer these would be FLAT atomics (slow, and counted on vmcnt).
VHP_LANE_FN int lds_cas(int* p, int expected, int desired) {
  int old = 0;
  if ((threadIdx.x & 63u) == 0) {
    int e = expected;
    __hip_atomic_compare_exchange_strong((lds_int*)p, &e, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = e;
  }
  return __builtin_amdgcn_readfirstlane(old);
}
VHP_LANE_FN int lds_add(int* p, int v) {
  int old = 0;
  if ((threadIdx.x & 63u) == 0) old = __hip_atomic_fetch_add((lds_int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(old);
}
VHP_LANE_FN int lds_or(int* p, int v) {
  int old = 0;
  if ((threadIdx.x & 63u) == 0) old = __hip_atomic_fetch_or((lds_int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(old);
}
VHP_LANE_FN int lds_and(int* p, int v) {
  int old = 0;
  if ((threadIdx.x & 63u) == 0) old = __hip_atomic_fetch_and((lds_int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(old);
}
// one global atomic add per wavefront (the unit queue), the old value as a uniform
VHP_LANE_FN int g_add(int* p, int v) {
  int old = 0;
  if ((threadIdx.x & 63u) == 0) old = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_amdgcn_readfirstlane(old);
}
VHP_LANE_FN unsigned long long g_add_u64(unsigned long long* p, unsigned long long v) {
  unsigned long long old = 0;
  if ((threadIdx.x & 63u) == 0) old = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)old), hi = __builtin_amdgcn_readfirstlane((unsigned)(old >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// a word that other workgroups update with atomics, read (not modified) as a uniform: an agent-scope load, no atomic
VHP_LANE_FN unsigned long long g_peek_u64(const unsigned long long* p) {
  const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// four consecutive ints at a uniform index (a 16-byte record that an earlier kernel wrote: one load, one round trip), as uniform values
VHP_LANE_FN void g_load_rec4(const int* base, int idx, int& a, int& b, int& c, int& d) {
  const int4 v = *reinterpret_cast<const int4*>(base + 4 * (size_t)idx);
  a = __builtin_amdgcn_readfirstlane(v.x);
  b = __builtin_amdgcn_readfirstlane(v.y);
  c = __builtin_amdgcn_readfirstlane(v.z);
  d = __builtin_amdgcn_readfirstlane(v.w);
}
VHP_LANE_FN void g_store_f64(double* base, vi idx, vd v) { base[idx] = v; }
VHP_LANE_FN void g_store_f64_if(bool p, double* base, vi idx, vd v) { if (p) base[idx] = v; }
template <typename T> VHP_LANE_FN void g_store_scalar_if(bool p, T* base, vi idx, T v) { if (p) base[idx] = v; }
VHP_LANE_FN int ffs_u32(uint32_t v) { return v ? __builtin_ctz(v) : -1; }
// Tagged 16-byte entries in global memory: one 16-byte store per lane (both words reach the L2 together); the loads are
// agent-scope (they miss the CU's L1, which a poll may have filled with the line's previous contents), tag first.
struct alignas(16) Tagged { double v; uint64_t tag; };
VHP_LANE_FN void g_store_tagged(Tagged* base, vi idx, vd v, uint64_t tag) { base[idx] = Tagged{v, tag}; }
VHP_LANE_FN bool g_load_tagged(const Tagged* base, vi idx, uint64_t tag, vd& v) {
  // tag FIRST, then the value: the writer stores both words with one 16-byte store, so a value loaded after a matching tag is
  // the one that came with it.  The hardware issues and returns a wavefront's loads in program order; the compiler barrier keeps
  // the program order that way (two relaxed atomic loads of different addresses may otherwise be swapped).
  const uint64_t t = __hip_atomic_load(&base[idx].tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
  v = __hip_atomic_load(&base[idx].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t == tag;
}
// (the loads are issued and nothing waits for them: the tag is compared where the caller needs the value)
VHP_LANE_FN void g_load_tagged_raw(const Tagged* base, vi idx, vu64& tag, vd& v) {
  tag = __hip_atomic_load(&base[idx].tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
  v = __hip_atomic_load(&base[idx].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
VHP_LANE_FN bool tags_are(vu64 tag, uint64_t want) { return tag == want; }
// A tagged entry for a reader that may sit behind ANOTHER L2 (another XCD): an agent-scope store, which goes through to memory.  A
// plain 16-byte store stays in the writer's L2 until the kernel ends -- a reader behind the same L2 sees it, any other never does (found
// when a unit's workgroups were dealt to neighbouring XCDs: the launch hung).
VHP_LANE_FN void g_store_tagged_device(Tagged* base, vi idx, vd v, uint64_t tag) {
#ifdef VHP_TAGGED_STORE_TWO_HALVES
  __hip_atomic_store(&base[idx].v, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
  __hip_atomic_store(&base[idx].tag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  // (one 16-byte store with the scope bit of an agent-scope atomic store -- the language has none that wide; the two halves as atomic
  // stores cost the build eleven vector registers it does not have.  The s_nop behind it: a wide store's data registers must not be
  // written by the next vector instruction, and the compiler does not look into an asm statement for that.)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
  u32x4 d;
  d.x = (unsigned)vb; d.y = (unsigned)(vb >> 32); d.z = (unsigned)tag; d.w = (unsigned)(tag >> 32);
  Tagged* p = base + idx;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
#endif
}
VHP_LANE_FN void g_store_tagged_if(bool p, Tagged* base, vi idx, vd v, uint64_t tag) { if (p) g_store_tagged_device(base, idx, v, tag); }
VHP_LANE_FN bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == ~0ull; }
VHP_LANE_FN void lds_store_i_if(bool p, int* base, vi idx, int v) { if (p) ((lds_int*)base)[idx] = v; }
VHP_LANE_FN void backoff() { __builtin_amdgcn_s_sleep(VHP_BACKOFF_SLEEP); }
VHP_LANE_FN void ready_backoff() { __builtin_amdgcn_s_sleep(VHP_READY_SLEEP); }
#ifndef VHP_TAKEN_SLEEP
#define VHP_TAKEN_SLEEP 1
#endif
VHP_LANE_FN void taken_backoff() { __builtin_amdgcn_s_sleep(VHP_TAKEN_SLEEP); }   // (a sweeper in the middle of its window's first step, waiting for its storer's word)
VHP_LANE_FN void short_backoff() { __builtin_amdgcn_s_sleep(1); }   // (64 cycles: a wavefront whose answer another one is waiting for)
// waits until every global store (and load) this wavefront has issued has completed
VHP_LANE_FN void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// acquire: the poll's value has arrived (the branch on it waited for lgkmcnt); later LDS reads are issued after it, in order
VHP_LANE_FN void lds_acquire() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
#endif

// RN(num/den) for integers 0 <= num < den <= 16384, given rden = RN(1/den): Markstein's correction, proved
// bit-identical to the division by oracle/markstein_check.c.  num is a lane vector, den and rden are uniform.
template <typename D, typename R>
VHP_LANE_FN vd ratio(vd num, D den, R rden) {
  VHP_DIAG_NOMATH_RETURN(num)
  const vd q = num * vd(rden);
  const vd r = vfma(-vd(den), q, num);
  return vfma(r, vd(rden), q);
}
// the reference's update (solver.cpp:592-594 / 598-600): a - c*(a - b), no contraction
VHP_LANE_FN vd stencil(vd a, vd b, vd c) {
  VHP_DIAG_NOMATH_RETURN(a)
  const vd t = a - b;
  const vd u = c * t;
  return a - u;
}

}  // namespace lanes
}  // namespace vhp
