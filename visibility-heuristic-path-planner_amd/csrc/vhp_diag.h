// vhp_diag.h -- every diagnostic build switch of the kernels, in one place.
//
// The product is built with NONE of these defined, and then this header defines empty macros only.  tools/build_exp.sh
// builds exp/libvhp_<NAME>.so with one of them for tools/ab_libs.py, tools/ab_slowfast.py, tools/launch_timeline.py and tools/lat_timeline.py.
// Several produce WRONG results on purpose (they take a cost away to measure it); none is reachable from the C ABI.
//
//   VHP_DIAG_NOSTORE    all the work, none of the field stores                        (every batch kernel)
//   VHP_DIAG_PLAINSTORE the field stores of the batch kernels without the nt bit (what a launch took until round 4)
//   VHP_DIAG_NOXSTORE   ... none of the x-major strips' (pool sweep, both builds)
//   VHP_DIAG_NOYSTORE   ... none of the y-major strips' (pool sweep, both builds)
//   VHP_DIAG_NOMATH     the stencil and the ratio return an operand: the traffic without the arithmetic
//   VHP_DIAG_NOWAIT     no strip waits for the strip below or for its seeds: the launch's stores at full speed (pool sweep)
//   VHP_DIAG_TIMELINE   bytes swept and strips running per 10 us of a launch, per workgroup (pool sweep; tools/launch_timeline.py)
//   VHP_DIAG_POOLPROF   per-strip stamps (latency sweep; tools/lat_timeline.py)
//   VHP_DIAG_WINPROF    cycle accounts inside the x-major windows (with POOLPROF; ~250 cycles per probe)   (latency sweep)
//   VHP_DIAG_NODEATH    no strip ever declares itself dead: what the early exits are worth                  (latency sweep)
//   VHP_DIAG_NODIAGSTORE  a strip that is growing along its diagonal stores nothing: the bound on what handing its stores to
//                       another wavefront could buy (C2: 98.8 -> 79 us)                                     (latency sweep)
// A/B switches of round 5 (correct results; the product is the build WITHOUT them):
//   VHP_POOL_X8 / VHP_POOL_Y8   the pool sweep's strips in the 8-step windows of rounds 3-4 (both kinds / the y-major ones only)
//   VHP_POOL_SPLIT_ENDS=0         the ANYW build's unpredicated y-major stores all with the nt bit (the ends of a row piece too)
//   VHP_POOL_ANYW_Y16           the 16-step y-major strips also in the build for the other widths (measured slower: vhp_pool.hpp)
//   VHP_POOL_NO_HDR_POLL        a waiting strip re-reads the 17 boundary values with every poll, not the header alone
//   VHP_PRIO_DIAG / VHP_PRIO_TASK = 0   no instruction priorities by phase;  VHP_PRIO_MARCH = n: ... and by what is left of a march
//   VHP_EPI_BLOCKS / VHP_EPI_THREADS    the planner epilogue's launch shape (vhp_planner.hip.h)
//
// Experiments that are over were deleted together with their switches.  Round 3: FLATPOLL, MASKPUB, HEAVYSYNC, NOLOAD, YDRAIN,
// NOREFILL, SLOTTIME, SMALLSTORE, NOSTORE_X/_Y of the front sweep, PRIO, the back-off lengths as -D values.  Round 4: NOPARTIAL, DROP_XPRED / _YPRED / _XRAGGED, WHOLELINES, NOLINES, YALIGNED, NOBANDLOAD / NOBANDTASK (the seam band
// itself: `git show a1f0eca`), WGTIME (it went with the streaming sweep), the pool sweep's cycle accounts (POOLPROF there: one
// s_memtime per probe slowed the launch by a third; the timeline took their place).  What they measured is in DESIGN.md,
// appendix A (A.4b, A.4c, A.7) and section 6.  The kernels' sources contain the macros below and no #if of these switches (the latency sweep's
// stamps and window accounts excepted: vhp_lat.hpp).
#pragma once

// field stores of the lane-vector kernels (vhp_lanes.hpp g_store2 / g_store2_if): a, b = the values, off = the byte offset
#ifdef VHP_DIAG_NOSTORE
#define VHP_DIAG_STORE_GUARD(a, b, off) { asm volatile("" :: "v"(a), "v"(b), "v"(off)); return; }
#define VHP_DIAG_FRONT_STORE_GUARD if (nx != 0x7fffffff) return;   // front sweep (StoreEmit): nx is never that
#else
#define VHP_DIAG_STORE_GUARD(a, b, off)
#define VHP_DIAG_FRONT_STORE_GUARD
#endif

// one kind of strip of the pool sweep stores nothing (wrong results): what the other kind's stores cost by themselves
#ifdef VHP_DIAG_NOXSTORE
#define VHP_DIAG_NOXSTORE_RETURN return;
#else
#define VHP_DIAG_NOXSTORE_RETURN
#endif
#ifdef VHP_DIAG_NOYSTORE
#define VHP_DIAG_NOYSTORE_RETURN return;
#define VHP_DIAG_NOYSTORE_GUARD if (m.nx == 0x7fffffff)
#else
#define VHP_DIAG_NOYSTORE_RETURN
#define VHP_DIAG_NOYSTORE_GUARD
#endif

// the field stores of the batch kernels without the nt bit
#ifdef VHP_DIAG_PLAINSTORE
#define VHP_FIELD_STORE_PLAIN 1
#endif

// no strip waits for another (wrong results): fetch() returns at once, find_work() claims whatever is unclaimed
#ifdef VHP_DIAG_NOWAIT
#define VHP_DIAG_NOWAIT_RETURN return;
#define VHP_DIAG_WAITS false
#else
#define VHP_DIAG_NOWAIT_RETURN
#define VHP_DIAG_WAITS true
#endif

// the arithmetic of a step replaced by one of its operands (wrong results)
#ifdef VHP_DIAG_NOMATH
#define VHP_DIAG_NOMATH_RETURN(x) return (x);
#else
#define VHP_DIAG_NOMATH_RETURN(x)
#endif

// The timeline of a pool-sweep launch: cells swept (= bytes stored, one window late) and strips running, per 10 us of wall clock
// since the order pre-kernel, one atomic per 64-step block of a strip into the histogram of the wavefront's own workgroup (one
// histogram for the chip was 3000 wavefronts adding to one address: +30 % launch time).  Costs 5-8 % of the launch.
#if defined(VHP_DIAG_TIMELINE) && !defined(VHP_SIM)
#define VHP_DIAG_TL_DECLARE                                                                                                     \
  constexpr int kPpBins = 256;                                                                                                  \
  static __device__ unsigned long long g_pp_hist[256 * 2 * kPpBins];                                                            \
  static __device__ unsigned long long g_pp_t0;                                                                                 \
  static __device__ unsigned g_pp_unit[16384 * 2];   /* per unit: installed / finished, in 10 ns ticks since the order pre-kernel */ \
  static __device__ __forceinline__ unsigned long long* pp_slot(int which) {                                                    \
    const unsigned long long b_ = (wall_clock64() - g_pp_t0) / 1000;                                                            \
    return g_pp_hist + ((size_t)(blockIdx.x & 255) * 2 + which) * kPpBins + (b_ < kPpBins - 1 ? b_ : kPpBins - 1);              \
  }
#define VHP_DIAG_TL_ADD(which, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(pp_slot(which), (unsigned long long)(long long)(n)); } while (0)
#define VHP_DIAG_TL_STRIPS(d) VHP_DIAG_TL_ADD(1, d);
#define VHP_DIAG_TL_UNIT(unit, which) do { if ((threadIdx.x & 63) == 0 && (unit) < 16384) g_pp_unit[2 * (unit) + (which)] = (unsigned)(wall_clock64() - g_pp_t0); } while (0);
#define VHP_DIAG_TL_XBLOCK(lo, hi, rows_here, j0, cb) { long c_ = 0; for (int i_ = (lo); i_ <= (hi); ++i_) c_ += ((rows_here) < i_ - (j0) + 1 ? (rows_here) : i_ - (j0) + 1); VHP_DIAG_TL_ADD(0, c_ * (cb)); }
#define VHP_DIAG_TL_YBLOCK(lo, hi, i0, ycols, ni, cb) { long c_ = 0; for (int j_ = (lo); j_ <= (hi); ++j_) { int t_ = (i0) + (ycols) - 1; if ((ni) - 1 < t_) t_ = (ni) - 1; if (j_ < t_) t_ = j_; t_ -= ((i0) > 0 ? (i0) : 0) - 1; if (t_ > 0) c_ += t_; } VHP_DIAG_TL_ADD(0, c_ * (cb)); }
#define VHP_DIAG_TL_RESET for (int k_ = threadIdx.x; k_ < 256 * 2 * kPpBins; k_ += blockDim.x) g_pp_hist[k_] = 0; if (threadIdx.x == 0) g_pp_t0 = wall_clock64();
#else
#define VHP_DIAG_TL_DECLARE
#define VHP_DIAG_TL_STRIPS(d)
#define VHP_DIAG_TL_UNIT(unit, which)
#define VHP_DIAG_TL_XBLOCK(lo, hi, rows_here, j0, cb)
#define VHP_DIAG_TL_YBLOCK(lo, hi, i0, ycols, ni, cb)
#define VHP_DIAG_TL_RESET
#endif

// Instruction-arbitration priority by phase (s_setprio; round 5): a strip that is growing along its diagonal is the chain of its unit
// -- the strip above cannot start before it has got there --, so it issues ahead of the strips in their steady phase that share its
// SIMD; the diagonal task of a y-major unit (every strip of the unit waits for its seeds) ahead of both; idle wavefronts polling for
// work behind everybody.  Measured (tools/ab_slowfast.py, 256 sources at 1000^2, fast buffer): 0.446 -> 0.432 ms; 128 sources at
// 2048^2 1.039 -> 1.000; 96 at 1000^2 0.304 -> 0.286; nothing where the memory bounds the launch (slow buffer 0.583, C5 3.53).
// -DVHP_PRIO_DIAG=0 -DVHP_PRIO_TASK=0: off.
#ifndef VHP_PRIO_DIAG
#define VHP_PRIO_DIAG 1
#endif
#ifndef VHP_PRIO_TASK
#define VHP_PRIO_TASK 2
#endif
// EXPERIMENT: a steady strip's priority by what is left of its march (steps): the strips that end last issue first
#if !defined(VHP_SIM) && defined(VHP_PRIO_MARCH)
#define VHP_EXP_PRIO_SET_LEFT(is_diag, left) wave_priority((is_diag) ? 3 : (left) >= VHP_PRIO_MARCH ? 2 : (left) >= VHP_PRIO_MARCH / 2 ? 1 : 0);
#define VHP_EXP_PRIO_SET(is_diag) wave_priority((is_diag) ? 3 : 0);
#define VHP_EXP_PRIO_TASK_BEGIN wave_priority(3);
#define VHP_EXP_PRIO_END wave_priority(0);
#elif !defined(VHP_SIM) && (VHP_PRIO_DIAG != 0 || VHP_PRIO_TASK != 0)
#define VHP_EXP_PRIO_SET_LEFT(is_diag, left) wave_priority((is_diag) ? VHP_PRIO_DIAG : 0);
#define VHP_EXP_PRIO_SET(is_diag) wave_priority((is_diag) ? VHP_PRIO_DIAG : 0);
#define VHP_EXP_PRIO_TASK_BEGIN wave_priority(VHP_PRIO_TASK);
#define VHP_EXP_PRIO_END wave_priority(0);
#else
#define VHP_EXP_PRIO_SET_LEFT(is_diag, left)
#define VHP_EXP_PRIO_SET(is_diag)
#define VHP_EXP_PRIO_TASK_BEGIN
#define VHP_EXP_PRIO_END
#endif

// back-off of a wavefront that waits (s_sleep units of 64 cycles): measured in round 2, 12 for a hand-off that is not
// ready, 4 for a dependency that usually is (lesson 3)
#ifndef VHP_BACKOFF_SLEEP
#define VHP_BACKOFF_SLEEP 12
#endif
#ifndef VHP_READY_SLEEP
#define VHP_READY_SLEEP 4
#endif
