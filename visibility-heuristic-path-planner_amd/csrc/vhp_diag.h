// vhp_diag.h -- every diagnostic build switch of the kernels, in one place.
//
// The product is built with NONE of these defined, and then this header defines empty macros only.  tools/build_exp.sh
// builds exp/libvhp_<NAME>.so with one of them for tools/ab_libs.py, tools/stream_timeline.py, tools/pool_timeline.py and tools/lat_timeline.py.
// Several produce WRONG results on purpose (they take a cost away to measure it); none is reachable from the C ABI.
//
//   VHP_DIAG_NOSTORE    all the work, none of the field stores                        (every batch kernel)
//   VHP_DIAG_NOXSTORE   no x-major field stores / VHP_DIAG_NOYSTORE no y-major ones      (pool sweep)
//   VHP_DIAG_NOMATH     the stencil and the ratio return an operand: the traffic without the arithmetic
//   VHP_DIAG_NOPARTIAL  no predicated field store (the partially written sectors)        (streaming and pool sweep)
//   VHP_DIAG_DROP_XPRED / _YPRED / _XRAGGED   the predicated x-major flushes / y-major stores / ragged x-major rows only (pool sweep)
//   VHP_DIAG_NOWAIT     no strip waits for the strip below or for its seeds: the launch's stores at full speed (pool sweep)
//   VHP_DIAG_WGTIME     per-workgroup times and per-wavefront cycle accounts             (streaming sweep)
//   VHP_DIAG_POOLPROF   per-wavefront cycle accounts, per-unit install / finish times    (pool sweep); per-strip stamps (latency sweep)
//   VHP_DIAG_WINPROF    cycle accounts inside the x-major windows (with POOLPROF; ~250 cycles per probe)   (latency sweep)
//   VHP_DIAG_NODEATH    no strip ever declares itself dead: what the early exits are worth                  (latency sweep)
//   VHP_DIAG_NODIAGSTORE  a strip that is growing along its diagonal stores nothing: the bound on what handing its stores to
//                       another wavefront could buy (C2: 98.8 -> 79 us)                                     (latency sweep)
//
// Experiments that are over were deleted together with their switches (round 3): FLATPOLL, MASKPUB, HEAVYSYNC, NOLOAD,
// YDRAIN, NOREFILL, SLOTTIME, SMALLSTORE, NOSTORE_X/_Y of the front sweep, PRIO, the back-off lengths as -D values.
// What they measured is in DESIGN.md sections 4 and 4b.
#pragma once

// field stores of the lane-vector kernels (vhp_lanes.hpp g_store2 / g_store2_if): a, b = the values, off = the byte offset
#ifdef VHP_DIAG_NOSTORE
#define VHP_DIAG_STORE_GUARD(a, b, off) { asm volatile("" :: "v"(a), "v"(b), "v"(off)); return; }
#define VHP_DIAG_FRONT_STORE_GUARD if (nx != 0x7fffffff) return;   // front sweep (StoreEmit): nx is never that
#else
#define VHP_DIAG_STORE_GUARD(a, b, off)
#define VHP_DIAG_FRONT_STORE_GUARD
#endif

// every PREDICATED field store of the lane-vector kernels dropped (wrong results): what the partially written sectors at
// octant diagonals, quadrant axes and ragged edges cost the memory system
#ifdef VHP_DIAG_NOPARTIAL
#define VHP_DIAG_PARTIAL_GUARD(a, b, off) { asm volatile("" :: "v"(a), "v"(b), "v"(off)); return; }
#else
#define VHP_DIAG_PARTIAL_GUARD(a, b, off)
#endif

// back-off of a wavefront that waits (s_sleep units of 64 cycles): measured in round 2, 12 for a hand-off that is not
// ready, 4 for a dependency that usually is (lesson 3)
#define VHP_BACKOFF_SLEEP 12
#define VHP_READY_SLEEP 4
