"""ctypes binding of tests/sim/libvhp_pool_sim.so -- TEST INFRASTRUCTURE ONLY.

The simulator compiles the pool sweep's and the latency sweep's own source (csrc/vhp_pool.hpp, vhp_lat.hpp) for the CPU with
64 explicit lanes per wavefront, so that their schedules and index arithmetic can be checked against the oracle without a
GPU.  Nothing in the product imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, "tests", "sim")
_pool = None


def load_pool():
    global _pool
    if _pool is None:
        name = os.environ.get("VHP_SIM_LIB", "libvhp_pool_sim.so")   # (libvhp_pool_sim_asan.so: the AddressSanitizer build)
        subprocess.check_call(["make", "-s", "-C", SIM_DIR, name])
        lib = C.CDLL(os.path.join(SIM_DIR, name))
        lib.vhp_sim_pool_sweep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_uint, C.c_void_p]
        lib.vhp_sim_lat_sweep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint,
                                          C.c_void_p]
        lib.vhp_sim_band_sweep.argtypes = lib.vhp_sim_lat_sweep.argtypes
        _pool = lib
    return _pool


# pool simulator policies: who runs next (& 7), and whether the marked race points switch wavefronts (always / at random)
POOL_ROUND_ROBIN, POOL_BACKWARD, POOL_RANDOM, POOL_GREEDY, POOL_BURSTS = 0, 1, 2, 3, 4
POOL_POINTS_ALWAYS, POOL_POINTS_RANDOM, POOL_SHUFFLED_QUEUE, POOL_TIGHT_BUSY_CAP = 8, 16, 32, 64


def pool_sweep(occ, sources, dtype=np.float64, W=12, C=4, G=1, policy=0, seed=1, out_offset=0):
    """Fields [n, ny, nx] of the simulated pool kernel (csrc/vhp_pool.hpp; pre-filled with NaN) and the stats dict.
    out_offset: the fields start that many cells into a 128-byte aligned buffer (off the line grid, or off the 16-byte grid)."""
    lib = load_pool()
    occ = np.ascontiguousarray(occ, np.uint8)
    ny, nx = occ.shape
    src = np.ascontiguousarray(sources, np.int32).reshape(-1, 2)
    item = np.dtype(dtype).itemsize
    raw = np.full(len(src) * ny * nx + out_offset + 128 // item + 32, np.nan, dtype)
    lead = (-raw.ctypes.data % 128) // item + out_offset
    out = raw[lead: lead + len(src) * ny * nx].reshape(len(src), ny, nx)
    stats = np.zeros(14, np.int64)
    rc = lib.vhp_sim_pool_sweep(occ.ctypes.data, nx, ny, src.ctypes.data, len(src), 0 if dtype == np.float64 else 1, out.ctypes.data,
                                W, C, G, policy, seed, stats.ctypes.data)
    assert rc == 0, rc
    assert np.isnan(raw[:lead]).all() and np.isnan(raw[lead + out.size:]).all(), "a store outside the fields"
    return out, dict(switches=int(stats[0]), progress=int(stats[1]), deadlock=int(stats[2]), st16=int(stats[3]), st8=int(stats[4]),
                     err=int(stats[5]), pulled=int(stats[6]), from_ring=int(stats[7]), from_global=int(stats[8]), too_far=int(stats[9]),
                     overwritten=int(stats[10]), lines_whole=int(stats[11]), lines_part=int(stats[12]), static_round=int(stats[13]))


def lat_sweep(occ, sources, dtype=np.float64, W=12, policy=0, seed=1, bands=None, halves=None):
    """Fields [n, ny, nx] of the simulated latency sweep (a workgroup of W wavefronts per unit) and the stats dict.
    bands: the sweep in bands (csrc/vhp_band.hpp, what the library launches) or in strips of rows (csrc/vhp_lat.hpp).
    halves: workgroups per unit of the band sweep (1, or 2: the bands of an octant dealt out to two workgroups, LatArgs::halves;
    default: the environment's VHP_SIM_HALVES, else 1)."""
    lib = load_pool()
    if bands is None:
        bands = os.environ.get("VHP_SIM_LAT", "bands") == "bands"
    if halves is None:
        halves = int(os.environ.get("VHP_SIM_HALVES", "1"))
    lib.vhp_sim_set_lat_halves(int(halves))
    occ = np.ascontiguousarray(occ, np.uint8)
    ny, nx = occ.shape
    src = np.ascontiguousarray(sources, np.int32).reshape(-1, 2)
    out = np.full((len(src), ny, nx), np.nan, dtype)
    stats = np.zeros(11, np.int64)
    rc = (lib.vhp_sim_band_sweep if bands else lib.vhp_sim_lat_sweep)(occ.ctypes.data, nx, ny, src.ctypes.data, len(src), 0 if dtype == np.float64 else 1,
                                                                      out.ctypes.data, W, policy, seed, stats.ctypes.data)
    assert rc == 0, rc
    return out, dict(switches=int(stats[0]), progress=int(stats[1]), deadlock=int(stats[2]), st16=int(stats[3]), st8=int(stats[4]),
                     err=int(stats[5]), died=int(stats[6]), from_ring=int(stats[7]), from_global=int(stats[8]), too_far=int(stats[9]),
                     overwritten=int(stats[10]))


LAZY_FLUSH = 8   # order flag: the flushers of x-major strips run as late as the hand-off allows
TWO_SLOTS = 16   # order flag: two tile slots (a plain hand-off) instead of three
def tile_slots(n):
    """order flag: n tile slots (4, 6 or 8: what a launch with one workgroup per CU may use)"""
    return n << 8


SMALL_Y_TEAM = 32  # order flag: y-major units swept by W - 1 wavefronts (two such units share a workgroup on the GPU)


def lds_bytes(nx, ny, W=4, tile_slots=3):
    return load().vhp_sim_lds_bytes(nx, ny, W, tile_slots)
