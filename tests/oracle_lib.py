"""ctypes binding of oracle/libvhp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg and
by nothing else; the product package never touches it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
UNLABELLED = 1000000000000000

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(fast=False):
    target = "libvhp_oracle_fast.so" if fast else "all"
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, target])


def load(path=None):
    path = path or os.path.join(ORACLE_DIR, "libvhp_oracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.vhp_oracle_sweep_full.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _f64p]
    lib.vhp_oracle_sweep_full_offset.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _f64p]
    lib.vhp_oracle_planner_solve_offset.argtypes = [
        _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_double,
        _u64p, _f64p, _f64p, _i32p, C.POINTER(C.c_uint32)]
    lib.vhp_oracle_sweep_queue.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _f64p]
    lib.vhp_oracle_planner_solve.argtypes = [
        _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64,
        _u64p, _f64p, _f64p, _i32p, C.POINTER(C.c_uint32), C.c_void_p]
    lib.vhp_oracle_planner_step.argtypes = [
        _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64,
        _i32p, _u64p, _f64p, _f64p, _i32p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.vhp_oracle_reconstruct_path.restype = C.c_double
    lib.vhp_oracle_reconstruct_path.argtypes = [
        _u64p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.vhp_oracle_generate_env.argtypes = [
        C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _u8p]
    lib.vhp_oracle_raycast_all.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _f64p]
    if hasattr(lib, "vhp_oracle_sweep_matlab"):  # both libraries are built from both sources; tolerate an older fast build
        lib.vhp_oracle_sweep_matlab.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _f64p]
        lib.vhp_oracle_planner_matlab.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                  C.c_uint64, _u64p, _f64p, _f64p, _i32p, C.POINTER(C.c_uint32)]
    lib.vhp_oracle_time_sweeps.restype = C.c_double
    lib.vhp_oracle_time_sweeps.argtypes = [_u8p, C.c_int, C.c_int, _i32p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    return lib


class Oracle:
    """Arrays are indexed [y, x] (row-major, x fastest) like the reference's Field."""

    def __init__(self, path=None):
        self.lib = load(path)

    def sweep_full(self, occ, sx, sy, vis=None):
        ny, nx = occ.shape
        vis = np.zeros((ny, nx), np.float64) if vis is None else vis
        rc = self.lib.vhp_oracle_sweep_full(np.ascontiguousarray(occ, np.uint8), nx, ny, sx, sy, vis)
        assert rc == 0, rc
        return vis

    def sweep_full_offset(self, occ, sx, sy, offset):
        """TEST-ONLY: computeVisibility() with the reference's `offset` local (solver.cpp:573) set to `offset`."""
        ny, nx = occ.shape
        vis = np.zeros((ny, nx), np.float64)
        rc = self.lib.vhp_oracle_sweep_full_offset(np.ascontiguousarray(occ, np.uint8), nx, ny, sx, sy, float(offset), vis)
        assert rc == 0, rc
        return vis

    def sweep_queue(self, occ, sx, sy, vis=None):
        ny, nx = occ.shape
        vis = np.zeros((ny, nx), np.float64) if vis is None else vis
        rc = self.lib.vhp_oracle_sweep_queue(np.ascontiguousarray(occ, np.uint8), nx, ny, sx, sy, vis)
        assert rc == 0, rc
        return vis

    def sweep_matlab(self, occ, sx, sy, alpha=1.0, fac=1.0):
        """getAccessibilityMap.m restated (oracle/vhp_oracle_matlab.cpp; unpinned against MATLAB itself)."""
        ny, nx = occ.shape
        vis = np.full((ny, nx), np.nan)
        rc = self.lib.vhp_oracle_sweep_matlab(np.ascontiguousarray(occ, np.uint8), nx, ny, sx, sy, float(alpha), float(fac), vis)
        assert rc == 0, rc
        return vis

    def solve_matlab(self, occ, start, end, threshold, alpha, max_iter):
        ny, nx = occ.shape
        lab = np.zeros((ny, nx), np.uint64)
        uni = np.zeros((ny, nx), np.float64)
        loc = np.zeros((ny, nx), np.float64)
        way = np.zeros((max_iter + 3, 2), np.int32)
        n = C.c_uint32(0)
        rc = self.lib.vhp_oracle_planner_matlab(np.ascontiguousarray(occ, np.uint8), nx, ny, start[0], start[1], end[0], end[1],
                                                float(threshold), float(alpha), int(max_iter), lab, uni, loc, way, C.byref(n))
        return dict(status=rc, label=lab, map_builder=uni, local=loc, waypoints=way[: n.value].copy())

    def solve(self, occ, start, end, threshold, max_iter):
        ny, nx = occ.shape
        came = np.zeros((ny, nx), np.uint64)
        vg = np.zeros((ny, nx), np.float64)
        vl = np.zeros((ny, nx), np.float64)
        piv = np.zeros((max_iter + 2, 2), np.int32)
        n = C.c_uint32(0)
        rc = self.lib.vhp_oracle_planner_solve(
            np.ascontiguousarray(occ, np.uint8), nx, ny, start[0], start[1], end[0], end[1],
            float(threshold), int(max_iter), came, vg, vl, piv, C.byref(n), None)
        return dict(status=rc, came_from=came, vis_global=vg, vis_local=vl,
                    pivots=piv[: n.value + 1].copy(), n_pivots=n.value)

    def planner_step(self, occ, pivot, end, threshold, label, pivots, came, vg):
        ny, nx = occ.shape
        vl = np.zeros((ny, nx), np.float64)
        top = np.zeros(2, np.int32)
        h = C.c_double(0)
        npush = C.c_uint64(0)
        rc = self.lib.vhp_oracle_planner_step(
            np.ascontiguousarray(occ, np.uint8), nx, ny, pivot[0], pivot[1], end[0], end[1],
            float(threshold), int(label), np.ascontiguousarray(pivots, np.int32), came, vg, vl,
            top, C.byref(h), C.byref(npush))
        return dict(status=rc, vis_local=vl, top=(int(top[0]), int(top[1])), top_h=h.value,
                    n_pushed=npush.value)

    def reconstruct_path(self, came, pivots, end):
        ny, nx = came.shape
        cap = 4096
        path = np.zeros((cap, 2), np.int32)
        n = C.c_uint32(0)
        d = self.lib.vhp_oracle_reconstruct_path(
            came, np.ascontiguousarray(pivots, np.int32), nx, ny, end[0], end[1], path, cap, C.byref(n))
        return d, path[: n.value].copy()

    def generate_env(self, nx, ny, nb, min_w, max_w, min_h, max_h, seed):
        occ = np.zeros((ny, nx), np.uint8)
        rc = self.lib.vhp_oracle_generate_env(nx, ny, nb, min_w, max_w, min_h, max_h, seed, occ)
        assert rc == 0
        return occ

    def time_sweeps(self, occ, sources, n_threads=1):
        """(wall seconds for all sweeps, best single-sweep seconds); steady clock around the sweep only."""
        ny, nx = occ.shape
        src = np.ascontiguousarray(sources, np.int32).reshape(-1, 2)
        best = C.c_double(0)
        wall = self.lib.vhp_oracle_time_sweeps(np.ascontiguousarray(occ, np.uint8), nx, ny, src, len(src),
                                               int(n_threads), C.byref(best))
        return wall, best.value

    def raycast_all(self, occ, sx, sy):
        ny, nx = occ.shape
        ray = np.ones((ny, nx), np.float64)
        self.lib.vhp_oracle_raycast_all(np.ascontiguousarray(occ, np.uint8), nx, ny, sx, sy, ray)
        return ray
