"""visibility-heuristic-path-planner_amd -- MI355X-native visibility sweep + visibility-heuristic planner.

Python is plumbing only: this module binds the C ABI of include/vhp.h (libvhp_hip.so, HIP
kernels for gfx950) with ctypes so tests, bench.py and the torch.distributed sharding helper
can drive it.  There is no CPU implementation behind these calls: if the library or a HIP
device is missing they raise.

The directory name is not a Python identifier; import it through the `vhp_amd` shim at the
repo root (``import vhp_amd``) or ``importlib.import_module("visibility-heuristic-path-planner_amd")``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VHP_LIB") or os.path.join(_HERE, "libvhp_hip.so")  # VHP_LIB: experiment builds
CSRC = os.path.join(_HERE, "csrc")

VHP_OK = 0
VHP_ERR_ARG = 1
VHP_ERR_SOURCE_OOB = 2
VHP_ERR_NOTHING_LIT = 3
VHP_ERR_START_OOB = 10
VHP_ERR_END_OOB = 11
VHP_ERR_START_OCCUPIED = 12
VHP_ERR_END_OCCUPIED = 13
VHP_ERR_MAX_ITER = 20
VHP_ERR_HIP = 100
VHP_ERR_NO_MAP = 101
VHP_ERR_TOO_LARGE = 102
SWEEP_FULL, SWEEP_QUEUE = 0, 1
F64, F32 = 0, 1
UNLABELLED = 1000000000000000

# every symbol include/vhp.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = (
    "vhp_create", "vhp_destroy", "vhp_last_error", "vhp_set_stream", "vhp_set_map", "vhp_set_map_device",
    "vhp_sweep_batch", "vhp_sweep_batch_device", "vhp_sync", "vhp_planner_solve", "vhp_reconstruct_path",
    "vhp_raycast_all", "vhp_timing", "vhp_timing_collect", "vhp_set_option", "vhp_sweep_batch_variant", "vhp_planner_solve_variant",
    "vhp_planner_solve_device", "vhp_planner_results_device", "vhp_last_sweep_kernel",
    "vhp_last_elapsed_ms", "vhp_version", "vhp_sweep_batch_offset", "vhp_planner_solve_speculative", "vhp_probe_stores", "vhp_alloc_output", "vhp_alloc_output_cost", "vhp_free_output",
    "vhp_multi_create", "vhp_multi_destroy", "vhp_multi_last_error", "vhp_multi_devices", "vhp_multi_context", "vhp_multi_shard_bounds",
    "vhp_multi_set_map", "vhp_multi_sweep_batch", "vhp_multi_allgather_fields", "vhp_multi_allgather_plan", "vhp_multi_use_rccl",
    "vhp_union_fields_device", "vhp_union_partials_device", "vhp_multi_union_fields",
)


class VhpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vhp status %d: %s" % (code, msg))
        self.code = code


def build_library(force=False):
    """Compile libvhp_hip.so in-tree with hipcc (cross-compiles gfx950 without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libvhp_hip.so.  Import torch first if you use it: both then share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libvhp_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C %s`" % CSRC)
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
    lib.vhp_create.argtypes = [i32, C.POINTER(vp)]
    lib.vhp_destroy.argtypes = [vp]
    lib.vhp_last_error.argtypes = [vp]
    lib.vhp_last_error.restype = C.c_char_p
    lib.vhp_set_stream.argtypes = [vp, vp]
    lib.vhp_set_map.argtypes = [vp, vp, i32, i32]
    lib.vhp_set_map_device.argtypes = [vp, vp, i32, i32]
    lib.vhp_sweep_batch.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.vhp_sweep_batch_device.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.vhp_sync.argtypes = [vp]
    lib.vhp_planner_solve.argtypes = [vp, i32, i32, i32, i32, f64, u64, vp, vp, vp, vp, C.POINTER(u32)]
    lib.vhp_reconstruct_path.argtypes = [vp, vp, u32, i32, i32, i32, i32, vp, u32, C.POINTER(u32), C.POINTER(f64)]
    lib.vhp_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    lib.vhp_planner_solve_speculative.argtypes = [vp, i32, i32, i32, i32, f64, u64, i32, i32, vp, vp, vp, vp, C.POINTER(u32), vp]
    lib.vhp_planner_solve_device.argtypes = [vp, i32, i32, i32, i32, f64, u64, C.POINTER(u32)]
    lib.vhp_planner_results_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.vhp_sweep_batch_variant.argtypes = [vp, vp, i32, f64, f64, vp]
    lib.vhp_planner_solve_variant.argtypes = [vp, i32, i32, i32, i32, f64, f64, u64, vp, vp, vp, vp, C.POINTER(u32)]
    lib.vhp_sweep_batch_offset.argtypes = [vp, vp, i32, f64, vp]
    lib.vhp_raycast_all.argtypes = [vp, i32, i32, vp]
    lib.vhp_timing.argtypes = [vp, i32]
    lib.vhp_timing_collect.argtypes = [vp, vp, i32, C.POINTER(i32)]
    lib.vhp_last_elapsed_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.vhp_version.restype = C.c_char_p
    lib.vhp_last_sweep_kernel.argtypes = [vp]
    lib.vhp_probe_stores.argtypes = [vp, vp, C.c_ulonglong, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.vhp_alloc_output.argtypes = [vp, C.c_ulonglong, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.vhp_free_output.argtypes = [vp, vp]
    lib.vhp_alloc_output_cost.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]
    lib.vhp_multi_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    lib.vhp_multi_destroy.argtypes = [vp]
    lib.vhp_multi_last_error.argtypes = [vp]
    lib.vhp_multi_last_error.restype = C.c_char_p
    lib.vhp_multi_devices.argtypes = [vp]
    lib.vhp_multi_context.argtypes = [vp, i32]
    lib.vhp_multi_context.restype = vp
    lib.vhp_multi_shard_bounds.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.vhp_multi_shard_bounds.restype = None
    lib.vhp_multi_set_map.argtypes = [vp, vp, i32, i32]
    lib.vhp_multi_sweep_batch.argtypes = [vp, vp, i32, i32, i32, C.POINTER(vp)]
    lib.vhp_multi_allgather_fields.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(vp)]
    lib.vhp_multi_allgather_plan.argtypes = [i32, i32, vp, vp, vp, vp, vp, i32]
    lib.vhp_multi_use_rccl.argtypes = [vp, i32]
    lib.vhp_union_fields_device.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.vhp_union_partials_device.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.vhp_multi_union_fields.argtypes = [vp, i32, i32, vp, vp, vp]
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU, one occupancy map.  Mirrors the lifetime of vbs::visibilityBasedSolver."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.vhp_create(int(device), C.byref(h))
        if rc != VHP_OK:
            raise VhpError(rc, "vhp_create failed (no usable HIP device %d?)" % device)
        self.h = h
        self.nx = self.ny = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.vhp_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc, ok=(VHP_OK,)):
        if rc not in ok:
            raise VhpError(rc, (self.lib.vhp_last_error(self.h) or b"").decode())
        return rc

    def set_stream(self, stream_handle):
        self._check(self.lib.vhp_set_stream(self.h, C.c_void_p(stream_handle or 0)))

    def set_map(self, occ):
        occ = np.ascontiguousarray(occ, np.uint8)
        self.ny, self.nx = occ.shape
        self._check(self.lib.vhp_set_map(self.h, _ptr(occ), self.nx, self.ny))

    def set_map_device(self, dptr, nx, ny):
        self.nx, self.ny = nx, ny
        self._check(self.lib.vhp_set_map_device(self.h, C.c_void_p(dptr), nx, ny))

    def sweep_batch(self, sources, variant=SWEEP_FULL, dtype=F64):
        """Host-buffer form.  sources int32 [n, 2] (x, y) -> fields [n, ny, nx]."""
        src = np.ascontiguousarray(sources, np.int32).reshape(-1, 2)
        out = np.empty((len(src), self.ny, self.nx), np.float64 if dtype == F64 else np.float32)
        self._check(self.lib.vhp_sweep_batch(self.h, _ptr(src), len(src), variant, dtype, _ptr(out)))
        return out

    def sweep_batch_device(self, d_src, n_src, d_out, variant=SWEEP_FULL, dtype=F64):
        """Device-resident, asynchronous on the context stream.  d_src / d_out are raw device pointers."""
        self._check(self.lib.vhp_sweep_batch_device(self.h, C.c_void_p(d_src), n_src, variant, dtype, C.c_void_p(d_out)))

    def union_fields_device(self, d_fields, n_fields, d_best, d_arg, first_index=0, dtype=F64):
        """Max-union of n_fields device-resident fields and the (lowest) source index that attains it, into d_best / d_arg (int32);
        raw device pointers, asynchronous on the context stream (include/vhp.h vhp_union_fields_device)."""
        self._check(self.lib.vhp_union_fields_device(self.h, C.c_void_p(d_fields), n_fields, dtype, first_index, C.c_void_p(d_best), C.c_void_p(d_arg)))

    def union_partials_device(self, d_bests, d_args, n_parts, d_best, d_arg, dtype=F64):
        """The same reduction over partial unions with their label fields (vhp_union_partials_device)."""
        self._check(self.lib.vhp_union_partials_device(self.h, C.c_void_p(d_bests), C.c_void_p(d_args), n_parts, dtype, C.c_void_p(d_best), C.c_void_p(d_arg)))

    def sync(self):
        self._check(self.lib.vhp_sync(self.h))

    def set_option(self, key, value):
        """Launch-shape override (include/vhp.h vhp_set_option); 0 / -1 = automatic."""
        self._check(self.lib.vhp_set_option(self.h, key.encode(), int(value)))

    def last_sweep_kernel(self):
        """1 = front sweep, 3 = pool sweep, 4 = latency sweep: what the last batch sweep (or planner solve) launched."""
        return int(self.lib.vhp_last_sweep_kernel(self.h))

    def timing(self, enable=True, prealloc=0):
        """prealloc > 1: event pairs created now, so that launches inside a timed loop create none."""
        self._check(self.lib.vhp_timing(self.h, max(int(prealloc), 1) if enable else 0))

    def timing_collect(self, cap=4096):
        """Durations (ms) of the sweep kernels launched since timing(True), oldest first."""
        buf = np.zeros(cap, np.float32)
        n = C.c_int(0)
        self._check(self.lib.vhp_timing_collect(self.h, _ptr(buf), cap, C.byref(n)))
        return buf[: n.value].copy()

    def probe_stores(self, d_ptr, n_bytes):
        """(whole-line TB/s, split-line TB/s) of the memory behind a device buffer (vhp_probe_stores; overwrites it)."""
        a, b = C.c_float(0), C.c_float(0)
        self._check(self.lib.vhp_probe_stores(self.h, C.c_void_p(d_ptr), int(n_bytes), C.byref(a), C.byref(b)))
        return a.value, b.value

    def alloc_output(self, n_bytes, max_candidates=24):
        """(device pointer, whole-line TB/s, split-line TB/s, allocations tried): a result buffer placed by the library (vhp_alloc_output)."""
        p, a, b, n = C.c_void_p(), C.c_float(0), C.c_float(0), C.c_int(0)
        self._check(self.lib.vhp_alloc_output(self.h, int(n_bytes), int(max_candidates), C.byref(p), C.byref(a), C.byref(b), C.byref(n)))
        return p.value, a.value, b.value, n.value

    def alloc_output_cost(self):
        """(wall-clock ms, peak bytes held) of the last alloc_output of this context (vhp_alloc_output_cost)."""
        ms, peak = C.c_double(0), C.c_ulonglong(0)
        self._check(self.lib.vhp_alloc_output_cost(self.h, C.byref(ms), C.byref(peak)))
        return ms.value, peak.value

    def free_output(self, d_ptr):
        self._check(self.lib.vhp_free_output(self.h, C.c_void_p(d_ptr)))

    def last_elapsed_ms(self):
        ms = C.c_float(0)
        self._check(self.lib.vhp_last_elapsed_ms(self.h, C.byref(ms)))
        return ms.value

    def planner_solve(self, start, end, threshold, max_iter):
        n = self.nx * self.ny
        came = np.empty((self.ny, self.nx), np.uint64)
        vg = np.empty((self.ny, self.nx), np.float64)
        vl = np.empty((self.ny, self.nx), np.float64)
        piv = np.zeros((int(max_iter) + 2, 2), np.int32)
        npiv = C.c_uint32(0)
        rc = self.lib.vhp_planner_solve(self.h, start[0], start[1], end[0], end[1], float(threshold), int(max_iter),
                                        _ptr(came), _ptr(vg), _ptr(vl), _ptr(piv), C.byref(npiv))
        if rc in (VHP_ERR_HIP, VHP_ERR_NO_MAP, VHP_ERR_ARG, VHP_ERR_TOO_LARGE):
            self._check(rc)
        return dict(status=rc, came_from=came, vis_global=vg, vis_local=vl, pivots=piv[: npiv.value + 1].copy(),
                    n_pivots=npiv.value)

    def planner_solve_speculative(self, start, end, threshold, max_iter, k=4, mode=0, outputs=True):
        """vhp_planner_solve_speculative: mode 0 = exact (same outputs as planner_solve), mode 1 = fast (commits all k).
        Adds stats: cache hits, sweeping iterations, fields swept."""
        came = np.empty((self.ny, self.nx), np.uint64) if outputs else None
        vg = np.empty((self.ny, self.nx), np.float64) if outputs else None
        vl = np.empty((self.ny, self.nx), np.float64) if outputs else None
        piv = np.zeros((int(max_iter) + 2 + 8, 2), np.int32)
        npiv = C.c_uint32(0)
        st = np.zeros(3, np.int32)
        rc = self.lib.vhp_planner_solve_speculative(self.h, start[0], start[1], end[0], end[1], float(threshold), int(max_iter), int(k), int(mode),
                                                    _ptr(came) if outputs else None, _ptr(vg) if outputs else None, _ptr(vl) if outputs else None,
                                                    _ptr(piv), C.byref(npiv), _ptr(st))
        if rc in (VHP_ERR_HIP, VHP_ERR_NO_MAP, VHP_ERR_ARG, VHP_ERR_TOO_LARGE):
            self._check(rc)
        return dict(status=rc, came_from=came, vis_global=vg, vis_local=vl, pivots=piv[: npiv.value + 1].copy(), n_pivots=npiv.value,
                    hits=int(st[0]), sweeps=int(st[1]), fields_swept=int(st[2]))

    def planner_solve_device(self, start, end, threshold, max_iter):
        """Planner solve with the results left on the device.  Returns (status, n_pivots, dict of raw device pointers:
        labels uint32 [ny, nx] (0xFFFFFFFF = unlabelled), vis_global / vis_local float64 [ny, nx], pivots int32 [n_pivots+1, 2])."""
        npiv = C.c_uint32(0)
        rc = self.lib.vhp_planner_solve_device(self.h, start[0], start[1], end[0], end[1], float(threshold), int(max_iter), C.byref(npiv))
        if rc in (VHP_ERR_HIP, VHP_ERR_NO_MAP, VHP_ERR_ARG, VHP_ERR_TOO_LARGE):
            self._check(rc)
        ptrs = {}
        if rc in (VHP_OK, VHP_ERR_MAX_ITER):
            p = [C.c_void_p() for _ in range(4)]
            self._check(self.lib.vhp_planner_results_device(self.h, *[C.byref(q) for q in p]))
            ptrs = dict(labels=p[0].value, vis_global=p[1].value, vis_local=p[2].value, pivots=p[3].value)
        return rc, npiv.value, ptrs

    def sweep_batch_variant(self, sources, alpha=1.0, fac=1.0):
        """MATLAB-flavoured sweep (getAccessibilityMap.m): fields [n, ny, nx] float64."""
        src = np.ascontiguousarray(sources, np.int32).reshape(-1, 2)
        out = np.empty((len(src), self.ny, self.nx), np.float64)
        self._check(self.lib.vhp_sweep_batch_variant(self.h, _ptr(src), len(src), float(alpha), float(fac), _ptr(out)))
        return out

    def sweep_batch_offset(self, sources, offset):
        """computeVisibility() with the reference's `offset` local exposed (include/vhp.h): fields [n, ny, nx] float64."""
        src = np.ascontiguousarray(sources, np.int32).reshape(-1, 2)
        out = np.empty((len(src), self.ny, self.nx), np.float64)
        self._check(self.lib.vhp_sweep_batch_offset(self.h, _ptr(src), len(src), float(offset), _ptr(out)))
        return out

    def planner_solve_variant(self, start, end, threshold, alpha, max_iter):
        lab = np.empty((self.ny, self.nx), np.uint64)
        uni = np.empty((self.ny, self.nx), np.float64)
        loc = np.empty((self.ny, self.nx), np.float64)
        way = np.zeros((int(max_iter) + 3, 2), np.int32)
        n = C.c_uint32(0)
        rc = self.lib.vhp_planner_solve_variant(self.h, start[0], start[1], end[0], end[1], float(threshold), float(alpha), int(max_iter),
                                                _ptr(lab), _ptr(uni), _ptr(loc), _ptr(way), C.byref(n))
        if rc in (VHP_ERR_HIP, VHP_ERR_NO_MAP, VHP_ERR_ARG, VHP_ERR_TOO_LARGE):
            self._check(rc)
        return dict(status=rc, label=lab, map_builder=uni, local=loc, waypoints=way[: n.value].copy())

    def raycast_all(self, sx, sy):
        out = np.empty((self.ny, self.nx), np.float64)
        self._check(self.lib.vhp_raycast_all(self.h, int(sx), int(sy), _ptr(out)))
        return out

    def reconstruct_path(self, came_from, pivots, end):
        return reconstruct_path(came_from, pivots, end)


def reconstruct_path(came_from, pivots, end):
    """vhp_reconstruct_path (host-only: needs no context and no GPU).

    pivots: [n_pivots + 1, 2] as planner_solve returns them (the last entry is `end`).  Returns (length, path)."""
    lib = load_library()
    ny, nx = came_from.shape
    piv = np.ascontiguousarray(pivots, np.int32).reshape(-1, 2)
    n_pivots = len(piv) - 1
    cap = n_pivots + 3
    path = np.zeros((cap, 2), np.int32)
    n = C.c_uint32(0)
    d = C.c_double(0)
    came = np.ascontiguousarray(came_from, np.uint64)
    rc = lib.vhp_reconstruct_path(_ptr(came), _ptr(piv), n_pivots, nx, ny, end[0], end[1], _ptr(path), cap,
                                  C.byref(n), C.byref(d))
    if rc != VHP_OK:
        raise VhpError(rc, "vhp_reconstruct_path: inconsistent came_from / pivots (or a path longer than %d points)" % cap)
    return d.value, path[: n.value].copy()


def version():
    return load_library().vhp_version().decode()
