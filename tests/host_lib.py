"""ctypes binding of libvhp_host.so (host-side surface: config parser, environment, writers)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "visibility-heuristic-path-planner_amd")
HOST_LIB = os.path.join(PKG, "libvhp_host.so")
CLI = os.path.join(PKG, "vhp")
REF_PARSER = os.path.join(ROOT, "oracle", "_ref", "libref_parser.so")


class HostConfig(C.Structure):
    _fields_ = [("mode", C.c_int32)] + [(n, C.c_uint64) for n in (
        "ncols", "nrows", "nb_of_obstacles", "minWidth", "maxWidth", "minHeight", "maxHeight")] + [
        ("randomSeed", C.c_int32), ("seedValue", C.c_int32), ("imagePath", C.c_char * 1024),
        ("start_x", C.c_int32), ("start_y", C.c_int32), ("end_x", C.c_int32), ("end_y", C.c_int32),
        ("max_iter", C.c_uint64), ("visibilityThreshold", C.c_double), ("lightStrength", C.c_float)] + [
        (n, C.c_int32) for n in ("timer", "saveResults", "saveLocalVisibility", "saveCameFrom", "saveLightSources",
                                 "saveGlobalVisibility", "saveVisibilityField", "silent", "ballRadius")]

    def as_dict(self):
        return {n: (getattr(self, n).decode() if n == "imagePath" else getattr(self, n)) for n, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "host"), HOST_LIB])


def load():
    if not os.path.exists(HOST_LIB):
        build()
    lib = C.CDLL(HOST_LIB)
    lib.vhp_host_parse_config.argtypes = [C.c_char_p, C.POINTER(HostConfig)]
    lib.vhp_host_generate_env.argtypes = [C.c_uint64] * 7 + [C.c_int, C.c_void_p]
    lib.vhp_host_load_image.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.vhp_host_write_matrix_f64.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.vhp_host_write_matrix_u64.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.vhp_host_save_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    return lib


def parse_config(lib, path):
    c = HostConfig()
    ok = lib.vhp_host_parse_config(path.encode(), C.byref(c))
    return bool(ok), c.as_dict()


def parse_config_reference(path):
    """The REFERENCE's own parser (oracle/_ref, built from /root/reference); None if not built."""
    if not os.path.exists(REF_PARSER):
        return None
    lib = C.CDLL(REF_PARSER)
    lib.ref_parse_config.argtypes = [C.c_char_p, C.POINTER(HostConfig)]
    c = HostConfig()
    ok = lib.ref_parse_config(path.encode(), C.byref(c))
    return bool(ok), c.as_dict()


def generate_env(lib, nx, ny, nb, min_w, max_w, min_h, max_h, seed):
    occ = np.zeros((ny, nx), np.uint8)
    lib.vhp_host_generate_env(nx, ny, nb, min_w, max_w, min_h, max_h, seed, occ.ctypes.data_as(C.c_void_p))
    return occ


def load_image(lib, path):
    nx, ny = C.c_int32(0), C.c_int32(0)
    if lib.vhp_host_load_image(path.encode(), None, 0, C.byref(nx), C.byref(ny)) != 0:
        return None
    occ = np.zeros((ny.value, nx.value), np.uint8)
    rc = lib.vhp_host_load_image(path.encode(), occ.ctypes.data_as(C.c_void_p), occ.size, C.byref(nx), C.byref(ny))
    assert rc == 0
    return occ


def format_matrix(a, flip=False):
    """What the reference's writers produce (solver.cpp:1042-1057): default ostream formatting
    (== C's %g for doubles), one space after every token, newline per row."""
    rows = a[::-1] if flip else a
    if a.dtype.kind == "f":
        return "".join("".join("%g " % v for v in r) + "\n" for r in rows)
    return "".join("".join("%d " % v for v in r) + "\n" for r in rows)
