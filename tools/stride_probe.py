"""Is the slow state of an output buffer an aliasing of the launch's regular strides (fields 8 000 000 B apart) in physical memory?
The C3 launch on hipMalloc'ed outputs with the fields packed and with padded field strides; VMM outputs of 2 MiB handles next
to hipMalloc outputs, alternating.  Diagnostic only.   usage: stride_probe.py [n buffers]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, side = 256, 1000
CELLS = side * side
PADS = [0, 16, 64, 256, 512, 512 + 16, 1024, 4096, 8192, 8192 + 512, 65536, 131072, 131072 + 8192 + 512]
BYTES = n * (CELLS + max(PADS)) * 8
occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
hip = C.CDLL("libamdhip64.so")
vp, sz = C.c_void_p, C.c_size_t
hip.hipMalloc.argtypes = [C.POINTER(vp), sz]
hip.hipFree.argtypes = [vp]
def run(ptr, pad=0, reps=6, k=3):
    c.set_option("kernel", k)
    c.set_option("field_stride", CELLS + pad if pad else 0)
    for _ in range(2):
        c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize()
    c.timing(True)
    for _ in range(reps):
        c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize()
    t = c.timing_collect(reps)
    c.timing(False)
    return float(np.median(t))
bufs = []
for i in range(nbuf):
    p = vp()
    if hip.hipMalloc(C.byref(p), BYTES) != 0:
        break
    bufs.append(p.value)
t0 = [run(p) for p in bufs]
print("packed fields, launch ms per hipMalloc buffer:", " ".join("%.3f" % t for t in t0), flush=True)
order = np.argsort(t0)
picks = list(order[-3:][::-1]) + list(order[:2])
print("pad (cells):        " + " ".join("%7d" % p for p in PADS))
for k in picks:
    print("buffer %2d (%.3f):  " % (k, t0[k]) + " ".join("%7.3f" % run(bufs[k], pad) for pad in PADS), flush=True)
print("front sweep (kernel 1):")
for k in picks[:2] + picks[-1:]:
    print("buffer %2d (%.3f):  " % (k, t0[k]) + " ".join("%7.3f" % run(bufs[k], pad, 5, 1) for pad in PADS), flush=True)
for p in bufs:
    hip.hipFree(p)
# ---- VMM outputs of 2 MiB handles next to hipMalloc outputs, alternating
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle", C.c_int), ("loc_type", C.c_int), ("loc_id", C.c_int), ("win32", vp),
                ("comp", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]
class Access(C.Structure):
    _fields_ = [("loc_type", C.c_int), ("loc_id", C.c_int), ("flags", C.c_int)]
prop = Prop(1, 0, 1, 0, None, 0, 0, 0)
hip.hipMemAddressReserve.argtypes = [C.POINTER(vp), sz, sz, vp, C.c_ulonglong]
hip.hipMemCreate.argtypes = [C.POINTER(vp), sz, C.POINTER(Prop), C.c_ulonglong]
hip.hipMemMap.argtypes = [vp, sz, sz, vp, C.c_ulonglong]
hip.hipMemSetAccess.argtypes = [vp, sz, C.POINTER(Access), sz]
def vmm(chunk):
    size = (n * CELLS * 8 + chunk - 1) // chunk * chunk
    va = vp()
    assert hip.hipMemAddressReserve(C.byref(va), size, 1 << 21, None, 0) == 0
    for off in range(0, size, chunk):
        h = vp()
        assert hip.hipMemCreate(C.byref(h), chunk, C.byref(prop), 0) == 0
        assert hip.hipMemMap(va.value + off, chunk, 0, h, 0) == 0
    acc = Access(1, 0, 3)
    assert hip.hipMemSetAccess(va, size, C.byref(acc), 1) == 0
    return va.value
res = {"hipMalloc": [], "vmm 2 MiB": [], "vmm 64 KiB": [], "vmm 32 MiB": []}
for rep in range(10):
    p = vp(); assert hip.hipMalloc(C.byref(p), n * CELLS * 8) == 0
    res["hipMalloc"].append(run(p.value))
    res["vmm 2 MiB"].append(run(vmm(1 << 21)))
    if rep < 4:
        res["vmm 64 KiB"].append(run(vmm(1 << 16)))
    res["vmm 32 MiB"].append(run(vmm(1 << 25)))
for k, v in res.items():
    print("%-12s launch ms: %s" % (k, " ".join("%.3f" % t for t in v)), flush=True)
