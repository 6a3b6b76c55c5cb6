"""What an unprivileged process can still choose about the output allocation: the C3 launch timed on
outputs from hipMalloc (as many 2 GB buffers as fit: the map of fast and slow allocations), hipExtMallocWithFlags (default,
fine-grained, uncached, contiguous), the virtual-memory API (hipMemCreate at minimum / recommended granularity, one handle or
many), hipMallocAsync, and hipMalloc again after everything has been freed.  Diagnostic only.
usage: alloc_map.py [max hipMalloc buffers] [kernel]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
max_bufs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, side = 256, 1000
BYTES = n * side * side * 8
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
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(vp), sz, C.c_uint]
hip.hipMemGetInfo.argtypes = [C.POINTER(sz), C.POINTER(sz)]
hip.hipMemsetAsync.argtypes = [vp, C.c_int, sz, vp]
def run(ptr, reps=8, k=kernel):
    c.set_option("kernel", k)
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
def fill(ptr):
    s = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hip.hipMemsetAsync(ptr, 0, BYTES, s)
    e0.record()
    for _ in range(4):
        hip.hipMemsetAsync(ptr, 0, BYTES, s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4
def report(name, ptr):
    print("%-44s pool %.3f ms   front %.3f ms   memset %.3f ms (%.2f TB/s)" % (name, run(ptr, 8, 3), run(ptr, 6, 1), fill(ptr), BYTES / fill(ptr) / 1e9), flush=True)
free, total = sz(), sz()
hip.hipMemGetInfo(C.byref(free), C.byref(total))
print("free %.1f GB of %.1f GB" % (free.value / 1e9, total.value / 1e9), flush=True)
# ---- A: the map of hipMalloc allocations
nb = min(max_bufs, int((free.value - 24e9) // BYTES))
bufs = []
for i in range(nb):
    p = vp()
    if hip.hipMalloc(C.byref(p), BYTES) != 0:
        break
    bufs.append(p.value)
tA = [run(p, 5) for p in bufs]
print("A. %d hipMalloc buffers in allocation order, launch ms (kernel %d):" % (len(bufs), kernel))
for i in range(0, len(bufs), 10):
    print("   %3d: " % i + " ".join("%.3f" % t for t in tA[i:i + 10]), flush=True)
print("   virtual addresses (GB): " + " ".join("%.1f" % (p / 2**30) for p in bufs[:12]), "...", flush=True)
tA2 = [run(p, 5) for p in bufs]
print("   again: max |difference| %.3f ms; fast (< 0.62 ms) %d of %d" % (max(abs(a - b) for a, b in zip(tA, tA2)), sum(t < 0.62 for t in tA), len(tA)), flush=True)
if bufs:
    k_slow, k_fast = int(np.argmax(tA)), int(np.argmin(tA))
    report("A. slowest hipMalloc buffer (%d)" % k_slow, bufs[k_slow])
    report("A. fastest hipMalloc buffer (%d)" % k_fast, bufs[k_fast])
for p in bufs:
    hip.hipFree(p)
# ---- B: after freeing everything
for rep in range(3):
    p = vp(); assert hip.hipMalloc(C.byref(p), BYTES) == 0
    report("B. hipMalloc after free-all #%d" % rep, p.value)
    hip.hipFree(p)
# ---- C: hipExtMallocWithFlags
for name, flag in (("default", 0), ("finegrained", 1), ("uncached", 3), ("contiguous", 4)):
    ps = []
    for rep in range(3):
        p = vp()
        rc = hip.hipExtMallocWithFlags(C.byref(p), BYTES, flag)
        if rc != 0:
            print("C. hipExtMallocWithFlags(%s): error %d" % (name, rc), flush=True)
            break
        ps.append(p.value)
        report("C. hipExtMallocWithFlags(%s) #%d" % (name, rep), p.value)
    for q in ps:
        hip.hipFree(q)
# ---- D: virtual-memory API
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle", C.c_int), ("loc_type", C.c_int), ("loc_id", C.c_int), ("win32", vp),
                ("comp", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]
class Access(C.Structure):
    _fields_ = [("loc_type", C.c_int), ("loc_id", C.c_int), ("flags", C.c_int)]
prop = Prop(1, 0, 1, 0, None, 0, 0, 0)
hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(sz), C.POINTER(Prop), C.c_int]
hip.hipMemAddressReserve.argtypes = [C.POINTER(vp), sz, sz, vp, C.c_ulonglong]
hip.hipMemCreate.argtypes = [C.POINTER(vp), sz, C.POINTER(Prop), C.c_ulonglong]
hip.hipMemMap.argtypes = [vp, sz, sz, vp, C.c_ulonglong]
hip.hipMemSetAccess.argtypes = [vp, sz, C.POINTER(Access), sz]
hip.hipMemUnmap.argtypes = [vp, sz]
hip.hipMemRelease.argtypes = [vp]
hip.hipMemAddressFree.argtypes = [vp, sz]
gmin, grec = sz(), sz()
r0 = hip.hipMemGetAllocationGranularity(C.byref(gmin), C.byref(prop), 0)
r1 = hip.hipMemGetAllocationGranularity(C.byref(grec), C.byref(prop), 1)
print("D. granularity: minimum %d (rc %d), recommended %d (rc %d)" % (gmin.value, r0, grec.value, r1), flush=True)
def vmm(chunk, align, label):
    size = (BYTES + chunk - 1) // chunk * chunk
    va = vp()
    rc = hip.hipMemAddressReserve(C.byref(va), size, align, None, 0)
    if rc != 0:
        print("D. %s: reserve error %d" % (label, rc)); return
    handles = []
    for off in range(0, size, chunk):
        h = vp()
        rc = hip.hipMemCreate(C.byref(h), chunk, C.byref(prop), 0)
        if rc != 0:
            print("D. %s: create error %d at %d" % (label, rc, off)); return
        rc = hip.hipMemMap(va.value + off, chunk, 0, h, 0)
        if rc != 0:
            print("D. %s: map error %d" % (label, rc)); return
        handles.append(h)
    acc = Access(1, 0, 3)
    rc = hip.hipMemSetAccess(va, size, C.byref(acc), 1)
    if rc != 0:
        print("D. %s: set-access error %d" % (label, rc)); return
    report("D. " + label + " va %% 1 GiB = %d MiB" % ((va.value % 2**30) >> 20), va.value)
    hip.hipMemUnmap(va, size)
    for h in handles:
        hip.hipMemRelease(h)
    hip.hipMemAddressFree(va, size)
G = max(grec.value, 1 << 21)
for rep in range(2):
    vmm((BYTES + G - 1) // G * G, 0, "VMM one handle, #%d" % rep)
    vmm((BYTES + G - 1) // G * G, 1 << 30, "VMM one handle, VA aligned to 1 GiB #%d" % rep)
    vmm(1 << 30, 1 << 30, "VMM handles of 1 GiB #%d" % rep)
    vmm(64 << 20, 1 << 30, "VMM handles of 64 MiB #%d" % rep)
    vmm(G, 1 << 21, "VMM handles of %d KiB #%d" % (G >> 10, rep))
# ---- E: stream-ordered allocator, torch
hip.hipMallocAsync.argtypes = [C.POINTER(vp), sz, vp]
p = vp()
rc = hip.hipMallocAsync(C.byref(p), BYTES, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
if rc == 0:
    report("E. hipMallocAsync", p.value)
t = torch.empty((n, side, side), dtype=torch.float64, device="cuda")
report("E. torch.empty", t.data_ptr())
