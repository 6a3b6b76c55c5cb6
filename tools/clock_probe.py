"""Does a single-source launch run at the chip's full clock?  Times the latency sweep alone and beside a heavy kernel on another
stream (which keeps the clocks up), and prints what rocm-smi reports meanwhile.  Diagnostic only."""
import os, sys, subprocess, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
side = 1000
occ = np.ones((side, side), np.uint8)
c = mod.Context(0)
s1 = torch.cuda.Stream()
c.set_stream(s1.cuda_stream)
c.set_map(occ)
d_src = torch.from_numpy(np.array([[500, 500]], np.int32)).cuda()
out = torch.empty((1, side, side), dtype=torch.float64, device="cuda")
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in r.splitlines() if "sclk" in l or "mclk" in l or "fclk" in l)[:300]
    except Exception as e:
        return repr(e)
def run(n, heavy):
    torch.cuda.synchronize()
    c.timing(True)
    for i in range(n):
        if heavy and i % 4 == 0:
            torch.mm(a, a)
        with torch.cuda.stream(s1):
            c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
    torch.cuda.synchronize()
    t = np.array(c.timing_collect(4096))
    c.timing(False)
    return np.median(t) * 1e3, t.min() * 1e3
for heavy in (False, True, False):
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("smi", smi()))
    th.start()
    med, mn = run(3000 if not heavy else 400, heavy)
    th.join()
    print("heavy neighbour %s: latency sweep median %.1f us, min %.1f us;  %s" % (heavy, med, mn, res.get("smi")))
