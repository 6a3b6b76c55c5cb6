"""How fast does this box take 2 GB of plain stores right now?  torch fill_ / copy_ on the bench-sized buffer, and the
sweep's own launch on the same buffer.  Diagnostic only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
n = 256
out = torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda")
src = torch.ones((n, 1000, 1000), dtype=torch.float64, device="cuda")
for name, fn in (("fill_", lambda: out.fill_(1.0)), ("copy_", lambda: out.copy_(src)), ("zero_", lambda: out.zero_())):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s: %.3f ms per 2.048 GB  (%.0f GB/s written)" % (name, ms, 2.048 / ms * 1e3))
