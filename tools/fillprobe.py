"""Achievable HBM write rate on this GPU: torch fill_ and copy_ of a C3-sized field batch. Diagnostic only."""
import torch, time
x = torch.empty((256, 1000, 1000), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for name, fn, nbytes in [("fill_", lambda: x.fill_(1.0), x.numel() * 8), ("zero_", lambda: x.zero_(), x.numel() * 8),
                         ("copy_ (r+w)", lambda: y.copy_(x), 2 * x.numel() * 8)]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("%-12s %.3f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
