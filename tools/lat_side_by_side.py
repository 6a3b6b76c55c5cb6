"""Two contexts on two streams of one device, latency-sweep launches of several workgroups per unit side by side: time and equality with the
launches alone (a launch sized for the whole device gets part of it).  Diagnostic only.  usage: lat_side_by_side.py <side> <sources per context>"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
side, n = int(sys.argv[1]), int(sys.argv[2])
occ = synth.random_rect_map(side, side, 50, 80, 400, 80, 400, seed=1)
srcs = [synth.free_sources(occ, n, seed=7), synth.free_sources(occ, n, seed=8)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs, outs, d_src = [], [], []
for k in range(2):
    c = mod.Context(0); c.set_stream(streams[k].cuda_stream); c.set_map(occ); c.set_option("kernel", 4)
    ctxs.append(c)
    outs.append(torch.full((n, side, side), float("nan"), dtype=torch.float64, device="cuda"))
    d_src.append(torch.from_numpy(np.ascontiguousarray(srcs[k], np.int32)).cuda())
torch.cuda.synchronize()
# reference: one after the other
refs = []
for k in range(2):
    ctxs[k].sweep_batch_device(d_src[k].data_ptr(), n, outs[k].data_ptr()); torch.cuda.synchronize(); refs.append(outs[k].clone()); outs[k].fill_(float("nan"))
torch.cuda.synchronize()
import time
t0 = time.time()
for it in range(10):
    for k in range(2):
        ctxs[k].sweep_batch_device(d_src[k].data_ptr(), n, outs[k].data_ptr())
torch.cuda.synchronize()
print("side %d, %d sources per context, two contexts on two streams, 10 launches each side by side: %.1f ms; equal to the launches alone: %s %s" % (
    side, n, (time.time() - t0) * 1e3, bool(torch.equal(outs[0], refs[0])), bool(torch.equal(outs[1], refs[1]))))
