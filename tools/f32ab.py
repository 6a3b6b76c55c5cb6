import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
n = 256
occ = synth.random_rect_map(1000, 1000, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
ctx = vhp_amd.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_map(occ)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
out = torch.empty((n, 1000, 1000), dtype=torch.float32, device="cuda")
ref = None
for rep in range(4):
    for cfg in ["2 8", "1 8"]:
        os.environ["VHP_R"], os.environ["VHP_W"] = cfg.split()
        for _ in range(3): ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F32)
        torch.cuda.synchronize(); ctx.timing(True)
        for _ in range(25): ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F32)
        torch.cuda.synchronize(); k = ctx.timing_collect(25); ctx.timing(False)
        if ref is None: ref = out.clone()
        print("f32 shape", cfg, "median %.4f ms" % np.median(k), "same bytes as first:", bool(torch.equal(ref, out)), flush=True)
