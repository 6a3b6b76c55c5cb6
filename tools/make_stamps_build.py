"""Diagnostic build: per-wavefront cycle stamps around the window kinds of the sweep kernel.
Writes exp/stamps.so (not part of the product)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "visibility-heuristic-path-planner_amd", "csrc")
B = "/tmp/expbuild"
os.makedirs(B, exist_ok=True)
s = open(os.path.join(CS, "vhp_sweep.hip.h")).read()
s = s.replace("struct UnitGeom {", "__device__ unsigned long long vhp_dbg[256];\nstruct UnitGeom {", 1)
def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)
rep('''          const bool steady = i >= j0 + S;
          if (p > 0) {''', '''          const bool steady = i >= j0 + S;
          unsigned long long tq = __builtin_amdgcn_s_memtime();
          if (p > 0) {''')
rep('''          const bool steady = all_cols && j >= i0 + S;
          if (p > 0) {''', '''          const bool steady = all_cols && j >= i0 + S;
          unsigned long long tq = __builtin_amdgcn_s_memtime();
          if (p > 0) {''')
rep('''            else fast_window(i, std::true_type(), std::false_type());
          }
          i += 8;''', '''            else fast_window(i, std::true_type(), std::false_type());
          }
          { const int o = steady ? 0 : 2; dacc[o] += __builtin_amdgcn_s_memtime() - tq; dacc[o + 1] += 1; }
          i += 8;''')
rep('''            else fast_window(j, std::true_type(), std::false_type());
          }
          j += 8;''', '''            else fast_window(j, std::true_type(), std::false_type());
          }
          { const int o = steady ? 0 : 2; dacc[o] += __builtin_amdgcn_s_memtime() - tq; dacc[o + 1] += 1; }
          j += 8;''')
rep('''          slow_step(i);
          i += 1;''', '''          unsigned long long tq = __builtin_amdgcn_s_memtime();
          slow_step(i);
          i += 1;
          dacc[4] += __builtin_amdgcn_s_memtime() - tq; dacc[5] += 1;''')
rep('''          slow_step(j);
          j += 1;''', '''          unsigned long long tq = __builtin_amdgcn_s_memtime();
          slow_step(j);
          j += 1;
          dacc[4] += __builtin_amdgcn_s_memtime() - tq; dacc[5] += 1;''')
rep("  for (int T = 0; T <= tmax; ++T) {\n    const int n = T - p;\n    if (strip_on && n >= nfirst && n <= nlast) {\n      int ilo, ihi;",
    "  unsigned long long dacc[8] = {0,0,0,0,0,0,0,0};\n  const int dbg_slot = p;\n  for (int T = 0; T <= tmax; ++T) {\n    const int n = T - p;\n    if (strip_on && n >= nfirst && n <= nlast) {\n      int ilo, ihi;")
rep("  for (int T = 0; T <= tmax; ++T) {\n    const int n = T - p - kYLag;\n    if (strip_on && n >= nfirst && n <= nlast) {\n      int jlo, jhi;",
    "  unsigned long long dacc[8] = {0,0,0,0,0,0,0,0};\n  const int dbg_slot = 8 + p;\n  for (int T = 0; T <= tmax; ++T) {\n    const int n = T - p - kYLag;\n    if (strip_on && n >= nfirst && n <= nlast) {\n      int jlo, jhi;")
rep("    __syncthreads();\n  }\n}",
    "    { unsigned long long tb = __builtin_amdgcn_s_memtime(); __syncthreads(); dacc[6] += __builtin_amdgcn_s_memtime() - tb; dacc[7] += 1; }\n  }\n  if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 8; ++k) vhp_dbg[dbg_slot * 8 + k] += dacc[k];\n}", 2)
open(os.path.join(B, "vhp_sweep.hip.h"), "w").write(s)
c = open(os.path.join(CS, "vhp_capi.hip")).read()
c = c.replace('}  // extern "C"', '''int vhp_debug_fetch(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(vhp::vhp_dbg), 256 * 8);
  if (reset) { unsigned long long z[256] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(vhp::vhp_dbg), z, 256 * 8); }
  return 0;
}
}  // extern "C"''')
open(os.path.join(B, "vhp_capi.hip"), "w").write(c)
for f in ("vhp_planner.hip.h", "vhp_queue.hip.h"):
    open(os.path.join(B, f), "w").write(open(os.path.join(CS, f)).read())
os.makedirs(os.path.join(ROOT, "exp"), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "--offload-arch=gfx950",
                       "-I" + os.path.join(ROOT, "include"), "-I.", "-shared", "-w", "-o", os.path.join(ROOT, "exp", "stamps.so"),
                       "vhp_capi.hip"] + sys.argv[1:], cwd=B)
print("built exp/stamps.so")
