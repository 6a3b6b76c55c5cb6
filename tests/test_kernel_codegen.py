"""Code-generation invariants of the two dataflow kernels (pool sweep, latency sweep), checked on the gfx950 assembly.

Their cross-wavefront protocols (progress words, ring headers, descriptors) are ordered by the LDS executing one
wavefront's DS instructions in issue order, with compiler barriers only.  That holds as long as every LDS access IS a DS
instruction: a pointer that loses its address space compiles to FLAT instructions, which are not ordered with the DS
ones (and count on vmcnt) -- silently, and the CPU simulator would still pass.  Scratch traffic in these kernels means a
spilled value is reloaded behind vmcnt, i.e. behind every store in flight.  hipcc cross-compiles without a GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visibility-heuristic-path-planner_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


_COMPILED = {}   # source file -> (assembly, the compiler's resource-usage remarks): one compilation per file and test session


def _compile(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    if src not in _COMPILED:
        out = str(tmp_path / (src + ".s"))
        p = subprocess.run([HIPCC, "-std=c++17", "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                            "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src), "-Rpass-analysis=kernel-resource-usage"],
                           stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, check=True)
        _COMPILED[src] = (open(out).read(), p.stderr)
    return _COMPILED[src]


def _asm(src, tmp_path):
    return _compile(src, tmp_path)[0]


def _kernels(asm, name):
    """{mangled name: body} of the kernels whose name contains `name`"""
    out = {}
    # (up to the end of the function, not to the first s_endpgm: an early return may have one of its own)
    for m in re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:" % name, asm, re.S | re.M):
        out[m.group(1)] = m.group(2)
    return out


@pytest.mark.parametrize("src,kernel", [("vhp_pool.hip", "vhp_pool_sweep"), ("vhp_lat.hip", "vhp_lat_sweep")])
def test_no_flat_and_no_scratch_instructions(tmp_path, src, kernel):
    asm = _asm(src, tmp_path)
    ks = _kernels(asm, kernel)
    # fp64 and fp32, each also in its build for the other widths (pool sweep: not a multiple of 8; latency sweep: odd); the latency sweep
    # all of that twice: for launches with one workgroup per unit and for those whose bands read across workgroups
    want = 8 if kernel == "vhp_lat_sweep" else 4
    assert len(ks) == want, "%d instantiations of %s expected, found %r" % (want, kernel, list(ks))
    for name, body in ks.items():
        flat = re.findall(r"^\s+flat_\w+", body, re.M)
        scratch = re.findall(r"^\s+scratch_\w+", body, re.M)
        assert not flat, "%s: FLAT instructions %r" % (name, sorted(set(flat)))
        assert not scratch, "%s: scratch instructions %r" % (name, sorted(set(scratch)))
        assert re.search(r"^\s+ds_write", body, re.M) and re.search(r"^\s+global_store_dwordx[24]", body, re.M)
    for name in ks:
        m = re.search(r"\.name:\s*%s\n\s*\.private_segment_fixed_size:\s*(\d+)" % re.escape(name), asm)
        assert m and int(m.group(1)) == 0, "%s uses scratch memory" % name


def test_latency_sweep_register_budget(tmp_path):
    """The latency sweep's workgroup is sixteen wavefronts (a sweeper and a storer per band in flight: csrc/vhp_band.hpp): four per
    SIMD, i.e. 128 vector registers each, with no scratch -- and its spilled scalars stay out of the window loops and few (the sweep
    in strips of rows, until round 6, spilled 559-677 of them at 256 vector registers and two wavefronts per SIMD)."""
    asm, remarks = _compile("vhp_lat.hip", tmp_path)
    blocks = re.split(r"remark: Function Name: ", remarks)[1:]
    seen = 0
    for blk in blocks:
        name = blk.split()[0]
        if "vhp_lat_sweep" not in name:
            continue
        seen += 1
        get = lambda key: int(re.search(key + r":\s*(\d+)", blk).group(1))
        assert get(r"\bVGPRs") <= 128, (name, get(r"\bVGPRs"))
        assert get(r"ScratchSize \[bytes/lane\]") == 0, name
        assert get(r"VGPRs Spill") == 0, name
        assert get(r"Occupancy \[waves/SIMD\]") >= 4, name
        # the build for launches with one workgroup per unit (C2, C4, the small batches: <..., false>): 170-202 at the end of round 6;
        # the build whose bands read across workgroups (sides above 1024: <..., true>) holds that protocol on top: 420-462
        multi = "Lb1EEEvNS0_7LatArgs" in name
        assert get(r"SGPRs Spill") <= (520 if multi else 240), (name, get(r"SGPRs Spill"))
    assert seen == 8, seen
    # ... and (next to) none of the spilled scalars is reloaded inside a window's sixteen steps (the blocks that hold the arithmetic)
    for name, body in _kernels(asm, "vhp_lat_sweep").items():
        for blk in re.split(r"^\.LBB\d+_\d+:", body, flags=re.M):
            fp64 = len(re.findall(r"v_(?:fma|mul|add|fmac)_f64", blk))
            if fp64 >= 90:   # (~300 instructions: a window's sixteen steps)
                n = len(re.findall(r"v_(?:readlane|writelane)_b32", blk))
                assert n <= 2, "%s: %d spilled scalars moved inside a window's steps" % (name, n)
