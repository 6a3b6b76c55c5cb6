"""The basic blocks of a kernel in hipcc's assembly with the most fp64 arithmetic: instructions, fp64 operations, v_readlane /
v_writelane (what a spilled SGPR costs when it is used), scalar instructions, global stores, LDS instructions.  Answers
"are the register spills inside the window loops?" without a GPU.   usage: asm_blocks.py <file.s> <kernel-name substring> [n]"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
for m in re.finditer(r'^(_Z\w*%s\w*):' % re.escape(key), s, re.M):
    i = m.start(); j = s.index('s_endpgm', i)
    body = s[i:j].split('\n')
    ins = [l for l in body if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    print("%s: %d instructions, %d v_readlane, %d v_writelane, %d s_waitcnt" % (m.group(1), len(ins), sum('v_readlane' in l for l in ins), sum('v_writelane' in l for l in ins), sum('s_waitcnt' in l for l in ins)))
    blocks, cur = [], None
    for l in body:
        if re.match(r'^\.LBB\d+_\d+:', l):
            cur = [l.strip().split(':')[0] + (' (' + l.split(';')[1].strip() + ')' if ';' in l else ''), []]; blocks.append(cur)
        elif cur is not None and l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
            cur[1].append(l.strip())
    rows = [(b[0], len(b[1]), sum(('v_fma_f64' in x) or ('v_mul_f64' in x) or ('v_add_f64' in x) for x in b[1]), sum('v_readlane' in x for x in b[1]),
             sum('v_writelane' in x for x in b[1]), sum(x.startswith('s_') for x in b[1]), sum('global_store' in x for x in b[1]), sum(x.startswith('ds_') for x in b[1])) for b in blocks]
    rows.sort(key=lambda t: -t[2])
    print("  %-52s %6s %5s %8s %9s %6s %7s %4s" % ("block", "instrs", "fp64", "readlane", "writelane", "scalar", "gstores", "ds"))
    for t in rows[:n]:
        print("  %-52s %6d %5d %8d %9d %6d %7d %4d" % t)
    inloop = sum(t[3] + t[4] for t in rows[:n])
    print("  v_readlane + v_writelane inside these %d blocks: %d of %d in the kernel" % (n, inloop, sum(t[3] + t[4] for t in rows)))
