#!/usr/bin/env python3
"""Counts instruction classes inside the loops of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only):
   tools/isa_loop_stats.py LISTING.s KERNEL_SUBSTRING
Loops = backward branches; printed largest first with the counts of v_readlane / v_writelane (SGPR spill traffic), s_nop, vector, scalar,
LDS and memory instructions between the branch target and the branch."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    t = open(path).read()
    i = t.index(key + ":") if (key + ":") in t else t.index(key)
    k = t.index(".Lfunc_end", i)
    lines = [l.strip() for l in t[i:k].splitlines()]
    label_at = {}
    ins = []          # (index, text)
    for l in lines:
        if not l or l.startswith((";", "//")):
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        if l.startswith("."):
            continue
        ins.append(l)
    loops = []
    for n, l in enumerate(ins):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB[0-9_]+)|^s_branch\s+(\.LBB[0-9_]+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in label_at and label_at[tgt] <= n:
                loops.append((n - label_at[tgt], label_at[tgt], n, tgt))
    loops.sort(reverse=True)
    print(f"{key[:60]}: {len(ins)} instructions")
    for size, a, b, tgt in loops[:6]:
        c = collections.Counter(x.split()[0] for x in ins[a:b + 1])
        v = sum(n for k2, n in c.items() if k2.startswith("v_"))
        s = sum(n for k2, n in c.items() if k2.startswith("s_"))
        print(f"  loop {tgt:>12} {size:6d} instr: v_readlane {c['v_readlane_b32']:4d} v_writelane {c['v_writelane_b32']:4d} s_nop {c['s_nop']:4d} vector {v:5d} scalar {s:5d} "
              f"ds {sum(n for k2, n in c.items() if k2.startswith('ds_')):4d} mem {sum(n for k2, n in c.items() if k2.startswith(('buffer_', 'global_', 'scratch_', 'flat_'))):4d} "
              f"s_waitcnt {c['s_waitcnt']:4d} branches {sum(n for k2, n in c.items() if k2.startswith(('s_cbranch', 's_branch'))):4d}")


if __name__ == "__main__":
    main()
