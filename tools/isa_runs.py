#!/usr/bin/env python3
"""Instruction-class run lengths of the hottest basic block (most MFMAs) of one kernel in a -save-temps .s file:
shows whether the compiler interleaved matrix, vector and memory instructions or clustered them.
usage: isa_runs.py file.s mangled_kernel_name"""
import re
import sys

t = open(sys.argv[1]).read()
i = t.index(sys.argv[2] + ":")
j = t.index(".end_amdhsa_kernel", i)
blocks, cur = [], []
for ln in t[i:j].split("\n"):
    l = ln.strip()
    if re.match(r"\.LBB\d+_\d+:", l):
        blocks.append(cur)
        cur = []
    elif l and not l.startswith((";", ".")):
        cur.append(l.split()[0])
blocks.append(cur)
ins = max(blocks, key=lambda b: sum(x.startswith("v_mfma") for x in b))


def cls(x):
    if x.startswith("v_mfma"):
        return "M"
    if x.startswith(("global_", "buffer_", "flat_")):
        return "L"
    if x.startswith("ds_"):
        return "D"
    if x.startswith("s_waitcnt"):
        return "W"
    if x.startswith("v_"):
        return "v"
    return "s"


runs, last, n = [], None, 0
for x in ins:
    c = cls(x)
    if c == last:
        n += 1
    else:
        if last:
            runs.append("%s%d" % (last, n))
        last, n = c, 1
runs.append("%s%d" % (last, n))
print(len(ins), "instructions")
print(" ".join(runs))
