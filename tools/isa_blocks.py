#!/usr/bin/env python3
"""Instruction mix of the big basic blocks of every kernel in a gfx950 .s file
(hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S x.hip -o x.s)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for m in re.finditer(r"^(_Z\w+):.*$", s, flags=re.M):
    name = m.group(1)
    i = m.end()
    j = s.index(".Lfunc_end", i)
    blocks, cur = {"entry": []}, "entry"
    for l in s[i:j].split("\n"):
        l = l.strip()
        if not l or l.startswith(";") or (l.startswith(".") and not l.startswith(".LBB")):
            continue
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            cur = mm.group(1)
            blocks[cur] = []
            continue
        blocks[cur].append(l.split()[0])
    for k, v in blocks.items():
        if len(v) > thr:
            c = Counter()
            for op in v:
                if op.startswith("v_pk"): c["v_pk"] += 1
                elif op.startswith("v_mfma"): c["mfma"] += 1
                elif op.startswith("v_"): c["valu"] += 1
                elif op.startswith("s_"): c["salu"] += 1
                elif op.startswith(("global_load", "buffer_load")): c["vmem_ld"] += 1
                elif op.startswith(("global_store", "buffer_store")): c["vmem_st"] += 1
                elif op.startswith("ds_"): c["lds"] += 1
                elif op.startswith("scratch"): c["scratch"] += 1
                else: c[op] += 1
            print(name[:60], k, len(v), dict(c))
