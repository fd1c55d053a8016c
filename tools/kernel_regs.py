#!/usr/bin/env python3
"""Register / spill / LDS summary of the kernels in a -save-temps .s file (or compile one .hip first):
    python tools/kernel_regs.py banet_amd/csrc/gather128s.hip [substring]"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if src.endswith(".hip"):
    d = tempfile.mkdtemp(prefix="kregs_")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-I" + os.path.join(root, "banet_amd/lib"),
                    "-I" + os.path.join(root, "banet_amd/csrc")] + os.environ.get("EXTRA_HIPCC_FLAGS", "").split() +
                   ["-c", os.path.abspath(src), "-save-temps", "-o", os.path.join(d, "x.o")], cwd=d, check=True, stderr=subprocess.DEVNULL)
    src = [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f][0]
    print("#", src)
txt = open(src).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: *\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\." + k + r": *(\S+)", blk).group(1)
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if flt not in name:
        continue
    print("%-70s vgpr %3s agpr %3s spill v%-3s s%-3s scratch %4s lds %6s" % (name[:70], g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"),
                                                                       g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
