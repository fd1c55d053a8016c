#!/usr/bin/env python3
"""Build-time check of ba_gather128s_kernel's register contract: the kernel reserves v224 .. v255 for source
features that are in flight behind counted s_waitcnt (named only inside inline asm) and limits the compiler to 224 registers with
amdgpu_num_vgpr.  A toolchain change that let the COMPILER allocate v224+ (or spill in the hot variants) would break it silently.
This compiles gather128s.hip to gfx950 assembly and asserts, for every ba_gather128s_kernel instantiation:
  * no instruction outside the inline-asm blocks (;;#ASMSTART .. ;;#ASMEND) names a VGPR >= 224;
  * the default variants (16-row segments, K <= 128, two-frame and frame-parallel K = 0) have no VGPR spills.
    python tools/check_strip_regs.py      (exit code 0 = ok)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_SPILL = ("ILi0ELi4ELb0ELi0E", "ILi1ELi4ELb0ELi0E", "ILi0ELi2ELb0ELi0E", "ILi1ELi2ELb0ELi0E", "ILi0ELi4ELb1ELi0E")     # <KV4, NCH, FP, OPT = 0> manglings


def main():
    d = tempfile.mkdtemp(prefix="stripregs_")
    out = os.path.join(d, "g.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "--cuda-device-only", "-S",
                    "-I" + os.path.join(ROOT, "banet_amd", "lib"), os.path.join(ROOT, "banet_amd", "csrc", "gather128s.hip"), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    bad, nk = [], 0
    for m in re.finditer(r"^(_ZN5banet20ba_gather128s_kernel\w+):", txt, re.M):
        name = m.group(1)
        body = txt[m.end():txt.index(".Lfunc_end", m.end())]
        nk += 1
        inasm = False
        for ln in body.split("\n"):
            t = ln.strip()
            if t.startswith(";APP") or t.startswith(";;#ASMSTART"):
                inasm = True
                continue
            if t.startswith(";NO_APP") or t.startswith(";;#ASMEND"):
                inasm = False
                continue
            if inasm or not t or t[0] in ";.":
                continue
            code = t.split(";")[0]
            for r in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", code):
                if int(r.group(1) or r.group(3)) >= 224:
                    bad.append("%s: compiler-emitted instruction names a reserved register: %s" % (name, code))
                    break
    for blk in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: *\d+", txt, re.S):
        b = blk.group(0)
        name = re.search(r"\.name: *(\S+)", b).group(1)
        if "ba_gather128s_kernel" in name and any(k in name for k in NO_SPILL):
            sp = int(re.search(r"\.vgpr_spill_count: *(\d+)", b).group(1))
            if sp:
                bad.append("%s: %d VGPR spills in a default variant" % (name, sp))
    for b in bad[:20]:
        print(b)
    print("check_strip_regs: %d ba_gather128s_kernel instantiations, %d violations" % (nk, len(bad)))
    return 1 if bad or nk == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
