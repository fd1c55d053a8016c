#!/usr/bin/env python3
"""The figures DESIGN.md section 4.8 quotes from the SQ counters of the backward's large kernels (profiles/r06_run4_backward_sq_counters.txt,
collected by tools/gpu_r6_sq.sh; means over the 40 / 20 launches of four 32-window training steps, all five levels):
VALU-pipe busy, what a wave does with its life, resident waves per SIMD.     python tools/derive_backward_counters.py [file]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x 4 = cycles); GRBM_GUI_ACTIVE is summed over
the 8 XCDs; 1024 SIMDs (256 CUs x 4)."""
import collections, os, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r06_run4_backward_sq_counters.txt")
v = collections.defaultdict(dict)
for line in open(f):
    m = re.match(r"void banet::\(anonymous namespace\)::(adj_\w+)<([^>]*)>.*?\s(\w+)\s+launches=\s*(\d+)\s+mean=([0-9.e+]+)", line)
    if m:
        v[m.group(1) + "<" + m.group(2) + ">"][m.group(3)] = float(m.group(5))
for k, c in v.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    wc = c["SQ_WAVE_CYCLES"]
    print("%s: mean launch %.3g cycles; VALU pipe busy %.0f %%; resident waves per SIMD %.2f; a wave: instruction active %.0f %% (VALU %.0f, scalar %.0f, "
          "LDS %.0f, other %.0f), stalled at issue %.0f %%, parked at s_waitcnt / barriers %.0f %%; LDS array busy %.0f %%; per launch %.3g VALU / "
          "%.3g scalar / %.3g LDS / %.3g vector-memory-read instructions" % (
              k, cyc, 100 * 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 4 * wc / (1024 * cyc), 100 * c["SQ_ACTIVE_INST_ANY"] / wc,
              100 * c["SQ_ACTIVE_INST_VALU"] / wc, 100 * c["SQ_ACTIVE_INST_SCA"] / wc, 100 * c["SQ_ACTIVE_INST_LDS"] / wc,
              100 * (c["SQ_ACTIVE_INST_VMEM"] + c["SQ_ACTIVE_INST_MISC"]) / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
              100 * (1 - (c["SQ_ACTIVE_INST_ANY"] + c["SQ_WAIT_INST_ANY"]) / wc), 100 * c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc),
              c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_LDS"], c["SQ_INSTS_VMEM_RD"]))
