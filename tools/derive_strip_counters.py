#!/usr/bin/env python3
"""The figures DESIGN.md section 4.1d derives from the SQ counters of the strip gather (profiles/r05_run1_strip_sq_counters_fp_vs_2frame.txt):
VALU-pipe busy, what a wave does with its life (instruction active / stalled at issue / parked), instructions per step.
    python tools/derive_strip_counters.py [file]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x 4 = cycles); GRBM_GUI_ACTIVE is summed
over the 8 XCDs; 1024 SIMDs (256 CUs x 4); a 640x480 x 32 launch has 32 x 1200 segments of 16 x 16 pixels = 64 steps x 4 ... i.e.
64 (pixel row, slice) steps per segment and target frame."""
import os
import re
import sys

f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles",
                                                        "r05_run1_strip_sq_counters_fp_vs_2frame.txt")
vals = {}
for line in open(f):
    m = re.match(r"void banet::ba_gather128s_kernel<1, 4, (true|false)>.*?\s(\w+)\s+launches=\s*\d+\s+mean=([0-9.e+]+)", line)
    if m:
        vals.setdefault("fp" if m.group(1) == "true" else "2f", {})[m.group(2)] = float(m.group(3))
for key, pairs, name in (("fp", 4, "frame-parallel, 4 target frames"), ("2f", 1, "2-frame")):
    v = vals[key]
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0                      # kernel duration in shader cycles
    steps = 32 * 1200 * pairs * 64.0                       # (pixel row, slice) steps of the launch
    wave = 4.0 * v["SQ_WAVE_CYCLES"]
    print("%s: kernel %.3g cycles; VALU pipe busy %.0f %%; per wave: instruction active %.0f %% (VALU %.0f, scalar %.0f, LDS %.0f, "
          "vector memory %.1f, other %.0f), stalled at issue %.1f %%, parked %.0f %%; LDS array busy %.0f %% (conflicts %.0f %% of it)" % (
              name, cyc, 100 * 4 * v["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"],
              100 * v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_ACTIVE_INST_SCA"] / v["SQ_WAVE_CYCLES"],
              100 * v["SQ_ACTIVE_INST_LDS"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_ACTIVE_INST_VMEM"] / v["SQ_WAVE_CYCLES"],
              100 * v["SQ_ACTIVE_INST_MISC"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
              100 * (1 - (v["SQ_ACTIVE_INST_ANY"] + v["SQ_WAIT_INST_ANY"]) / v["SQ_WAVE_CYCLES"]),
              100 * v["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
    print("    per step (all phases of a segment amortised): %.0f VALU + %.0f scalar + %.0f LDS + %.1f vector-memory instructions, "
          "%.0f wave-cycles" % (v["SQ_INSTS_VALU"] / steps, v["SQ_INSTS_SALU"] / steps, v["SQ_INSTS_LDS"] / steps,
                                v["SQ_INSTS_VMEM_RD"] / steps, wave / steps))
