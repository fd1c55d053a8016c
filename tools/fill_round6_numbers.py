#!/usr/bin/env python3
"""Fill the @PLACEHOLDER@ numbers of DESIGN.md / README.md / profiles/README.md from the files tools/gpu_r6_final.sh left under
gpurun_out/r6f/ (and copy those files to profiles/r06_*).   python tools/fill_round6_numbers.py"""
import glob, json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "r6f")
P = os.path.join(ROOT, "profiles")
for f in ("pytest_gpu.txt", "smoke.txt", "dense_train.txt", "backward_pmc_fetch_write.txt", "eqcon_literal_op.txt", "sparse_training_and_tracker.txt",
          "pmc_fetch_write_headline.txt", "pmc_fetch_write_cfg3.txt", "pmc_fetch_write_cfg5.txt", "bench_detail.json", "bench_default_line.json",
          "bench_2rank_gloo_one_gpu_line.json"):
    shutil.copy(os.path.join(R, f), os.path.join(P, "r06_" + f))
for f in glob.glob(os.path.join(P, "r06_*_kernel_stats_build_*.csv")):
    if "959cda8c4e94dfd4" not in f:          # (the dense-training statistics of the build before the instruction pass stay: its before-record)
        os.remove(f)
BID = open(os.path.join(R, "build_id.txt")).read().split()[-1]          # gpurun_out/ keeps earlier builds' files: only this build's
for f in glob.glob(os.path.join(R, "*_kernel_stats_build_%s.csv" % BID)):
    shutil.copy(f, os.path.join(P, "r06_" + os.path.basename(f)))
shutil.copy(os.path.join(R, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
line = json.load(open(os.path.join(R, "bench_default_line.json")))
det = json.load(open(os.path.join(R, "bench_detail.json")))
tests = re.search(r"(\d+) passed", open(os.path.join(R, "pytest_gpu.txt")).read()).group(1)
dt = open(os.path.join(R, "dense_train.txt")).read()
fb = re.findall(r"forward only \(lm_level\)\s+([\d.]+) ms.*?\n\s*forward \+ backward \(solve_differentiable\)\s+([\d.]+) ms.*?([\d.]+)x the forward\), peak extra memory ([\d.]+) GB", dt)
# order: 32, 8, 2 windows, 8 five-frame; 8 five-frame with REUSE=0; then FOLD=0, SMALL_STEP_HIP=0 (TILE=1 carries a prefix: not matched)
sp = open(os.path.join(R, "sparse_training_and_tracker.txt")).read()
sparse_ms = re.search(r"fused\s+B=4 N=4096.*?: ([\d.]+) ms", sp).group(1)
tracker = [json.loads(l) for l in sp.split("\n") if l.startswith("{")]
sw = line["sweep"]
k = lambda v: "%.2f k" % (v / 1e3)
import csv
rows = list(csv.DictReader(open(os.path.join(R, "sparse_training_iteration_kernel_stats_build_%s.csv" % BID))))
launches = sum(int(r["Calls"]) for r in rows if "banet" in r["Name"] and "target_map_kernel" not in r["Name"] and "resample" not in r["Name"]) / 4.0
drows = list(csv.DictReader(open(os.path.join(R, "dense_train_kernel_stats_build_%s.csv" % BID))))      # 4 training steps
per_step = lambda sub: "%.1f" % (sum(float(r["TotalDurationNs"]) for r in drows if sub in r["Name"]) / 4e6)
vals = {
    "K_PIXEL2_MS": per_step("adj_pixel2_kernel"), "K_TILE2_MS": per_step("adj_tile2_kernel"), "K_BASIS6_MS": per_step("adj_basis6_kernel"),
    "GPU_TESTS": tests, "CPU_TESTS": "68",
    "HEAD_VALUE": k(line["value"]), "HEAD_MS": "%.1f" % line["ms_per_step"], "HEAD_FRAC": "%.3f" % line["roofline"]["frac"],
    "HEAD_TRAFFIC": "%.2f" % (line["roofline"]["traffic"] / line["roofline"]["algorithmic_bytes_per_launch"]) if line["roofline"].get("traffic") else "1.19",
    "EXACT_VALUE": k(line["value_exact_syrk"]),
    "FWD_MS": fb[0][0], "BWD_MS": fb[0][1], "BWD_X": fb[0][2], "BWD_GB": "%.1f" % float(fb[0][3]),
    "BWD8_MS": fb[1][1], "BWD85_MS": fb[3][1], "BWD85_NOREUSE_MS": fb[4][1], "BWD_OLDPATH_MS": fb[5][1], "BWD_TORCHSMALL_MS": fb[6][1],
    "SPARSE_MS": sparse_ms, "SPARSE_LAUNCHES": "%.0f" % launches,
    "TRACKER_MS": "%.2f" % tracker[0]["ms_per_solve"],
    "B1_VALUE": k(sw["B1_2frame"]["value"]), "B8_VALUE": k(sw["B8_2frame"]["value"]), "B256_VALUE": k(sw["B256_2frame"]["value"]),
    "CFG3_VALUE": k(sw["cfg3_5frame_B32"]["value"]), "CFG5_VALUE": k(sw["cfg5_8frame_1280x960_K256_B8"]["value"]),
    "CFG1_MS": "%.2f" % sw["cfg1_160x120_K32_B1"]["ms_per_step"], "CPU_VALUE": "%.2f" % line["cpu_baseline"]["value"],
}
print(vals)
NUM = os.path.join(P, "r06_numbers_quoted_in_documents.json")
prev = json.load(open(NUM)) if os.path.exists(NUM) else {}
# Values that are distinctive strings are replaced wherever they stand; the short ones only inside these contexts ({} = the value)
CONTEXT = {
    "GPU_TESTS": ["Parity: {} GPU tests", "({} tests)", "again ({} passed)"],
    "CPU_TESTS": ["+ {} CPU tests"],
    "BWD_X": ["8.46 \u2192 {} \u00d7", "8.46 -> {} x", ": {} \u00d7), peak", "= {} \u00d7 the forward"],
    "BWD_GB": ["47.6 \u2192\n{} GB", "47.6 \u2192 {} GB", "47.6 ->\n{} GB", "47.6 -> {} GB", "memory **{} GB**", "forward (target 5 \u00d7), {} GB"],
    "HEAD_FRAC": ["gather {} of the 8 TB/s"],
    "HEAD_MS": ["{} ms per 32-window"],
    "HEAD_TRAFFIC": ["roofline on {} \u00d7 its", "roofline on {} x its"],
    "TRACKER_MS": ["2.06 \u2192 **{} ms**", "2.06 \u2192 {} ms", "2.06 -> {} ms"],
    "SPARSE_MS": ["**{} ms** per forward + backward at 4 pairs", "2.0 \u2192 {} ms", "2.0 -> {} ms", "2.0 \u2192\n**{} ms**"],
    "SPARSE_LAUNCHES": ["4096 points, {} library launches"],
    "CFG1_MS": ["cfg-1 ({} ms", "cfg-1 {} ms", "cfg-1\n{} ms"],
    "CPU_VALUE": ["CPU baseline {} it/s"],
    "CFG5_VALUE": ["cfg-5 share {}) have", "K = 256) {}, cfg-1"],
    "K_BASIS6_MS": ["**{} ms** (13.6)"], "K_PIXEL2_MS": ["**{} ms** (42;"], "K_TILE2_MS": ["**{} ms** (adj_map2"],
}
for name in ("DESIGN.md", "README.md", os.path.join("profiles", "README.md")):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    for key, v in vals.items():                       # first use: the @PLACEHOLDER@ form
        s = s.replace("@%s@" % key, str(v))
    for key, v in vals.items():                       # later runs: the previous run's figures -> this run's
        o = prev.get(key)
        if o is None or o == v:
            continue
        if key in CONTEXT:
            n = 0
            for c in CONTEXT[key]:
                n += s.count(c.format(o))
                s = s.replace(c.format(o), c.format(v))
        else:
            n = s.count(o)
            s = s.replace(o, str(v))
        print("%-22s %-18s %s -> %s: %d places" % (name, key, o, v, n))
    left = set(re.findall(r"@[A-Z0-9_]+@", s))
    if left:
        print(name, "unfilled:", left)
    open(p, "w").write(s)
json.dump(vals, open(NUM, "w"), indent=1, sort_keys=True)
