#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 1 --warmup 0`:
mean HBM-side bytes per gather launch (all levels), corrected as MI355X_MICROARCH.md's HBM section prescribes (FETCH_SIZE
x 2 on gfx950; WRITE_SIZE taken at face value).   python tools/make_pmc_traffic.py <fetch_dir> <write_dir> <windows> <out.json>
With a 5th argument NAME (+ FRAMES HEIGHT WIDTH K ITERS): the passes were taken on that sweep workload (`bench.py --frames ...`);
the record is merged into <out.json> under "workloads"[NAME] (bench.py's sweep entries read it, same build-id rule)."""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    acc[row.get("Kernel_Name", "?")].append(float(row.get("Counter_Value", 0)))
    return acc


def build_id():
    """banet_build_id() of the in-tree library (bench.py refuses a traffic file recorded on another build)"""
    import ctypes
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    lib = ctypes.CDLL(os.environ.get("BANET_HIP_LIB") or os.path.join(root, "banet_amd", "lib", "libbanet_hip.so"))
    lib.banet_build_id.restype = ctypes.c_char_p
    return lib.banet_build_id().decode()


def main(fetch_dir, write_dir, windows, out, name=None, frames=2, height=480, width=640, K=128, iters=10):
    frames, height, width, K, iters = int(frames), int(height), int(width), int(K), int(iters)
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    g = lambda acc: [(k, v) for k, v in acc.items() if "ba_gather128" in k]
    nl = sum(len(v) for _, v in g(fe))
    fetch_kb = sum(sum(v) for _, v in g(fe)) / max(nl, 1)
    nw = sum(len(v) for _, v in g(wr))
    write_kb = sum(sum(v) for _, v in g(wr)) / max(nw, 1)
    C = 128
    alg = sum(4 * (height // s) * (width // s) * (frames * C + K + 1) for s in (16, 8, 4, 2, 1)) * int(windows) * iters / (5.0 * iters)
    hbm = 2.0 * fetch_kb * 1024 + write_kb * 1024
    rec = {"kernel": "; ".join("%s x %d" % (k.split("(")[0].replace("void ", ""), len(v)) for k, v in g(fe)),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-sweep "
                     "--no-parity --no-cpu-baseline%s`: mean over its %d gather launches, all 5 levels" % (
                         " (headline workload)" if name is None else " --frames %d --height %d --width %d --basis %d --iters %d --windows %s"
                         % (frames, height, width, K, iters, windows), nl),
           "build_id": build_id(), "windows": int(windows), "fetch_size_kb_per_launch": round(fetch_kb, 1), "write_size_kb_per_launch": round(write_kb, 1),
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE "
                         "uncalibrated, taken at face value",
           "hbm_bytes_per_launch": int(round(hbm)), "algorithmic_bytes_per_launch": int(round(alg)),
           "overfetch": round(hbm / alg, 4),
           "syrk_fetch_kb_per_launch": round(sum(sum(v) for k, v in fe.items() if "ba_syrk" in k) /
                                             max(sum(len(v) for k, v in fe.items() if "ba_syrk" in k), 1), 1)}
    if name is not None:          # a sweep workload: merged under "workloads" of the (headline) file
        top = json.load(open(out)) if os.path.exists(out) else {}
        top.setdefault("workloads", {})[name] = rec
        json.dump(top, open(out, "w"), indent=1)
    else:
        old = json.load(open(out)) if os.path.exists(out) else {}
        if old.get("workloads"):    # keep the sweep workloads of the same build
            rec["workloads"] = {k: v for k, v in old["workloads"].items() if v.get("build_id") == rec["build_id"]}
        json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main(*sys.argv[1:])
