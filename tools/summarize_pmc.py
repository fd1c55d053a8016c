#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")[:70]
                acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
        print("==", f)
        for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:12]:
            for c, v in cs.items():
                print("%-72s %-12s launches=%5d mean=%.4g sum=%.4g" % (k, c, len(v), sum(v) / len(v), sum(v)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
