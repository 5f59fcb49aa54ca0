#!/usr/bin/env python3
"""Mean per-launch counter values per kernel from rocprofv3 --pmc CSV directories (development aid).

    python tools/pmc_summary.py gpurun_out/pmc_a
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                vals[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(vals):
        print(k)
        for c in sorted(vals[k]):
            v = vals[k][c]
            print(f"    {c:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
    for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
        print(open(f).read())


if __name__ == "__main__":
    main()
