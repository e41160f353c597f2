#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py <dir> [> summary.txt]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    root = sys.argv[1]
    sums = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add(row["Dispatch_Id"])
    for k in sorted(sums, key=lambda k: -sum(sums[k].values())):
        print(f"{k}  dispatches={len(calls[k])}")
        for c in sorted(sums[k]):
            print(f"    {c:32s} {sums[k][c]:.6g}   per-dispatch {sums[k][c] / max(1, len(calls[k])):.6g}")


if __name__ == "__main__":
    main()
