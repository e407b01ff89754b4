#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean of each counter per (kernel, grid).
Usage: python tools/pmc_summary.py <counter_collection.csv>"""
import csv
import re
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*\)$", "", name)[:60]
        key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, ctrs in acc.items():
        print(f"{key[0]} grid={key[1]}")
        for c, v in sorted(ctrs.items()):
            print(f"    {c:32s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()
