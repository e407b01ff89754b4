#!/usr/bin/env python3
"""Effective clock + MFMA duty of GEMM launches from one rocprofv3 pass with GRBM_GUI_ACTIVE and SQ_VALU_MFMA_BUSY_CYCLES
(+ kernel trace for durations): clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; duty = MFMA busy cycles / (1024 SIMDs x cycles).
Usage: python tools/pmc_clock.py <counter_collection.csv> <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict


def fam(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*\)$", "", n)[:70]


def main():
    dur = {}
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), fam(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")))
    ctr = defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    agg = defaultdict(list)
    for d, c in ctr.items():
        if d not in dur or "GRBM_GUI_ACTIVE" not in c:
            continue
        ns, name, grid = dur[d]
        if ns < 20000:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        ghz = cyc / ns
        duty = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else 0.0
        agg[(name, grid)].append((ns / 1e3, ghz, duty))
    for k, v in agg.items():
        n = len(v)
        print(f"{k[0]:72s} grid={k[1]:>8s} n={n:3d} us={sum(x[0] for x in v) / n:8.1f} clock={sum(x[1] for x in v) / n:5.2f} GHz mfma_duty={sum(x[2] for x in v) / n:5.3f}")


if __name__ == "__main__":
    main()
