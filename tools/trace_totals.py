#!/usr/bin/env python3
"""Per-(kernel, grid) totals of a rocprofv3 --kernel-trace CSV, optionally only the last `--last N` dispatches.
Usage: python tools/trace_totals.py <kernel_trace.csv> [--last N]"""
import csv
import re
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if "--last" in sys.argv:
        rows = rows[-int(sys.argv[sys.argv.index("--last") + 1]):]
    tot = defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*\)$", "", n)[:70]
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
        key = f"{n} wg={gx}"
        tot[key][0] += 1
        tot[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print(f"dispatches {len(rows)}  span {span / 1e3:.3f} ms  sum of kernels {sum(v[1] for v in tot.values()) / 1e3:.3f} ms")
    for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{k:86s} n={c:5d} total={us / 1e3:8.3f} ms avg={us / c:9.1f} us")


if __name__ == "__main__":
    main()
