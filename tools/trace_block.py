#!/usr/bin/env python3
"""Post-process a rocprofv3 --kernel-trace CSV of `bench.py`: per-kernel timeline of ONE transformer block in the
steady state (the 20th block of the last timed step) and totals per kernel family for that step.
Usage: python tools/trace_block.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:64]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
           int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)) for r in rows]
    # attention launches mark the blocks: 2 per block (self, cross)
    att = [i for i, e in enumerate(ev) if e[0].startswith("attn_fwd_kernel")]
    if len(att) < 96 * 2:
        print("not enough attention launches", len(att))
        return
    # last full step = last 96 attention launches; block 20 of it
    first = att[-96]
    b0 = att[-96 + 40]
    # walk back to the previous attention launch +? -> the block starts after the previous block's last GEMM (ff2); use the
    # kernel right after the previous cross-attention's three trailing kernels; simpler: print from previous cross attn + 1
    prev = att[-96 + 39]
    nxt = att[-96 + 41]
    print(f"{'kernel':66s} {'grid':>7s} {'us':>8s} {'gap_us':>7s}")
    t_prev_end = ev[prev][2]
    for i in range(prev + 1, nxt + 1):
        n, s, e, gx, wx = ev[i]
        print(f"{n:66s} {gx // max(wx, 1):7d} {(e - s) / 1e3:8.1f} {(s - t_prev_end) / 1e3:7.1f}")
        t_prev_end = e
    tot = defaultdict(lambda: [0, 0.0])
    last = len(ev)
    for i in range(first, last):
        n, s, e, gx, wx = ev[i]
        key = f"{n} g{gx // max(wx, 1)}"
        tot[key][0] += 1
        tot[key][1] += (e - s) / 1e3
    print("\nper-kernel totals from the first attention launch of the last step to the end of the trace:")
    for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k:78s} n={c:5d} total={us / 1e3:8.3f} ms avg={us / c:8.1f} us")


if __name__ == "__main__":
    main()
