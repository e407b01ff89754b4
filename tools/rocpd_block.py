#!/usr/bin/env python3
"""Timeline of ONE transformer block from a rocprofv3 --kernel-trace rocpd database: kernels in start order with their durations and
the gaps between them, taken from the middle of the last forward (blocks are delimited by attention launches, two per block).
Usage: python tools/rocpd_block.py <results.db> [block index from the end, default 20]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    return name[:70]


def main():
    con = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = list(con.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d "
                            "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    att = [i for i, r in enumerate(rows) if "attn_fwd" in r[0]]
    lo, hi = att[-2 * back - 1] + 1, att[-2 * back + 1] + 1  # behind the cross attention of block b-1 ... through the cross attention of block b
    prev_end = rows[lo - 1][2]
    t0 = rows[lo][1]
    tot = gap = 0
    for name, s, e, gx, wx in rows[lo:hi]:
        print(f"{(s - t0) / 1e3:9.2f} us  +{(s - prev_end) / 1e3:6.2f} gap  {(e - s) / 1e3:8.2f} us  grid {gx // max(wx, 1):5d} x {wx:4d}  {short(name)}")
        tot += e - s
        gap += max(0, s - prev_end)
        prev_end = e
    print(f"block: {hi - lo} launches, kernels {tot / 1e3:.1f} us, gaps {gap / 1e3:.1f} us, span {(rows[hi - 1][2] - rows[lo - 1][2]) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
