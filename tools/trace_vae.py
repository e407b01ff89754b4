#!/usr/bin/env python3
"""Post-process a rocprofv3 --kernel-trace CSV of `tools/bench_vae.py`: the launch timeline of the LAST VAE decode in the trace
(from the first kernel after the previous decode's conv_out to its own conv_out) with grid, duration and gap.
Usage: python tools/trace_vae.py <kernel_trace.csv>"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
           int(r.get("Grid_Size_X", 0) or 0) * max(int(r.get("Grid_Size_Y", 1) or 1), 1), int(r.get("Workgroup_Size_X", 1) or 1),
           int(r.get("Grid_Size_Y", 1) or 1)) for r in rows]
    # conv_out (the 256x64 tile instantiation; its epilogue writes the frames) is the last launch of a decode
    ends = [i for i, e in enumerate(ev) if e[0].startswith("gemm_bf16_kernel_v2<256, 64")]
    if len(ends) < 2:
        print("need two decodes in the trace")
        return
    a, b = ends[-2] + 1, ends[-1]
    print(f"{'kernel':62s} {'wgs':>7s} {'y':>3s} {'us':>8s} {'gap_us':>7s}")
    t_prev, tot = ev[a][1], 0.0
    for i in range(a, b + 1):
        n, s, e, g, w, gy = ev[i]
        print(f"{n:62s} {g // max(w, 1):7d} {gy:3d} {(e - s) / 1e3:8.1f} {(s - t_prev) / 1e3:7.1f}")
        t_prev = e
        tot += (e - s) / 1e3
    print(f"kernel time {tot / 1e3:.3f} ms over {b - a + 1} launches; span {(ev[b][2] - ev[a][1]) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
