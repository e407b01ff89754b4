#!/usr/bin/env python3
"""Kernel statistics (and PMC counter means) from a rocprofv3 run's rocpd sqlite database - the default output of this ROCm's
rocprofv3; `--output-format csv` hung on this pool in round 3.
Usage: python tools/rocpd_stats.py <results.db> [--csv out.csv] [--pmc]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = "rocpd_kernel_dispatch"
    ks = "rocpd_info_kernel_symbol"
    rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                            f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append(f"\"{name}\",{n},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}")
    out = "\n".join(lines) + "\n"
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out)
    else:
        sys.stdout.write(out)
    if "--pmc" in sys.argv:
        pt = [t for t in tabs if "pmc" in t.lower()]
        print("pmc tables:", pt, file=sys.stderr)


if __name__ == "__main__":
    main()
