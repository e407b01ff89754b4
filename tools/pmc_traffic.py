#!/usr/bin/env python3
"""Combine two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE - they do not fit one pass on gfx950) of the same bench
command into per-launch HBM-side traffic per kernel family. Corrections per MI355X_MICROARCH.md 'HBM': both counters
are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B stores.
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def load_db(path, counter):
    """rocpd sqlite database (the default output of this ROCm's rocprofv3)"""
    import sqlite3

    acc = defaultdict(list)
    cur = sqlite3.connect(path).cursor()
    q = ("select s.kernel_name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.event_id = e.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ?")
    for name, val in cur.execute(q, (counter,)):
        acc[family(demangle(name))].append(float(val))
    return acc


def demangle(name):
    """_ZN12_GLOBAL__N_119gemm_bf16_kernel_v2ILi192E...  ->  gemm_bf16_kernel_v2<...> (enough for family())"""
    name = name.replace("_ZN12_GLOBAL__N_1", "").replace("_Z", "")
    m = re.match(r"(\d+)(.*)", name)
    return m.group(2)[:int(m.group(1))] + ("<" + m.group(2)[int(m.group(1)):].split("EEv")[0] + ">" if "I" in m.group(2)[int(m.group(1)):][:1] else "") if m else name


def load(path, counter):
    if path.endswith(".db"):
        return load_db(path, counter)
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[family(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {"unit": "bytes per launch (mean over the launches of the run)", "corrections": "FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1", "kernels": {}}
    tot_f = tot_w = conv_f = conv_w = 0.0
    n_gemm = n_conv = 0
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])
        w = write.get(k, [])
        fb = 2.0 * 1024.0 * (sum(f) / len(f)) if f else 0.0
        wb = 1024.0 * (sum(w) / len(w)) if w else 0.0
        out["kernels"][k] = {"launches": max(len(f), len(w)), "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
        m = re.search(r"ELb([01])E", k)  # first bool template argument of gemm_bf16_kernel_v2 = CONV
        is_conv = k.startswith("conv3d_halo") or (k.startswith("gemm_bf16_kernel_v2") and m is not None and m.group(1) == "1")
        if is_conv:
            conv_f += 2.0 * 1024.0 * sum(f)
            conv_w += 1024.0 * sum(w)
            n_conv += max(len(f), len(w))
        elif k.startswith("gemm_bf16_kernel"):
            tot_f += 2.0 * 1024.0 * sum(f)
            tot_w += 1024.0 * sum(w)
            n_gemm += max(len(f), len(w))
        elif k.startswith("splitk_finish"):  # the finish pass belongs to its GEMM launch: bytes counted, no launch of its own
            tot_f += 2.0 * 1024.0 * sum(f)
            tot_w += 1024.0 * sum(w)
    if n_gemm:  # the dense (DiT) GEMM launches: bench.py's roofline.traffic
        out["gemm_all"] = {"launches": n_gemm, "hbm_bytes_per_launch": (tot_f + tot_w) / n_gemm,
                           "fetch_bytes_per_launch": tot_f / n_gemm, "write_bytes_per_launch": tot_w / n_gemm}
    if n_conv:  # the implicit-GEMM conv launches (VAE): bench.py's vae.roofline.traffic
        out["conv_all"] = {"launches": n_conv, "hbm_bytes_per_launch": (conv_f + conv_w) / n_conv,
                           "fetch_bytes_per_launch": conv_f / n_conv, "write_bytes_per_launch": conv_w / n_conv}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out.get("gemm_all", {})))


if __name__ == "__main__":
    main()
