#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output) of bench.py into per-kernel
HBM traffic per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE (reported in KiB, derived from TCC_EA0_RDREQ x 64 B) under-counts wide coalesced reads
by exactly 2x -> doubled; WRITE_SIZE is taken as reported (uncalibrated).

    python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv > profiles/rNN_pmc_traffic.json
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def main(fetch_csv, write_csv):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, n = f.get(k, (0.0, 0))
        wk, n2 = w.get(k, (0.0, 0))
        out[short(k)] = {"launches_sampled": max(n, n2),
                         "fetch_bytes_per_launch": fk * 1024.0 * 2.0,
                         "write_bytes_per_launch": wk * 1024.0,
                         "hbm_bytes_per_launch": fk * 1024.0 * 2.0 + wk * 1024.0}
    json.dump({"note": "FETCH_SIZE KiB x2 (gfx950 wide-read correction) + WRITE_SIZE KiB, mean per launch",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
