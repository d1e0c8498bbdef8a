#!/usr/bin/env python
"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass over bench.py (CSV output):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv ...
    python tools/pmc_mfma.py <..._counter_collection.csv> > profiles/rNN_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's MFMA pipe is busy, summed over all 1024 SIMDs
(MI355X_MICROARCH.md: "= 32 x N_mfma for 32x32x16 bf16"; checked here: = 16 x SQ_INSTS_MFMA for the
16x16x32 bf16 kernels).  GRBM_GUI_ACTIVE is reported SUMMED OVER THE 8 XCDs (one GRBM instance each):
GUI_ACTIVE / 8 / kernel duration = 2.0-2.1 GHz on every kernel of the r02 pass, whereas the raw value would
mean a 17 GHz clock.  So
    mfma_utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs)
= the fraction of the chip's MFMA issue slots the kernel used while it ran (the profiled run: kernels are
~15 % slower under rocprofv3 than un-profiled).  The raw sums are kept next to the derived fraction.
"""
import collections
import csv
import json
import re
import sys

SIMDS = 256 * 4


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def main(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, d in sorted(acc.items()):
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        n = max(len(v) for v in d.values())
        row = {"launches_sampled": n, **{c.lower() + "_per_launch": m for c, m in mean.items()}}
        busy, gui = mean.get("SQ_VALU_MFMA_BUSY_CYCLES"), mean.get("GRBM_GUI_ACTIVE")
        if busy is not None and gui:
            row["mfma_utilisation"] = busy / (gui / 8.0 * SIMDS)
        out[k] = row
    json.dump({"note": "mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), mean per launch",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
