#!/usr/bin/env python
"""One-screen summary of a bench.py JSON line read from stdin (tools/gpu.sh)."""
import json
import sys

d = json.loads(sys.stdin.read())
g = lambda k: (d.get(k) or {}).get("value")
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "single", g("single_stream"), "module", g("module_call"),
      "batch1", g("batch1"), "batch4", g("batch4"), "config3", g("config3"), "batch_exact", g("batch_exact"), "batch16", g("batch16"))
r = d["roofline"]
print("roofline", r["kernel"], r["bound"], round(r["achieved"], 1), r["unit"], "frac", round(r["frac"], 4), "avg us", round(r["avg_launch_us"], 2),
      "launches", r.get("launches_per_step", r.get("launches_per_frame")), "kernel sum ms", round(r.get("kernel_sum_ms_per_step", r.get("kernel_sum_ms_per_frame")), 3))
for k in ("config4", "config5", "fp32"):
    if d.get(k):
        v = dict(d[k])
        v.pop("kernels", None)
        print(k, json.dumps(v)[:900])
if d.get("op_surface"):
    for row in d["op_surface"]["rows"]:
        print("  ", row["op"], row["us"], "us", row["gbs"], "GB/s", row["frac"])
print("pcie", d.get("pcie_inclusive"), "cpu", d.get("cpu_baseline"))
