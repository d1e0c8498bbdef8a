#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a per-kernel stats table
(what `rocprofv3 --stats` prints in CSV mode): calls, total / average / min / max duration.

    python tools/rocpd_stats.py gpurun_out/prof_x/bench_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for n, cnt, tot, avg, mn, mx in rows:
        print(f"{cnt:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}  {n}")


if __name__ == "__main__":
    main(sys.argv[1])
