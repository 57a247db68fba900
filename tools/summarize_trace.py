#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, mean/total time.
usage: summarize_trace.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[(name, grid)]
    a[0] += 1
    a[1] += d
    tot += d
print("total kernel time %.2f ms over %d dispatches" % (tot / 1e3, len(rows)))
print("%-62s %-16s %6s %10s %10s %6s" % ("kernel", "grid(threads)", "calls", "mean_us", "total_ms", "%"))
for (name, grid), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-62s %-16s %6d %10.1f %10.2f %6.1f" % (name, grid, n, t / n, t / 1e3, 100 * t / tot))
