#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, mean/total time.
usage: summarize_trace.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def split_steps(rows):
    """Segments ending at an adam kernel (one training step each); what follows the last adam kernel is dropped."""
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "adam" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    return steps


# The table is over STEADY-STATE steps only: the first segment holds set-up kernels and the first step's
# one-off work (weight packing, first-touch page faults: a 20 ms outlier averaged into a 0.2 ms kernel); a trace without at
# least three adam kernels (single-kernel probes) is summarised whole.
all_steps = split_steps(rows)
steady = len(all_steps) >= 3
table_rows = [r for st in all_steps[1:] for r in st] if steady else rows
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in table_rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[(name, grid)]
    a[0] += 1
    a[1] += d
    tot += d
if steady:
    print("steady-state steps only: %d of %d steps (the first, with set-up and one-off work, is dropped: %d of %d dispatches kept)"
          % (len(all_steps) - 1, len(all_steps), len(table_rows), len(rows)))
else:
    print("whole trace (fewer than three adam kernels: no step structure to cut at)")
print("total kernel time %.2f ms over %d dispatches" % (tot / 1e3, len(table_rows)))
print("%-62s %-16s %6s %10s %10s %6s" % ("kernel", "grid(threads)", "calls", "mean_us", "total_ms", "%"))
for (name, grid), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-62s %-16s %6d %10.1f %10.2f %6.1f" % (name, grid, n, t / n, t / 1e3, 100 * t / tot))

# ---- persistent kernels: their grid is the CU count whatever the layer, so (kernel, grid) averages every
# layer and batch size that runs on one symbol.  A training step launches its kernels in a fixed order: the i-th launch of a
# symbol between two adam kernels is the same layer call in every step.  Per (symbol, ordinal): mean / min / max over the steps.
PERSISTENT = re.compile(r"conv_ring2?_bf16|wgrad_ring_bf16|cgemm_bins_kernel")
steps = all_steps[1:]                   # (the first segment holds set-up and warm-up differences)
if len(steps) >= 3:
    per = collections.defaultdict(list)
    counts = collections.Counter()
    for st in steps:
        seen = collections.Counter()
        for r in st:
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name)[:60]
            if not PERSISTENT.search(name):
                continue
            per[(name, seen[name])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            seen[name] += 1
        for k, v in seen.items():
            counts[(k, v)] += 1
    stable = {k for k in {n for n, _ in per} if len({c for (n, c) in counts if n == k}) == 1}
    print("\npersistent kernels by call ordinal within a step (%d steps; symbols whose launch count differs between steps are skipped)" % len(steps))
    print("%-62s %4s %10s %10s %10s" % ("kernel", "#", "mean_us", "min_us", "max_us"))
    for (name, i), v in sorted(per.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        if name in stable:
            print("%-62s %4d %10.1f %10.1f %10.1f" % (name, i, sum(v) / len(v), min(v), max(v)))
