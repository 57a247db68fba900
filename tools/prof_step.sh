#!/bin/bash
# GPU box: rocprofv3 kernel-trace statistics of the headline bench command (fp32 DtoD, B=20) -> per-kernel and per-(kernel,grid)
# tables under gpurun_out/prof_step_<tag>/ (copy the summaries into profiles/).   usage: prof_step.sh <tag> [bench args...]
set -u
tag=${1:-step}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/prof_step_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o step -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline "$@" > $out/bench.json 2> $out/bench.err
cd $R
tr=$(ls $out/t/*kernel_trace.csv $out/t/*/*kernel_trace.csv 2>/dev/null | head -1)
st=$(ls $out/t/*kernel_stats.csv $out/t/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -z "$tr" ]; then echo "no kernel trace"; tail -20 $out/bench.err; exit 1; fi
cp $st $out/kernel_stats.csv
python3 tools/summarize_trace.py $tr 60 > $out/by_kernel_and_grid.txt
python3 - $tr > $out/step_summary.txt <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into steps at the adam kernel
grp = collections.Counter()
tot = collections.Counter()
torch_k = 0
adam_at = []
sys.path.insert(0, "tools")
from kernel_family import family, is_ours
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "adam" in n:
        adam_at.append(i)
    fam = family(n)
    if fam.startswith("torch"):
        torch_k += 1
    grp[fam] += d
    tot[fam] += 1
steps = len(adam_at)
T = sum(grp.values())
print("steps seen: %d   dispatches: %d   torch (at::native) dispatches: %d" % (steps, len(rows), torch_k))
if steps >= 2:
    a, b = adam_at[-2], adam_at[-1]
    inner = rows[a + 1:b + 1]
    span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in inner) / 1e6
    foreign = [r["Kernel_Name"] for r in inner if not is_ours(r["Kernel_Name"])]
    print("steady-state step (between the last two adam kernels): %d dispatches, %.2f ms span, %.2f ms sum of kernel time, "
          "%d not from libgdn_hip.so%s" % (len(inner), span, busy, len(foreign), (": " + ", ".join(sorted(set(foreign))[:4])) if foreign else ""))
print("%-48s %10s %8s %8s" % ("family", "ms/step", "%", "launches/step"))
for k, v in grp.most_common():
    print("%-48s %10.2f %8.1f %8.1f" % (k, v / 1e3 / max(steps, 1), 100 * v / T, tot[k] / max(steps, 1)))
print("%-48s %10.2f" % ("sum of kernel time per step", T / 1e3 / max(steps, 1)))
PY
python3 tools/check_no_mfma16_beside_fft.py $tr --label "$tag ($*)" > $out/mfma16_vs_fft.txt; chk=$?
head -c 400 $out/bench.json; echo; cat $out/step_summary.txt; cat $out/mfma16_vs_fft.txt; head -45 $out/by_kernel_and_grid.txt
exit $chk
