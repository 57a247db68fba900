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
steps = 0
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "adam" in n:
        steps += 1
    if n.startswith("void at::native") or "at::native" in n:
        torch_k += 1
    fam = ("fft chain" if re.search(r"cgemm|fft", n) else "winograd" if "wino" in n else "direct conv (igemm/head/splitk)" if re.search(r"conv_igemm|conv_head|splitk", n)
           else "direct wgrad" if "wgrad" in n else "batchnorm" if re.search(r"bn_", n) else "losses/metrics" if re.search(r"berhu|sobel|smooth|sqdiff|finalize_sum|absdiff|zero_u32", n)
           else "adam" if "adam" in n else "other gdn" if "anonymous" in n or "kernel" in n and "at::" not in n else "torch")
    grp[fam] += d
    tot[fam] += 1
T = sum(grp.values())
print("steps seen: %d   dispatches: %d   torch (at::native) dispatches: %d" % (steps, len(rows), torch_k))
print("%-36s %10s %8s %8s" % ("family", "ms/step", "%", "launches/step"))
for k, v in grp.most_common():
    print("%-36s %10.2f %8.1f %8.1f" % (k, v / 1e3 / max(steps, 1), 100 * v / T, tot[k] / max(steps, 1)))
print("%-36s %10.2f" % ("sum of kernel time per step", T / 1e3 / max(steps, 1)))
PY
head -c 400 $out/bench.json; echo; cat $out/step_summary.txt; head -45 $out/by_kernel_and_grid.txt
