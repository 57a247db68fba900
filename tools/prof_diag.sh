#!/bin/bash
# GPU box: rocprofv3 kernel-trace of one diag script -> per-(kernel, grid) table.   usage: prof_diag.sh <tag> <script.py> [args...]
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/prof_diag_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o d -- python3 $R/"$@" > $out/stdout.txt 2> $out/stderr.txt
cd $R
tr=$(ls $out/t/*kernel_trace.csv $out/t/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -z "$tr" ]; then echo "no kernel trace"; tail -20 $out/stderr.txt; exit 1; fi
python3 tools/summarize_trace.py $tr 60 > $out/by_kernel_and_grid.txt
cat $out/stdout.txt; head -50 $out/by_kernel_and_grid.txt
