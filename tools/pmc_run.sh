#!/bin/bash
# usage: pmc_run.sh <tag> <prof_layer args...>   -- separate PMC passes (never combined with trace domains other than kernel-trace)
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 $R/tools/prof_layer.py "$@" > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p2 -- python3 $R/tools/prof_layer.py "$@" > $out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p3 -- python3 $R/tools/prof_layer.py "$@" > $out/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/p4 -- python3 $R/tools/prof_layer.py "$@" > $out/p4.log 2>&1
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    files = glob.glob(out + "/" + p + "/*/*counter_collection.csv")
    if not files:
        print(p, "no counter file", glob.glob(out + "/" + p + "/*/*")); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "conv_" not in k: continue
        print(p, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
    tr = glob.glob(out + "/" + p + "/*/*kernel_trace.csv")
    if tr and p == "p1":
        ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr[0])) if "conv_" in r["Kernel_Name"]]
        print("   mean duration us (profiled):", round(sum(ds) / len(ds), 1), "n", len(ds))
PY
