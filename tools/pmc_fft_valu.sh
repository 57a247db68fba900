#!/bin/bash
# GPU box: VALU / wait counters of the frequency-domain kernels (one PMC pass, kernel-trace only)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_fft_valu
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p -- python3 $R/tests/diag/fft_kernels_time.py > $out/p.log 2>&1
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + "/p/*/*counter_collection.csv")
if not files:
    print("no counter file; tail of the profiler log:")
    print("".join(open(out + "/p.log").readlines()[-30:]))
    sys.exit(1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[0])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-36s %9s %10s %12s %12s %10s %10s" % ("kernel", "grid", "waves", "valu/wave", "valu_busy%", "wait%", "gui_Mcyc"))
for k, d in sorted(agg.items()):
    if not ("fft" in k[0] or "gemm" in k[0]):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = m.get("SQ_WAVES", 1)
    print("%-36s %9d %10.0f %12.0f %12.1f %10.1f %10.2f" % (k[0], k[1], w, m.get("SQ_INSTS_VALU", 0) / w,
          100 * m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1),
          100 * m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("GRBM_GUI_ACTIVE", 0) / 1e6))
PY
