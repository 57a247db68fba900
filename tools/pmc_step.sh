#!/bin/bash
# GPU box: whole-step MFMA utilisation of the headline workload (fp32 DtoD, B=20): one PMC pass (kernel-trace + SQ/GRBM counters only)
# over 2 warm-up + 5 training steps -> gpurun_out/pmc_step/step_mfma_util.json (copy to profiles/rNN_step_mfma_util.json).
# GDN_COMMIT=<short hash> in the environment is recorded as collected_at (the GPU box has no .git).
#   mfma_util_pct = 100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (4 SIMDs * 256 CUs * sum(GRBM_GUI_ACTIVE) / 8 XCDs)
# i.e. MFMA-pipe busy cycles over ALL SIMD cycles of the kernels of the step (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/pmc_step
rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline "$@" > $out/bench.json 2> $out/bench.err
cd $R
python3 - $out <<'PY'
import csv, glob, json, os, sys, collections, re
sys.path.insert(0, "tools")
from kernel_family import family
out = sys.argv[1]
files = glob.glob(out + "/p/*/*counter_collection.csv") + glob.glob(out + "/p/*counter_collection.csv")
if not files:
    print("no counter file; tail of the log:"); print("".join(open(out + "/bench.err").readlines()[-30:])); sys.exit(1)
tot = collections.Counter()
fam = collections.defaultdict(collections.Counter)
nd = 0
for r in csv.DictReader(open(files[0])):
    n, c, v = r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
    tot[c] += v
    f = family(n)
    fam[f][c] += v
    nd += c == "GRBM_GUI_ACTIVE"
util = 100.0 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * tot["GRBM_GUI_ACTIVE"] / 8)
res = {"mfma_util_pct": round(util, 2), "steps": 7, "dispatches": nd,
       "sum_SQ_VALU_MFMA_BUSY_CYCLES": tot["SQ_VALU_MFMA_BUSY_CYCLES"], "sum_GRBM_GUI_ACTIVE": tot["GRBM_GUI_ACTIVE"],
       "by_family_pct": {k: round(100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (128 * v["GRBM_GUI_ACTIVE"]), 2) for k, v in fam.items() if v["GRBM_GUI_ACTIVE"]},
       "gui_share_pct": {k: round(100.0 * v["GRBM_GUI_ACTIVE"] / tot["GRBM_GUI_ACTIVE"], 1) for k, v in fam.items()},
       "note": "fp32 DtoD B=20 training steps (2 warm-up + 5), every kernel dispatch; MFMA-pipe busy cycles / (4 SIMDs x 256 CUs x "
               "GRBM_GUI_ACTIVE/8); kernels are serialised under counter collection, so inter-kernel gaps and the two-stream overlap "
               "of the fft backward are not in the denominator; fp32 MFMA peak = 100 %",
       "command": "bash tools/pmc_step.sh", "collected_at": os.environ.get("GDN_COMMIT") or None}
json.dump(res, open(out + "/step_mfma_util.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
