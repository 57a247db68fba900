#!/bin/bash
# GPU box: PMC passes (separate runs, kernel-trace only) of the per-bin complex GEMMs of one frequency-domain layer at B=20 -> JSON
# (gpurun_out/pmc_cgemm_<tag>/summary.json).  usage: pmc_cgemm.sh <tag> <k> <C> <H> <W> <which 0|1|2> [train 0|1]
tag=${1:-k9}; K=${2:-9}; C=${3:-64}; H=${4:-128}; W=${5:-416}; WHICH=${6:-0}; TRAIN=${7:-1}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_cgemm_$tag; rm -rf $out; mkdir -p $out
cat > /tmp/cg_drv.py <<'PY'
import sys, pathlib
R = pathlib.Path(sys.argv[1]); sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "gdn-pytorch_amd"))
K, C, H, W, WHICH, TRAIN = (int(v) for v in sys.argv[2:8])
import torch
from gdn_amd import ops
op = ops.Conv(C, C, K, 1, K // 2)
ws, bins, M, npnt = op.fft_cgemm_only(20, H, W, WHICH, train=bool(TRAIN))
for _ in range(5):
    op.fft_cgemm_only(20, H, W, WHICH, ws=ws, train=bool(TRAIN))
torch.cuda.synchronize()
print("bins", bins, "M", M, "np", npnt)
PY
A="$R $K $C $H $W $WHICH $TRAIN"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 /tmp/cg_drv.py $A > $out/$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/SQ -- python3 /tmp/cg_drv.py $A > $out/SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/SQ2 -- python3 /tmp/cg_drv.py $A > $out/SQ2.log 2>&1
cd $R
python3 - "$out" "$K" "$C" "$H" "$W" "$WHICH" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
K, C, H, W, WHICH = (int(v) for v in sys.argv[2:7])
res, dur, kn = {}, [], "?"
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2"):
    files = sorted(glob.glob(out + "/" + c + "/*/*counter_collection.csv") + glob.glob(out + "/" + c + "/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if "cgemm" not in r["Kernel_Name"]:
            continue
        kn = r["Kernel_Name"]
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if c == "SQ" and r["Counter_Name"] == "SQ_WAVES":
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for cn, v in agg.items():
        res[cn] = sum(v) / len(v)
fe, wr = res.get("FETCH_SIZE", 0), res.get("WRITE_SIZE", 0)
row = {"kernel": "%s  %dx%d %d ch, B=20 %dx%d, which %d" % (kn, K, K, C, H, W, WHICH), "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
       "traffic_bytes_per_launch": int((2 * fe + wr) * 1024), "mean_duration_us_profiled": round(sum(dur) / max(len(dur), 1), 1)}
for cn in sorted(res):
    if cn not in ("FETCH_SIZE", "WRITE_SIZE"):
        row[cn] = round(res[cn])
if res.get("GRBM_GUI_ACTIVE"):
    # fp32 MFMA 32x32x2: SQ_VALU_MFMA_BUSY_CYCLES counts 64 per instruction
    row["mfma_pipe_busy_pct"] = round(100.0 * res["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * res["GRBM_GUI_ACTIVE"] / 8), 1)
    if dur:
        row["clock_ghz_profiled"] = round(res["GRBM_GUI_ACTIVE"] / 8 / (sum(dur) / len(dur)) / 1e3, 2)
row["command"] = "bash tools/pmc_cgemm.sh " + " ".join(sys.argv[2:7])
json.dump(row, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(row, indent=1))
PY
