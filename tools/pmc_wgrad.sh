#!/bin/bash
# GPU box: PMC passes (separate runs, kernel-trace only) of the bf16 weight-gradient kernels on one layer at B=20 -> JSON
# (gpurun_out/pmc_wgrad_<tag>/summary.json; copy to profiles/rNN_wgrad_ring_pmc.json).  FETCH_SIZE is doubled per the gfx950 correction.
# usage: pmc_wgrad.sh <tag> <k> <C> <H> <W> [cfg]     e.g. pmc_wgrad.sh k9 9 64 128 416 4
tag=${1:-k9}; K=${2:-9}; C=${3:-64}; H=${4:-128}; W=${5:-416}; CFG=${6:-4}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_wgrad_$tag; rm -rf $out; mkdir -p $out
cat > /tmp/wg_drv.py <<'PY'
import sys, pathlib
R = pathlib.Path(sys.argv[1]); sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "gdn-pytorch_amd"))
K, C, H, W, CFG = (int(v) for v in sys.argv[2:7])
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
op = ops.Conv(C, C, K, 1, K // 2)
x = torch.randn(20, H, W, C, device=dev).bfloat16(); gy = torch.randn(20, H, W, C, device=dev).bfloat16()
dw = torch.empty(K * K, C, C, device=dev)
for _ in range(5):
    op.wgrad(x, gy, dw, cfg=CFG)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 /tmp/wg_drv.py $R $K $C $H $W $CFG > $out/$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/SQ -- python3 /tmp/wg_drv.py $R $K $C $H $W $CFG > $out/SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/SQ2 -- python3 /tmp/wg_drv.py $R $K $C $H $W $CFG > $out/SQ2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $out/SQ3 -- python3 /tmp/wg_drv.py $R $K $C $H $W $CFG > $out/SQ3.log 2>&1
cd $R
python3 - "$out" "$K" "$C" "$H" "$W" "$CFG" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
K, C, H, W, CFG = (int(v) for v in sys.argv[2:7])
res, dur = {}, []
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2", "SQ3"):
    files = sorted(glob.glob(out + "/" + c + "/*/*counter_collection.csv") + glob.glob(out + "/" + c + "/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if "wgrad" not in r["Kernel_Name"] or "reduce" in r["Kernel_Name"]:
            continue
        kn = r["Kernel_Name"]
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if c == "SQ" and r["Counter_Name"] == "SQ_WAVES":
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for cn, v in agg.items():
        res[cn] = sum(v) / len(v)
fe, wr = res.get("FETCH_SIZE", 0), res.get("WRITE_SIZE", 0)
row = {"kernel": "%s  %dx%d s1 %d->%d wgrad, B=20 %dx%d, bf16, cfg %d" % (kn, K, K, C, C, H, W, CFG), "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
       "traffic_bytes_per_launch": int((2 * fe + wr) * 1024),
       "algorithmic_bytes_per_launch": 20 * H * W * C * 2 * 2 + K * K * C * C * 4,
       "algorithmic_gflop": 2.0 * 20 * H * W * K * K * C * C / 1e9,
       "mean_duration_us_profiled": round(sum(dur) / max(len(dur), 1), 1)}
for cn in sorted(res):
    if cn not in ("FETCH_SIZE", "WRITE_SIZE"):
        row[cn] = round(res[cn])
if res.get("GRBM_GUI_ACTIVE"):
    row["mfma_pipe_busy_pct"] = round(100.0 * res["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * res["GRBM_GUI_ACTIVE"] / 8), 1)
    if dur:
        row["clock_ghz_profiled"] = round(res["GRBM_GUI_ACTIVE"] / 8 / (sum(dur) / len(dur)) / 1e3, 2)
        row["tflops_profiled"] = round(row["algorithmic_gflop"] / (sum(dur) / len(dur)) * 1e3, 1)
row["note"] = "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE reads half of a wide coalesced stream); L2 -> fabric requests, Infinity-Cache hits included"
row["command"] = "bash tools/pmc_wgrad.sh " + " ".join(sys.argv[2:7])
row["collected_at"] = os.environ.get("GDN_COMMIT") or None
json.dump(row, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(row, indent=1))
PY
