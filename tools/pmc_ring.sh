#!/bin/bash
# GPU box: PMC passes (separate runs, kernel-trace only) of conv_ring_bf16<64,9> on the 9x9 64->64 layer at B=20, 128x416 -> JSON
# (gpurun_out/pmc_ring/summary.json; copy to profiles/rNN_conv_ring_pmc.json).  FETCH_SIZE is doubled per the gfx950 correction.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_ring; rm -rf $out; mkdir -p $out
cat > /tmp/ring_drv.py <<'PY'
import sys, pathlib
R = pathlib.Path(sys.argv[1]); sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
op = ops.Conv(64, 64, 9, 1, 4)
x = torch.randn(20, 128, 416, 64, device=dev).bfloat16(); w = (torch.randn(81, 64, 64, device=dev) * 0.02).bfloat16()
for _ in range(5):
    op.fwd(x, w, stats=True)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 /tmp/ring_drv.py $R > $out/$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/SQ -- python3 /tmp/ring_drv.py $R > $out/SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/SQ2 -- python3 /tmp/ring_drv.py $R > $out/SQ2.log 2>&1
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
res, dur = {}, []
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2"):
    files = sorted(glob.glob(out + "/" + c + "/*/*counter_collection.csv") + glob.glob(out + "/" + c + "/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if "conv_ring2_bf16" not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if c == "SQ" and r["Counter_Name"] == "SQ_WAVES":
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for cn, v in agg.items():
        res[cn] = sum(v) / len(v)
fe, wr = res.get("FETCH_SIZE", 0), res.get("WRITE_SIZE", 0)
row = {"kernel": "conv_ring2_bf16<64, 9, 0>  9x9 s1 64->64 + BN-stats, B=20 128x416, bf16", "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
       "traffic_bytes_per_launch": int((2 * fe + wr) * 1024),
       "algorithmic_bytes_per_launch": 20 * 128 * 416 * 64 * 2 * 2 + 81 * 64 * 64 * 2,
       "mean_duration_us_profiled": round(sum(dur) / max(len(dur), 1), 1)}
for cn in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
           "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"):
    if cn in res:
        row[cn] = round(res[cn])
if res.get("GRBM_GUI_ACTIVE"):
    row["mfma_pipe_busy_pct"] = round(100.0 * res["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * res["GRBM_GUI_ACTIVE"] / 8), 1)
    if dur:
        row["clock_ghz_profiled"] = round(res["GRBM_GUI_ACTIVE"] / 8 / (sum(dur) / len(dur)) / 1e3, 2)
row["note"] = ("traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE reads half of a wide coalesced stream); L2 -> fabric requests, "
               "Infinity-Cache hits included; the weight tiles and the patch re-reads of the nine filter rows are L2 hits and do not appear")
row["command"] = "bash tools/pmc_ring.sh"
row["collected_at"] = os.environ.get("GDN_COMMIT") or None
json.dump(row, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(row, indent=1))
PY
