#!/bin/bash
# GPU box: PMC passes (separate runs, kernel-trace only) of the bf16 x 3 GEMM kernels on the level-3 shape, B=20 -> JSON summary
# (gpurun_out/pmc_x3/summary.json; copy to profiles/rNN_gemm_x3_pmc.json).  FETCH_SIZE is doubled per the gfx950 correction.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_x3
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $R/tests/diag/x3_pmc_driver.py > $out/$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/SQ -- python3 $R/tests/diag/x3_pmc_driver.py > $out/SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/SQ2 -- python3 $R/tests/diag/x3_pmc_driver.py > $out/SQ2.log 2>&1
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
res = collections.defaultdict(dict)
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2"):
    files = sorted(glob.glob(out + "/" + c + "/*/*counter_collection.csv") + glob.glob(out + "/" + c + "/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[-1])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
        k = (name, int(r["Grid_Size"]))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if c == "SQ" and r["Counter_Name"] == "SQ_WAVES":
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in agg.items():
        for cn, v in d.items():
            res[k][cn] = sum(v) / len(v)
rows = {}
for k, d in sorted(res.items()):
    if "x3" not in k[0]:
        continue
    fe, wr = d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0)
    row = {"FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1), "traffic_bytes_per_launch": int((2 * fe + wr) * 1024),
           "mean_duration_us_profiled": round(sum(dur[k]) / max(len(dur[k]), 1), 1)}
    for cn in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS",
               "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"):
        if cn in d:
            row[cn] = d[cn]
    if d.get("GRBM_GUI_ACTIVE"):
        row["mfma_pipe_busy_pct"] = round(100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128 * d["GRBM_GUI_ACTIVE"]), 1)
        row["clock_ghz_est"] = round(d["GRBM_GUI_ACTIVE"] / 8 / (row["mean_duration_us_profiled"] * 1e3), 2)
    rows["%s grid %d" % k] = row
json.dump(rows, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
