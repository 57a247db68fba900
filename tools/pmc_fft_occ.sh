#!/bin/bash
# GPU box: occupancy / memory-unit stall derived metrics of the frequency-domain kernels (separate passes) -> gpurun_out/pmc_fft_occ_<tag>.txt
tag=${1:-r06}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_fft_occ_$tag
rm -rf $out; mkdir -p $out
D=$R/tests/diag/fft_train_kernels.py
i=0
for set in ${PMC_SETS:+"__custom__"} "MeanOccupancyPerCU" "MemUnitStalled" "MemUnitBusy" "WriteUnitStalled" "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "SPI_RA_RES_STALL_CSN SPI_RA_WVLIM_STALL_CSN SPI_RA_TMP_STALL_CSN GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $D 3 > $out/p$i.log 2>&1
done
cd $R
python3 - $out > $R/gpurun_out/pmc_fft_occ_$tag.txt <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
rows = collections.defaultdict(dict)
for p in sorted(glob.glob(out + "/p*/")):
    files = glob.glob(p + "*/*counter_collection.csv") + glob.glob(p + "*counter_collection.csv")
    if not files:
        print("#", p, "no counter file:", open(p.rstrip("/") + ".log").read()[-300:].replace("\n", " | ")); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if re.search(r"fft|cgemm", n):
            agg[(re.sub(r"\(.*", "", n)[:34], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            rows[k][c] = sum(v) / len(v)
cols = sorted({c for d in rows.values() for c in d})
print("%-36s %9s " % ("kernel", "grid") + " ".join("%14s" % c[:14] for c in cols))
for k in sorted(rows):
    print("%-36s %9d " % k + " ".join("%14.4g" % rows[k].get(c, float("nan")) for c in cols))
PY
cat $R/gpurun_out/pmc_fft_occ_$tag.txt
