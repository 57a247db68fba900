#!/bin/bash
# GPU box: address-translation counters of the frequency-domain kernels (is tile-major access to the [bin][tile][C] spectra TLB-bound?)
tag=${1:-r06}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_fft_tlb_$tag
rm -rf $out; mkdir -p $out
D=$R/tests/diag/fft_train_kernels.py
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $D 3 > $out/p$i.log 2>&1
done
cd $R
python3 - $out > $R/gpurun_out/pmc_fft_tlb_$tag.txt <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
rows = collections.defaultdict(dict)
for p in sorted(glob.glob(out + "/p*/")):
    files = glob.glob(p + "*/*counter_collection.csv") + glob.glob(p + "*counter_collection.csv")
    if not files:
        print("#", p, "no counter file:", open(p.rstrip("/") + ".log").read()[-200:].replace("\n", " | ")); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if re.search(r"fft|cgemm_bins", n):
            agg[(re.sub(r"\(.*", "", n)[:30], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            rows[k][c] = sum(v) / len(v)
cols = sorted({c for d in rows.values() for c in d})
for k in sorted(rows):
    print("%-32s %8d " % k + "  ".join("%s=%.4g" % (c.replace("TCP_UTCL1_", "U1_").replace("_sum", ""), rows[k][c]) for c in cols if c in rows[k]))
PY
cat $R/gpurun_out/pmc_fft_tlb_$tag.txt
