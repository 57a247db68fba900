#!/bin/bash
# GPU box: HBM-side traffic of the frequency-domain kernels (separate PMC passes, kernel-trace only), level-0 9x9 shape
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_fft
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $R/tests/diag/fft_kernels_time.py > $out/$c.log 2>&1
done
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(out + "/" + c + "/*/*counter_collection.csv")
    if not files:
        print(c, "no counter file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        agg[(name.split("(")[0][:40], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = sum(v) / len(v)
print("kernel | grid | FETCH_SIZE KB (raw; x2 on gfx950 per the guide) | WRITE_SIZE KB")
for k, d in sorted(res.items()):
    if "fft" in k[0] or "gemm" in k[0]:
        print("%-60s %-10s fetch %.0f KB  write %.0f KB" % (k[0], k[1], d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0)))
PY
