cd $GRAFT_REPO_ROOT
timeout 600 python tests/diag/ring_probe.py 20 > gpurun_out/ring_probe.txt 2>&1; echo rc=$? >> gpurun_out/ring_probe.txt
cat gpurun_out/ring_probe.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_ring; rm -rf $out; mkdir -p $out
cat > /tmp/drv.py <<'PY'
import sys, pathlib
R = pathlib.Path(sys.argv[1]); sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
op = ops.Conv(64, 64, 9, 1, 4)
x = torch.randn(20, 128, 416, 64, device=dev).bfloat16(); w = (torch.randn(81, 64, 64, device=dev) * 0.02).bfloat16()
for cfg in (9, 10):
    for _ in range(3):
        op.fwd(x, w, stats=True, tile_cfg=cfg)
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 /tmp/drv.py $R > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/p2 -- python3 /tmp/drv.py $R > $out/p2.log 2>&1
cd $R
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    files = glob.glob(out + "/" + p + "/*/*counter_collection.csv") + glob.glob(out + "/" + p + "/*counter_collection.csv")
    if not files:
        print(p, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "conv_r" in k:
            print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
