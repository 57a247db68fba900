cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_ring2; rm -rf $out; mkdir -p $out
cat > /tmp/drv.py <<'PY'
import sys, pathlib
R = pathlib.Path(sys.argv[1]); sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
op = ops.Conv(64, 64, 9, 1, 4)
x = torch.randn(20, 128, 416, 64, device=dev).bfloat16(); w = (torch.randn(81, 64, 64, device=dev) * 0.02).bfloat16()
for knob in (0, 2, 4):
    for _ in range(3):
        op.fwd(x, w, stats=True, tile_cfg=10 | (knob << 12))
    op.fwd(x, w, stats=False, tile_cfg=9)      # separator (row-patch kernel)
torch.cuda.synchronize()
PY
i=0
for grp in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 /tmp/drv.py $R > $out/p$i.log 2>&1
done
cd $R
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in sorted(glob.glob(out + "/p*/")):
    files = glob.glob(p + "*/*counter_collection.csv") + glob.glob(p + "*counter_collection.csv")
    if not files:
        print(p, "no counter file"); continue
    rows = [r for r in csv.DictReader(open(files[0])) if "conv_r" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # dispatches in order: 3x knob0, sep, 3x knob2, sep, 3x knob4, sep
    disp = sorted({int(r["Dispatch_Id"]) for r in rows})
    label = {}
    ring = [d for d in disp if any("conv_ring" in r["Kernel_Name"] for r in rows if int(r["Dispatch_Id"]) == d)]
    for k, d in enumerate(ring):
        label[d] = ("both", "no_w", "no_a")[min(k // 3, 2)]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d in label:
            agg[r["Counter_Name"]][label[d]].append(float(r["Counter_Value"]))
    for c, dd in agg.items():
        print("%-40s" % c, "  ".join("%s %14.0f" % (k, sum(v) / len(v)) for k, v in dd.items()))
PY
