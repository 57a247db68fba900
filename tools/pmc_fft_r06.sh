#!/bin/bash
# GPU box: what bounds the frequency-domain TRANSFORM kernels of the headline step?  Separate rocprofv3 PMC passes
# (kernel-trace only) over tests/diag/fft_train_kernels.py: issue / wait counters, LDS, FETCH_SIZE, WRITE_SIZE (+ an unprofiled
# duration pass) -> gpurun_out/pmc_fft_r06/summary.json (copy to profiles/r06_fft_pmc.json).   usage: pmc_fft_r06.sh [tag]
tag=${1:-r06}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_fft_$tag
rm -rf $out; mkdir -p $out
D=$R/tests/diag/fft_train_kernels.py
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $D 5 > $out/t.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 $D 3 > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $out/p2 -- python3 $D 3 > $out/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p3 -- python3 $D 3 > $out/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p4 -- python3 $D 3 > $out/p4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --output-format csv -d $out/p5 -- python3 $D 3 > $out/p5.log 2>&1
cd $R
python3 tools/pmc_fft_summary.py $out > $out/summary.txt
cat $out/summary.txt
