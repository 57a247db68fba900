#!/bin/bash
# GPU box: the round 6 final records (profiles/r06f_*): whole-step MFMA utilisation, step traces of three configurations with the
# no-MFMA16-beside-FFT check, smoke, default bench line.   usage: GDN_COMMIT=<hash> bash tools/r06_final.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
bash tools/pmc_step.sh > gpurun_out/pmc_step.log 2>&1
bash tools/prof_step.sh r06f_dtod_fp32 > /dev/null 2>&1
bash tools/prof_step.sh r06f_rtod_bf16 --mode RtoD --dtype bf16 > /dev/null 2>&1
bash tools/prof_step.sh r06f_dtod_bf16 --dtype bf16 > /dev/null 2>&1
for t in dtod_fp32 rtod_bf16 dtod_bf16; do echo "== $t"; head -3 gpurun_out/prof_step_r06f_$t/step_summary.txt; cat gpurun_out/prof_step_r06f_$t/mfma16_vs_fft.txt 2>/dev/null | tail -3; done
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
python bench.py > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err
tail -c 6000 gpurun_out/r06f_bench.json
