#!/bin/bash
# GPU box: the round's final records (profiles/r04f_*): whole-step MFMA utilisation, step traces of five configurations, default bench line
cd /root/repo
GDN_COMMIT=22d62b5 bash tools/pmc_step.sh > gpurun_out/pmc_step.log 2>&1
bash tools/prof_step.sh r04f_dtod_fp32 > /dev/null 2>&1
bash tools/prof_step.sh r04f_rtod_bf16 --mode RtoD --dtype bf16 > /dev/null 2>&1
bash tools/prof_step.sh r04f_dtod_bf16 --dtype bf16 > /dev/null 2>&1
bash tools/prof_step.sh r04f_rtod_fp32 --mode RtoD > /dev/null 2>&1
bash tools/prof_step.sh r04f_dtod_fp32_graph --graph > /dev/null 2>&1
for t in dtod_fp32 rtod_bf16 dtod_bf16 rtod_fp32 dtod_fp32_graph; do echo "== $t"; head -3 gpurun_out/prof_step_r04f_$t/step_summary.txt; cat gpurun_out/prof_step_r04f_$t/mfma16_vs_fft.txt; done
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
python bench.py > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err
tail -c 3000 gpurun_out/r04f_bench.json
