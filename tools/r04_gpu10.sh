cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_bf16.py -x -q 2>&1 | tail -5 > gpurun_out/r04_t10.log
for cfg in "rtod_bf16:--mode RtoD --dtype bf16" "dtod_bf16:--dtype bf16"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  bash tools/prof_step.sh r04c_$tag $a > gpurun_out/r04c_prof_$tag.log 2>&1
  echo "$tag rc=$?" >> gpurun_out/r04_t10.log
  head -c 300 gpurun_out/prof_step_r04c_$tag/bench.json >> gpurun_out/r04_t10.log; echo >> gpurun_out/r04_t10.log
done
python bench.py --mode infer --dtype bf16 --steps 3 --warmup 1 2>/dev/null | head -c 400 >> gpurun_out/r04_t10.log
cat gpurun_out/r04_t10.log
cat gpurun_out/prof_step_r04c_rtod_bf16/step_summary.txt
head -24 gpurun_out/prof_step_r04c_rtod_bf16/by_kernel_and_grid.txt
