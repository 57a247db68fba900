cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "winoconv or wino2 or gemm_x3 or data_parallel or bench_gpus or rccl or adam or accumulation or train_step_b20" 2>&1 | tail -8 > gpurun_out/r04_t1.log
for cfg in "dtod_fp32:" "rtod_fp32:--mode RtoD" "rtod_bf16:--mode RtoD --dtype bf16" "dtod_fp32_graph:--graph" "dtod_bf16:--dtype bf16"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  bash tools/prof_step.sh r04a_$tag $a > gpurun_out/r04a_prof_$tag.log 2>&1
  echo "$tag rc=$?" >> gpurun_out/r04_t1.log
done
cat gpurun_out/r04_t1.log
cat gpurun_out/prof_step_r04a_*/mfma16_vs_fft.txt
