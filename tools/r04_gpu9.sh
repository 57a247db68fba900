cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_bf16.py -x -q -k "ring or fwd_dgrad" 2>&1 | tail -15 > gpurun_out/r04_t9.log
timeout 900 python tests/diag/ring_check.py 20 > gpurun_out/ring_check_b20.txt 2>&1
GDN_RING_TAIL=0 timeout 900 python tests/diag/ring_check.py 20 > gpurun_out/ring_check_b20_notail.txt 2>&1
cat gpurun_out/r04_t9.log gpurun_out/ring_check_b20.txt; echo "--- GDN_RING_TAIL=0"; cat gpurun_out/ring_check_b20_notail.txt
