cd $GRAFT_REPO_ROOT
timeout 600 python tests/diag/ring_check.py 2 > gpurun_out/ring_check_b2.txt 2>&1
timeout 600 python tests/diag/ring_probe.py 20 > gpurun_out/ring_probe.txt 2>&1; echo rc=$? >> gpurun_out/ring_probe.txt
timeout 900 python tests/diag/ring_check.py 20 > gpurun_out/ring_check_b20.txt 2>&1
cat gpurun_out/ring_check_b2.txt gpurun_out/ring_probe.txt gpurun_out/ring_check_b20.txt
