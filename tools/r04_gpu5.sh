cd $GRAFT_REPO_ROOT
timeout 600 python tests/diag/ring_probe.py 20 > gpurun_out/ring_probe.txt 2>&1; echo rc=$? >> gpurun_out/ring_probe.txt
cat gpurun_out/ring_probe.txt
