cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep "Counter_Name" | awk '{print $3}' | grep -i "^TA_\|^TCP_\|^TCC_\|^TD_\|LDS\|VMEM" | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
cat $GRAFT_REPO_ROOT/gpurun_out/counters.txt | head -c 8000
