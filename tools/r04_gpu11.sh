cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04_t11.log
GDN_COMMIT=c436299 bash tools/pmc_ring.sh > gpurun_out/r04_pmc_ring.log 2>&1
python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
cat gpurun_out/r04_t11.log; tail -30 gpurun_out/r04_pmc_ring.log; head -c 1500 gpurun_out/r04_bench_a.json
