B="python bench.py --mode RtoD --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-other-configs"
for i in 1 2 3; do
  echo "HEAD: $(GDN_HIP_LIB=$PWD/gdn-pytorch_amd/lib/ab/libgdn_head.so $B 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
  echo "NEW:  $($B 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
done
