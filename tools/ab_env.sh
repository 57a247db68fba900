#!/bin/bash
# GPU box: A/B of one environment switch on a bench configuration, interleaved, 3 rounds.
#   tools/ab_env.sh VAR "<bench args>"      e.g. tools/ab_env.sh GDN_FUSE_UP2X_BF16 "--mode RtoD --dtype bf16"
var=$1; shift
B="python bench.py $* --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-other-configs"
ms() { python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])'; }
for i in 1 2 3; do
  echo "$var=0: $(env $var=0 $B 2>/dev/null | ms)"
  echo "$var=1: $(env $var=1 $B 2>/dev/null | ms)"
done
