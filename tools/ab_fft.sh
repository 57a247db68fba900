#!/bin/bash
# GPU box: per-kernel times of the frequency-domain layers on their training plans (tests/diag/fft_train_kernels.py under
# rocprofv3 --kernel-trace), the library as built against A/B variants under gdn-pytorch_amd/lib/ab/ (tools/ab_variant.sh).
#   usage: ab_fft.sh <tag> [variant ...]        -> gpurun_out/ab_fft_<tag>.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/ab_fft_$tag.txt
: > $out
for v in main "$@"; do
  if [ "$v" = main ]; then unset GDN_HIP_LIB; else export GDN_HIP_LIB=$R/gdn-pytorch_amd/lib/ab/libgdn_$v.so; fi
  for rep in $(seq 1 ${AB_REPS:-2}); do
    bash $R/tools/prof_diag.sh abfft_${v}_$rep tests/diag/fft_train_kernels.py 10 > /dev/null 2>&1
    echo "=== $v (run $rep)" >> $out
    cat $R/gpurun_out/prof_diag_abfft_${v}_$rep/stdout.txt >> $out
    grep -E "fft2d|ifft|gather" $R/gpurun_out/prof_diag_abfft_${v}_$rep/by_kernel_and_grid.txt | awk '{printf "%-44s %-14s %5s %9s\n", $1" "$2, $3, $4, $5}' >> $out
  done
done
cat $out
