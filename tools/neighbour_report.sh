#!/bin/bash
# Evidence for DESIGN.md 2.10 "two processes on one MI355X": which kernels of a NEIGHBOUR process change the results of a
# process that shares the GPU.  Victims: a torch-only process (rocFFT, rocBLAS, elementwise kernels: tests/diag/torch_victim.py),
# this library's frequency-domain layers (tests/diag/fft_neighbour.py), and a whole tiny training run (tests/diag/dp_solo.py).
# Neighbours: tests/diag/mfma_neighbour.hip (a loop of matrix instructions and nothing else), vendor GEMMs, gemm_x3_nt / _tn.
#   bash tools/neighbour_report.sh > profiles/rNN_neighbour_mfma.txt     (about 4 GPU-minutes)
cd "$(dirname "$0")/.."
mkdir -p tests/diag/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tests/diag/_build/libmfma_neighbour.so tests/diag/mfma_neighbour.hip 2>/dev/null
echo "# victim: torch-only process (rocFFT rfft2 / rocBLAS sgemm / elementwise / softmax / sum), 200 repeats compared with the first"
echo "# (MIOpen's fp32 conv2d is left out: it is not reproducible run to run on its own)"
for a in none mfma0 mfma1 mfma2 mfma3 mfma4 mfma5 hugebf16 hugefp32 bignt bigtn; do
  case $a in
    mfma0) d="loop of v_mfma_f32_32x32x16_bf16, 4 accumulators, no barrier";;
    mfma1) d="the same, one accumulator (dependent chain)";;
    mfma2) d="loop of v_mfma_f32_16x16x32_bf16, 4 accumulators";;
    mfma3) d="loop of v_mfma_f32_32x32x2_f32, 4 accumulators";;
    mfma4) d="loop of v_mfma_f32_32x32x16_f16, 4 accumulators";;
    mfma5) d="loop of v_mfma_f32_32x32x16_bf16 with a workgroup barrier every 12 instructions";;
    hugebf16) d="torch.mm 8192^3 bf16 (vendor GEMM)";;
    hugefp32) d="torch.mm 8192^3 fp32 (vendor GEMM)";;
    bignt) d="gemm_x3_nt, 16 bins x 2080 x 512 x 512";;
    bigtn) d="gemm_x3_tn, 16 bins, 2080 rows, 512 x 512";;
    *) d="no neighbour";;
  esac
  echo "== neighbour: $a ($d)"
  python3 tests/diag/torch_victim.py $a 200 2>&1 | grep -v conv_fp32 | tail -4
done
echo
echo "# victim: this library's frequency-domain layers (forward + backward of three layer shapes), 300 repeats"
for a in none bignt mfma5 mfma3; do echo "== neighbour: $a"; python3 tests/diag/fft_neighbour.py $a 300 2>&1 | tail -2; done
echo
echo "# victim: a 6-step training run of the 32 x 64 test model; two INDEPENDENT trainers at once, compared with each alone"
echo "== both with the bf16 x 3 GEMMs"; python3 tests/diag/dp_solo.py 6 8 2>&1 | tail -1
echo "== both without (GDN_X3=0)"; GDN_X3=0 python3 tests/diag/dp_solo.py 6 8 2>&1 | tail -1
echo "== trainer 0 without, trainer 1 with: which of the two changes?"; GDN_X3_R0=0 GDN_X3_R1=1 python3 tests/diag/dp_solo.py 6 8 2>&1 | grep -E "DIFFERS|differ" | sed -E "s/ [0-9.]+ [0-9.]+ [0-9.]+ [0-9.]+'/ ...'/g" | tail -9
