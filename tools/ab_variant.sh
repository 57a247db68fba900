#!/bin/bash
# Build an A/B variant of the library from an alternative source of ONE translation unit:
#   tools/ab_variant.sh <name> <unit, e.g. conv_fft> <path/to/alternative.hip> [extra compiler flags, e.g. -DGDN_X=1]
#                                                                                  -> gdn-pytorch_amd/lib/ab/libgdn_<name>.so
# run with GDN_HIP_LIB=gdn-pytorch_amd/lib/ab/libgdn_<name>.so (the main library must be built first: its other objects are linked in)
set -eu
name=$1; unit=$2; src=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/gdn-pytorch_amd/lib
mkdir -p $L/ab
cp "$src" $L/ab/${unit}_$name.hip
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -I$R/include -I$R/gdn-pytorch_amd/csrc "$@" -c $L/ab/${unit}_$name.hip -o $L/ab/${unit}_$name.o
objs=$(ls $L/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/ab/libgdn_$name.so $objs $L/ab/${unit}_$name.o
rm -f $L/ab/${unit}_$name.hip
echo $L/ab/libgdn_$name.so
