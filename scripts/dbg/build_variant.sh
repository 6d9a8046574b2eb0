#!/bin/bash
# build_variant.sh NAME "-DFLAGS": a timing variant of the chain kernels (results are WRONG) into scripts/dbg/exp/libqcnet_NAME.so
set -e
cd "$(dirname "$0")/../../tempenslearnagg-qcnet_amd/csrc"
OUT=../../scripts/dbg/exp
mkdir -p $OUT
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -I../../include -I. $2"
/opt/rocm/bin/hipcc $FL -fno-slp-vectorize -c fourier.hip -o $OUT/fourier_$1.o &
/opt/rocm/bin/hipcc $FL -c nodeops.hip -o $OUT/nodeops_$1.o &
wait
OBJS=$(ls *.o | grep -v -e '^fourier.o$' -e '^nodeops.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libqcnet_$1.so $OBJS $OUT/fourier_$1.o $OUT/nodeops_$1.o
rm -f $OUT/fourier_$1.o $OUT/nodeops_$1.o
