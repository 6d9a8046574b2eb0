#!/bin/bash
# build_all_variant.sh NAME "-DFLAGS": every source of the library rebuilt with FLAGS into scripts/dbg/exp/libqcnet_NAME.so
# (e.g. -DQCN_DEBUG_KNOBS: the A/B knobs of the launchers; load with QCN_HIP_LIB=... / QCN_LIB=...)
set -e
cd "$(dirname "$0")/../../tempenslearnagg-qcnet_amd/csrc"
OUT=../../scripts/dbg/exp
mkdir -p $OUT/obj_$1
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -I../../include -I. $2"
for f in *.hip; do
  X=""; [ $f = fourier.hip ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $FL $X -c $f -o $OUT/obj_$1/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libqcnet_$1.so $OUT/obj_$1/*.o
rm -rf $OUT/obj_$1
