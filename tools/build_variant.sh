#!/bin/bash
# An experimental build of libsf_hip.so with extra compiler flags, beside the product library (A/B runs: SF_HIP_LIB).
#   tools/build_variant.sh NAME "EXTRA FLAGS" [SOURCE_DIR]   ->   staticfusion_amd/csrc/libsf_hip_NAME.so
# SOURCE_DIR defaults to this tree's staticfusion_amd/csrc (give a checkout of another revision to build that one).
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=${2:-}; SRC=${3:-$ROOT/staticfusion_amd/csrc}
OUT=$ROOT/staticfusion_amd/csrc; OBJ=/tmp/sf_variant_$NAME; rm -rf $OBJ; mkdir -p $OBJ
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function $EXTRA"
cd $SRC
for f in sf_hip sf_hip_solver sf_hip_input sf_hip_model; do [ -f $f.hip ] && /opt/rocm/bin/hipcc $FL -c -o $OBJ/$f.o $f.hip & done
/opt/rocm/bin/hipcc $FL -DSF_NT=256 -c -o $OBJ/f256.o sf_frame_kernels.hip &
/opt/rocm/bin/hipcc $FL -DSF_NT=256 ${O5FLAGS:--DSF_OCC=5} -DSF_VARIANT_TAG=256o5 -c -o $OBJ/f256o5.o sf_frame_kernels.hip &
/opt/rocm/bin/hipcc $FL -DSF_NT=1024 -c -o $OBJ/f1024.o sf_frame_kernels.hip &
/opt/rocm/bin/hipcc $FL -DSF_NT=1024 -DSF_CLUSTER=1 -DSF_VARIANT_TAG=cluster -c -o $OBJ/fcl.o sf_frame_kernels.hip &
wait
/opt/rocm/bin/hipcc $FL -shared -o $OUT/libsf_hip_$NAME.so $OBJ/sf_hip*.o $OBJ/f256.o $OBJ/f256o5.o $OBJ/f1024.o $OBJ/fcl.o
echo built $OUT/libsf_hip_$NAME.so
