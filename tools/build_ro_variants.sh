#!/bin/bash
# The attribution builds of profiles/PARITY.md (round 4): the reference-order build (sf_reforder.h) with ONE of the product's
# arithmetic shortcuts switched back on each -> staticfusion_amd/csrc/libsf_hip_ro_<name>.so (git-ignored; they travel to the
# GPU box with the snapshot). tools/diag/attribution_hunt.py runs them side by side on the same seeds.
set -e
cd "$(dirname "$0")/../staticfusion_amd/csrc"
build() {  # name, make variables...
    local name=$1; shift
    make -j8 RO_TAG=ro_$name "$@" libsf_hip_ro_$name.so > /dev/null
    echo "built libsf_hip_ro_$name.so ($*)"
}
build splat_int        RO_FLAGS="-DSF_RO_SPLAT=0"
build splat_int_fastdiv RO_FLAGS="-DSF_RO_SPLAT=0" RO_FAST_NORMALISE=1
build splat_int_fine   RO_FLAGS="-DSF_RO_SPLAT_MIN_LEVEL=2"   # integer sums at image levels 0, 1 only (ordered floats at the coarse levels)
build splat_int_coarse RO_FLAGS="-DSF_RO_SPLAT_MAX_LEVEL=1"   # integer sums at image levels >= 2 only
build rows_fact        RO_FLAGS="-DSF_RO_ROWS=0"
build rows_fact_fma    RO_FLAGS="-DSF_RO_ROWS=0" RO_ROWS_FMA=1
build fast_weights     RO_FAST_WEIGHTS=1
build p1_fp32          RO_FLAGS="-DSF_RO_P1_FP64=0"
build labsum_int       RO_FLAGS="-DSF_RO_LABSUM=0"
build fp64_lane_sums   RO_FLAGS="-DSF_RO_SEQ64=0"   # the [C1] sums as per-lane partial sums: 1 frame of 73 600 differs in the last bit (PARITY.md)
build jacobi_rr        RO_FLAGS="-DSF_RO_JACOBI=0"
build init_res         RO_FLAGS="-DSF_RO_INIT_RES=0"
build behind           RO_FLAGS="-DSF_RO_BEHIND=0"
build all_shortcuts    RO_FLAGS="-DSF_RO_SPLAT=0 -DSF_RO_ROWS=0 -DSF_RO_P1_FP64=0 -DSF_RO_LABSUM=0 -DSF_RO_JACOBI=0 -DSF_RO_INIT_RES=0 -DSF_RO_BEHIND=0" RO_FAST_WEIGHTS=1 RO_ROWS_FMA=1 RO_FAST_NORMALISE=1
# the product's arithmetic everywhere EXCEPT the warp / residual splat of the coarse levels (image levels >= 2: 6 300 of 102 300 pixels at
# QVGA), which adds the reference's floats in the reference's order: what an ordered coarse splat in the product would buy
build all_but_coarse_splat RO_FLAGS="-DSF_RO_SPLAT_MIN_LEVEL=2 -DSF_RO_ROWS=0 -DSF_RO_P1_FP64=0 -DSF_RO_LABSUM=0 -DSF_RO_JACOBI=0 -DSF_RO_INIT_RES=0 -DSF_RO_BEHIND=0" RO_FAST_WEIGHTS=1 RO_ROWS_FMA=1 RO_FAST_NORMALISE=1
# round 5: the floor an ordered splat at ALL levels could reach -- every product shortcut on EXCEPT the
# integer splat -- and the same with the integer splat kept at image level 0 only (ordered floats at levels >= 1)
build all_but_splat        RO_FLAGS="-DSF_RO_ROWS=0 -DSF_RO_P1_FP64=0 -DSF_RO_LABSUM=0 -DSF_RO_JACOBI=0 -DSF_RO_INIT_RES=0 -DSF_RO_BEHIND=0" RO_FAST_WEIGHTS=1 RO_ROWS_FMA=1 RO_FAST_NORMALISE=1
build all_but_splat_above0 RO_FLAGS="-DSF_RO_SPLAT_MIN_LEVEL=1 -DSF_RO_ROWS=0 -DSF_RO_P1_FP64=0 -DSF_RO_LABSUM=0 -DSF_RO_JACOBI=0 -DSF_RO_INIT_RES=0 -DSF_RO_BEHIND=0" RO_FAST_WEIGHTS=1 RO_ROWS_FMA=1 RO_FAST_NORMALISE=1
