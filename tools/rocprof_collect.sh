#!/bin/bash
# Collect the rocprofv3 evidence for one workload: kernel-trace stats + HBM byte counters
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes). Run ON THE GPU BOX via gpurun.
# usage: tools/rocprof_collect.sh <tag> <workload> <batch>
set -u
TAG=$1; WL=$2; B=$3
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export SF_TIMED_LAUNCH_PER_FRAME=1  # bytes per FRAME of every stream: one launch per frame under the counters
# (sequences with SF_PROF_ONE_LAUNCH=K in the environment: ONE launch of K frames, the bench's own shape -- tools/measure_traffic.sh)
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/prof_run.py --workload $WL --batch $B --steps 3"
# default output of this rocprofv3 is a rocpd sqlite database; tools/rocprof_summarise.py reads it
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
# (round 6) what the kernel is bound by when it is not HBM: vector-ALU issue. SQ counters in a pass of their own, the GPU-active
# cycles (GRBM) in another -- SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md): busy = 4 x ACTIVE_INST_VALU / (cycles x SIMDs)
if [ "${SF_PROF_SQ:-0}" = "1" ]; then
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/grbm -o grbm -- $CMD > $OUT/grbm.log 2>&1
fi
find $OUT -name "*.csv" | head -20
