#!/bin/bash
# one rocprofv3 --pmc pass over a full-frame launch with the given counters; prints per-launch sums (run on the GPU box)
# usage: tools/diag/pmc_probe.sh <workload> <batch> COUNTER [COUNTER ...]
WL=$1; B=$2; shift 2
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=/tmp/pmcp_$$; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc "$@" -d $OUT/pmc1 -o pmc -- python tools/stage_counters.py --stage frame --batch $B --reps 3 --workload $WL > $OUT/log 2>&1 || tail -5 $OUT/log
python tools/stage_counters_sum.py $OUT 3
