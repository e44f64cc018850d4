#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS / busy cycles of one full-frame launch for a given build of the library (run on the GPU box).
# usage: tools/diag/valu_count.sh <lib.so> [workload] [batch]
LIB=$1; WL=${2:-sphere}; B=${3:-4096}
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=/tmp/valu_$$; rm -rf $OUT; mkdir -p $OUT
SF_HIP_LIB=$LIB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc1 -o pmc -- python tools/stage_counters.py --stage frame --batch $B --reps 3 --workload $WL > $OUT/log 2>&1
python tools/stage_counters_sum.py $OUT 3 | python -c "
import sys, json
j = json.load(sys.stdin)
print('$LIB $WL: per stream VALU %.1fk SALU %.1fk LDS %.1fk VMEM rd %.1fk wr %.1fk' % tuple(j[k] / $B / 1e3 for k in ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR']))
"
