#!/bin/bash
# VALU / SALU / LDS / VMEM instruction counts per stream of the separately launchable stage groups (run on the GPU box)
# usage: tools/diag/valu_by_stage.sh [workload] [batch]
WL=${1:-sphere}; B=${2:-5120}
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for st in pyramid kmeans solver residuals frame; do
  OUT=/tmp/vbs_$st; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc -- python tools/stage_counters.py --stage $st --batch $B --reps 3 --workload $WL > $OUT/log 2>&1
  python tools/stage_counters_sum.py $OUT 3 | python -c "
import sys, json
j = json.load(sys.stdin)
print('%-10s per stream: VALU %7.1fk SALU %6.1fk LDS %6.1fk VMEM rd %5.1fk wr %5.1fk' % (('$st',) + tuple(j[k] / $B / 1e3 for k in ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR'])))
"
done
