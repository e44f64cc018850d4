#!/bin/bash
# SQ counters of one stage (see tools/stage_counters.py). Run ON THE GPU BOX via gpurun.
# usage: tools/stage_counters.sh <tag> <stage> <batch> [workload] [variant]
set -u
TAG=$1; STAGE=$2; B=$3; WL=${4:-sphere}; VAR=${5:-throughput}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ctr_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/stage_counters.py --stage $STAGE --batch $B --reps 3 --workload $WL --variant $VAR"
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1
done
python tools/stage_counters_sum.py $OUT 3 > /dev/null
python - <<PY
import json
r = json.load(open("$OUT/counters.json")); r["batch"] = $B; r["workload"] = "$WL"; r["variant"] = "$VAR"
json.dump(r, open("$OUT/counters.json", "w"), indent=1); print(json.dumps(r, indent=1))
PY
