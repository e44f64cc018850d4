#!/bin/bash
# A/B two builds of libsf_hip.so on the same box, interleaved: tools/ab.sh <a.so> <b.so> [workload] [rounds]
A=$1; B=$2; WL=${3:-static}; R=${4:-3}
for i in $(seq $R); do
  for lib in $A $B; do
    echo -n "$lib: "; SF_HIP_LIB=$lib timeout 100 python tools/stage_profile.py --batch ${BATCH:-4096} --workload $WL | grep workload
  done
done
