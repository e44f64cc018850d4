#!/bin/bash
# round 3, first GPU call: A/B of the pass-1 flush / label-load fix / accumulator scope, b distance per build, the GPU
# test suite on the new build, the hunts with JSON records
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
C=staticfusion_amd/csrc
{
for lib in libsf_hip.so libsf_hip_k32.so libsf_hip_noflush.so libsf_hip_agent.so; do
  SF_HIP_LIB=$PWD/$C/$lib timeout 300 python tools/diag/b_summary.py 6 throughput
done
SF_HIP_LIB=$PWD/$C/libsf_hip.so timeout 300 python tools/diag/b_summary.py 6 latency
SF_HIP_LIB=$PWD/$C/libsf_hip.so timeout 300 python tools/diag/b_summary.py 6 cluster
for r in 1 2; do for w in static sphere; do for lib in libsf_hip_head.so libsf_hip.so libsf_hip_k32.so libsf_hip_noflush.so libsf_hip_agent.so; do
  echo -n "$lib: "; SF_HIP_LIB=$PWD/$C/$lib timeout 200 python tools/stage_profile.py --batch 5120 --workload $w | grep workload
done; done; done
for w in static sphere; do SF_HIP_LIB=$PWD/$C/libsf_hip.so timeout 200 python tools/stage_profile.py --batch 5120 --workload $w; done
} > gpurun_out/r03a_ab.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_gputest.log 2>&1
tail -5 gpurun_out/r03a_gputest.log
timeout 900 python tools/diag/sequence_hunt.py --first 5000 --count 240 --json gpurun_out/r03a_hunt_160x120_s5000_n240.json > gpurun_out/r03a_hunt_s5000.log 2>&1
timeout 1500 python tools/diag/sequence_hunt.py --first 20000 --count 1000 --json gpurun_out/r03a_hunt_160x120_s20000_n1000.json > gpurun_out/r03a_hunt_s20000.log 2>&1
timeout 900 python tools/diag/sequence_hunt.py --first 7000 --count 60 --size 640x480 --json gpurun_out/r03a_hunt_qvga_s7000_n60.json > gpurun_out/r03a_hunt_qvga.log 2>&1
tail -1 gpurun_out/r03a_hunt_s5000.log gpurun_out/r03a_hunt_s20000.log gpurun_out/r03a_hunt_qvga.log | cut -c1-600
cat gpurun_out/r03a_ab.txt
