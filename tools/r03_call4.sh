#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
C=staticfusion_amd/csrc
{
for r in 1 2; do for w in sphere static; do for lib in libsf_hip_prev.so libsf_hip.so libsf_hip_slownorm.so; do
  echo -n "$lib: "; SF_HIP_LIB=$PWD/$C/$lib timeout 200 python tools/stage_profile.py --batch 5120 --workload $w --steps 6 | grep workload
done; done; done
for r in 1 2; do for o in 4 5; do echo -n "static wg/cu $o: "; SF_THROUGHPUT_WG_PER_CU=$o timeout 200 python tools/stage_profile.py --batch 5120 --workload static --steps 6 | grep workload; done; done
} > gpurun_out/r03d_ab.txt 2>&1
cat gpurun_out/r03d_ab.txt
timeout 900 python -m pytest tests/test_multi_frame.py tests/test_gpu_edge_rules.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03d_tests.log 2>&1
tail -8 gpurun_out/r03d_tests.log
