#!/bin/bash
# more parity evidence on the final build (GPU box): a soak of the multi-frame hand-over, fresh hunt seeds, a parameter sweep
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03h}
timeout -k 10 900 python tools/diag/multi_frame_soak.py 12 4000 12 > gpurun_out/${T}_multi_frame_soak.txt 2>&1; tail -3 gpurun_out/${T}_multi_frame_soak.txt
timeout -k 10 900 python tools/diag/sequence_hunt.py --first 40000 --count 2000 --json gpurun_out/${T}_hunt_160x120_s40000_n2000.json > gpurun_out/${T}_hunt_s40000.log 2>&1
tail -n 1 gpurun_out/${T}_hunt_s40000.log | cut -c1-700
timeout -k 10 900 python tools/diag/sequence_hunt.py --first 9000 --count 200 --size 640x480 --json gpurun_out/${T}_hunt_qvga_s9000_n200.json > gpurun_out/${T}_hunt_qvga_s9000.log 2>&1
tail -n 1 gpurun_out/${T}_hunt_qvga_s9000.log | cut -c1-700
SF_SWEEP_CASES=1300:1800 timeout -k 10 1500 python -m pytest tests/test_gpu_parity_sweep.py -m gpu -q -k test_random_parameter_sweep > gpurun_out/${T}_sweep_1300_1800.log 2>&1
tail -15 gpurun_out/${T}_sweep_1300_1800.log | cut -c1-300
