#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 2400 python tools/diag/sequence_hunt.py --first 8000 --count 600 --size 640x480 --json gpurun_out/r03o_hunt_qvga_s8000_n600.json > gpurun_out/r03o_hunt_qvga.log 2>&1
tail -n 1 gpurun_out/r03o_hunt_qvga.log | cut -c1-600
timeout -k 10 1200 python tools/diag/sequence_hunt.py --first 8000 --count 300 --size 640x480 --no-seg --json gpurun_out/r03o_hunt_qvga_noseg_s8000_n300.json > gpurun_out/r03o_hunt_qvga_noseg.log 2>&1
tail -n 1 gpurun_out/r03o_hunt_qvga_noseg.log | cut -c1-600
