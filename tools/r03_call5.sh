#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/diag/behind_camera_case.py throughput 2>&1 | tail -40
