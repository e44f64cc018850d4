#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multi_frame.py tests/test_gpu_edge_rules.py tests/test_cluster_timeout.py -m gpu -x -q > gpurun_out/r03e_tests.log 2>&1
tail -6 gpurun_out/r03e_tests.log
for b in 16384 4096; do for f in 0 1; do
  if [ $f = 1 ]; then export SF_NO_PYRAMID_FLIP=1; else unset SF_NO_PYRAMID_FLIP; fi
  timeout 600 python bench.py --workload sequences --batch $b --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seq batch $b noflip=$f', round(d['value']), round(d['frames_per_s']), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
done; done
unset SF_NO_PYRAMID_FLIP
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03e_gputest.log 2>&1
tail -5 gpurun_out/r03e_gputest.log
