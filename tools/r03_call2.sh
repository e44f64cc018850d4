#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 300 python tools/diag/b_summary.py 6 throughput
SF_ORACLE_EXACT_WARP=1 timeout 300 python tools/diag/b_summary.py 6 throughput
SF_ORACLE_EXACT_WARP=1 timeout 300 python tools/diag/b_summary.py 6 cluster
} > gpurun_out/r03b_b_summary.txt 2>&1
cat gpurun_out/r03b_b_summary.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_gputest.log 2>&1
tail -15 gpurun_out/r03b_gputest.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r03b_bench_default.json 2> gpurun_out/r03b_bench_default.err
tail -3 gpurun_out/r03b_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03b_bench_default.json'))
print('static', d['value'], d['frames_per_s'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['irls_passes'])
f=d['full_solver']; print('sphere', f['value'], f['frames_per_s'], f['ms_per_step'], f['roofline']['frac'])
for q in d['sequences']: print('seq', q['streams_per_gpu'], q['value'], q['frames_per_s'], q['ms_per_step'], q['roofline']['frac'], q['iterations_per_frame_spread'])
print(d['cpu_baseline']['value'], d['configs_unavailable'][0][:40])
PY
bash tools/traffic_by_stage.sh sphere 4096 > gpurun_out/r03b_traffic_by_stage_sphere.txt 2>&1
cat gpurun_out/r03b_traffic_by_stage_sphere.txt
