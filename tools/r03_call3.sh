#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multi_frame.py tests/test_cluster_timeout.py -m gpu -x -q > gpurun_out/r03c_newtests.log 2>&1
tail -25 gpurun_out/r03c_newtests.log
{
SF_ORACLE_EXACT_WARP=1 SF_ORACLE_EXACT_SUMS=1 timeout 300 python tools/diag/b_summary.py 6 throughput
SF_ORACLE_EXACT_SUMS=1 timeout 300 python tools/diag/b_summary.py 6 throughput
} > gpurun_out/r03c_b_summary.txt 2>&1
cat gpurun_out/r03c_b_summary.txt
for mode in "" "--launch-per-frame"; do
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $mode > gpurun_out/r03c_bench$mode.json 2> gpurun_out/r03c_bench$mode.err
tail -2 gpurun_out/r03c_bench$mode.err
python - "$mode" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r03c_bench%s.json'%sys.argv[1]))
print(sys.argv[1] or 'one-launch', 'static', round(d['value']), round(d['frames_per_s']), d['ms_per_step'], round(d['roofline']['frac'],4))
f=d['full_solver']; print('   sphere', round(f['value']), round(f['frames_per_s']), f['ms_per_step'], round(f['roofline']['frac'],4))
for q in d['sequences']: print('   seq', q['streams_per_gpu'], round(q['value']), round(q['frames_per_s']), q['ms_per_step'], round(q['roofline']['frac'],4), q['iterations_per_frame_spread'])
PY
done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_gputest.log 2>&1
tail -6 gpurun_out/r03c_gputest.log
