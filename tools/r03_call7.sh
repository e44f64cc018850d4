#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/diag/residual_dev.py > gpurun_out/r03f_residual_dev.txt 2>&1; cat gpurun_out/r03f_residual_dev.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03f_bench_default.json 2> gpurun_out/r03f_bench_default.err
tail -2 gpurun_out/r03f_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03f_bench_default.json'))
print('static', round(d['value']), round(d['frames_per_s']), d['ms_per_step'], round(d['roofline']['frac'],4), {k:round(v['frac'],3) for k,v in d['roofline']['irls_passes'].items()})
f=d['full_solver']; print('sphere', round(f['value']), round(f['frames_per_s']), f['ms_per_step'], round(f['roofline']['frac'],4))
for q in d['sequences']: print('seq', q['streams_per_gpu'], round(q['value']), round(q['frames_per_s']), q['ms_per_step'], round(q['roofline']['frac'],4))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'], d['full_solver']['cpu_baseline']['value'])
PY
bash tools/stage_counters.sh r03f_frame_sphere frame 5120 sphere throughput > gpurun_out/r03f_sq_counters_sphere.txt 2>&1
tail -40 gpurun_out/r03f_sq_counters_sphere.txt
