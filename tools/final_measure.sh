set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
T=${1:-r02e}
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rocprof_bench -o bench -- python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_bench_under_rocprof.err
find gpurun_out/${T}_rocprof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_rocprofv3_stats_bench.csv \;
timeout 900 python bench.py --workload sequences --batch 4096 --steps 10 --warmup 2 > gpurun_out/${T}_bench_sequences_b4096.json 2> gpurun_out/${T}_bench_sequences.err
timeout 900 python bench.py --workload sequences --batch 16384 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_sequences_b16384.json 2>> gpurun_out/${T}_bench_sequences.err
for b in 1 8; do timeout 300 python bench.py --workload sequences --variant cluster --batch $b --steps 50 --warmup 5 --no-cpu-baseline 2>> gpurun_out/${T}_bench_sequences.err | tail -1; done > gpurun_out/${T}_bench_sequences_cluster_b1_b8.json
{ for v in cluster latency throughput; do for w in sphere static; do timeout 100 python tools/stage_profile.py --workload $w --batch 1 --variant $v --steps 30 | grep -v "  0.0 us"; done; done; for b in 2 4 8; do timeout 100 python tools/stage_profile.py --workload sphere --batch $b --variant cluster --steps 30 | grep workload; done; for w in sphere static; do timeout 200 python tools/stage_profile.py --workload $w --batch 4096 --variant throughput | grep -v "  0.0 us"; done; } > gpurun_out/${T}_stage_profiles.txt 2>&1
timeout 600 python tools/parity_report.py --out gpurun_out/${T}_parity_report.json > gpurun_out/${T}_parity_report.md 2>&1
python -c "
import json
d=json.load(open('gpurun_out/${T}_bench_default.json')); f=d['full_solver']
print('static', d['value'], d['frames_per_s'], d['roofline']['frac'], d['roofline']['traffic_provenance'])
print('sphere', f['value'], f['frames_per_s'], f['roofline']['frac'], f['roofline']['traffic_provenance'])
print(d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'])
s=json.load(open('gpurun_out/${T}_bench_sequences_b4096.json')); print('seq', s['value'], s['frames_per_s'], s['iterations_per_frame'], s['iterations_per_frame_spread'], s['roofline']['frac'], s['pose_delta_vs_cpu'])
s=json.load(open('gpurun_out/${T}_bench_sequences_b16384.json')); print('seq16k', s['value'], s['frames_per_s'], s['ms_per_step'], s['roofline']['frac'])
for l in open('gpurun_out/${T}_bench_sequences_cluster_b1_b8.json'):
    s=json.loads(l); print('seq cluster', s['config']['streams_per_gpu'], s['frames_per_s'], s['ms_per_step'])
"
head -5 gpurun_out/${T}_rocprofv3_stats_bench.csv
grep -E "workload" gpurun_out/${T}_stage_profiles.txt
