#!/bin/bash
# second half of tools/final_measure.sh (round 3: the first run was cut by its time limit after a profiling step hung)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out profiles
T=${1:-r03g}
export TMPDIR=/tmp
{ for w in sphere static; do timeout -k 10 300 python tools/stage_profile.py --workload $w --batch 5120 --variant throughput --steps 10 | grep -v "  0.0 us"; done; } > gpurun_out/${T}_stage_profiles.txt 2>&1
timeout -k 10 300 python tools/pass_microbench.py --batch 512 > gpurun_out/${T}_pass_microbench_b512.txt 2>&1
timeout -k 10 600 python tools/parity_report.py --out gpurun_out/${T}_parity_report.json > gpurun_out/${T}_parity_report.md 2>&1
{ timeout -k 10 300 python tools/diag/b_summary.py 6 throughput; SF_ORACLE_EXACT_WARP=1 timeout -k 10 300 python tools/diag/b_summary.py 6 throughput; SF_ORACLE_EXACT_WARP=1 timeout -k 10 300 python tools/diag/b_summary.py 6 cluster; } > gpurun_out/${T}_b_summary.txt 2>&1
timeout -k 10 600 python tools/diag/sequence_hunt.py --first 5000 --count 240 --json gpurun_out/${T}_hunt_160x120_s5000_n240.json > gpurun_out/${T}_hunt_s5000.log 2>&1
timeout -k 10 900 python tools/diag/sequence_hunt.py --first 20000 --count 1000 --json gpurun_out/${T}_hunt_160x120_s20000_n1000.json > gpurun_out/${T}_hunt_s20000.log 2>&1
timeout -k 10 600 python tools/diag/sequence_hunt.py --first 7000 --count 60 --size 640x480 --json gpurun_out/${T}_hunt_qvga_s7000_n60.json > gpurun_out/${T}_hunt_qvga.log 2>&1
timeout -k 10 600 python tools/diag/sequence_hunt.py --first 7000 --count 60 --size 640x480 --no-seg --json gpurun_out/${T}_hunt_qvga_noseg_s7000_n60.json > gpurun_out/${T}_hunt_qvga_noseg.log 2>&1
for f in s5000 s20000 qvga qvga_noseg; do tail -n 1 gpurun_out/${T}_hunt_$f.log | cut -c1-600; done
timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputest.log 2>&1
tail -3 gpurun_out/${T}_gputest.log
cp gpurun_out/traffic_static_b16384.json gpurun_out/traffic_sphere_b16384.json profiles/ 2>/dev/null
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequences > gpurun_out/${T}_bench_with_traffic.json 2> gpurun_out/${T}_bench_with_traffic.err
python - "$T" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench_with_traffic.json" % sys.argv[1]).read().strip().splitlines()[-1]); fs = d["full_solver"]
print("with traffic: static", round(d["roofline"]["frac"], 4), d["roofline"]["traffic_provenance"], "| sphere", round(fs["roofline"]["frac"], 4), fs["roofline"]["traffic_provenance"])
PY
bash tools/traffic_by_stage.sh sphere 4096 > gpurun_out/${T}_traffic_by_stage_sphere.txt 2>&1
bash tools/traffic_by_stage.sh static 4096 > gpurun_out/${T}_traffic_by_stage_static.txt 2>&1
tail -9 gpurun_out/${T}_traffic_by_stage_sphere.txt
# last, so that nothing depends on it: the sequences workload under the counters
timeout -k 10 900 bash tools/measure_traffic.sh sequences 4096 > gpurun_out/${T}_traffic_sequences_b4096.log 2>&1
tail -2 gpurun_out/${T}_traffic_sequences_b4096.log | cut -c1-400
timeout -k 10 900 bash tools/measure_traffic.sh sequences 16384 > gpurun_out/${T}_traffic_sequences_b16384.log 2>&1
tail -2 gpurun_out/${T}_traffic_sequences_b16384.log | cut -c1-400
