#!/bin/bash
# The evidence bundle of a round, ON THE GPU BOX (gpurun --timeout 3000 -- 'bash tools/final_measure.sh r03g'); tools/collect_bundle.py <tag>
# copies what it leaves under gpurun_out/ into profiles/ and prints the numbers DESIGN.md section 7 quotes. About 15 GPU-minutes.
#   1 PMC traffic per frame of every stream for the four workloads of the bench line (tagged with the source hash and HEAD)
#   2 the driver's command (bench line: configs[1], full solver, sequences at two batch sizes, isolated IRLS passes, CPU legs),
#     then the same command under rocprofv3 --kernel-trace --stats
#   3 stage profiles, bytes per stage group, the isolated passes
#   4 parity: distance report, the hunts (product, precise and reference-order builds side by side) with JSON records, the GPU test suite
# Every step has its own timeout that kills the whole process group (a profiling step that hung in round 3 left orphans that kept
# the GPU busy for everything after it); nothing under rocprofv3 uses a process pool.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out profiles
T=${1:-r06w}
export TMPDIR=/tmp
STEPS=20  # the driver's --steps: the sequences blocks are measured as ONE launch of that many frames, like the bench times them
for spec in "static 16384 0" "sphere 16384 0" "sequences 4096 $STEPS" "sequences 16384 $STEPS"; do
  set -- $spec
  SF_PROF_SQ=1 timeout -k 10 1200 bash tools/measure_traffic.sh $1 $2 throughput $3 > gpurun_out/${T}_traffic_$1_b$2.log 2>&1
done
cp gpurun_out/traffic_*_b*.json profiles/ 2>/dev/null; rm -f profiles/traffic_*_summary.json
timeout -k 10 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
timeout -k 10 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rocprof_bench -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequences > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_bench_under_rocprof.err
find gpurun_out/${T}_rocprof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_rocprofv3_stats_bench.csv \;
{ for w in sphere static; do timeout -k 10 300 python tools/stage_profile.py --workload $w --batch 5120 --variant throughput --steps 10 | grep -v "  0.0 us"; done; } > gpurun_out/${T}_stage_profiles.txt 2>&1
bash tools/traffic_by_stage.sh sphere 4096 > gpurun_out/${T}_traffic_by_stage_sphere.txt 2>&1
bash tools/traffic_by_stage.sh static 4096 > gpurun_out/${T}_traffic_by_stage_static.txt 2>&1
timeout -k 10 300 python tools/pass_microbench.py --batch 512 > gpurun_out/${T}_pass_microbench_b512.txt 2>&1
timeout -k 10 600 python tools/parity_report.py --out gpurun_out/${T}_parity_report.json > gpurun_out/${T}_parity_report.md 2>&1
{ timeout -k 10 300 python tools/diag/b_summary.py 6 throughput; SF_ORACLE_EXACT_WARP=1 timeout -k 10 300 python tools/diag/b_summary.py 6 throughput; SF_ORACLE_EXACT_WARP=1 timeout -k 10 300 python tools/diag/b_summary.py 6 cluster; } > gpurun_out/${T}_b_summary.txt 2>&1
C=staticfusion_amd/csrc
LIBS="product=$C/libsf_hip.so,precise=$C/libsf_hip_precise.so,reforder=$C/libsf_hip_reforder.so"
for b in throughput latency; do
  timeout -k 10 900 python tools/diag/attribution_hunt.py --first 8000 --count 600 --size 640x480 --build $b --libs $LIBS --json gpurun_out/${T}_hunt_qvga_s8000_n600_$b.json > gpurun_out/${T}_hunt_qvga_$b.log 2>&1
  timeout -k 10 900 python tools/diag/attribution_hunt.py --first 50000 --count 5000 --size 320x240 --build $b --libs $LIBS --json gpurun_out/${T}_hunt_160x120_s50000_n5000_$b.json > gpurun_out/${T}_hunt_160x120_$b.log 2>&1
done
timeout -k 10 900 python tools/diag/sequence_hunt.py --first 8000 --count 600 --size 640x480 --builds cluster --json gpurun_out/${T}_hunt_qvga_s8000_n600_cluster.json > gpurun_out/${T}_hunt_qvga_cluster.log 2>&1
timeout -k 10 600 python tools/diag/sequence_hunt.py --first 7000 --count 300 --size 640x480 --no-seg --json gpurun_out/${T}_hunt_qvga_noseg_s7000_n300.json > gpurun_out/${T}_hunt_qvga_noseg.log 2>&1
# round 5: sequences of the length BASELINE's configs have (tools/diag/long_sequence_hunt.py: N x 200 QVGA frames against the oracle)
timeout -k 10 900 python tools/diag/long_sequence_hunt.py --seeds 2000:2128 --variant throughput --out gpurun_out/${T}_long_hunt_throughput_s2000_n128.json > gpurun_out/${T}_long_hunt_throughput.log 2>&1
timeout -k 10 900 python tools/diag/long_sequence_hunt.py --seeds 2000:2128 --variant latency --out gpurun_out/${T}_long_hunt_latency_s2000_n128.json > gpurun_out/${T}_long_hunt_latency.log 2>&1
timeout -k 10 900 python tools/diag/long_sequence_hunt.py --seeds 2000:2032 --variant cluster --chunk 8 --out gpurun_out/${T}_long_hunt_cluster_s2000_n32.json > gpurun_out/${T}_long_hunt_cluster.log 2>&1
SF_HIP_LIB=$PWD/$C/libsf_hip_reforder.so timeout -k 10 900 python tools/diag/long_sequence_hunt.py --seeds 2000:2032 --variant throughput --out gpurun_out/${T}_long_hunt_reforder_s2000_n32.json > gpurun_out/${T}_long_hunt_reforder.log 2>&1
for f in throughput latency cluster reforder; do grep -h "^episodes\|\"total\"" gpurun_out/${T}_long_hunt_$f.log | cut -c1-400; done
timeout -k 10 1800 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputest.log 2>&1
tail -3 gpurun_out/${T}_gputest.log
for f in qvga_throughput qvga_latency 160x120_throughput 160x120_latency; do tail -n 5 gpurun_out/${T}_hunt_$f.log | cut -c1-200; done
for f in qvga_cluster qvga_noseg; do tail -n 1 gpurun_out/${T}_hunt_$f.log | cut -c1-500; done
head -4 gpurun_out/${T}_rocprofv3_stats_bench.csv
