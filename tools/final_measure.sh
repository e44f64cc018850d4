#!/bin/bash
# The evidence bundle of a round, ON THE GPU BOX (gpurun -- 'bash tools/final_measure.sh r03g'); tools/collect_bundle.py <tag> copies
# what it leaves under gpurun_out/ into profiles/ and prints the numbers DESIGN.md section 7 quotes.
#   1 the driver's command (bench line: configs[1], full solver, sequences at two batch sizes, isolated IRLS passes, CPU legs)
#   2 the same command under rocprofv3 --kernel-trace --stats
#   3 PMC traffic per frame of every stream for the workloads of the line (tagged with the source hash; the NEXT bench run reports it)
#   4 stage profiles, bytes per stage group, SQ counters of the whole frame, the isolated passes
#   5 parity: distance report, the hunts with JSON records, the GPU test suite
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
T=${1:-r03g}
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rocprof_bench -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_bench_under_rocprof.err
find gpurun_out/${T}_rocprof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_rocprofv3_stats_bench.csv \;
for spec in "static 16384" "sphere 16384" "sequences 16384" "sequences 4096"; do
  set -- $spec
  timeout 1500 bash tools/measure_traffic.sh $1 $2 > gpurun_out/${T}_traffic_$1_b$2.log 2>&1
done
# the bench line once more, now with roofline.traffic of THESE sources
mkdir -p profiles; cp gpurun_out/traffic_*.json profiles/ 2>/dev/null
timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_with_traffic.json 2> gpurun_out/${T}_bench_with_traffic.err
{ for w in sphere static; do timeout 300 python tools/stage_profile.py --workload $w --batch 5120 --variant throughput --steps 10 | grep -v "  0.0 us"; done; } > gpurun_out/${T}_stage_profiles.txt 2>&1
bash tools/traffic_by_stage.sh sphere 4096 > gpurun_out/${T}_traffic_by_stage_sphere.txt 2>&1
bash tools/traffic_by_stage.sh static 4096 > gpurun_out/${T}_traffic_by_stage_static.txt 2>&1
timeout 300 python tools/pass_microbench.py --batch 512 > gpurun_out/${T}_pass_microbench_b512.txt 2>&1
timeout 600 python tools/parity_report.py --out gpurun_out/${T}_parity_report.json > gpurun_out/${T}_parity_report.md 2>&1
{ timeout 300 python tools/diag/b_summary.py 6 throughput; SF_ORACLE_EXACT_WARP=1 timeout 300 python tools/diag/b_summary.py 6 throughput; SF_ORACLE_EXACT_WARP=1 timeout 300 python tools/diag/b_summary.py 6 cluster; } > gpurun_out/${T}_b_summary.txt 2>&1
timeout 900 python tools/diag/sequence_hunt.py --first 5000 --count 240 --json gpurun_out/${T}_hunt_160x120_s5000_n240.json > gpurun_out/${T}_hunt_s5000.log 2>&1
timeout 1500 python tools/diag/sequence_hunt.py --first 20000 --count 1000 --json gpurun_out/${T}_hunt_160x120_s20000_n1000.json > gpurun_out/${T}_hunt_s20000.log 2>&1
timeout 900 python tools/diag/sequence_hunt.py --first 7000 --count 60 --size 640x480 --json gpurun_out/${T}_hunt_qvga_s7000_n60.json > gpurun_out/${T}_hunt_qvga.log 2>&1
timeout 900 python tools/diag/sequence_hunt.py --first 7000 --count 60 --size 640x480 --no-seg --json gpurun_out/${T}_hunt_qvga_noseg_s7000_n60.json > gpurun_out/${T}_hunt_qvga_noseg.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputest.log 2>&1
tail -3 gpurun_out/${T}_gputest.log
python - "$T" <<'PY'
import json, sys
T = sys.argv[1]
for f in ("bench_default", "bench_under_rocprof", "bench_with_traffic"):
    try:
        d = json.loads(open("gpurun_out/%s_%s.json" % (T, f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    fs = d["full_solver"]
    print(f, d["build"], "static", round(d["value"]), round(d["frames_per_s"]), round(d["ms_per_step"], 2), round(d["roofline"]["frac"], 4),
          "traffic", d["roofline"]["traffic"] and round(d["roofline"]["traffic_provenance"]["ratio_to_algorithmic"], 3),
          "| sphere", round(fs["value"]), round(fs["frames_per_s"]), round(fs["ms_per_step"], 2), round(fs["roofline"]["frac"], 4),
          "traffic", fs["roofline"]["traffic"] and round(fs["roofline"]["traffic_provenance"]["ratio_to_algorithmic"], 3))
    for q in d["sequences"]:
        print("   seq", q["streams_per_gpu"], round(q["value"]), round(q["frames_per_s"]), round(q["ms_per_step"], 2), round(q["roofline"]["frac"], 4),
              "traffic", q["roofline"]["traffic"] and round(q["roofline"]["traffic_provenance"]["ratio_to_algorithmic"], 3))
    print("   passes", {k: round(v["frac"], 3) for k, v in d["roofline"]["irls_passes"].items()})
PY
for f in s5000 s20000 qvga qvga_noseg; do tail -n 1 gpurun_out/${T}_hunt_$f.log | cut -c1-700; done
head -4 gpurun_out/${T}_rocprofv3_stats_bench.csv
