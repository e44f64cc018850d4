#!/bin/bash
# after a change of the device sources: PMC traffic of the four workloads of the bench line (tagged with the new source hash),
# the bench line with it, the GPU test suite, the distance of both library builds from the oracle
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out profiles; export TMPDIR=/tmp
T=${1:-r03i}
for spec in "static 16384" "sphere 16384" "sequences 4096" "sequences 16384"; do
  set -- $spec
  timeout -k 10 900 bash tools/measure_traffic.sh $1 $2 > gpurun_out/${T}_traffic_$1_b$2.log 2>&1
  tail -n 1 gpurun_out/${T}_traffic_$1_b$2.log | cut -c1-250
done
cp gpurun_out/traffic_*_b*.json profiles/ 2>/dev/null; rm -f profiles/traffic_*_summary.json
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
python - "$T" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench_default.json" % sys.argv[1]).read().strip().splitlines()[-1]); fs = d["full_solver"]
tr = lambda r: r["traffic_provenance"] and (round(r["traffic_provenance"]["ratio_to_algorithmic"], 3), r["traffic_provenance"]["head"])
print(d["build"], "static", round(d["value"]), round(d["frames_per_s"]), round(d["roofline"]["frac"], 4), tr(d["roofline"]), "| sphere", round(fs["frames_per_s"]), round(fs["roofline"]["frac"], 4), tr(fs["roofline"]))
for q in d["sequences"]: print("  seq", q["streams_per_gpu"], round(q["frames_per_s"]), round(q["roofline"]["frac"], 4), tr(q["roofline"]))
print("  passes", {k: round(v["frac"], 3) for k, v in d["roofline"]["irls_passes"].items()})
PY
C=staticfusion_amd/csrc
{ timeout -k 10 300 python tools/diag/b_summary.py 6 throughput; SF_HIP_LIB=$PWD/$C/libsf_hip_precise.so timeout -k 10 300 python tools/diag/b_summary.py 6 throughput; } > gpurun_out/${T}_b_summary_product_vs_precise.txt 2>&1
cat gpurun_out/${T}_b_summary_product_vs_precise.txt
timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputest.log 2>&1
tail -3 gpurun_out/${T}_gputest.log
