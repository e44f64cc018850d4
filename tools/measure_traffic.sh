#!/bin/bash
# HBM bytes per launch of the frame kernel from the PMC counters, tagged with the identity of the sources measured:
# what bench.py reports as roofline.traffic (and refuses to report when the tag does not match the running sources).
# Run ON THE GPU BOX via gpurun; copy gpurun_out/traffic_<workload>_b<batch>.json to profiles/.
# usage: tools/measure_traffic.sh <workload: static|sphere|sequences> <batch> [variant]
set -u
WL=$1; B=$2; VAR=${3:-throughput}
cd "$(dirname "$0")/.."
TAG=traffic_${WL}_b${B}
bash tools/rocprof_collect.sh $TAG $WL $B > gpurun_out/${TAG}_collect.log 2>&1
python tools/rocprof_summarise.py gpurun_out/prof_$TAG > gpurun_out/${TAG}_summary.json
python - "$WL" "$B" "$VAR" <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
wl, B, var = sys.argv[1], int(sys.argv[2]), sys.argv[3]
s = json.load(open("gpurun_out/traffic_%s_b%d_summary.json" % (wl, B)))
out = {"workload": wl, "batch": B, "variant": var, "kernel": s["kernel"], "src_sha": bench.source_sha(), "head": bench.git_head(),
       "hbm_bytes_per_launch": s["hbm_bytes_per_launch"], "fetch_size_bytes_raw": s["FETCH_SIZE_KB_avg"] * 1024.0,
       "write_size_bytes": s["WRITE_SIZE_KB_avg"] * 1024.0, "kernel_ms_avg": s["duration_ms_avg_timed"],
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/measure_traffic.sh); read bytes doubled per "
               "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts wide coalesced reads at half size; tools/rocprof_calibrate.sh: raw/expected = 0.500)"}
json.dump(out, open("gpurun_out/traffic_%s_b%d.json" % (wl, B), "w"), indent=1)
print(json.dumps(out))
PY
