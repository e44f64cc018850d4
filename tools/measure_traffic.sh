#!/bin/bash
# HBM bytes per launch of the frame kernel from the PMC counters, tagged with the identity of the sources measured:
# what bench.py reports as roofline.traffic (and refuses to report when the tag does not match the running sources).
# Run ON THE GPU BOX via gpurun; copy gpurun_out/traffic_<workload>_b<batch>.json to profiles/.
# usage: tools/measure_traffic.sh <workload: static|sphere|sequences> <batch> [variant] [frames per launch]
# frames per launch (sequences only): K > 0 measures ONE launch of K frames of every stream -- what bench.py times (the launch
# skips createImagePyramid(true) for the frames whose pyramid buffers swap and advances the streams itself) -- instead of one
# launch per frame; hbm_bytes_per_launch is then divided by K, and bench.py only accepts the file for the same K.
set -u
WL=$1; B=$2; VAR=${3:-throughput}; K=${4:-0}
cd "$(dirname "$0")/.."
TAG=traffic_${WL}_b${B}
LAST=3
if [ "$K" -gt 0 ]; then export SF_PROF_ONE_LAUNCH=$K; LAST=1; fi
bash tools/rocprof_collect.sh $TAG $WL $B > gpurun_out/${TAG}_collect.log 2>&1
python tools/rocprof_summarise.py gpurun_out/prof_$TAG --last $LAST > gpurun_out/${TAG}_summary.json
python - "$WL" "$B" "$VAR" "$K" <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
wl, B, var, K = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
s = json.load(open("gpurun_out/traffic_%s_b%d_summary.json" % (wl, B)))
out = {"workload": wl, "batch": B, "variant": var, "kernel": s["kernel"], "src_sha": bench.source_sha(), "head": bench.git_head(),
       "hbm_bytes_per_launch": s["hbm_bytes_per_launch"] / max(K, 1), "frames_per_launch": max(K, 1),
       "fetch_size_bytes_raw": s["FETCH_SIZE_KB_avg"] * 1024.0 / max(K, 1),
       "write_size_bytes": s["WRITE_SIZE_KB_avg"] * 1024.0 / max(K, 1), "kernel_ms_avg": s["duration_ms_avg_timed"] / max(K, 1),
       "valu_issue": ({"busy": s["valu_busy"], "insts": s["valu_insts_per_launch"] / max(K, 1), "quad_cycles_per_inst": s["valu_quad_cycles_per_inst"],
                       "gpu_active_cycles": s["gpu_active_cycles_per_launch"] / max(K, 1),
                       "note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... and --pmc GRBM_GUI_ACTIVE in passes of their own; busy = 2 cycles x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): a wave64 fp32 / integer-add / compare instruction occupies its SIMD for two cycles on gfx950, integer multiplies, fp64 and packed fp32 for four (tools/micro/valu_rate.hip, int_rate.hip) -- a LOWER bound of the occupancy; + 4 % dummy instructions cost the full solver 0 - 0.5 % (profiles/r06x_ab_dummy_valu.txt). Until round 6 this field assumed four cycles and read twice as much."} if "valu_busy" in s else None),
       "per": "frame of every stream" + (" (one launch of %d frames, divided by %d)" % (K, K) if K else " (one launch per frame)"),
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/measure_traffic.sh); read bytes doubled per "
               "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts wide coalesced reads at half size; tools/rocprof_calibrate.sh: raw/expected = 0.500)"}
json.dump(out, open("gpurun_out/traffic_%s_b%d.json" % (wl, B), "w"), indent=1)
print(json.dumps(out))
PY
