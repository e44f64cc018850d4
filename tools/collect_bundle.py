"""Copy the artefacts tools/final_measure.sh <tag> left under gpurun_out/ (merged back by gpurun) into profiles/ and print the
numbers DESIGN.md section 7 quotes. Run here (no GPU needed): python tools/collect_bundle.py <tag>"""
import csv, json, os, shutil, sys

T = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
last = lambda path: json.loads(open(path).read().strip().splitlines()[-1])


def cp(src, dst=None):
    a = os.path.join(G, src)
    if os.path.exists(a):
        shutil.copy(a, os.path.join(P, dst or src))
        return True
    print("missing", src)
    return False


for f in ("bench_default.json", "bench_under_rocprof.json", "rocprofv3_stats_bench.csv", "stage_profiles.txt",
          "traffic_by_stage_sphere.txt", "traffic_by_stage_static.txt", "pass_microbench_b512.txt", "parity_report.md", "parity_report.json",
          "b_summary.txt"):
    cp("%s_%s" % (T, f))
# the hunts: per-frame records compressed, the summaries beside them
import gzip
for f in sorted(os.listdir(G)):
    if f.startswith("%s_hunt_" % T) and f.endswith(".json"):
        dd = json.load(open(os.path.join(G, f)))
        with gzip.open(os.path.join(P, f + ".gz"), "wt", compresslevel=9) as g:
            json.dump(dd, g)
        json.dump({k: dd[k] for k in ("meta", "summary") if k in dd}, open(os.path.join(P, f.replace(".json", "_summary.json")), "w"), indent=1)
cp("%s_gputest.log" % T, "%s_gputest_three_variants.log" % T)
for f in sorted(os.listdir(G)):
    if f.startswith("%s_long_hunt_" % T) and f.endswith(".json"):
        cp(f)
for f in os.listdir(G):
    if f.startswith("traffic_") and f.endswith(".json") and "summary" not in f:
        cp(f)

d = last(os.path.join(G, "%s_bench_default.json" % T))
print("src_sha", d["build"])
# rocprofv3 kernel trace: every launch of the frame kernels, in order; the timed launch is the one that covers `steps` frames
trace = os.path.join(G, "%s_rocprof_bench" % T)
rows = []
for root, _, files in os.walk(trace):
    for f in files:
        if f.endswith("kernel_trace.csv"):
            rows += [r for r in csv.DictReader(open(os.path.join(root, f))) if "sf_frame_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
b = last(os.path.join(G, "%s_bench_under_rocprof.json" % T))
res = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline (tools/final_measure.sh %s)" % T,
       "build": b["build"], "note": "per handle: 1 first frame + 4 history priming launches + 5 warm-up launches + ONE timed launch of 20 frames of every "
       "stream (sf_process_frames) + 1 statistics launch; the sequences handles likewise with their bootstrap"}
blocks = {"static": b, "sphere": b["full_solver"]}
for name, key in (("sf_frame_kernel_nt256(", "static"), ("sf_frame_kernel_nt256o5(", "all full-solver handles (configs[2] + sequences 16384 + 4096)")):
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith(name)]
    res[key] = {"kernel": name.rstrip("("), "launch_ms_in_order": [round(x, 3) for x in dd]}
steps = b["steps"]
res["timed_launches"] = {
    "static": {"trace_ms": max((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith("sf_frame_kernel_nt256(")),
               "bench_hip_events_ms": b["roofline"]["kernel_launch_ms"]},
    "full_solver": {"bench_hip_events_ms": b["full_solver"]["roofline"]["kernel_launch_ms"]},
    "sequences": [{"streams": q["streams_per_gpu"], "bench_hip_events_ms": q["roofline"]["kernel_launch_ms"]} for q in b.get("sequences", [])],
}
o5 = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith("sf_frame_kernel_nt256o5(")), reverse=True)
res["timed_launches"]["full_solver"]["trace_ms_candidates_longest_first"] = [round(x, 3) for x in o5[:3]]
json.dump(res, open(os.path.join(P, "%s_rocprofv3_trace_summary.json" % T), "w"), indent=1)
print("timed launches, trace vs HIP events:", json.dumps(res["timed_launches"]))
for f in ("bench_default", "bench_under_rocprof"):
    x = last(os.path.join(G, "%s_%s.json" % (T, f)))
    fs = x["full_solver"]
    tr = lambda r: r["traffic_provenance"] and round(r["traffic_provenance"]["ratio_to_algorithmic"], 3)
    print(f, "static", round(x["value"]), round(x["frames_per_s"]), round(x["ms_per_step"], 2), round(x["roofline"]["frac"], 4), tr(x["roofline"]),
          "| sphere", round(fs["value"]), round(fs["frames_per_s"]), round(fs["ms_per_step"], 2), round(fs["roofline"]["frac"], 4), tr(fs["roofline"]))
    for q in x.get("sequences", []):
        print("    sequences", q["streams_per_gpu"], round(q["value"]), round(q["frames_per_s"]), round(q["ms_per_step"], 2), round(q["roofline"]["frac"], 4), tr(q["roofline"]))
    print("    irls passes", {k: round(v["frac"], 3) for k, v in x["roofline"]["irls_passes"].items()})
print("cpu 1 core it/s, frames/s:", round(d["cpu_baseline"]["value"]), round(d["cpu_baseline"]["frames_per_s"], 1), "| sphere",
      round(d["full_solver"]["cpu_baseline"]["value"]), round(d["full_solver"]["cpu_baseline"]["frames_per_s"], 1), "| all cores",
      round(d["cpu_baseline_all_cores"]["value"]), d["cpu_baseline_all_cores"]["cores"])
for f in ("hunt_160x120_s5000_n240", "hunt_160x120_s20000_n1000", "hunt_qvga_s7000_n60", "hunt_qvga_noseg_s7000_n60"):
    p = os.path.join(G, "%s_%s.json" % (T, f))
    if os.path.exists(p):
        s = json.load(open(p))["summary"]
        print(f, s["frames"], "labels", s["label_mismatch_frames"], "decisions", s["decision_mismatch_frames"], "counts", s["count_mismatch_frames"],
              "flips", s["threshold_flips"], "pose_over", s["pose_over"], "b_img_over", s["b_img_over"], "worst", {k: "%.2e" % v for k, v in s["worst"].items()})
