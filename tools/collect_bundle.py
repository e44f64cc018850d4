"""Copy the artefacts tools/final_measure.sh <tag> left under gpurun_out/ (merged back by gpurun) into profiles/ and print the
numbers DESIGN.md section 7.1 quotes. Run here (no GPU needed): python tools/collect_bundle.py <tag>"""
import csv, json, os, shutil, sys

T = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for f in ("bench_default.json", "bench_under_rocprof.json", "rocprofv3_stats_bench.csv", "bench_sequences_b4096.json",
          "bench_sequences_b16384.json", "bench_sequences_cluster_b1_b8.json"):
    shutil.copy(os.path.join(G, "%s_%s" % (T, f)), os.path.join(P, "%s_%s" % (T, f)))
for f in ("traffic_static_b16384.json", "traffic_sphere_b16384.json"):
    shutil.copy(os.path.join(G, f), os.path.join(P, f))
if os.path.exists(os.path.join(G, "%s_gputest.log" % T)):
    shutil.copy(os.path.join(G, "%s_gputest.log" % T), os.path.join(P, "%s_gputest_three_variants.log" % T))
shutil.copy(os.path.join(G, "%s_parity_report.md" % T), os.path.join(P, "r02_parity_report.md"))
shutil.copy(os.path.join(G, "%s_parity_report.json" % T), os.path.join(P, "r02_parity_report.json"))
last = lambda path: json.loads(open(path).read().strip().splitlines()[-1])
d = last(os.path.join(G, "%s_bench_default.json" % T))
sha = d["build"]["src_sha"]
heads = ["cluster build (G = 24 workgroups per stream), one stream", "cluster build, one stream",
         "latency build (1024 threads, one workgroup), one stream", "latency build, one stream",
         "throughput build (256 threads, one workgroup), one stream", "throughput build, one stream",
         "cluster build, 2 streams (G = 24)", "cluster build, 4 streams (G = 24)", "cluster build, 8 streams (G = 24: 192 workgroups)",
         "throughput build, 4096 streams (5 workgroups per CU: 3.2 rounds of 1280)", "throughput build, 4096 streams (4 workgroups per CU: 4 rounds of 1024)"]
out = ["# tools/final_measure.sh %s: tools/stage_profile.py (in-kernel stage timers, 30 steps), one MI355X, src_sha %s" % (T, sha),
       "# us/frame = per stream and frame for batch 1; for batch 4096 per workgroup slot (wall time of one workgroup per frame)", ""]
k = 0
for l in open(os.path.join(G, "%s_stage_profiles.txt" % T)).read().splitlines():
    if l.startswith("workload"):
        out.append("## " + heads[k]); k += 1
    out.append(l)
open(os.path.join(P, "r02_cluster_latency.txt"), "w").write("\n".join(out) + "\n")
rows = [r for r in csv.DictReader(open(os.path.join(G, "%s_rocprof_bench" % T, "bench_kernel_trace.csv"))) if "sf_frame_kernel" in r["Kernel_Name"]]
b = last(os.path.join(G, "%s_bench_under_rocprof.json" % T))
res = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline (tools/final_measure.sh %s)" % T}
for name, key, blk in (("sf_frame_kernel_nt256(", "static", b), ("sf_frame_kernel_nt256o5(", "sphere", b["full_solver"])):
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith(name)]
    res[key] = {"kernel": name.rstrip("("), "launch_ms_in_order": [round(x, 3) for x in dd],
                "launches": "1 first frame + 4 history priming (no 5-frame residual stage) + 2 warm-up + 10 timed + 1 statistics launch",
                "timed_launches_7_to_16_avg_ms": sum(dd[7:17]) / 10, "bench_hip_event_avg_ms": blk["roofline"]["kernel_ms_avg"]}
json.dump(res, open(os.path.join(P, "%s_rocprofv3_trace_summary.json" % T), "w"), indent=1)
print("src_sha", sha)
print("trace vs events", {k_: (round(v["timed_launches_7_to_16_avg_ms"], 2), round(v["bench_hip_event_avg_ms"], 2)) for k_, v in res.items() if k_ != "command"})
for f in ("bench_default", "bench_under_rocprof"):
    x = last(os.path.join(G, "%s_%s.json" % (T, f)))
    fs = x["full_solver"]
    print(f, "static", round(x["value"]), round(x["frames_per_s"]), round(x["ms_per_step"], 1), round(x["roofline"]["frac"], 3),
          "| sphere", round(fs["value"]), round(fs["frames_per_s"]), round(fs["ms_per_step"], 2), round(fs["roofline"]["frac"], 3))
print("traffic GB", round(d["roofline"]["traffic"] / 1e9, 1), round(d["full_solver"]["roofline"]["traffic"] / 1e9, 1))
print("cpu 1 core it/s, frames/s:", round(d["cpu_baseline"]["value"]), round(d["cpu_baseline"]["frames_per_s"], 1), "| sphere",
      round(d["full_solver"]["cpu_baseline"]["value"]), round(d["full_solver"]["cpu_baseline"]["frames_per_s"], 1), "| all cores", round(d["cpu_baseline_all_cores"]["value"]))
for f in ("bench_sequences_b4096", "bench_sequences_b16384"):
    x = last(os.path.join(G, "%s_%s.json" % (T, f)))
    print(f, round(x["value"]), round(x["frames_per_s"]), round(x["ms_per_step"], 1), round(x["roofline"]["frac"], 3))
for l in open(os.path.join(G, "%s_bench_sequences_cluster_b1_b8.json" % T)):
    x = json.loads(l)
    print("sequences on the cluster build", x["config"]["streams_per_gpu"], round(x["value"]), round(x["frames_per_s"]), round(x["ms_per_step"], 3))
