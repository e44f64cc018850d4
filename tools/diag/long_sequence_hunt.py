#!/usr/bin/env python3
"""Long sequences against the oracle: N synthetic sequences of F frames (staticfusion_amd/synth.py: make_sequence, the
generator of bench.py's `sequences` workload), every frame of every sequence compared -- what a tie of the oracle's stopping
tests at frame t does to frames t + 1 ... F ("next round" item 1).

    python tools/diag/long_sequence_hunt.py --seeds 1000:1064 --variant throughput --out gpurun_out/long_hunt.json   # GPU box
    python tools/diag/long_sequence_hunt.py --seeds 1000:1064 --control 2 --out profiles/r05_long_control_gemm2.json  # CPU only
    SF_HIP_LIB=.../libsf_hip_reforder.so python tools/diag/long_sequence_hunt.py ...                                  # another build

--control M compares the oracle with ITSELF under another reading of the GEMM order the reference leaves open (gemm mode M):
the rate of events and the decay after them that any other faithful build of the reference would show.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import long_sequences as ls  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1000:1016", help="first:last (exclusive)")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--rows", type=int, default=240)
    ap.add_argument("--cols", type=int, default=320)
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--control", type=int, default=0, help="oracle vs oracle under gemm mode M (no GPU)")
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=64, help="sequences per HIP handle / frame pool")
    ap.add_argument("--cache", default="/tmp")
    ap.add_argument("--out", required=True)
    return ap.parse_args()


def cpu_count():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def hip_runs(args, seeds):
    import staticfusion_amd as sf
    from staticfusion_amd.synth import sequence_arrays

    api = sf.load()
    runs = {}
    for c0 in range(0, len(seeds), args.chunk):
        chunk = seeds[c0:c0 + args.chunk]
        arrs = [sequence_arrays(s, args.frames, args.rows, args.cols, cache_dir=args.cache) for s in chunk]
        pd, pi = ls.DevicePool(np.concatenate([a[0] for a in arrs])), ls.DevicePool(np.concatenate([a[1] for a in arrs]))
        r = ls.Runner(api, pd, pi, len(chunk), args.frames, args.rows, args.cols, variant=args.variant)
        recs = [[] for _ in chunk]
        for _ in range(args.frames - 1):
            r.step()
            for q, rec in enumerate(r.frame_records(images=False)):
                recs[q].append(rec)
        variant = r.s.variant()
        r.close()
        pd.free()
        pi.free()
        for q, s in enumerate(chunk):
            runs[s] = recs[q]
    return runs, api.backend_name(), variant


def main():
    args = parse()
    a, b = [int(x) for x in args.seeds.split(":")]
    seeds = list(range(a, b))
    procs = args.procs or cpu_count()
    t0 = time.time()
    with mp.get_context("spawn").Pool(min(procs, len(seeds))) as pool:
        ref = pool.map(ls.oracle_sequence, [(s, args.frames, args.rows, args.cols, args.cache) for s in seeds], chunksize=1)
        t_ref = time.time() - t0
        if args.control:
            got = {r["seed"]: r["recs"] for r in pool.map(ls.oracle_sequence, [(s, args.frames, args.rows, args.cols, args.cache, args.control) for s in seeds], chunksize=1)}
            what = {"got": "oracle, gemm mode %d" % args.control, "variant": None}
    if not args.control:
        got, backend, variant = hip_runs(args, seeds)
        what = {"got": backend, "variant": list(variant), "lib": os.environ.get("SF_HIP_LIB", "libsf_hip.so")}
    from oracle import binding

    thr = float(binding.load().default_params_struct().irls_delta_threshold)
    streams, tot = [], {"frames": 0, "frames_past_bar": 0, "count_mismatches": 0, "label_mismatch_frames": 0, "decision_mismatch_frames": 0,
                        "b24_over_1e-5": 0, "b24_over_1e-4": 0, "bit_identical_frames": 0, "sequences_with_an_event": 0}
    for run in ref:
        s = run["seed"]
        recs = [ls.compare(x, y, thr) for x, y in zip(run["recs"], got[s])]
        summ = ls.summarise_stream(recs)
        # the whole trajectory: product vs oracle, and both against the generator's ground truth
        A_ref, A_got = ls.chain([x["T"] for x in run["recs"]]), ls.chain([y["T"] for y in got[s]])
        A_gt = ls.chain(list(run["T_gt"][1:]))
        summ["trajectory"] = {"got_vs_ref": ls.pose_delta(A_ref, A_got), "ref_vs_gt": ls.pose_delta(A_gt, A_ref), "got_vs_gt": ls.pose_delta(A_gt, A_got)}
        summ["seed"] = s
        summ["dist"] = [float("%.3g" % max(r["rot"], r["trans"])) for r in recs]
        streams.append(summ)
        for k in tot:
            if k == "sequences_with_an_event":
                tot[k] += 1 if summ["events"] else 0
            else:
                tot[k] += summ[k]
    back = [s["frames_until_back_under_bar"] for s in streams if s["events"]]
    eps = [dict(e, seed=s["seed"]) for s in streams for e in s["episodes"]]
    hist = {}
    for e in eps:
        hist[e["length"]] = hist.get(e["length"], 0) + 1
    out = {"what": dict(what, ref="oracle [C1]", seeds=[a, b], frames=args.frames, rows=args.rows, cols=args.cols, pose_bar=ls.POSE_BAR,
                        tie_rel_margin=ls.TIE_REL_MARGIN, oracle_seconds=round(t_ref, 1), wall_s=round(time.time() - t0, 1)),
           "total": tot, "frames_until_back_under_bar": back,
           # an EPISODE = a run of disturbed frames (past the bar or with a count mismatch) at most 2 frames apart; length = frames from
           # its first frame to its last frame past the bar; `after` = what the frames behind it look like
           "episodes": {"count": len(eps), "length_histogram": {str(k): v for k, v in sorted(hist.items())},
                        "peak_max": max([e["peak"] for e in eps] or [0.0]),
                        "first_frame_after_max": max([e["after"][0] for e in eps if e["after"]] or [0.0]),
                        "second_frame_after_max": max([e["after"][1] for e in eps if len(e["after"]) > 1] or [0.0]), "list": eps},
           "worst_frame": max(s["worst"] for s in streams),
           "worst_trajectory_got_vs_ref": [max(s["trajectory"]["got_vs_ref"][0] for s in streams), max(s["trajectory"]["got_vs_ref"][1] for s in streams)],
           "streams": streams}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f)
    print(json.dumps({k: out[k] for k in ("what", "total", "frames_until_back_under_bar", "worst_frame", "worst_trajectory_got_vs_ref")}))
    print("episodes:", json.dumps({k: v for k, v in out["episodes"].items() if k != "list"}))
    for s in streams:
        if s["events"]:
            print("seed %d: events %s; back under the bar after %s frames; after: %s" % (
                s["seed"], [(e["frame"], e["kind"], "%.1e" % e["dist"]) for e in s["events"][:6]], s["frames_until_back_under_bar"],
                " ".join("%.1e" % v for v in s["after_first_event"][:12])))


if __name__ == "__main__":
    main()
