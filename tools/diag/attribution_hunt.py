"""Attribution hunt: the SAME random eight-frame sequences (tests/sequence_cases.py) through SEVERAL builds of the library
against the oracle, one summary per build -- which of the product's arithmetic shortcuts buys which share of the
excursions (profiles/PARITY.md, round 4).

    python tools/diag/attribution_hunt.py --first 8000 --count 600 --size 640x480 \
        --libs reforder=staticfusion_amd/csrc/libsf_hip_reforder.so,product=staticfusion_amd/csrc/libsf_hip.so \
        --json profiles/r04_attribution_qvga_s8000_n600.json

The oracle runs once per seed (worker processes), every library runs the seed on the GPU with the named build of the frame
kernel (--build, default throughput). JSON: {"summary": {name: {...}}, "frames": [every frame worth a look, with "lib"]}.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "diag"))

import numpy as np  # noqa: E402


def new_summary():
    pose_bins, b_bins = (1e-7, 1e-6, 1e-5, 1e-4), (1e-6, 1e-5, 1e-4, 1e-3, 1e-2)
    return {"frames": 0, "runs": 0, "bit_identical_frames": 0, "label_mismatch_frames": 0, "decision_mismatch_frames": 0,
            "count_mismatch_frames": 0, "threshold_flips": 0, "other_count_mismatches": 0,
            "pose_over": {str(t): 0 for t in pose_bins}, "b24_over": {str(t): 0 for t in b_bins}, "b_img_over": {str(t): 0 for t in b_bins},
            "worst": {"rot": 0.0, "trans": 0.0, "b24": 0.0, "b_img": 0.0}, "pose_over_1e-4_not_flip": 0, "seconds_gpu": 0.0}


_LIBS = None


def _seed_worker(job):
    """One seed in one process: the case, the oracle's frames, then every library on the GPU (this process's own HIP context:
    the synchronous copies of the getters serialise the streams of ONE process, so the parallelism is across processes)."""
    seed, W, H, seg, lib_items, build, thr = job
    global _LIBS
    import staticfusion_amd as sf
    from oracle import binding
    from sequence_cases import compare_frames, make_case, run_case

    if _LIBS is None:
        _LIBS = [(name, sf.Api(path, "sf_").with_variant(build)) for name, path in lib_items]
    ora = binding.load()
    case = make_case(seed, W, H, seg)
    ref = run_case(ora, case)
    out = []
    for name, api in _LIBS:
        tg = time.time()
        got = run_case(api, case)
        dt = time.time() - tg
        recs = compare_frames(ref, got, thr)
        for k, rec in enumerate(recs):
            rec["bit_identical"] = bool(np.array_equal(ref[k]["T"], got[k]["T"]) and np.array_equal(ref[k]["b"], got[k]["b"])
                                        and np.array_equal(ref[k]["b_img"], got[k]["b_img"]) and ref[k]["counts"] == got[k]["counts"])
        out.append((name, dt, recs))
    return seed, case["scale"], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=8000)
    ap.add_argument("--count", type=int, default=20)
    ap.add_argument("--size", default="320x240", help="RENDERED size; the solver sees half of it")
    ap.add_argument("--no-seg", action="store_true")
    ap.add_argument("--build", default="throughput", help="frame-kernel build every library runs (sf_create_ex)")
    ap.add_argument("--libs", required=True, help="name=path[,name=path...]")
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--keep-pose", type=float, default=1e-5)
    ap.add_argument("--keep-b", type=float, default=1e-3)
    a = ap.parse_args()
    W, H = (int(x) for x in a.size.split("x"))
    import staticfusion_amd as sf
    from oracle import binding

    binding.build()
    thr = float(binding.load().default_params_struct().irls_delta_threshold)
    lib_items = []
    for item in a.libs.split(","):
        name, path = item.split("=", 1)
        lib_items.append((name, os.path.join(ROOT, path) if not os.path.isabs(path) else path))
    cap = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        cap = None if q == "max" else int(float(q) / float(p))
    except Exception:
        pass
    procs = a.procs or max(1, min(len(os.sched_getaffinity(0)), cap or 64))

    summ = {name: new_summary() for name, _ in lib_items}
    backends = {name: sf.Api(path, "sf_").backend_name() for name, path in lib_items}
    kept = []
    t0 = time.time()
    jobs = [(s, W, H, not a.no_seg, lib_items, a.build, thr) for s in range(a.first, a.first + a.count)]
    with mp.get_context("spawn").Pool(procs) as pool:
        done = 0
        for seed, scale, out in pool.imap_unordered(_seed_worker, jobs, chunksize=1):
            done += 1
            if a.json and done % 100 == 0:  # a checkpoint: a run that is cut off still leaves what it had
                with open(a.json + ".partial", "w") as f:
                    json.dump({"seeds_done": done, "summary": summ, "frames": kept}, f)
            for name, dt, recs in out:
                sm = summ[name]
                sm["seconds_gpu"] += dt
                sm["runs"] += 1
                after_flip = False
                for rec in recs:
                    sm["frames"] += 1
                    pose = max(rec["rot"], rec["trans"])
                    sm["bit_identical_frames"] += rec["bit_identical"]
                    for t in sm["pose_over"]:
                        sm["pose_over"][t] += pose > float(t)
                    for t in sm["b24_over"]:
                        sm["b24_over"][t] += rec["b24"] > float(t)
                        sm["b_img_over"][t] += rec["b_img"] > float(t)
                    for q in ("rot", "trans", "b24", "b_img"):
                        sm["worst"][q] = max(sm["worst"][q], rec[q])
                    sm["label_mismatch_frames"] += rec["label_px"] > 0
                    sm["decision_mismatch_frames"] += rec["decision_px"] > 0
                    flip = rec.get("flip")
                    if flip:
                        sm["count_mismatch_frames"] += 1
                        if not after_flip:
                            sm["threshold_flips" if flip["kind"] in ("threshold", "level-exit") else "other_count_mismatches"] += 1
                    rec["after_flip"] = after_flip
                    if pose > 1e-4 and not (flip or after_flip):
                        sm["pose_over_1e-4_not_flip"] += 1
                    if flip:
                        after_flip = True
                    odd = not rec["bit_identical"] and "reference-order" in backends[name]  # that build is expected to be identical: keep every exception
                    if odd:
                        print("seed %d %s frame %d: NOT bit-identical (rot %.2e trans %.2e b24 %.2e b_img %.2e)" % (
                            seed, name, rec["frame"], rec["rot"], rec["trans"], rec["b24"], rec["b_img"]), flush=True)
                    if odd or rec["label_px"] or rec["decision_px"] or flip or pose > a.keep_pose or rec["b24"] > a.keep_b:
                        rec.update(seed=seed, lib=name, motion_scale=scale)
                        kept.append(rec)
                        if rec["label_px"] or rec["decision_px"] or flip or pose > 1e-4:
                            print("seed %d %s frame %d: rot %.2e trans %.2e labels %d px decisions %d px counts %s vs %s %s" % (
                                seed, name, rec["frame"], rec["rot"], rec["trans"], rec["label_px"], rec["decision_px"], rec["counts"],
                                rec["counts_ref"], (flip or {}).get("kind", "")), flush=True)
    from bench import git_head, source_sha

    meta = {"first_seed": a.first, "count": a.count, "solver_size": "%dx%d" % (W // 2, H // 2), "segmentation": not a.no_seg,
            "build": a.build, "irls_delta_threshold": thr, "seconds": round(time.time() - t0, 1), "head": git_head(), "src_sha": source_sha(),
            "libs": backends}
    for name, sm in summ.items():
        sm["seconds_gpu"] = round(sm["seconds_gpu"], 1)
        print(name, json.dumps(sm))
    print("%-22s %8s %9s %8s %8s %8s %9s %9s %9s" % ("build", "frames", "identical", "counts", "ties", ">1e-4", ">1e-5", "b>1e-4", "worst m"))
    for name, sm in summ.items():
        print("%-22s %8d %9d %8d %8d %8d %9d %9d %9.2e" % (name, sm["frames"], sm["bit_identical_frames"], sm["count_mismatch_frames"],
              sm["threshold_flips"], sm["pose_over"]["0.0001"], sm["pose_over"]["1e-05"], sm["b24_over"]["0.0001"],
              max(sm["worst"]["rot"], sm["worst"]["trans"])))
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump({"meta": meta, "summary": summ, "frames": kept}, f, indent=1)


if __name__ == "__main__":
    main()
