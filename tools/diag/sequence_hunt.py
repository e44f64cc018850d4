"""Parity hunt: N random eight-frame sequences (tests/sequence_cases.py: carried state, 5-frame residuals from frame 5 on)
through sf_process_frame on every build of the frame kernel against the oracle -- or, with --control, the ORACLE AGAINST
ITSELF under another summation convention (no GPU needed) -- and a JSON record of every frame that is worth a look.

    python tools/diag/sequence_hunt.py --first 20000 --count 1000 --json profiles/r03_hunt_160x120_s20000.json
    python tools/diag/sequence_hunt.py --first 20000 --count 1000 --control gemm2 --json profiles/r03_control_gemm2.json
    python tools/diag/sequence_hunt.py --first 7000 --count 60 --size 640x480 ...        (the solver then sees QVGA)

--control MODE  compare oracle [C1] (float operands, fp64 sums) with the oracle in MODE:
    gemm1  AtA / AtB accumulated in ONE float per entry          gemm2  four interleaved float partial sums (SSE packets)
    gemm3  [C1] over the rows in reverse order                   exact  the per-cluster fp32 sums in fp64 (sfo_test_set_exact_sums)
    warp   the warp's scatter sums in fp64 from exact products (sfo_test_set_exact_warp)
JSON: {"summary": {...}, "frames": [every frame with a discrete mismatch, a pose distance > --keep-pose or b > --keep-b]}.
"""
import argparse
import ctypes
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

CONTROL_MODES = {"gemm1": ("gemm", 1), "gemm2": ("gemm", 2), "gemm3": ("gemm", 3), "exact": ("exact", 1), "warp": ("warp", 1)}


def _control_prepare(mode):
    kind, val = CONTROL_MODES[mode]

    def prepare(solver):
        lib = solver.api.lib
        fn = {"gemm": lib.sfo_test_set_gemm_mode, "exact": lib.sfo_test_set_exact_sums, "warp": lib.sfo_test_set_exact_warp}[kind]
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert fn(solver.h, val) == 0

    return prepare


def _worker(job):
    """One seed on the CPU: the case, the oracle's frames, and (control) the second oracle's frames."""
    seed, W, H, seg, control = job
    from oracle import binding
    from sequence_cases import make_case, run_case

    ora = binding.load()
    case = make_case(seed, W, H, seg)
    ref = run_case(ora, case)
    other = run_case(ora, case, prepare=_control_prepare(control)) if control else None
    return seed, case, ref, other


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=5000)
    ap.add_argument("--count", type=int, default=20)
    ap.add_argument("--size", default="320x240", help="RENDERED size; the solver sees half of it")
    ap.add_argument("--no-seg", action="store_true", help="pure odometry (BASELINE configs[1] parameters)")
    ap.add_argument("--builds", default="throughput,latency,cluster")
    ap.add_argument("--control", choices=sorted(CONTROL_MODES), default=None)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--keep-pose", type=float, default=1e-5)
    ap.add_argument("--keep-b", type=float, default=1e-3)
    a = ap.parse_args()
    W, H = (int(x) for x in a.size.split("x"))
    from oracle import binding
    from sequence_cases import compare_frames

    binding.build()
    thr = float(binding.load().default_params_struct().irls_delta_threshold)
    builds = [] if a.control else [b for b in a.builds.split(",") if b]
    if builds:
        import staticfusion_amd as sf
        from sequence_cases import run_case

        hip = sf.load()
    cap = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        cap = None if q == "max" else int(float(q) / float(p))
    except Exception:
        pass
    procs = a.procs or max(1, min(len(os.sched_getaffinity(0)), cap or 64) - (1 if builds else 0))

    pose_bins, b_bins = (1e-6, 1e-5, 1e-4), (1e-5, 1e-4, 1e-3, 1e-2)
    summ = {"frames": 0, "runs": 0, "label_mismatch_frames": 0, "decision_mismatch_frames": 0, "count_mismatch_frames": 0,
            "threshold_flips": 0, "other_count_mismatches": 0, "pose_over": {str(t): 0 for t in pose_bins},
            "b24_over": {str(t): 0 for t in b_bins}, "b_img_over": {str(t): 0 for t in b_bins},
            "worst": {"rot": 0.0, "trans": 0.0, "b24": 0.0, "b_img": 0.0}, "pose_over_1e-4_not_flip": 0}
    kept = []
    t0 = time.time()
    jobs = [(s, W, H, not a.no_seg, a.control) for s in range(a.first, a.first + a.count)]
    with mp.get_context("spawn").Pool(procs) as pool:
        for seed, case, ref, other in pool.imap(_worker, jobs, chunksize=1):
            runs = [("oracle:" + a.control, other)] if a.control else [(b, run_case(hip.with_variant(b), case)) for b in builds]
            for name, got in runs:
                summ["runs"] += 1
                after_flip = False
                for rec in compare_frames(ref, got, thr):
                    summ["frames"] += 1
                    pose = max(rec["rot"], rec["trans"])
                    for t in pose_bins:
                        summ["pose_over"][str(t)] += pose > t
                    for t in b_bins:
                        summ["b24_over"][str(t)] += rec["b24"] > t
                        summ["b_img_over"][str(t)] += rec["b_img"] > t
                    for k in ("rot", "trans", "b24", "b_img"):
                        summ["worst"][k] = max(summ["worst"][k], rec[k])
                    summ["label_mismatch_frames"] += rec["label_px"] > 0
                    summ["decision_mismatch_frames"] += rec["decision_px"] > 0
                    flip = rec.get("flip")
                    if flip:
                        summ["count_mismatch_frames"] += 1
                        if flip["kind"] in ("threshold", "level-exit") and not after_flip:
                            summ["threshold_flips"] += 1
                        elif not after_flip:
                            summ["other_count_mismatches"] += 1
                    # a flipped frame changes the carried state: later frames of that run are not independent evidence
                    rec["after_flip"] = after_flip
                    if pose > 1e-4 and not (flip or after_flip):
                        summ["pose_over_1e-4_not_flip"] += 1
                    if flip:
                        after_flip = True
                    if rec["label_px"] or rec["decision_px"] or flip or pose > a.keep_pose or rec["b24"] > a.keep_b:
                        rec.update(seed=seed, build=name, motion_scale=case["scale"])
                        kept.append(rec)
                        if rec["label_px"] or rec["decision_px"] or flip or pose > 1e-4:
                            print("seed %d %s frame %d: rot %.2e trans %.2e labels %d px decisions %d px counts %s vs %s %s" % (
                                seed, name, rec["frame"], rec["rot"], rec["trans"], rec["label_px"], rec["decision_px"], rec["counts"],
                                rec["counts_ref"], (flip or {}).get("kind", "")), flush=True)
    summ.update(first_seed=a.first, count=a.count, solver_size="%dx%d" % (W // 2, H // 2), segmentation=not a.no_seg,
                compared="oracle [C1] vs oracle %s" % a.control if a.control else "HIP builds %s vs oracle" % ",".join(builds),
                irls_delta_threshold=thr, seconds=round(time.time() - t0, 1))
    if builds:
        from bench import git_head, source_sha

        summ["build"] = {"head": git_head(), "src_sha": source_sha(), "backend": hip.backend_name()}
    print(json.dumps(summ))
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump({"summary": summ, "frames": kept}, f, indent=1)


if __name__ == "__main__":
    main()
