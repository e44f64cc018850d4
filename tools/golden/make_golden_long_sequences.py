#!/usr/bin/env python3
"""tests/golden/long_sequences_qvga.npz: the ORACLE's frame-by-frame results on the four 200-frame synthetic QVGA sequences that
bench.py's `sequences` blocks play (seeds 1000-1003, staticfusion_amd/synth.py) and on one more whose frame 190 is a stopping-threshold
tie (seed 2059, tests/long_sequences.py: EVENT_SEEDS) -- pose, the 24 b values, outer / IRLS counts,
a CRC of the level-0 label image and the packed static / dynamic decisions of every frame.

What it is for: (i) it pins the oracle itself (tests/test_long_sequences.py re-runs a prefix on the CPU and every frame on the
GPU box and demands these bits: a change of the restatement, of the generator or of the host's libm shows up), (ii) it is what
bench.py's sequences blocks compare the timed frames' poses with (`pose_delta_vs_cpu`) without running the oracle inside the
bench. It is the oracle's output, NOT a reference-pinned vector: the reference cannot be run here.

    python tools/golden/make_golden_long_sequences.py            # ~1 min on 4 cores
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import long_sequences as ls  # noqa: E402


def main():
    jobs = [(seed, ls.FRAMES, ls.ROWS, ls.COLS, "/tmp") for seed in ls.SEEDS + ls.EVENT_SEEDS]
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        runs = pool.map(ls.oracle_sequence, jobs)
    out = {
        "seeds": np.array(ls.SEEDS + ls.EVENT_SEEDS), "frames": np.array(ls.FRAMES),
        "T": np.stack([[r["T"] for r in run["recs"]] for run in runs]).astype(np.float32),
        "b": np.stack([[r["b"] for r in run["recs"]] for run in runs]).astype(np.float32),
        "counts": np.array([[r["counts"] for r in run["recs"]] for run in runs], dtype=np.int32),
        "status": np.array([[r["status"] for r in run["recs"]] for run in runs], dtype=np.int32),
        "label_crc": np.array([[r["label_crc"] for r in run["recs"]] for run in runs], dtype=np.uint32),
        "decisions": np.stack([[r["decisions"] for r in run["recs"]] for run in runs]),
        "T_gt": np.stack([run["T_gt"] for run in runs]),
    }
    path = os.path.join(ROOT, "tests", "golden", "long_sequences_qvga.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
