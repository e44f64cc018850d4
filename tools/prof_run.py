"""Minimal driver for rocprofv3: N timed steps of sf_process_frame on a synthetic batch (no torch). With
SF_TIMED_LAUNCH_PER_FRAME=1 (tools/rocprof_collect.sh sets it) every step is its own launch of the frame kernel, so that
the PMC counters give bytes per FRAME of every stream. --workload sequences: the frame-to-frame replay of bench.py's
sequences workload (two distinct sequences, streams staggered), one launch per step as well."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch, make_sequence
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--workload", default="static")
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
api = sf.load()
p = bench.make_params(api, a.workload)
s = sf.Solver(api, 240, 320, a.batch, p)
if a.workload == "sequences":
    D, F = 2, 12 + max(a.steps, int(os.environ.get("SF_PROF_ONE_LAUNCH", "0")))
    # no process pool here: rocprofv3 follows every child process, and a pool of spawned workers under it never came back
    # (round 3: the orphans kept the GPU busy for everything that ran after them)
    seqs = [make_sequence(1000 + q, F, sphere=True) for q in range(D)]
    col = lambda x: np.ascontiguousarray(np.asarray(x, np.float32).T).ravel()
    hiprt = ctypes.CDLL("libamdhip64.so")
    ptrs = []
    for ch in (0, 1):
        h = np.stack([col(f[ch]) for sq in seqs for f in sq["frames"]])
        ptr = ctypes.c_void_p()
        assert hiprt.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(h.nbytes)) == 0
        assert hiprt.hipMemcpy(ptr, h.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(h.nbytes), 1) == 0
        ptrs.append(ptr.value)
    phase = (np.arange(a.batch) // D * 3) % 5
    idx = lambda step: ((np.arange(a.batch) % D) * F + phase + step).astype(np.int32)
    s.advance_sequences_device(ptrs[0], ptrs[1], idx(0), D * F); s.push_history(0)
    for step in range(1, 7):
        s.advance_sequences_device(ptrs[0], ptrs[1], idx(step), D * F); s.process_frame(step)
    s.synchronize()
    K = int(os.environ.get("SF_PROF_ONE_LAUNCH", "0"))
    if K:  # the bench's own shape: K frames of every stream in ONE launch (sf_process_sequence_frames_device) -- the LAST dispatch
        assert F >= 7 + K
        s.process_sequence_frames_device(ptrs[0], ptrs[1], np.stack([idx(7 + q) for q in range(K)]), D * F, 7)
        s.synchronize()
        print("one launch of %d frames, batch %d: %.3f ms per frame" % (K, a.batch, s.last_solver_kernel_ms() / K))
        sys.exit(0)
    ms = 0.0
    for step in range(7, 7 + a.steps):  # the advance is its own (small) kernel here; the frame kernel is what the counters are read for
        s.advance_sequences_device(ptrs[0], ptrs[1], idx(step), D * F); s.process_frame(step); s.synchronize()
        ms += s.last_solver_kernel_ms()
else:
    pairs = make_batch(8, sphere=(a.workload == "sphere"), distinct=8)
    for b in range(a.batch):
        s.set_current(b, *pairs[b % 8]["new"]); s.set_prediction(b, *pairs[b % 8]["old"])
    for im in range(5):
        s.process_frame(im)
    s.synchronize()
    ms = s.timed_process_frames(5, a.steps)
print("steps %d batch %d: %.3f ms/step" % (a.steps, a.batch, ms / a.steps))
