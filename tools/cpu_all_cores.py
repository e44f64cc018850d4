"""The CPU restatement on ALL host cores, one independent stream per process (the fair multi-core
comparator for the one-GPU number). Same workload and sample as bench.py's single-core cpu_baseline."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import multiprocessing as mp


def worker(args):
    workload, seconds, seed = args
    import numpy as np
    import staticfusion_amd as sf
    from oracle import binding
    from staticfusion_amd.synth import make_batch
    import bench

    ora = binding.load()
    p = bench.make_params(ora, workload)
    pairs = make_batch(1, base_seed=1234 + seed, sphere=(workload == "sphere"), distinct=1)
    s = sf.Solver(ora, 240, 320, 1, p)
    s.set_current(0, *pairs[0]["new"]); s.set_prediction(0, *pairs[0]["old"])
    for im in range(5):
        s.process_frame(im)
    im, iters, frames = 5, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        s.process_frame(im); im += 1; frames += 1
        iters += s.stats(0).n_irls
    return iters, frames, time.perf_counter() - t0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="static")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    a = ap.parse_args()
    from oracle import binding
    binding.build()
    with mp.Pool(a.procs) as pool:
        res = pool.map(worker, [(a.workload, a.seconds, k % 8) for k in range(a.procs)])
    it = sum(r[0] / r[2] for r in res); fr = sum(r[1] / r[2] for r in res)
    print("%s, %d processes (host has %d CPUs): %.0f solver iterations/s, %.0f frames/s in total; %.0f it/s per process" % (
        a.workload, a.procs, os.cpu_count(), it, fr, it / a.procs))
