#!/usr/bin/env python3
"""Kernel time of ONE launch of K sequence frames as a function of K (and of the stream count): the fixed part of a launch.

    python tools/diag/launch_overhead.py [--batch 4096] [--ks 1,2,3,4,6,8,12,20,40,60] [--variant auto]

Prints, per K, the launch's duration by the handle's HIP events (sf_last_solver_kernel_ms), the duration per frame, and a
least-squares line t(K) = a + b K over the K >= 8 points: b is the steady-state cost of a frame of every stream, a what a
launch pays once (ramp-up of the resident workgroups in lock-step, the drain of the last frames).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--ks", default="1,2,3,4,6,8,12,20,40,60")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--stages", action="store_true", help="print the stage timers of every launch of the first repetition")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--workload", "sequences", "--batch", str(a.batch), "--variant", a.variant, "--no-cpu-baseline"]
    args = bench.parse()
    hx = bench.Harness(args)
    import staticfusion_amd as sf

    pool = bench.synthetic_sequence_pool(hx, args)
    api = sf.load()
    params = bench.make_params(api, "sequences")
    B, D, F = a.batch, pool["D"], pool["F"]
    solver = sf.Solver(api, pool["rows"], pool["cols"], B, params, device=hx.dev_index, variant=a.variant)
    phase = (np.arange(B) // D * 7) % (F - 1)
    seq_of = np.arange(B) % D

    def index_at(step):
        return (seq_of * F + (phase + step) % F).astype(np.int32)

    pd, pi = pool["d"].data_ptr(), pool["i"].data_ptr()
    solver.advance_sequences_device(pd, pi, index_at(0), D * F)
    solver.push_history(0)
    step = 1
    for _ in range(8):
        solver.advance_sequences_device(pd, pi, index_at(step), D * F)
        solver.process_frame(step)
        step += 1
    solver.synchronize()
    ks = [int(k) for k in a.ks.split(",")]
    resident = min(solver.resident_workgroups()[1], B)
    rows = []
    for rep in range(a.repeat):
        for K in ks:
            p0 = solver.stage_profile()
            r0 = solver.shader_clock_counters()
            solver.process_sequence_frames_device(pd, pi, np.stack([index_at(step + q) for q in range(K)]), D * F, step)
            step += K
            solver.synchronize()
            ms = solver.last_solver_kernel_ms()
            p1 = solver.stage_profile()
            mhz = solver.shader_clock_mhz(r0, solver.shader_clock_counters())  # inside the stream-frames of this launch
            rows.append((K, ms))
            # in-kernel timers: the time the workgroups spent INSIDE stream-frames, against resident workgroups x launch time
            busy = (p1["total"] - p0["total"]) * 1e3
            print("rep %d  K %3d  launch %9.2f ms  per frame %8.3f ms   workgroup-frame %7.3f ms   inside stream-frames %5.1f %% of %d workgroups x launch   shader clock %6.0f MHz"
                  % (rep, K, ms, ms / K, busy / (B * K), 100 * busy / (resident * ms), resident, mhz), flush=True)
            if rep == 0 and a.stages:
                print("        " + "  ".join("%s %.0f" % (n, 1e6 * (p1[n] - p0[n]) / (B * K)) for n in solver.STAGES if p1[n] - p0[n] > 0) + "  (us per workgroup-frame)")
    x = np.array([r[0] for r in rows if r[0] >= 8], float)
    y = np.array([r[1] for r in rows if r[0] >= 8], float)
    b, c = np.polyfit(x, y, 1)
    print("batch %d variant %s: t(K) = %.2f ms + %.3f ms * K   (K >= 8)" % (B, solver.variant()[0], c, b))
    for K, ms in rows:
        print("  K %3d  measured - b K = %7.2f ms" % (K, ms - b * K))
    solver.close()
    hx.close()


if __name__ == "__main__":
    main()
