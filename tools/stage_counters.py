"""SQ counters of ONE stage of the frame kernel (rocprofv3 --pmc): the stage is launched alone, `reps` times, after the
stream state it needs has been prepared; tools/stage_counters.sh wraps this in the rocprofv3 passes and sums the counters
of the last `reps` dispatches.

    python tools/stage_counters.py --stage kmeans --batch 4096 --reps 3
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--stage", default="kmeans", choices=["kmeans", "pyramid", "solver", "residuals", "frame"])
ap.add_argument("--workload", default="sphere")
ap.add_argument("--variant", default="throughput")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
api = sf.load().with_variant(a.variant)
p = bench.make_params(api, a.workload)
pairs = make_batch(8, sphere=(a.workload == "sphere"), distinct=8)
s = sf.Solver(api, 240, 320, a.batch, p)
for b in range(a.batch):
    s.set_current(b, *pairs[b % 8]["new"]); s.set_prediction(b, *pairs[b % 8]["old"])
for im in range(6):
    s.process_frame(im)
s.synchronize()
for r in range(a.reps):
    if a.stage == "kmeans":
        s.kmeans()
    elif a.stage == "pyramid":
        s.build_pyramid(True)
    elif a.stage == "solver":
        s.run_solver(False)
    elif a.stage == "residuals":
        s.residuals_vs_history(6)
    else:
        s.process_frame(6 + r)
s.synchronize()
print("stage %s x %d on %d streams done; last solver kernel %.3f ms" % (a.stage, a.reps, a.batch, s.last_solver_kernel_ms() if a.stage in ("solver", "frame") else -1))
