"""Minimal driver for rocprofv3: N timed steps of sf_process_frame on a synthetic batch (no torch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--workload", default="static")
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
api = sf.load()
p = bench.make_params(api, a.workload)
pairs = make_batch(8, sphere=(a.workload == "sphere"), distinct=8)
s = sf.Solver(api, 240, 320, a.batch, p)
for b in range(a.batch):
    s.set_current(b, *pairs[b % 8]["new"]); s.set_prediction(b, *pairs[b % 8]["old"])
for im in range(5):
    s.process_frame(im)
s.synchronize()
ms = s.timed_process_frames(5, a.steps)
print("steps %d batch %d: %.3f ms/step" % (a.steps, a.batch, ms / a.steps))
