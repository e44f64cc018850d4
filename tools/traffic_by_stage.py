"""HBM bytes per stage group: the per-frame calls of include/sf.h issued one by one (each is its own launch of
sf_frame_kernel with a stage mask), to be run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`.
The last 7 dispatches are, in order: pyramid(old) | pyramid(new) | K-means (or pyramid(new) again without segmentation) | K-means + solver | residuals | segm image | ring push."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--workload", default="static")
a = ap.parse_args()
api = sf.load()
p = bench.make_params(api, a.workload)
pairs = make_batch(8, sphere=(a.workload == "sphere"), distinct=8)
s = sf.Solver(api, 240, 320, a.batch, p)
for b in range(a.batch):
    s.set_current(b, *pairs[b % 8]["new"]); s.set_prediction(b, *pairs[b % 8]["old"])
for im in range(6):
    s.process_frame(im)
s.synchronize()
s.build_pyramid(True)
s.build_pyramid(False)
if p.segmentation_enabled:
    s.kmeans()
else:
    s.build_pyramid(False)  # keeps the number of launches the same
s.run_solver(False)
s.residuals_vs_history(6)
s.build_segm_image()
s.push_history(6)
s.synchronize()
