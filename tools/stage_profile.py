"""Per-stage time breakdown from the in-kernel timers (100 MHz wall clock, lane 0 of each workgroup)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--workload", default="static")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--lib", default=None, help="another build of the library (e.g. staticfusion_amd/csrc/libsf_hip_kmprof.so)")
ap.add_argument("--variant", default="auto")
ap.add_argument("--res-factor", type=int, default=2, help="2: QVGA 320x240 (the drivers' setting), 1: VGA 640x480")
a = ap.parse_args()
api = (sf.load() if a.lib is None else sf.Api(a.lib, "sf_")).with_variant(a.variant)
rows, cols = 480 // a.res_factor, 640 // a.res_factor
p = bench.make_params(api, a.workload)
if a.res_factor == 1 and a.workload == "static":
    p.ctf_levels = 4  # one more level than at QVGA, the same coarsest resolution
if a.res_factor == 1 and a.workload == "sphere":
    p.ctf_levels = 6  # the constructor's log2(cols / 40) + 2
pairs = make_batch(8, sphere=(a.workload == "sphere"), distinct=8, out_rows=rows, out_cols=cols)
s = sf.Solver(api, rows, cols, a.batch, p)
for b in range(a.batch):
    s.set_current(b, *pairs[b % 8]["new"]); s.set_prediction(b, *pairs[b % 8]["old"])
for im in range(5):
    s.process_frame(im)
s.synchronize()
p0 = s.stage_profile(); c0 = s.counters()
ms = s.timed_process_frames(5, a.steps)
p1 = s.stage_profile(); c1 = s.counters()
frames = c1[0] - c0[0]
print("workload %s %dx%d batch %d: %.2f ms/step, %.0f frames/s, %.0f it/s" % (a.workload, cols, rows, a.batch, ms / a.steps, frames / (ms * 1e-3), (c1[1] - c0[1]) / (ms * 1e-3)))
tot = p1["total"] - p0["total"]
for k in s.STAGES:
    d = p1[k] - p0[k]
    print("  %-11s %8.1f us/frame  %5.1f %%" % (k, 1e6 * d / frames, 100 * d / tot))

if a.lib and "filterprof" in a.lib:
    import ctypes
    t0 = (ctypes.c_int64 * 32)()
    api.check(api.get_stage_profile(s.h, t0))
    print("  filter: 6x6 inverse %.1f us, Jacobi %.1f us, rest (log / exp / products on one lane) %.1f us per frame" % tuple(1e-2 * t0[q] / (frames + 5 * a.batch) for q in (21, 22, 23)))
if a.lib and "kmcfine" in a.lib:
    import ctypes
    t0 = (ctypes.c_int64 * 32)()
    api.check(api.get_stage_profile(s.h, t0))
    names = ["loads+counts", "barrier 1", "offsets", "scatter", "barrier 2", "sums", "publish"]
    print("  cluster k-means collect+sum, shader cycles per frame (divide by ~2300 for us): " + ", ".join("%s %.0fk" % (n, 1e-3 * t0[16 + i] / (frames + 5 * a.batch)) for i, n in enumerate(names)))
if a.lib and "kmprof" in a.lib:
    import ctypes
    t = (ctypes.c_int64 * 32)()
    api.check(api.get_stage_profile(s.h, t))
    print("  Lloyd search trips (wave 0): %d over %d chunks = %.2f per chunk; k-means iterations of stream 0: %d" % (t[21], t[22], t[21] / max(1, t[22]), s.stats(0).kmeans_iters))
