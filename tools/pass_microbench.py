"""The IRLS streaming passes in isolation (sf_irls_pass_kernel): achieved GB/s per pass and ablations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
os.environ.setdefault("SF_VARIANT", "throughput")  # the 256-thread build the numbers in DESIGN.md §5.1 refer to (a batch of 512 would select the other)
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--workload", default="static")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--slices", type=int, default=1, help="workgroups per stream (experiment: records resident in the Infinity Cache)")
a = ap.parse_args()
api = sf.load()
p = bench.make_params(api, a.workload)
pairs = make_batch(8, sphere=(a.workload == "sphere"), distinct=8)
s = sf.Solver(api, 240, 320, a.batch, p)
for b in range(a.batch):
    s.set_current(b, *pairs[b % 8]["new"]); s.set_prediction(b, *pairs[b % 8]["old"])
s.process_frame(0); s.synchronize()
npx = 240 * 320
for which in (1, 2):
    for variant, name in ((0, "product"), (1, "loads only"), (2, "no accumulation")):
        s.microbench_pass(which, variant | (a.slices << 8), 2)
        ms = s.microbench_pass(which, variant | (a.slices << 8), a.reps)
        px = a.batch * a.reps * npx
        bpp = 29.0 if p.segmentation_enabled else 28.0  # 7 float planes (+ 1 label byte with segmentation)
        print("pass %d %-16s %8.3f ms  %6.2f Gpx/s  streamed(%d B/px) %7.1f GB/s  algorithmic(30 B/px/pass) %7.1f GB/s" % (
            which, name, ms, px / ms / 1e6, bpp, bpp * px / ms / 1e6, 30.0 * px / ms / 1e6))
import ctypes as C
t = (C.c_int64 * 32)()
api.check(api.get_stage_profile(s.h, t))
if t[23] > 0:
    print("shader clock during the last pass launch: %.0f MHz (s_memtime ticks %d / 100 MHz ticks %d)" % (100.0 * t[22] / t[23], t[22], t[23]))
