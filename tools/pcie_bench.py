"""PCIe-inclusive throughput of the solver step: every step's depthCurrent / intensityCurrent come from HOST memory.
  serial   : sf_upload_current_async + sf_commit_upload + sf_process_frame + synchronize, one after the other
  overlap  : the upload of step k+1 (second HIP stream, page-locked host buffers) runs while step k is solved
The prediction is the previous frame (frame-to-frame mode: sf_current_to_prediction on the device), so one step
moves 2 x 307 KB per stream over PCIe."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, ctypes as C
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--workload", default="static")
a = ap.parse_args()
api = sf.load()
p = bench.make_params(api, a.workload)
pairs = make_batch(8, sphere=(a.workload == "sphere"), distinct=8)
B, n0 = a.batch, 240 * 320
s = sf.Solver(api, 240, 320, B, p)
fp = C.POINTER(C.c_float)
bufs = []
for which in ("old", "new"):  # two host frames per stream, alternating: A, B, A, B ... (page-locked)
    pd, pi = C.c_void_p(), C.c_void_p()
    api.check(api.alloc_pinned(4 * n0 * B, C.byref(pd))); api.check(api.alloc_pinned(4 * n0 * B, C.byref(pi)))
    d = np.ctypeslib.as_array(C.cast(pd, fp), shape=(B, n0)); i = np.ctypeslib.as_array(C.cast(pi, fp), shape=(B, n0))
    for b in range(B):
        d[b] = np.ascontiguousarray(pairs[b % 8][which][0].T).ravel(); i[b] = np.ascontiguousarray(pairs[b % 8][which][1].T).ravel()
    bufs.append((C.cast(pd, fp), C.cast(pi, fp)))
def upload(k):
    api.check(api.upload_current_async(s.h, *bufs[k % 2]))
# prime: frame 0 becomes the prediction, history filled
upload(0); api.check(api.commit_upload(s.h)); s.current_to_prediction()
for im in range(5):
    upload(im + 1); api.check(api.commit_upload(s.h)); s.process_frame(im); s.current_to_prediction()
s.synchronize()
im = 5
# serial
t0 = time.perf_counter()
for k in range(a.steps):
    upload(k); api.check(api.commit_upload(s.h)); s.process_frame(im + k); s.current_to_prediction(); s.synchronize()
t_serial = (time.perf_counter() - t0) / a.steps
im += a.steps
# overlapped: upload(k+1) is issued before process_frame(k) is waited for
upload(0)
t0 = time.perf_counter()
for k in range(a.steps):
    api.check(api.commit_upload(s.h))
    s.process_frame(im + k)
    s.current_to_prediction()
    upload(k + 1)  # runs on the copy stream while the frame kernel above executes
s.synchronize()
t_overlap = (time.perf_counter() - t0) / a.steps
api.check(api.commit_upload(s.h)); s.synchronize()
ms_solver = s.timed_process_frames(im + a.steps, 3) / 3
gb = 2 * 4 * n0 * B / 1e9
print("batch %d, %s: solver alone %.1f ms/step; PCIe-inclusive serial %.1f ms/step (%.0f frames/s), overlapped %.1f ms/step (%.0f frames/s); %.2f GB per step over PCIe" % (
    B, a.workload, ms_solver, t_serial * 1e3, B / t_serial, t_overlap * 1e3, B / t_overlap, gb))
