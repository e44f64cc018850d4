"""Throughput of the input stage (sf_load_frame_device + sf_filter_depth) with the decoded frames resident in HBM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import argparse, ctypes as C
import numpy as np
import staticfusion_amd as sf
from make_golden_input import synth_frame

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--calls", type=int, default=5)
a = ap.parse_args()
api = sf.load()
H, W, res = 480, 640, 2
frames = [synth_frame(H, W, 100 + k) for k in range(4)]
col = np.ascontiguousarray(np.stack([frames[b % 4][0] for b in range(a.batch)]))
dep = np.ascontiguousarray(np.stack([frames[b % 4][1] for b in range(a.batch)]))
hiprt = C.CDLL("libamdhip64.so")
ptrs = []
for arr in (col, dep):
    ptr = C.c_void_p()
    assert hiprt.hipMalloc(C.byref(ptr), C.c_size_t(arr.nbytes)) == 0
    assert hiprt.hipMemcpy(ptr, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0
    ptrs.append(ptr)
p = api.default_params_struct()
s = sf.Solver(api, H // res, W // res, a.batch, p)
ms = C.c_float()
api.check(api.timed_input_stage(s.h, ptrs[0], ptrs[1], H, W, res, 1, C.byref(ms)))
api.check(api.timed_input_stage(s.h, ptrs[0], ptrs[1], H, W, res, a.calls, C.byref(ms)))
fps = a.batch * a.calls / (ms.value * 1e-3)
taps = 169.0 * (H // res) * (W // res)
print("input stage batch %d: %.3f ms per call, %.0f frames/s, %.2f G filter taps/s" % (a.batch, ms.value / a.calls, fps, fps * taps / 1e9))
