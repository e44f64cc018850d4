"""Throughput of the frame-to-model prediction (sf_predict_from_model_device) with the surfel buffer resident in HBM."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import argparse, ctypes as C
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd.synth import DEFAULT_XI, se3_exp
from test_model_prediction import surfels_from_frame, synthetic_view, prime_stream, ROWS, COLS

ap = argparse.ArgumentParser()
ap.add_argument("--copies", type=int, default=16, help="the 76 800-surfel view replicated this many times (a map of ~1.2 M surfels)")
ap.add_argument("--calls", type=int, default=20)
a = ap.parse_args()
api = sf.load()
s = sf.Solver(api, ROWS, COLS, 1, api.default_params_struct())
mp = s.default_model_params()
depth0, rgb0 = synthetic_view(np.eye(4))
T1 = se3_exp(np.array(DEFAULT_XI) * 3.0).astype(np.float32)
prime_stream(s, depth0, rgb0, 0.9)
base = surfels_from_frame(depth0, rgb0, np.eye(4), mp, step=1, seed=1)
rng = np.random.default_rng(0)
model = np.concatenate([base + np.pad(rng.normal(0, 0.002, (base.shape[0], 3)), ((0, 0), (0, 9))).astype(np.float32) for _ in range(a.copies)])
hiprt = C.CDLL("libamdhip64.so")
ptr = C.c_void_p()
assert hiprt.hipMalloc(C.byref(ptr), C.c_size_t(model.nbytes)) == 0
assert hiprt.hipMemcpy(ptr, model.ctypes.data_as(C.c_void_p), C.c_size_t(model.nbytes), 1) == 0
Tcm = np.ascontiguousarray(T1.T)
fp = C.POINTER(C.c_float)
def call():
    api.check(api.predict_from_model_device(s.h, 0, ptr, model.shape[0], Tcm.ctypes.data_as(fp), C.byref(mp)))
call(); s.synchronize()
t0 = time.perf_counter()
for _ in range(a.calls):
    call()
s.synchronize()
dt = (time.perf_counter() - t0) / a.calls
print("prediction from %d surfels: %.3f ms per call, %.1f M surfels/s, %.0f predictions/s" % (model.shape[0], dt * 1e3, model.shape[0] / dt / 1e6, 1 / dt))
