"""End-to-end latency of ONE stream, per frame, the way a live driver would call the library: decoded VGA frame in
host memory -> sf_load_frame -> sf_filter_depth -> sf_process_frame -> T_odometry and b_segm_perpixel back on the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd.synth import DEFAULT_XI, Scene, se3_exp

api = sf.load()
p = api.default_params_struct(); p.kb = 1.05
s = sf.Solver(api, 240, 320, 1, p)
scene = Scene(seed=5, sphere=True)
frames, T = [], np.eye(4)
for k in range(12):
    depth, inten = scene.render(T, 640, 480, sphere_offset=(0.01 * k, 0, 0))
    d_mm = np.clip(np.rint(depth * 1000.0), 0, 65535).astype(np.uint16)[::-1].copy()
    g8 = np.clip(np.rint(inten * 255.0), 0, 255).astype(np.uint8)[::-1]
    frames.append((np.ascontiguousarray(np.repeat(g8[:, :, None], 3, axis=2)), d_mm))
    T = T @ se3_exp(np.array(DEFAULT_XI) * 0.5)
s.load_frame(0, *frames[0], 2); s.filter_depth(); s.current_to_prediction(); s.push_history(0)
ts = {"load": 0.0, "filter+solve": 0.0, "read back": 0.0}
n = 0
for rep in range(4):
    for k in range(1, 12):
        t0 = time.perf_counter()
        s.load_frame(0, *frames[k], 2)
        t1 = time.perf_counter()
        s.filter_depth(); s.process_frame(rep * 12 + k); s.synchronize()
        t2 = time.perf_counter()
        Tk = s.T(); b = s.b_image()
        t3 = time.perf_counter()
        s.current_to_prediction()
        if rep:  # the first round warms up
            ts["load"] += t1 - t0; ts["filter+solve"] += t2 - t1; ts["read back"] += t3 - t2; n += 1
tot = sum(ts.values()) / n
print("one stream, per frame: " + ", ".join("%s %.2f ms" % (k, 1e3 * v / n) for k, v in ts.items()) + "; total %.2f ms = %.0f frames/s" % (1e3 * tot, 1 / tot))
