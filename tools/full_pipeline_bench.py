"""The reference's whole frame loop (StaticFusion-imagesequenceassoc.cpp:140-191) for N sequences on one MI355X, every stage on
the GPU and batched over the sequences:

    getPredictedImages (sf_map_predict_frames)  ->  load + decimate + bilateral filter (sf_load_frame_device, sf_filter_depth)
    ->  createImagePyramid / runSolver / residuals / buildSegmImage (sf_process_frame)  ->  fuseFrame (sf_map_fuse_frames)

The decoded VGA frames of a synthetic walk are resident in HBM (every sequence sees the same walk; the work does not depend on
that). Reports full-pipeline frames/s and the share of each stage (host wall time around synchronised calls).

usage: python tools/full_pipeline_bench.py [--streams 256] [--frames 10] [--sphere]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import staticfusion_amd as sf
from staticfusion_amd.synth import Scene, pose_delta, se3_exp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--sphere", action="store_true")
    ap.add_argument("--capacity", type=int, default=0)
    a = ap.parse_args()
    api = sf.load()
    rows, cols, res = 240, 320, 2
    n = a.streams
    s = sf.Solver(api, rows, cols, n, api.default_params_struct())
    maps = [sf.SurfelMap(s, a.capacity or 4 * rows * cols) for _ in range(n)]
    mp = s.default_model_params()
    streams = list(range(n))
    scene = Scene(seed=77, sphere=a.sphere)
    xi = np.array([0.010, 0.004, 0.006, 0.002, -0.004, 0.003]) * 0.6
    hiprt = C.CDLL("libamdhip64.so")

    def to_device(arr):
        ptr = C.c_void_p()
        assert hiprt.hipMalloc(C.byref(ptr), C.c_size_t(arr.nbytes)) == 0
        assert hiprt.hipMemcpy(ptr, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0
        return ptr

    T, gts, dev = np.eye(4), [], []
    for k in range(a.frames):  # the same decoded frame for every sequence, replicated in HBM
        depth, inten = scene.render(T, 640, 480, sphere_offset=(0.02 * k, 0, 0))
        g = np.clip(np.rint(inten * 255), 0, 255).astype(np.uint8)
        col = np.ascontiguousarray(np.broadcast_to(np.repeat(g[::-1, :, None], 3, axis=2), (n, 480, 640, 3)))
        dep = np.ascontiguousarray(np.broadcast_to(np.clip(np.rint(depth[::-1] * 1000), 0, 65535).astype(np.uint16), (n, 480, 640)))
        dev.append((to_device(col), to_device(dep)))
        gts.append(T.copy())
        T = T @ se3_exp(xi)

    def load(k, filtered):
        api.check(api.load_frame_device(s.h, dev[k][0], dev[k][1], 480, 640, res))
        if filtered:
            s.filter_depth()

    # bootstrap (:102-137)
    load(0, False)
    s.current_to_prediction()
    s.push_history(0)
    s.set_kb(1.05)
    load(1, False)
    s.process_frame(1)
    Tq = s.batch_results()[0]
    s.filter_depth()
    sf.SurfelMap.fuse_frames(s, streams, maps, list(Tq), 1.0, mp)
    stages = dict(predict=0.0, input=0.0, solve=0.0, fuse=0.0)
    timed = 0
    t_start = None
    for k in range(2, a.frames):
        if k == 3:  # frame 2 is the warm-up of the steady state
            s.synchronize()
            t_start = time.perf_counter()
            stages = dict.fromkeys(stages, 0.0)
            timed = 0
        s.set_kb(1.05 if k == 2 else 1.5)
        t0 = time.perf_counter()
        sf.SurfelMap.predict_frames(s, streams, maps, mp)
        s.synchronize()
        t1 = time.perf_counter()
        load(k, True)
        s.synchronize()
        t2 = time.perf_counter()
        s.process_frame(k)
        Tq = s.batch_results()[0]
        t3 = time.perf_counter()
        sf.SurfelMap.fuse_frames(s, streams, maps, list(Tq), 1.0, mp)
        t4 = time.perf_counter()
        for name, dt in zip(("predict", "input", "solve", "fuse"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            stages[name] += dt
        timed += 1
    total = time.perf_counter() - t_start
    info = maps[0].info()
    err = pose_delta(info["pose"], gts[a.frames - 1])
    print("full pipeline, %d sequences x %d timed frames (%s): %.2f ms per step, %.0f frames/s; map %d surfels; pose error vs ground truth %.2e rad %.2e m"
          % (n, timed, "moving sphere" if a.sphere else "static", 1e3 * total / timed, n * timed / total, info["count"], err[0], err[1]))
    for name in ("predict", "input", "solve", "fuse"):
        print("  %-8s %8.2f ms per step  %5.1f %%   %7.1f us per sequence" % (name, 1e3 * stages[name] / timed, 100 * stages[name] / total, 1e6 * stages[name] / timed / n))


if __name__ == "__main__":
    main()
