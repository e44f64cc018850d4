"""Wall time of one Reconstruction::fuseFrame + getPredictedImages on the HIP surfel map (sf_map_fuse_frame, sf_map_predict).
The call ends with a 32-byte read-back of the new surfel count, so host wall time = device time + one synchronisation.

usage: python tools/fusion_bench.py [--frames 12] [--pad-surfels 2000000] [--res-factor 2] [--maps N --capacity C]
--maps N: N sequences in one handle, each with its own map, fused and predicted by ONE sf_map_fuse_frames /
sf_map_predict_frames call per frame (the many-sequences-per-GPU form); reports maps/s.
--pad-surfels N: before timing, append N surfels that lie behind the camera (they cost the per-surfel kernels their
streaming time but never project), to see how the frame time grows with the size of the map.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import staticfusion_amd as sf
from staticfusion_amd.synth import Scene, se3_exp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--pad-surfels", type=int, default=0)
    ap.add_argument("--res-factor", type=int, default=2)
    ap.add_argument("--maps", type=int, default=1)
    ap.add_argument("--capacity", type=int, default=0, help="surfels per map (0: the reference's 3072 x 3072; with --maps > 1 default 4 frames' worth)")
    a = ap.parse_args()
    if a.maps > 1:
        return batched(a)
    api = sf.load()
    rows, cols = 480 // a.res_factor, 640 // a.res_factor
    p = api.default_params_struct()
    if a.res_factor == 1:
        p.ctf_levels = 6
    s = sf.Solver(api, rows, cols, 1, p)
    m = sf.SurfelMap(s, 0)
    mp = s.default_model_params()
    scene = Scene(seed=99, sphere=False)
    xi = np.array([0.010, 0.004, 0.006, 0.002, -0.004, 0.003])
    T = np.eye(4)
    labels = np.zeros((rows, cols), np.int32)
    t_fuse, t_pred = [], []
    for k in range(a.frames):
        depth, inten = scene.render(T, 640, 480)
        g = np.clip(np.rint(inten * 255), 1, 255).astype(np.uint8)
        full_c = np.repeat(g[::-1, :, None], 3, axis=2)
        full_d = np.clip(np.rint(depth[::-1] * 1000), 0, 65535).astype(np.uint16)
        s.load_frame(0, full_c, full_d, a.res_factor)
        s.filter_depth()
        s.set_segm_state(0, labels, np.full(24, 0.9, np.float32), np.ones(24, np.float32))
        s.build_segm_image()
        s.synchronize()
        t0 = time.perf_counter()
        m.fuse_frame(0, None if k == 0 else se3_exp(xi), 1.0, mp)
        t1 = time.perf_counter()
        m.predict(0, mp)
        s.synchronize()
        t2 = time.perf_counter()
        info = m.info()
        if k >= 2:
            t_fuse.append(t1 - t0)
            t_pred.append(t2 - t1)
        print("frame %2d: %8d surfels, emitted %6d associated %6d, fuse %.3f ms, predict %.3f ms" % (k, info["count"], info["stats"][0], info["stats"][1],
                                                                                                 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
        if k == 1 and a.pad_surfels:
            sfl = m.download()
            pad = np.zeros((a.pad_surfels, 12), np.float32)
            pad[:, 2] = -5.0  # behind the camera
            pad[:, 3] = 0.9
            pad[:, 6] = pad[:, 7] = 1
            pad[:, 10] = 1
            pad[:, 11] = 0.01
            m.upload(np.concatenate([sfl, pad]), info["pose"], info["tick"])
        T = T @ se3_exp(xi)
    print("median over frames 2..: fuse %.3f ms, predict %.3f ms (%dx%d, %d surfels)" % (1e3 * np.median(t_fuse), 1e3 * np.median(t_pred), cols, rows, m.info()["count"]))


def batched(a):
    api = sf.load()
    rows, cols = 480 // a.res_factor, 640 // a.res_factor
    p = api.default_params_struct()
    if a.res_factor == 1:
        p.ctf_levels = 6
    n = a.maps
    s = sf.Solver(api, rows, cols, n, p)
    cap = a.capacity or 4 * rows * cols
    maps = [sf.SurfelMap(s, cap) for _ in range(n)]
    mp = s.default_model_params()
    scene = Scene(seed=99, sphere=False)
    xi = np.array([0.010, 0.004, 0.006, 0.002, -0.004, 0.003])
    T = np.eye(4)
    labels = np.zeros((rows, cols), np.int32)
    streams = list(range(n))
    t_fuse, t_pred = [], []
    for k in range(a.frames):
        depth, inten = scene.render(T, 640, 480)
        g = np.clip(np.rint(inten * 255), 1, 255).astype(np.uint8)
        full_c = np.repeat(g[::-1, :, None], 3, axis=2)
        full_d = np.clip(np.rint(depth[::-1] * 1000), 0, 65535).astype(np.uint16)
        for q in streams:
            s.load_frame(q, full_c, full_d, a.res_factor)
            s.set_segm_state(q, labels, np.full(24, 0.9, np.float32), np.ones(24, np.float32))
        s.filter_depth()
        s.build_segm_image()
        s.synchronize()
        t0 = time.perf_counter()
        sf.SurfelMap.fuse_frames(s, streams, maps, None if k == 0 else [se3_exp(xi)] * n, 1.0, mp)
        t1 = time.perf_counter()
        sf.SurfelMap.predict_frames(s, streams, maps, mp)
        s.synchronize()
        t2 = time.perf_counter()
        if k >= 2:
            t_fuse.append(t1 - t0)
            t_pred.append(t2 - t1)
        T = T @ se3_exp(xi)
    info = maps[-1].info()
    f, pr = np.median(t_fuse), np.median(t_pred)
    print("%d maps of %d surfels (%dx%d), one batched call per frame: fuse %.3f ms (%.1f us per map, %.0f maps/s), predict %.3f ms (%.1f us per map); "
          "fuse + predict %.0f maps/s" % (n, info["count"], cols, rows, 1e3 * f, 1e6 * f / n, n / f, 1e3 * pr, 1e6 * pr / n, n / (f + pr)))


if __name__ == "__main__":
    main()
