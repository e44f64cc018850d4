"""Long run of the full loop (prediction -> input stage -> solver -> fusion) for one stream on a synthetic orbit: the map must
stay bounded and finite, the pose must stay near the ground truth, device memory must not grow.

usage: python tools/soak_fusion.py [--frames 300]
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


def free_bytes(hiprt):
    f, t = C.c_size_t(), C.c_size_t()
    assert hiprt.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    a = ap.parse_args()
    api = sf.load()
    hiprt = C.CDLL("libamdhip64.so")
    s = sf.Solver(api, 240, 320, 1, api.default_params_struct())
    m = sf.SurfelMap(s, 2_000_000)
    mp = s.default_model_params()
    scene = Scene(seed=5, sphere=True)
    T, worst, counts, mems = np.eye(4), (0.0, 0.0), [], []
    t0 = time.time()
    mem0 = None
    for k in range(a.frames):
        # a slow sway: the increments change sign every 40 frames, so the camera revisits what it has mapped
        sgn = 1.0 if (k // 40) % 2 == 0 else -1.0
        xi = sgn * np.array([0.006, 0.002, 0.003, 0.001, -0.004, 0.002])
        depth, inten = scene.render(T, 640, 480, sphere_offset=(0.01 * np.sin(k / 15.0), 0, 0))
        g = np.clip(np.rint(inten * 255), 1, 255).astype(np.uint8)
        full_c = np.repeat(g[::-1, :, None], 3, axis=2)
        full_d = np.clip(np.rint(depth[::-1] * 1000), 0, 65535).astype(np.uint16)
        if k >= 2:
            s.set_kb(1.05 if k == 2 else 1.5)
            m.predict(0, mp)
        s.load_frame(0, full_c, full_d, 2)
        if k == 0:
            s.current_to_prediction()
            s.push_history(0)
            s.set_kb(1.05)
            T = T @ se3_exp(xi)
            continue
        if k >= 2:
            s.filter_depth()
        s.process_frame(k)
        if k == 1:
            s.filter_depth()
        m.fuse_frame(0, s.T(), 1.0, mp)
        info = m.info()
        counts.append(info["count"])
        err = pose_delta(info["pose"], T)
        worst = (max(worst[0], err[0]), max(worst[1], err[1]))
        if k == 20:
            mem0 = free_bytes(hiprt)
        if k in (100, 200):
            mems.append(mem0 - free_bytes(hiprt))
        if k % 50 == 0 or k == a.frames - 1:
            sfl = m.download()
            assert np.isfinite(sfl[:, :4]).all(), "non-finite position / confidence in the map"
            print("frame %4d: %7d surfels (stable %6d), pose error %.2e rad %.2e m, %.1f frames/s incl. rendering"
                  % (k, info["count"], int((sfl[:, 3] > mp.conf_high).sum()), err[0], err[1], k / (time.time() - t0)))
        T = T @ se3_exp(xi)
    mems.append(mem0 - free_bytes(hiprt))
    print("worst pose error %.2e rad %.2e m; map %d..%d surfels; device memory in use beyond frame 20's, at frames 100 / 200 / end: %s bytes"
          % (worst[0], worst[1], min(counts), max(counts), mems))
    # the HIP runtime grows its own pools in 2 MB steps early on; a leak would keep growing
    assert worst[0] < 2e-2 and worst[1] < 5e-2 and max(counts) < 1_000_000 and mems[-1] <= mems[0] + (2 << 20)
    print("soak ok")


if __name__ == "__main__":
    main()
