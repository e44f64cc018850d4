"""Per-cluster 5-frame residuals (A10): largest relative distance HIP vs oracle over the frames of the sequence test, against
the plain oracle and against the oracle with its per-cluster float sums in fp64 (sfo_test_set_exact_sums)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from conftest import driver_params, make_solver
from oracle import binding
from staticfusion_amd.synth import DEFAULT_XI, Scene, quantise_and_decimate, se3_exp
ora = binding.load()
ora.lib.sfo_test_set_exact_sums.argtypes = [ctypes.c_void_p, ctypes.c_int]
for variant in ("throughput", "cluster"):
    hip = sf.load().with_variant(variant)
    for seed, scale in ((77, 0.6), (78, 1.5), (79, 0.3)):
        scene = Scene(seed=seed, sphere=True)
        xi = np.array(DEFAULT_XI) * scale
        frames, T = [], np.eye(4)
        for k in range(9):
            frames.append(quantise_and_decimate(*scene.render(T, 640, 480, sphere_offset=(0.02 * k, 0, 0))))
            T = T @ se3_exp(xi)
        S = [make_solver(hip, 240, 320, driver_params(hip, kb=1.5)), make_solver(ora, 240, 320, driver_params(ora, kb=1.5)), make_solver(ora, 240, 320, driver_params(ora, kb=1.5))]
        ora.lib.sfo_test_set_exact_sums(S[2].h, 1)
        for s in S:
            s.set_current(0, *frames[0]); s.current_to_prediction(); s.push_history(0)
        w = [0.0, 0.0, 0.0]
        for k in range(1, 9):
            for s in S:
                s.set_prediction(0, *frames[k - 1]); s.set_current(0, *frames[k]); s.process_frame(k)
            c = [s.cluster_residuals() for s in S]
            m = ~np.isnan(c[1])
            if m.any():
                w[0] = max(w[0], float(np.abs(c[0][m] / c[1][m] - 1).max()))
                w[1] = max(w[1], float(np.abs(c[0][m] / c[2][m] - 1).max()))
                w[2] = max(w[2], float(np.abs(c[1][m] / c[2][m] - 1).max()))
        print("%s seed %d scale %.1f: cluster residuals rel: HIP-oracle %.2e | HIP-oracle(fp64 sums) %.2e | oracle-oracle(fp64 sums) %.2e" % (variant, seed, scale, *w))
