"""Where does the b deviation between the HIP path and the oracle arise? Per outer iteration: b, b_prior, aver_res, var."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from conftest import driver_params, make_solver, trace_array
from oracle import binding
from staticfusion_amd.synth import make_pair
ora = binding.load(); hip = sf.load().with_variant("throughput")
ora.lib.sfo_test_set_exact_sums.argtypes = [ctypes.c_void_p, ctypes.c_int]
for seed in range(1234, 1240):
    pr = make_pair(seed=seed, sphere=True, out_rows=240, out_cols=320)
    S = []
    for api, ex in ((hip, 0), (ora, 0), (ora, 1)):
        s = make_solver(api, 240, 320, driver_params(api), pr)
        if ex: ora.lib.sfo_test_set_exact_sums(s.h, 1)
        s.build_pyramid(True); s.run_solver(True); S.append(s.stats())
    a, b, x = S
    print("seed", seed, "n_outer", a.n_outer)
    for i in range(a.n_outer):
        db = np.abs(np.array(a.outer[i].b_segm[:]) - np.array(x.outer[i].b_segm[:]))
        dbo = np.abs(np.array(b.outer[i].b_segm[:]) - np.array(x.outer[i].b_segm[:]))
        dp = np.abs(np.array(a.outer[i].b_prior[:]) - np.array(x.outer[i].b_prior[:]))
        l = int(db.argmax())
        print("  outer %2d level %d k %d nv %6d/%6d it %d | db hip-exact %.2e (label %2d, b=%.4f lt=%.3f) ora32-exact %.2e | dprior %.2e | aver_res rel hip %.2e ora32 %.2e | dvar %.2e" % (
            i, a.outer[i].level, a.outer[i].k, a.outer[i].n_valid, x.outer[i].n_valid, a.outer[i].irls_iters, db.max(), l, x.outer[i].b_segm[l], x.outer[i].lambda_t_w[l], dbo.max(), dp.max(),
            abs(a.outer[i].aver_res / x.outer[i].aver_res - 1), abs(b.outer[i].aver_res / x.outer[i].aver_res - 1),
            np.abs(np.array(a.outer[i].var[:]) - np.array(x.outer[i].var[:])).max()))
