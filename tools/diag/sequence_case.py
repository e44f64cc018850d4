"""One sequence of tools/diag/sequence_hunt.py in detail: HIP (one build) vs the oracle vs the oracle with exact (fp64)
cross-pixel sums, frame by frame -- whose rounding is it when a frame disagrees?
usage (GPU box): python tools/diag/sequence_case.py <seed> [variant]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from oracle import binding
from staticfusion_amd.synth import DEFAULT_XI, LCG64, Scene, pose_delta, quantise_and_decimate, se3_exp
from conftest import driver_params, make_solver

seed = int(sys.argv[1]); variant = sys.argv[2] if len(sys.argv) > 2 else "throughput"
binding.build()
ora = binding.load()
ora.lib.sfo_test_set_exact_sums.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = LCG64(seed)
scene = Scene(seed=seed, sphere=True, sphere_seed=seed + 17)
scale = g.uniform(0.2, 2.5)
xi = np.array(DEFAULT_XI) * scale * np.array([g.uniform(0.5, 1.5) * (1 if g.uniform() < 0.5 else -1) for _ in range(6)])
step = (g.uniform(-0.03, 0.03), g.uniform(-0.01, 0.01), g.uniform(-0.01, 0.01))
frames, T = [], np.eye(4)
for k in range(9):
    d, i = scene.render(T, 320, 240, sphere_offset=tuple(k * s for s in step))
    frames.append(quantise_and_decimate(d, i))
    T = T @ se3_exp(xi)
rows, cols = frames[0][0].shape
kb = g.uniform(1.0, 1.6)
print("seed %d: motion scale %.2f, |xi| %.4f m / %.4f rad per frame" % (seed, scale, np.linalg.norm(xi[:3]), np.linalg.norm(xi[3:])))
so = make_solver(ora, rows, cols, driver_params(ora, kb=kb))
sx = make_solver(ora, rows, cols, driver_params(ora, kb=kb)); ora.lib.sfo_test_set_exact_sums(sx.h, 1)
sg = make_solver(sf.load().with_variant(variant), rows, cols, driver_params(sf.load(), kb=kb))
for s in (so, sx, sg):
    s.set_current(0, *frames[0]); s.current_to_prediction(); s.push_history(0)
for k in range(1, 9):
    for s in (so, sx, sg):
        s.set_prediction(0, *frames[k - 1]); s.set_current(0, *frames[k]); s.process_frame(k)
    r1, t1 = pose_delta(so.T(), sg.T()); r2, t2 = pose_delta(sx.T(), sg.T()); r3, t3 = pose_delta(sx.T(), so.T())
    if len(sys.argv) > 3 and int(sys.argv[3]) == k:
        a, b = sg.stats(), so.stats()
        for i in range(a.n_outer):
            oa, ob = a.outer[i], b.outer[i]
            print("   outer %d level %d k %d: n_valid %d/%d irls %d/%d aver_res %.6e/%.6e |dT| %.2e |dvar| %.2e |db| %.2e |db_prior| %.2e" % (
                i, oa.level, oa.k, oa.n_valid, ob.n_valid, oa.irls_iters, ob.irls_iters, oa.aver_res, ob.aver_res,
                np.abs(np.array(oa.T) - np.array(ob.T)).max(), np.abs(np.array(oa.var) - np.array(ob.var)).max(),
                np.abs(np.array(oa.b_segm) - np.array(ob.b_segm)).max(), np.abs(np.array(oa.b_prior) - np.array(ob.b_prior)).max()))
            la, lb = np.array(oa.lambda_t_w), np.array(ob.lambda_t_w)
            j = int(np.argmax(np.abs(np.array(oa.b_segm) - np.array(ob.b_segm))))
            print("      cluster %d: b %.6f / %.6f, b_prior %.6f / %.6f, lambda_t_w %.8f / %.8f; lambdas nearest to 0.1: %s" % (
                j, oa.b_segm[j], ob.b_segm[j], oa.b_prior[j], ob.b_prior[j], la[j], lb[j], np.sort(np.abs(la - 0.1))[:2]))
    print("frame %d: HIP-oracle %.2e rad %.2e m | HIP-exact %.2e %.2e | oracle-exact %.2e %.2e | n_irls %d %d %d" % (
        k, r1, t1, r2, t2, r3, t3, sg.stats().n_irls, so.stats().n_irls, sx.stats().n_irls))
