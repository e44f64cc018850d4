"""Where does the reference-order build first leave the oracle on one hunt case?  usage: ro_diff.py <seed> [WxH rendered]
Runs the case on both, frame by frame, and prints every trace field of every outer iteration that is not bit-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd import _capi as capi
from oracle import binding
from sequence_cases import make_case, _params, N_FRAMES
from conftest import trace_array

seed = int(sys.argv[1]); W, H = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "320x240").split("x"))
binding.build()
ora = binding.load()
ro = sf.Api(os.path.join(os.path.dirname(sf.LIB), "libsf_hip_reforder.so"), "sf_").with_variant("throughput")
case = make_case(seed, W, H, True)
frames = case["frames"]; rows, cols = frames[0][0].shape
F = ("level", "k", "n_valid", "irls_iters", "aver_res", "delta_sol_max", "var", "twist_level", "b_segm", "T", "b_prior", "lambda_t_w", "AtA", "AtB")
ss = []
for api in (ro, ora):
    p = _params(api, case["kb"], case["over"]); p.debug_planes = 1
    s = sf.Solver(api, rows, cols, 1, p)
    s.set_current(0, *frames[0]); s.current_to_prediction(); s.push_history(0)
    ss.append(s)
for k in range(1, N_FRAMES):
    for s in ss:
        s.set_prediction(0, *frames[k - 1]); s.set_current(0, *frames[k]); s.process_frame(k)
    a, b = ss[0].stats(), ss[1].stats()
    same = np.array_equal(ss[0].T(), ss[1].T()) and np.array_equal(ss[0].b(), ss[1].b())
    print("frame", k, "identical" if same else "DIFFERENT", (a.n_outer, a.n_irls), (b.n_outer, b.n_irls), flush=True)
    for f in F:
        x, y = trace_array(a, f), trace_array(b, f)
        if not np.array_equal(x, y):
            x = np.asarray(x, np.float64).reshape(len(x), -1); y = np.asarray(y, np.float64).reshape(len(y), -1)
            for o in range(min(len(x), len(y))):
                if not np.array_equal(x[o], y[o]):
                    j = np.flatnonzero(x[o] != y[o])
                    print("   outer", o, "level", a.outer[o].level, f, "elements", j[:8], "gpu", x[o][j[:4]], "oracle", y[o][j[:4]], flush=True)
    for L in range(ss[0].levels):
        for ch, nm in ((capi.CH_DEPTH, "depth_w"), (capi.CH_INTENSITY, "inten_w"), (capi.CH_XX, "xx_w"), (capi.CH_YY, "yy_w")):
            x, y = ss[0].plane(capi.SET_WARPED, ch, L), ss[1].plane(capi.SET_WARPED, ch, L)
            if not np.array_equal(x, y):
                j = np.argwhere(x != y)
                print("   warped", nm, "level", L, len(j), "cells", j[:4].tolist(), x[tuple(j[0])], y[tuple(j[0])], flush=True)
    cg, co = ss[0].cluster_residuals(), ss[1].cluster_residuals()
    if not np.array_equal(cg, co, equal_nan=True):
        print("   cluster residuals differ", np.argwhere(~((cg == co) | (np.isnan(cg) & np.isnan(co))))[:6].tolist())
