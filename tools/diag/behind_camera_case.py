"""The near-patch pair of tests/test_gpu_edge_rules.py in detail: HIP vs the oracle with the HIP build's behind-the-camera
rule, per outer iteration -- which cluster's prior differs, and where the warped / linearisation planes differ."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from conftest import driver_params, make_solver, trace_array
from oracle import binding
from staticfusion_amd import capi
from test_gpu_edge_rules import _near_patch_pair
ora = binding.load(); hip = sf.load().with_variant(sys.argv[1] if len(sys.argv) > 1 else "throughput")
ora.lib.sfo_test_set_hip_behind_camera_rule.argtypes = [ctypes.c_void_p, ctypes.c_int]
pr = _near_patch_pair()
for max_outer in (None,):
    S = []
    for api in (hip, ora):
        p = driver_params(api); p.debug_planes = 1
        s = make_solver(api, 120, 160, p, pr)
        if api is ora: ora.lib.sfo_test_set_hip_behind_camera_rule(s.h, 1)
        s.build_pyramid(True); s.run_solver(True); S.append(s)
    g, o = S[0].stats(), S[1].stats()
    bp_g, bp_o = trace_array(g, "b_prior"), trace_array(o, "b_prior")
    lt_g, lt_o = trace_array(g, "lambda_t_w"), trace_array(o, "lambda_t_w")
    for i in range(g.n_outer):
        d = np.abs(bp_g[i] - bp_o[i]); l = int(d.argmax())
        print("outer %d level %d k %d n_valid %d/%d: max |db_prior| %.2e at cluster %d (%.6f / %.6f), lambda %.6f / %.6f, |dT| %.2e" % (
            i, g.outer[i].level, g.outer[i].k, g.outer[i].n_valid, o.outer[i].n_valid, d.max(), l, bp_g[i][l], bp_o[i][l], lt_g[i][l], lt_o[i][l],
            np.abs(np.array(g.outer[i].T[:]) - np.array(o.outer[i].T[:])).max()))
    # planes of the LAST outer iteration (level 0)
    L = 0
    dw_g, dw_o = S[0].plane(capi.SET_WARPED, capi.CH_DEPTH, L), S[1].plane(capi.SET_WARPED, capi.CH_DEPTH, L)
    dd = np.abs(dw_g.astype(np.float64) - dw_o)
    print("warped depth level 0: max |d| %.3e, cells > 1e-5: %d, sign differs: %d, zero pattern differs: %d, negative cells %d / %d" % (
        dd.max(), int((dd > 1e-5).sum()), int((np.sign(dw_g) != np.sign(dw_o)).sum()), int(((dw_g == 0) != (dw_o == 0)).sum()), int((dw_g < 0).sum()), int((dw_o < 0).sum())))
    idx = np.argwhere(dd > 1e-5)[:12]
    for v, u in idx:
        print("   (v %d, u %d): hip %.7f oracle %.7f" % (v, u, dw_g[v, u], dw_o[v, u]))
    lab = S[1].labels(0)
    if len(idx):
        print("   labels of those cells:", sorted(set(int(lab[v, u]) for v, u in np.argwhere(dd > 1e-5))))
