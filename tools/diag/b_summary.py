"""Largest distance of the HIP path from the oracle on BASELINE configs[2] at QVGA, per build of the library (SF_HIP_LIB):
max over seeds, outer iterations and clusters of |b - b_oracle| (the trace's b after every outer iteration), of the AtA / AtB
entries relative to the largest entry, of the 6-vector solution and of the pose.
usage (GPU box): SF_HIP_LIB=... python tools/diag/b_summary.py [seeds] [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from conftest import driver_params, make_solver
from oracle import binding
from staticfusion_amd.synth import make_pair, pose_delta
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
variant = sys.argv[2] if len(sys.argv) > 2 else "throughput"
import ctypes
ora = binding.load(); hip = sf.load().with_variant(variant)
EXACT_WARP = os.environ.get("SF_ORACLE_EXACT_WARP") == "1"  # the oracle's warp sums in fp64 (sfo_test_set_exact_warp)
ora.lib.sfo_test_set_exact_warp.argtypes = [ctypes.c_void_p, ctypes.c_int]
EXACT_SUMS = os.environ.get("SF_ORACLE_EXACT_SUMS") == "1"  # ... and its per-cluster float sums (sfo_test_set_exact_sums)
ora.lib.sfo_test_set_exact_sums.argtypes = [ctypes.c_void_p, ctypes.c_int]
w = dict(b=0.0, b_last=0.0, ata=0.0, atb=0.0, var=0.0, rot=0.0, trans=0.0, aver=0.0)
for seed in range(1234, 1234 + n_seeds):
    pr = make_pair(seed=seed, sphere=True, out_rows=240, out_cols=320)
    S = []
    for api in (hip, ora):
        s = make_solver(api, 240, 320, driver_params(api), pr)
        if api is ora and EXACT_WARP: ora.lib.sfo_test_set_exact_warp(s.h, 1)
        if api is ora and EXACT_SUMS: ora.lib.sfo_test_set_exact_sums(s.h, 1)
        s.build_pyramid(True); s.run_solver(True); S.append((s.stats(), s.T().copy())); s.close()
    (a, Ta), (o, To) = S
    assert a.n_outer == o.n_outer and a.n_irls == o.n_irls, (seed, a.n_outer, o.n_outer, a.n_irls, o.n_irls)
    for i in range(a.n_outer):
        x, y = a.outer[i], o.outer[i]
        db = float(np.abs(np.array(x.b_segm[:]) - np.array(y.b_segm[:])).max())
        w["b"] = max(w["b"], db)
        if i == a.n_outer - 1: w["b_last"] = max(w["b_last"], db)
        A, B = np.array(x.AtA[:]), np.array(y.AtA[:])
        w["ata"] = max(w["ata"], float(np.abs(A - B).max() / np.abs(B).max()))
        A, B = np.array(x.AtB[:]), np.array(y.AtB[:])
        w["atb"] = max(w["atb"], float(np.abs(A - B).max() / np.abs(B).max()))
        w["var"] = max(w["var"], float(np.abs(np.array(x.var[:]) - np.array(y.var[:])).max()))
        w["aver"] = max(w["aver"], abs(x.aver_res / y.aver_res - 1))
    r, t = pose_delta(To, Ta); w["rot"] = max(w["rot"], r); w["trans"] = max(w["trans"], t)
print("%s [%s] vs oracle%s, %d seeds: |b-b_o| max %.2e (last outer %.2e) | AtA rel %.2e AtB rel %.2e | var %.2e | aver_res rel %.2e | pose %.2e rad %.2e m" % (
    os.path.basename(sf.LIB), variant, (" (fp64 warp sums)" if EXACT_WARP else "") + (" (fp64 cluster sums)" if EXACT_SUMS else ""), n_seeds, w["b"], w["b_last"], w["ata"], w["atb"], w["var"], w["aver"], w["rot"], w["trans"]))
