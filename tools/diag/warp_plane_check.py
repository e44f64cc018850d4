"""Warped depth / intensity planes of every level, one library against the oracle: touched cells and the largest difference.
usage: python tools/diag/warp_plane_check.py [lib.so ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd import _capi as capi
from staticfusion_amd.synth import make_pair
from oracle import binding

def run(api):
    p = api.default_params_struct(); p.kb = 1.05; p.debug_planes = 1
    pr = make_pair(seed=3, sphere=True, out_rows=240, out_cols=320)
    s = sf.Solver(api, 240, 320, 1, p, variant="throughput" if api.prefix == "sf_" else None)
    s.set_current(0, *pr["new"]); s.set_prediction(0, *pr["old"])
    s.build_pyramid(True); s.run_solver(True)
    st = s.stats()
    planes = {(L, ch): s.plane(capi.SET_WARPED, ch, L).copy() for L in range(s.levels) for ch in (0, 1)}
    return (st.n_outer, st.n_irls, st.status, [int(st.outer[i].n_valid) for i in range(st.n_outer)]), planes

ref_stats, ref = run(binding.load())
print("oracle", ref_stats)
for lib in sys.argv[1:]:
    st, pl = run(sf.Api(os.path.join(ROOT, lib), "sf_"))
    print(lib, st)
    for L in range(5):
        a, b = pl[(L, 0)], ref[(L, 0)]
        print("  level %d: touched %d (oracle %d), max |d depth| %.3g, identical %s; intensity identical %s" % (
            L, int((a != 0).sum()), int((b != 0).sum()), float(np.abs(a - b).max()), np.array_equal(a, b), np.array_equal(pl[(L, 1)], ref[(L, 1)])))
