"""Debug helper: which debug planes differ between the HIP path and the oracle for one pair."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from conftest import driver_params, make_solver, trace_array
import staticfusion_amd as sf
from staticfusion_amd import capi
from staticfusion_amd.synth import make_pair, pose_delta
import oracle.binding as ob

hip = sf.load(); ob.build(); ora = ob.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
pr = make_pair(seed=seed, sphere=True, out_rows=240, out_cols=320)
out = []
for api in (hip, ora):
    s = make_solver(api, 240, 320, driver_params(api, debug_planes=1), pr)
    s.build_pyramid(True); s.run_solver(True); s.build_segm_image(); out.append(s)
sg, so = out
a, b = sg.stats(), so.stats()
print("outer", a.n_outer, b.n_outer, "irls", a.n_irls, b.n_irls)
print("irls iters", trace_array(a, "irls_iters"), trace_array(b, "irls_iters"))
print("pose delta", pose_delta(sg.T(), so.T()))
print("T diff trace", np.abs(trace_array(a, "T") - trace_array(b, "T")).max(axis=tuple(range(1, trace_array(a, "T").ndim))))
for which in range(capi.LIN_NULL + 1):
    d = np.abs(sg.lin_plane(which).astype(np.float64) - so.lin_plane(which))
    print("lin", which, "frac ok", (d <= 5e-5).mean(), "max", d.max())
for pset in (capi.SET_WARPED, capi.SET_INTER):
    for ch in range(4):
        d = np.abs(sg.plane(pset, ch, 0).astype(np.float64) - so.plane(pset, ch, 0))
        print("plane", pset, ch, "frac ok", (d <= 5e-5).mean(), "max", d.max())
g, o = sg.lin_plane(capi.LIN_WD), so.lin_plane(capi.LIN_WD)
r = g[o > 0] / o[o > 0]
print("WD ratio quantiles", np.quantile(r, [0, 0.01, 0.25, 0.5, 0.75, 0.99, 1]))
print("WD max", g.max(), o.max(), "argmax", np.unravel_index(g.argmax(), g.shape), np.unravel_index(o.argmax(), o.shape))
