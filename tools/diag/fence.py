"""Picket-fence warp case (tests/test_gpu_parity.py::test_warp_with_targets_outside_the_tile_windows) on a given build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd import capi
from staticfusion_amd.synth import make_pair
from conftest import driver_params, make_solver

variant = sys.argv[1] if len(sys.argv) > 1 else "throughput"
hip = sf.load().with_variant(variant)
from oracle import binding
binding.build()
ora = binding.load()
pr = make_pair(seed=21, out_rows=240, out_cols=320, xi=(0.05, 0.0, 0.0, 0.0, 0.0, 0.0))
d_old = pr["old"][0].copy()
patch = np.zeros((240, 320), bool)
patch[90:150, 130:190] = True
patch &= ((np.arange(320) // 3) % 2 == 0)[None, :]
d_old[patch] *= 0.3
fence = {"new": pr["new"], "old": (d_old, pr["old"][1])}
out = []
for api in (hip, ora):
    s = make_solver(api, 240, 320, driver_params(api, debug_planes=1), fence)
    s.build_pyramid(True)
    s.run_solver(True)
    out.append(s)
sg, so = out
a, b = sg.stats(), so.stats()
print("outer", a.n_outer, b.n_outer, "irls", a.n_irls, b.n_irls, "status", a.status, b.status)
for i in range(a.n_outer):
    print(" outer", i, "level", a.outer[i].level, "k", a.outer[i].k, "n_valid", a.outer[i].n_valid, b.outer[i].n_valid,
          "dT", np.abs(np.array(a.outer[i].T) - np.array(b.outer[i].T)).max())
for L in range(5):
    for ch in range(2):
        g, o = sg.plane(capi.SET_WARPED, ch, L), so.plane(capi.SET_WARPED, ch, L)
        d = np.abs(g.astype(np.float64) - o.astype(np.float64))
        print("level", L, "ch", ch, "within 5e-5: %.4f" % (d <= 5e-5).mean(), "max", d.max(), "nonzero g/o", (g != 0).mean(), (o != 0).mean())
print("replayed tiles", sg.splat_replays())
