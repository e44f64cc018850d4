"""Which in-plane rotations send a coarse level of the product's ordered tile splat to the per-cell lists (slot 25 of the stage
profile counts them), per build.   usage: roll_fallbacks.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd.synth import make_pair, pose_delta
from conftest import driver_params, make_solver

for variant in ("throughput", "latency"):
    api = sf.load().with_variant(variant)
    # (xi = (v, w) with the optical axis as z: an in-plane rotation is xi[5]; xi[3] -- what round 4's "strong roll" tests set -- is a PITCH)
    for roll, fwd in ((0.15, 0.0), (0.25, 0.0), (0.3, 0.0), (0.4, 0.0), (0.5, 0.0), (0.3, 0.1)):
        pr = make_pair(seed=17, sphere=True, out_rows=240, out_cols=320, xi=(0.0, 0.0, fwd, 0.0, 0.0, roll))
        s = make_solver(api, 240, 320, driver_params(api), pr)
        s.build_pyramid(True); s.run_solver(True)
        st = s.stats()
        rot, tr = pose_delta(pr["T_gt"], s.T())
        print(variant, "roll %.2f fwd %.2f: fallbacks %d replays %d outer %d irls %d err vs truth %.3f rad %.3f m" % (roll, fwd, s.ordered_fallbacks(), s.splat_replays(), st.n_outer, st.n_irls, rot, tr), flush=True)
        s.close()
