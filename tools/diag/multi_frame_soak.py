"""Soak of the per-stream frame hand-over inside multi-frame launches (sf_process_frames): many streams of unequal cost, more
than the resident workgroups, so that consecutive frames of a stream run on different CUs / XCDs while the chip is loaded
unevenly -- the condition under which a missing release / acquire shows (MI355X_MICROARCH.md: stale hand-offs at 1e-4 rates).
Every stream's pose after every frame is compared bit for bit with the same frames launched one by one.
usage (GPU box): python tools/diag/multi_frame_soak.py [rounds] [streams] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from conftest import driver_params
from staticfusion_amd.synth import DEFAULT_XI, make_pair
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
api = sf.load().with_variant("throughput")
pairs = [make_pair(seed=61 + q, sphere=True, out_rows=60, out_cols=80, xi=tuple(s * np.array(DEFAULT_XI))) for q, s in enumerate((1.0, 3.0, 0.5, 1.7, 2.4, 1.0, 0.3))]
which = lambda b: pairs[(b * 5 + b // 13) % len(pairs)]
def fresh():
    s = sf.Solver(api, 60, 80, B, driver_params(api))
    for b in range(B):
        s.set_current(b, *which(b)["new"]); s.set_prediction(b, *which(b)["old"])
    return s
ref = fresh()
T_ref = []
for k in range(K):
    ref.process_frame(k); T_ref.append(ref.batch_results()[0].copy())
T_ref = np.stack(T_ref)
bad = 0
t0 = time.time()
for r in range(rounds):
    s = fresh()
    T = s.process_frames(0, K, trajectory=True)
    diff = np.argwhere((T != T_ref).any(axis=(2, 3)))
    bad += len(diff)
    if len(diff):
        print("round %d: %d (frame, stream) pairs differ, first %s" % (r, len(diff), diff[:5].tolist()))
    s.close()
print("multi-frame soak: %d rounds x %d streams x %d frames = %d hand-overs, %d differing poses, %.0f s" % (rounds, B, K, rounds * B * (K - 1), bad, time.time() - t0))
sys.exit(1 if bad else 0)
