"""Soak run on the GPU: 300 frames x 64 streams through sf_process_frame (carried state, ring wrap-around), twice;
poses stay finite and the two runs are bit-identical."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import staticfusion_amd as sf
from staticfusion_amd.synth import make_batch
api = sf.load()
p = api.default_params_struct(); p.kb = 1.05
pairs = make_batch(4, sphere=True, distinct=4)
res = []
for run in range(2):
    s = sf.Solver(api, 240, 320, 64, p)
    for b in range(64):
        s.set_current(b, *pairs[b % 4]["old"])
    s.current_to_prediction(); s.push_history(0)
    acc = []
    for k in range(1, 301):
        # alternate the two frames of each pair: forward, backward, forward ...
        which = "new" if k % 2 else "old"
        for b in range(64):
            s.set_current(b, *pairs[b % 4][which])
        s.process_frame(k)
        s.current_to_prediction()
        if k % 50 == 0:
            T, n_irls, n_outer, pix = s.batch_results()
            assert np.isfinite(T).all()
            acc.append((T[:4].copy(), n_irls[:4].copy()))
    res.append(acc)
for a, b in zip(res[0], res[1]):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
print("soak ok: 300 frames x 64 streams twice, finite and run-to-run identical; last n_irls", res[0][-1][1])
