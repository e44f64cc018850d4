"""One-off hunt: N random 8-frame sequences (carried state, 5-frame residuals from frame 5 on) at 160 x 120 through
sf_process_frame on every build of the frame kernel against the oracle. Prints the worst deviations and every mismatch.
usage (GPU box): python tools/diag/sequence_hunt.py [first_seed] [count] [render_width render_height]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from oracle import binding
from staticfusion_amd.synth import DEFAULT_XI, LCG64, Scene, pose_delta, quantise_and_decimate, se3_exp
from conftest import driver_params, make_solver

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 5000), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (320, 240)  # rendered size; the solver sees half of it
binding.build()
ora = binding.load()
worst = {"rot": 0.0, "trans": 0.0, "b": 0.0}
bad = 0
frames_total = 0
b_over = {1e-4: 0, 1e-3: 0, 1e-2: 0}
pose_over = {1e-6: 0, 1e-5: 0}
for seed in range(first, first + count):
    g = LCG64(seed)
    scene = Scene(seed=seed, sphere=True, sphere_seed=seed + 17)
    scale = g.uniform(0.2, 2.5)
    xi = np.array(DEFAULT_XI) * scale * np.array([g.uniform(0.5, 1.5) * (1 if g.uniform() < 0.5 else -1) for _ in range(6)])
    step = (g.uniform(-0.03, 0.03), g.uniform(-0.01, 0.01), g.uniform(-0.01, 0.01))
    frames, T = [], np.eye(4)
    for k in range(9):
        d, i = scene.render(T, W, H, sphere_offset=tuple(k * s for s in step))
        frames.append(quantise_and_decimate(d, i))
        T = T @ se3_exp(xi)
    rows, cols = frames[0][0].shape
    kb = g.uniform(1.0, 1.6)
    over = dict(segmentation_enabled=0, ctf_levels=3) if os.environ.get("SF_HUNT_SEG") == "0" else {}  # pure odometry (configs[1])
    so = make_solver(ora, rows, cols, driver_params(ora, kb=kb, **over))
    ref = []
    so.set_current(0, *frames[0]); so.current_to_prediction(); so.push_history(0)
    for k in range(1, 9):
        so.set_prediction(0, *frames[k - 1]); so.set_current(0, *frames[k]); so.process_frame(k)
        ref.append((so.T().copy(), so.labels(0).copy(), so.b_image().copy(), (so.stats().n_outer, so.stats().n_irls)))
    for variant in ("throughput", "latency", "cluster"):
        sg = make_solver(sf.load().with_variant(variant), rows, cols, driver_params(sf.load(), kb=kb, **over))
        sg.set_current(0, *frames[0]); sg.current_to_prediction(); sg.push_history(0)
        for k in range(1, 9):
            sg.set_prediction(0, *frames[k - 1]); sg.set_current(0, *frames[k]); sg.process_frame(k)
            T_o, lab_o, b_o, cnt_o = ref[k - 1]
            rot, trans = pose_delta(T_o, sg.T())
            db = float(np.abs(sg.b_image() - b_o).max())
            worst["rot"], worst["trans"], worst["b"] = max(worst["rot"], rot), max(worst["trans"], trans), max(worst["b"], db)
            frames_total += 1
            for t in b_over:
                b_over[t] += db > t
            for t in pose_over:
                pose_over[t] += max(rot, trans) > t
            ok = rot <= 1e-4 and trans <= 1e-4 and np.array_equal(sg.labels(0), lab_o) and np.array_equal(sg.b_image() > 0.5, b_o > 0.5) \
                and (sg.stats().n_outer, sg.stats().n_irls) == cnt_o
            if not ok:
                bad += 1
                print("MISMATCH seed %d %s frame %d: rot %.2e trans %.2e labels %d px decisions %d px counts %s vs %s" % (
                    seed, variant, k, rot, trans, int((sg.labels(0) != lab_o).sum()), int(((sg.b_image() > 0.5) != (b_o > 0.5)).sum()),
                    (sg.stats().n_outer, sg.stats().n_irls), cnt_o))
                break
        sg.close()
    so.close()
print("%d sequences x 3 builds x 8 frames at %dx%d: %d mismatching runs; worst pose %.2e rad %.2e m, worst |b - b_oracle| %.2e" % (
    count, cols, rows, bad, worst["rot"], worst["trans"], worst["b"]))
print("frames %d; |b - b_oracle| above %s; pose difference above %s" % (frames_total, dict(b_over), dict(pose_over)))
