"""Where a hunt frame leaves the oracle: the outer-iteration traces of one frame of one sequence (tests/sequence_cases.py), HIP build
against oracle, side by side.   python tools/diag/frame_trace_diff.py <seed> <frame 1..8> [WxH rendered, default 640x480] [builds]"""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import staticfusion_amd as sf
from oracle import binding
from sequence_cases import N_FRAMES, make_case, _params
from staticfusion_amd.synth import pose_delta

seed, frame = int(sys.argv[1]), int(sys.argv[2])
W, H = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "640x480").split("x"))
builds = (sys.argv[4] if len(sys.argv) > 4 else "throughput,cluster").split(",")
case = make_case(seed, W, H)
hip, ora = sf.load(), binding.load()


def run(api, variant=None):
    frames = case["frames"]
    rows, cols = frames[0][0].shape
    s = sf.Solver(api, rows, cols, 1, _params(api, case["kb"], case["over"]), variant=variant)
    s.set_current(0, *frames[0]); s.current_to_prediction(); s.push_history(0)
    out = None
    for k in range(1, frame + 1):
        s.set_prediction(0, *frames[k - 1]); s.set_current(0, *frames[k]); s.process_frame(k)
        if k == frame - 1:
            prev = (s.T().copy(), s.b().copy(), s.twist_old().copy())
    st = s.stats()
    tr = [dict(level=o.level, k=o.k, irls=o.irls_iters, delta=o.delta_sol_max, n_valid=o.n_valid, aver_res=o.aver_res,
               twist=np.array(o.twist_level[:]), T=np.array(o.T[:]), b=np.array(o.b_segm[:])) for o in (st.outer[i] for i in range(st.n_outer))]
    return dict(T=s.T().copy(), b=s.b().copy(), trace=tr, prev=prev if frame > 1 else None, cres=s.cluster_residuals().copy())


ref = run(ora)
print("seed %d frame %d motion scale %.2f kb %.3f" % (seed, frame, case["scale"], case["kb"]))
for bname in builds:
    got = run(hip, bname)
    if ref["prev"]:
        print(bname, "state before the frame: T %.2e b %.2e twist_old %.2e" % tuple(np.abs(a - b).max() for a, b in zip(ref["prev"], got["prev"])))
    for i, (a, b) in enumerate(zip(ref["trace"], got["trace"])):
        print("  outer %d level %d k %d irls %d/%d delta %.6e/%.6e n_valid %d/%d aver_res %.6e/%.6e |dtwist| %.2e |dT| %.2e |db| %.2e (|twist| %.3e)" % (
            i, a["level"], a["k"], a["irls"], b["irls"], a["delta"], b["delta"], a["n_valid"], b["n_valid"], a["aver_res"], b["aver_res"],
            np.abs(a["twist"] - b["twist"]).max(), np.abs(a["T"] - b["T"]).max(), np.abs(a["b"] - b["b"]).max(), np.linalg.norm(a["twist"])))
    print("  final pose distance rot %.2e trans %.2e; b %.2e; cluster residuals %.2e" % (*pose_delta(ref["T"], got["T"]), np.abs(ref["b"] - got["b"]).max(),
          np.nanmax(np.abs(ref["cres"] - got["cres"])) if np.isfinite(ref["cres"]).any() else 0.0))
