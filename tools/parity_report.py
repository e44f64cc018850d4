#!/usr/bin/env python3
"""Measured deviations of the HIP path from the CPU oracle, per library build and frame-kernel variant.

    python tools/parity_report.py [--out gpurun_out/parity_report.json]

Builds compared: libsf_hip.so (hardware v_rsq_f32 / v_rcp_f32 in the per-pixel IRLS weights, the product) and
libsf_hip_precise.so (-DSF_FAST_WEIGHTS=0: IEEE division / square root), each on the throughput and the latency
variant. Cases: BASELINE configs[1] (static pair, 3 levels, no segmentation) and configs[2] (moving sphere, full
solver) at QVGA over several seeds, and an 8-frame sequence with carried state. Reported: the maximum over the cases of
pose delta (rad, m), |b - b_oracle|, |b image - oracle's|, pixels whose (b > 0.5) decision differs, per-iteration twist
increments and solutions. DESIGN.md section 6 quotes this table. Needs a GPU (test infrastructure: uses oracle/)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    ap.add_argument("--seeds", type=int, default=6)
    args = ap.parse_args()

    import staticfusion_amd as sf
    from conftest import config2_params, driver_params, make_solver, trace_array
    from oracle import binding
    from staticfusion_amd import Api
    from staticfusion_amd.synth import DEFAULT_XI, Scene, make_pair, pose_delta, quantise_and_decimate, se3_exp

    ora = binding.load()
    libs = {"fast (product)": sf.LIB, "precise (-DSF_FAST_WEIGHTS=0)": os.path.join(os.path.dirname(sf.LIB), "libsf_hip_precise.so")}

    import ctypes

    def exact(s):
        """test hook of the oracle: its per-cluster sums in fp64 instead of the reference's sequential fp32"""
        ora.lib.sfo_test_set_exact_sums.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert ora.lib.sfo_test_set_exact_sums(s.h, 1) == 0
        return s

    def one_pair(api, mk, pr, exact_sums=False):
        s = make_solver(api, 240, 320, mk(api), pr)
        if exact_sums:
            exact(s)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        return s

    def upd(acc, key, v):
        acc[key] = max(acc.get(key, 0.0), float(v))

    def compare(acc, sg, so, sx=None):
        if sx is not None:  # against the oracle with exact (fp64) per-cluster sums, and that oracle against the plain one
            a, x, b = sg.stats(), sx.stats(), so.stats()
            if (a.n_outer, a.n_irls) == (x.n_outer, x.n_irls):
                upd(acc, "b_trace_vs_exact_sums", np.abs(trace_array(a, "b_segm") - trace_array(x, "b_segm")).max())
                upd(acc, "b_prior_vs_exact_sums", np.abs(trace_array(a, "b_prior") - trace_array(x, "b_prior")).max())
            if (b.n_outer, b.n_irls) == (x.n_outer, x.n_irls):
                upd(acc, "oracle_fp32_sums_vs_exact", np.abs(trace_array(b, "b_segm") - trace_array(x, "b_segm")).max())
        rot, trans = pose_delta(so.T(), sg.T())
        upd(acc, "pose_rot_rad", rot)
        upd(acc, "pose_trans_m", trans)
        a, b = sg.stats(), so.stats()
        assert (a.n_outer, a.n_irls, a.kmeans_iters) == (b.n_outer, b.n_irls, b.kmeans_iters)
        upd(acc, "b", np.abs(sg.b() - so.b()).max())
        upd(acc, "b_trace", np.abs(trace_array(a, "b_segm") - trace_array(b, "b_segm")).max())
        upd(acc, "b_prior_trace", np.abs(trace_array(a, "b_prior") - trace_array(b, "b_prior")).max())
        upd(acc, "twist_level_trace", np.abs(trace_array(a, "twist_level") - trace_array(b, "twist_level")).max())
        upd(acc, "var_trace", np.abs(trace_array(a, "var") - trace_array(b, "var")).max())
        upd(acc, "aver_res_rel", np.abs(trace_array(a, "aver_res") / trace_array(b, "aver_res") - 1).max())
        bg, bo = sg.b_image(), so.b_image()
        upd(acc, "b_image", np.abs(bg - bo).max())
        acc["decision_mismatch_px"] = acc.get("decision_mismatch_px", 0) + int(((bg > 0.5) != (bo > 0.5)).sum())
        acc["label_mismatch_px"] = acc.get("label_mismatch_px", 0) + sum(int((sg.labels(L) != so.labels(L)).sum()) for L in range(sg.levels))

    report = {}
    oracle_cache = {}
    for lib_name, path in libs.items():
        base = Api(path, "sf_")
        for variant in ("throughput", "latency", "cluster"):
            api = base.with_variant(variant)
            for cfg, mk, sphere in (("configs[1] static, seg off", lambda a: config2_params(a, levels=3), False),
                                    ("configs[2] sphere, full solver", lambda a: driver_params(a), True)):
                acc = {}
                for seed in range(1234, 1234 + args.seeds):
                    pr = make_pair(seed=seed, sphere=sphere, out_rows=240, out_cols=320)
                    key = (cfg, seed)
                    if key not in oracle_cache:
                        oracle_cache[key] = (one_pair(ora, mk, pr), one_pair(ora, mk, pr, exact_sums=True))
                    compare(acc, one_pair(api, mk, pr), *oracle_cache[key])
                report["%s | %s | %s" % (lib_name, variant, cfg)] = acc
            # 8-frame sequence with carried state (twist_old, b_segm, the 5-frame ring)
            acc = {}
            scene = Scene(seed=77, sphere=True)
            xi = np.array(DEFAULT_XI) * 0.6
            frames, T = [], np.eye(4)
            for k in range(9):
                frames.append(quantise_and_decimate(*scene.render(T, 640, 480, sphere_offset=(0.02 * k, 0, 0))))
                T = T @ se3_exp(xi)
            solvers = [make_solver(a, 240, 320, driver_params(a, kb=1.5)) for a in (api, ora, ora)]
            exact(solvers[2])
            for s in solvers:
                s.set_current(0, *frames[0])
                s.current_to_prediction()
                s.push_history(0)
            for k in range(1, 9):
                for s in solvers:
                    s.set_prediction(0, *frames[k - 1])
                    s.set_current(0, *frames[k])
                    s.process_frame(k)
                compare(acc, *solvers)
            report["%s | %s | 8-frame sequence" % (lib_name, variant)] = acc

    keys = ["pose_rot_rad", "pose_trans_m", "b", "b_trace", "b_trace_vs_exact_sums", "oracle_fp32_sums_vs_exact", "b_prior_trace",
            "b_prior_vs_exact_sums", "b_image", "decision_mismatch_px", "label_mismatch_px",
            "twist_level_trace", "var_trace", "aver_res_rel"]
    print("| build | variant | case | " + " | ".join(keys) + " |")
    print("|" + "---|" * (3 + len(keys)))
    for name, acc in report.items():
        print("| " + name + " | " + " | ".join(("%d" % acc.get(k, 0)) if k.endswith("_px") else ("%.2e" % acc.get(k, float("nan"))) for k in keys) + " |")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
