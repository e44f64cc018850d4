"""Tiny images (level 0 of at most 2048 pixels: the product's ordered float splat runs at EVERY level, the residual stage included):
product builds and the reference-order build against the oracle over a short sequence; then a batch larger than the number of
resident workgroups (the ordered splat's scratch is per workgroup there) against the single stream.   usage: tiny_case.py ROWSxCOLS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import staticfusion_amd as sf
from oracle import binding
from staticfusion_amd.synth import Scene, pose_delta, quantise_and_decimate, se3_exp
from conftest import driver_params, make_solver

rows, cols = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30x40").split("x"))
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
binding.build(); ora = binding.load()
scene = Scene(seed=41, sphere=True)
xi = np.array((0.006, -0.004, 0.005, 0.01, -0.004, 0.003))
frames, T = [], np.eye(4)
for k in range(7):
    frames.append(quantise_and_decimate(*scene.render(T, 2 * cols, 2 * rows, sphere_offset=(0.02 * k, 0, 0))))
    T = T @ se3_exp(xi)
def run(api, batch=1, variant=None):
    a = api.with_variant(variant) if variant else api
    s = make_solver(a, rows, cols, driver_params(a, kb=1.5, ctf_levels=levels), batch=batch)
    out = []
    for b in range(batch): s.set_current(b, *frames[0])
    s.current_to_prediction()
    s.push_history(0)
    for k in range(1, 7):
        for b in range(batch):
            s.set_prediction(b, *frames[k - 1]); s.set_current(b, *frames[k])
        s.process_frame(k)
        st = s.stats()
        out.append(dict(T=s.T().copy(), Tl=s.T(batch - 1).copy() if batch > 1 else None, lab=s.labels(0).copy(), cnt=(st.n_outer, st.n_irls), status=st.status,
                        b=s.b().copy(), bimg=s.b_image().copy(), cr=s.cluster_residuals().copy()))
    return out
ref = run(ora)
libs = [("product", sf.Api(sf.LIB, "sf_"), v) for v in ("throughput", "latency", "cluster")]
libs.append(("reforder", sf.Api(os.path.join(os.path.dirname(sf.LIB), "libsf_hip_reforder.so"), "sf_"), "throughput"))
for name, api, v in libs:
    got = run(api, 1, v)
    for k, (r, g) in enumerate(zip(ref, got)):
        rot, tr = pose_delta(r["T"], g["T"])
        ok = ~np.isnan(r["cr"])
        print(name, v, "frame", k + 1, "pose %.1e %.1e" % (rot, tr), "counts", g["cnt"], r["cnt"], "status", g["status"], r["status"], "labels", bool(np.array_equal(r["lab"], g["lab"])),
              "b %.1e bimg %.1e" % (np.abs(r["b"] - g["b"]).max(), np.abs(r["bimg"] - g["bimg"]).max()),
              "cr nan-eq", bool(np.array_equal(np.isnan(r["cr"]), np.isnan(g["cr"]))), "cr rel %.1e" % (np.abs(g["cr"][ok] / r["cr"][ok] - 1).max() if ok.any() else 0),
              "IDENT" if np.array_equal(r["T"], g["T"]) and np.array_equal(r["b"], g["b"]) and np.array_equal(r["bimg"], g["bimg"]) else "", flush=True)
api = sf.Api(sf.LIB, "sf_")
one = run(api, 1, "throughput")
many = run(api, 3000, "throughput")
print("batch 3000 vs 1: stream 0 identical", all(np.array_equal(a["T"], b["T"]) for a, b in zip(one, many)), "last stream identical", all(np.array_equal(a["T"], b["Tl"]) for a, b in zip(one, many)))
