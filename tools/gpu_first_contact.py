"""First contact with the GPU: compare HIP vs oracle stage by stage on small and QVGA inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd import capi
from oracle import binding
from staticfusion_amd.synth import make_pair, pose_delta

api, ora = sf.load(), binding.load()

def run(a, pair, rows, cols, p, seg_image=True):
    s = sf.Solver(a, rows, cols, 1, p)
    s.set_current(0, *pair["new"]); s.set_prediction(0, *pair["old"])
    t = time.time(); s.build_pyramid(True); s.run_solver(True)
    if seg_image: s.build_segm_image()
    s.synchronize(); dt = time.time() - t
    return s, dt

for (rows, cols, sphere) in [(60, 80, False), (120, 160, True), (240, 320, True)]:
    pair = make_pair(seed=11, sphere=sphere, out_rows=rows, out_cols=cols)
    for cfg in ("driver", "config2"):
        if cfg == "driver":
            pa, po = api.default_params_struct(), ora.default_params_struct()
            pa.kb = po.kb = 1.05
        else:
            pa, po = api.ctor_params_struct(), ora.ctor_params_struct()
            pa.segmentation_enabled = po.segmentation_enabled = 0
            if cols >= 160:
                pa.ctf_levels = po.ctf_levels = 3
        pa.debug_planes = po.debug_planes = 1
        sg, tg = run(api, pair, rows, cols, pa)
        so, to = run(ora, pair, rows, cols, po)
        print("==== %dx%d sphere=%s cfg=%s  gpu %.1f ms  cpu %.1f ms" % (rows, cols, sphere, cfg, tg * 1e3, to * 1e3))
        for L in range(sg.levels):
            for setn, st in (("new", capi.SET_NEW), ("pred", capi.SET_PRED)):
                for chn, ch in (("d", 0), ("i", 1), ("xx", 2), ("yy", 3)):
                    g, o = sg.plane(st, ch, L), so.plane(st, ch, L)
                    if not np.array_equal(g, o):
                        print("  PYR MISMATCH L%d %s %s max|d|=%g n=%d" % (L, setn, chn, np.abs(g - o).max(), (g != o).sum()))
        if cfg == "driver":
            print("  kmeans centres equal:", np.array_equal(sg.kmeans_centres(), so.kmeans_centres()),
                  " iters", sg.stats().kmeans_iters, so.stats().kmeans_iters)
            for L in range(sg.levels):
                g, o = sg.labels(L), so.labels(L)
                print("  labels L%d equal: %s (diff %d)" % (L, np.array_equal(g, o), (g != o).sum()))
            print("  connectivity equal:", np.array_equal(sg.connectivity(), so.connectivity()))
        stg, sto = sg.stats(), so.stats()
        print("  n_outer %d/%d n_irls %d/%d status %d/%d" % (stg.n_outer, sto.n_outer, stg.n_irls, sto.n_irls, stg.status, sto.status))
        for i in range(min(stg.n_outer, sto.n_outer)):
            a_, b_ = stg.outer[i], sto.outer[i]
            print("   it%d L%d k%d nvalid %d/%d irls %d/%d aver %.6g/%.6g dVar %.2e dtw %.2e db %.2e dT %.2e" % (
                i, a_.level, a_.k, a_.n_valid, b_.n_valid, a_.irls_iters, b_.irls_iters, a_.aver_res, b_.aver_res,
                np.abs(np.array(a_.var[:]) - np.array(b_.var[:])).max(),
                np.abs(np.array(a_.twist_level[:]) - np.array(b_.twist_level[:])).max(),
                np.abs(np.array(a_.b_segm[:]) - np.array(b_.b_segm[:])).max(),
                np.abs(np.array(a_.T[:]) - np.array(b_.T[:])).max()))
        for nm, w in (("dcu", 0), ("dcv", 1), ("dct", 2), ("ddu", 3), ("ddv", 4), ("ddt", 5), ("wc", 6), ("wd", 7), ("null", 8)):
            g, o = sg.lin_plane(w), so.lin_plane(w)
            print("   lin %-4s max|d| %.3e  (max|o| %.3e)" % (nm, np.abs(g - o).max(), np.abs(o).max()))
        for L in range(sg.levels):
            for setn, st in (("warped", capi.SET_WARPED), ("inter", capi.SET_INTER)):
                d = [np.abs(sg.plane(st, ch, L) - so.plane(st, ch, L)).max() for ch in range(4)]
                print("   L%d %s max|d| d=%.2e i=%.2e xx=%.2e yy=%.2e" % (L, setn, *d))
        print("  pose delta gpu vs oracle:", pose_delta(so.T(), sg.T()), " vs gt:", pose_delta(pair["T_gt"], sg.T()))
        print("  b max diff:", np.abs(sg.b() - so.b()).max(), " b image diff:", np.abs(sg.b_image() - so.b_image()).max())
        sg.close(); so.close()
