"""Valid pixels against streamed pixels of the IRLS passes (per outer iteration: image level, n_valid / N_L, IRLS iterations) on the bench\nworkloads: the passes stream every pixel of a level, `pixel_iters` counts the valid ones.  python tools/diag/valid_fraction.py (GPU box)"""
import sys, os
sys.path.insert(0, os.getcwd())
import staticfusion_amd as sf, bench
from staticfusion_amd.synth import make_batch
api = sf.load().with_variant("throughput")
for wl in ("sphere", "static"):
    p = bench.make_params(api, wl)
    pairs = make_batch(8, sphere=(wl == "sphere"), distinct=8, out_rows=240, out_cols=320)
    s = sf.Solver(api, 240, 320, 8, p)
    for b in range(8):
        s.set_current(b, *pairs[b]["new"]); s.set_prediction(b, *pairs[b]["old"])
    for im in range(6):
        s.process_frame(im)
    s.synchronize()
    tot_all = tot_valid = 0
    for b in range(8):
        st = s.stats(b)
        line = []
        for o in range(st.n_outer):
            t = st.outer[o]
            levels = max(o_.level for o_ in [st.outer[q] for q in range(st.n_outer)]) + 1
            L = levels - 1 - t.level  # image level
            nL = (240 >> L) * (320 >> L)
            line.append("L%d:%d/%d x%d" % (L, t.n_valid, nL, t.irls_iters))
            tot_all += nL * t.irls_iters; tot_valid += t.n_valid * t.irls_iters
        if b < 3: print(wl, b, st.n_irls, st.pixel_iters, " ".join(line))
    print(wl, "pixel-iterations streamed %d, valid %d, ratio %.3f" % (tot_all, tot_valid, tot_all / tot_valid))
