"""One frame of a long sequence, outer iteration by outer iteration: the product (or the oracle under its gemm2 reading: --control 2)
against the oracle -- where inside a frame an excursion of tools/diag/long_sequence_hunt.py begins.
usage: long_frame_trace.py SEED FRAME [--variant throughput] [--control M]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import long_sequences as ls
from oracle import binding
from staticfusion_amd.synth import sequence_arrays, pose_delta

ap = argparse.ArgumentParser()
ap.add_argument("seed", type=int); ap.add_argument("frame", type=int)
ap.add_argument("--variant", default="throughput"); ap.add_argument("--control", type=int, default=0)
a = ap.parse_args()
F = a.frame + 3
d, i, _ = sequence_arrays(a.seed, F, cache_dir="/tmp")
ora = binding.load()
ref = ls.Runner(ora, ls.HostPool(d), ls.HostPool(i), 1, F)
if a.control:
    got = ls.Runner(ora, ls.HostPool(d), ls.HostPool(i), 1, F)
    fn = ora.lib.sfo_test_set_gemm_mode; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]; assert fn(got.s.h, a.control) == 0
else:
    import staticfusion_amd as sf
    got = ls.Runner(sf.load(), ls.DevicePool(d), ls.DevicePool(i), 1, F, variant=a.variant)
for k in range(1, F):
    ref.step(); got.step()
    if k < a.frame - 1:
        continue
    x, y = ref.s.stats(), got.s.stats()
    print("frame", k, "pose delta %.2e %.2e" % pose_delta(ref.s.T(), got.s.T()), "counts", (x.n_outer, x.n_irls), (y.n_outer, y.n_irls),
          "b24 %.2e" % np.abs(ref.s.b() - got.s.b()).max(), "b image %.2e" % np.abs(ref.s.b_image() - got.s.b_image()).max())
    for q in range(min(x.n_outer, y.n_outer)):
        p, r = x.outer[q], y.outer[q]
        f = lambda n: np.abs(np.array(getattr(p, n)[:], dtype=np.float64) - np.array(getattr(r, n)[:], dtype=np.float64)).max()
        print("   outer", q, "level", p.level, "k", p.k, "irls", p.irls_iters, r.irls_iters, "n_valid", p.n_valid, r.n_valid,
              "delta_sol %.4e %.4e" % (p.delta_sol_max, r.delta_sol_max), "| d var %.1e twist_level %.1e b %.1e b_prior %.1e T %.1e" % (f("var"), f("twist_level"), f("b_segm"), f("b_prior"), f("T")),
              "| |twist_level| %.4f %.4f" % (np.linalg.norm(p.twist_level[:]), np.linalg.norm(r.twist_level[:])))
