"""How well does a stream's IRLS iteration count of one frame predict the next? (the key of sf_order_kernel)"""
import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import staticfusion_amd as sf
from staticfusion_amd.synth import make_sequence
import bench
api = sf.load().with_variant("throughput")
p = bench.make_params(api, "sphere")
B, D, F = 2048, 4, 60
from multiprocessing import Pool
seqs = [make_sequence(1000 + i, F, True, 240, 320) for i in range(D)]
s = sf.Solver(api, 240, 320, B, p)
rng = np.random.RandomState(1)
start = rng.randint(0, F - 12, size=B)
for b in range(B):
    sq = seqs[b % D]
    s.set_current(b, *sq["frames"][start[b]])
s.current_to_prediction(); 
hist = []
for k in range(1, 10):
    for b in range(B):
        sq = seqs[b % D]
        s.set_prediction(b, *sq["frames"][start[b] + k - 1])
        s.set_current(b, *sq["frames"][start[b] + k])
    s.process_frame(k)
    T, n_irls, n_outer, pix = s.batch_results()
    hist.append(n_irls.copy())
h = np.array(hist, dtype=np.float64)
for k in range(1, len(h)):
    print("frame %d vs %d: corr %.3f  mean %.1f max %d" % (k, k + 1, np.corrcoef(h[k - 1], h[k])[0, 1], h[k].mean(), h[k].max()))
