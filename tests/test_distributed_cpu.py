"""The N > 1 path of bench.py on CPU: world_size 2, gloo backend, 127.0.0.1 rendezvous.
Each rank owns its own batch of independent streams (no data-path collective); the ranks only meet
in the timing barrier and in the MAX / SUM reductions of the result line."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist

    import bench
    import staticfusion_amd as sf
    from oracle import binding
    from staticfusion_amd.synth import make_batch

    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = binding.load()  # CPU stand-in for the device library: same ABI, same harness code path
    p = bench.make_params(api, "static")
    pairs = make_batch(2, base_seed=1234 + 100000 * rank, distinct=2, out_rows=60, out_cols=80)
    s = sf.Solver(api, 60, 80, 2, p)
    for b in range(2):
        s.set_current(b, *pairs[b]["new"])
        s.set_prediction(b, *pairs[b]["old"])
    c0 = s.counters()
    dist.barrier()
    s.timed_process_frames(0, 2)
    dist.barrier()
    c1 = s.counters()
    elapsed = 0.25 * (rank + 1)  # fake, rank-dependent: MAX must pick rank 1's
    t_max, iters_all, frames_all = bench.reduce_over_ranks(dist, torch.device("cpu"), elapsed, c1[1] - c0[1], c1[0] - c0[0])
    T, _, _, _ = s.batch_results()
    per_rank = bench.gather_per_rank(dist, torch.device("cpu"), elapsed, c1[1] - c0[1], c1[0] - c0[0])
    # the sequences workload: per-rank seeds 1000 + rank * D + q, frames resident in the "device"
    # pool (host memory for the CPU stand-in), every step = advance (prediction := current, current := next frame) + frame
    from staticfusion_amd.synth import make_sequence, pose_delta

    D, F = 2, 6
    seqs = [make_sequence(1000 + rank * D + k, F, sphere=True, out_rows=60, out_cols=80) for k in range(D)]
    col = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).T).ravel()
    pool_d = np.stack([col(f[0]) for sq in seqs for f in sq["frames"]])
    pool_i = np.stack([col(f[1]) for sq in seqs for f in sq["frames"]])
    sq_solver = sf.Solver(api, 60, 80, D, bench.make_params(api, "sequences"))
    idx = lambda step: (np.arange(D) * F + step).astype(np.int32)
    sq_solver.advance_sequences_device(pool_d.ctypes.data, pool_i.ctypes.data, idx(0), D * F)
    sq_solver.push_history(0)
    its, err = 0, 0.0
    for step in range(1, F):
        sq_solver.advance_sequences_device(pool_d.ctypes.data, pool_i.ctypes.data, idx(step), D * F)
        d_pred, _ = sq_solver.prediction(0)
        assert np.array_equal(d_pred, seqs[0]["frames"][step - 1][0])  # prediction is the previous frame
        sq_solver.process_frame(step)
        Ts, n_irls, _, _ = sq_solver.batch_results()
        its += int(n_irls.sum())
        err = max(err, max(pose_delta(seqs[k]["T_gt"][step], Ts[k])[1] for k in range(D)))
    seq_sum = bench.reduce_over_ranks(dist, torch.device("cpu"), 1.0, its, D * (F - 1))
    q.put((rank, t_max, iters_all, frames_all, c1[1] - c0[1], float(np.abs(T).sum()), per_rank, its, seq_sum, err,
           float(np.abs(pool_d).sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_reduction():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, t0, it0, fr0, own0, sig0, pr0, sq_it0, sq_sum0, err0, pool0), (r1, t1, it1, fr1, own1, sig1, pr1, sq_it1, sq_sum1, err1, pool1) = res
    assert pr0 == pr1 and len(pr0) == 2 and [p[0] for p in pr0] == [0.25, 0.5] and [p[1] for p in pr0] == [own0, own1]  # per-rank evidence
    assert sq_sum0 == sq_sum1 and sq_sum0[1] == sq_it0 + sq_it1 and sq_sum0[2] == 2 * 2 * 5   # sequences: SUM over ranks
    assert pool0 != pool1 and sq_it0 > 0 and sq_it1 > 0      # every rank renders its own sequences (seeds 1000 + rank D ...)
    assert err0 < 0.02 and err1 < 0.02                       # and tracks them (frame-to-frame, metres)
    assert t0 == t1 == pytest.approx(0.5)            # MAX over ranks
    assert it0 == it1 == own0 + own1                  # SUM over ranks
    assert fr0 == fr1 == 8                            # 2 ranks x 2 streams x 2 steps
    assert own0 > 0 and own1 > 0 and sig0 != sig1     # different seeds per rank: independent streams
